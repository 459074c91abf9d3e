"""Load the committed golden fixtures (tests/golden/*.npz, produced by
tests/golden/make_golden.py from the reference's own scripts)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RTOL, ATOL = 1e-3, 1e-5          # BASELINE.json north_star tolerance (fp32)


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.z = z
        self.meta = json.loads(str(z["meta"]))
        self.names = self.meta["param_names"]

    def __getitem__(self, k):
        return self.z[k]

    def params(self, prefix):
        """[(W, b)] in layer order from '<prefix>__input_layer__0__weight' keys."""
        ws = [self.z[prefix + "__" + n.replace(".", "__")] for n in self.names]
        return [(ws[i], ws[i + 1]) for i in range(0, len(ws), 2)]

    def list(self, prefix):
        n = len(self.names)
        arr = [self.z["%s__%d" % (prefix, i)] for i in range(n)]
        return [(arr[i], arr[i + 1]) for i in range(0, n, 2)]

    def calls(self):
        flat, off, run = self.z["calls_flat"], self.z["calls_off"], self.z["calls_run"]
        return [(flat[off[i]:off[i + 1]], int(run[i])) for i in range(len(run))]

    def relu_flags(self):
        """ReLU follows every Linear except the last of each Sequential."""
        names = [n for n in self.names if n.endswith("weight")]
        flags = []
        for i, n in enumerate(names):
            last_of_block = (i + 1 == len(names)) or (names[i + 1].split(".")[0] != n.split(".")[0])
            flags.append(not last_of_block)
        return flags


def close(a, b, rtol=RTOL, atol=ATOL):
    return np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


def max_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (ATOL + RTOL * np.abs(b))))
