import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/mui-deepautoencoder_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
from codae import hip as H
for mode in ("native", "x3"):
    os.environ["CODAE_F32_GEMM"] = mode; H.check(H.lib().codae_reload_env())
    tr, orc = T._oracle_vs_engine("f32", 3, 128, 1024, 3, 1e-3)
    tot = bad = 0; worst = 0.0
    for l, (w, b) in enumerate(orc.params):
        got = tr.engine.weight(l).cpu().numpy()
        d = np.abs(got - w) - (1e-5 + 1e-3 * np.abs(w))
        tot += d.size; bad += int((d > 0).sum()); worst = max(worst, float(np.abs(got - w).max()))
    print(mode, "elements", tot, "outside rtol 1e-3 / atol 1e-5:", bad, "largest |diff| %.3e" % worst)
