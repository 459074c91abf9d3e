"""Deviation of the bf16 (throughput) engine from the REFERENCE's own runs (tests/golden/embedding_wide_*.npz):
prints, per fixture, the largest per-step loss / grad-norm deviation, the per-epoch metric deviations and the
relative error of the parameter update.  The tolerances asserted in tests/test_gpu_parity.py (BF16_*) and stated in
DESIGN.md section 3 are derived from this output.  Needs a GPU:  python tools/bf16_curve_dev.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mui-deepautoencoder_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import test_gpu_parity as T  # noqa: E402

for name in ("embedding_wide_square", "embedding_wide_taper"):
    print(name, json.dumps(T.bf16_replay_deviations(name)))
    g = T.Golden(name)
    ad = T.FusedTrainerAdapter(g, "bf16")
    idx, run = g.calls()[0]
    eng = ad.t.engine
    batch = ad.t._batch(T.torch.tensor(idx, dtype=T.torch.int32, device=T.DEV), run)
    hyper = eng.hyper(g.meta["lr"], g.meta["weight_decay"], 1.0, global_rows=len(idx))
    eng.step_forward_loss(batch, hyper)
    eng.step_backward(len(idx), 0, eng.L)
    T.torch.cuda.synchronize()
    print(name, "grad0 rel-L2 dW", ["%.2e" % T._rel_l2(eng.weight_grad(l).cpu().numpy(), gw) for l, (gw, gb) in enumerate(g.list("grad0"))])
    print(name, "grad0 rel-L2 db", ["%.2e" % T._rel_l2(eng.bias_grad(l).cpu().numpy(), gb) for l, (gw, gb) in enumerate(g.list("grad0"))])

# the bf16 engine against the bf16-ROUNDED oracle (same arithmetic, different summation order): per-step deviations
import math  # noqa: E402
import numpy as np  # noqa: E402
from oracle import dae_oracle as O  # noqa: E402
for name in ("embedding_wide_square", "embedding_wide_taper"):
    g = T.Golden(name)
    m = g.meta
    ad = T.FusedTrainerAdapter(g, "bf16")
    orc = O.EmbeddingTrainer(g.params("init"), g.relu_flags(), m["lr"], m["weight_decay"], quant=O.bf16_round)
    eng = ad.t.engine
    for step, (idx, run) in enumerate(g.calls()[:9]):
        _, fmask = O.get_masks(g["binary_masks"], g["nb_missing_per_run"], g["mask_to_use"], 1, idx, run)
        ro = orc.step(g["data"][idx], fmask)
        eng.zero_metric_sums()
        ad.t.train_batch(T.torch.tensor(idx, dtype=T.torch.int32, device=T.DEV), run=run)
        sq, sqp, gsq, loss = eng.read_scalars()
        gw = max(T._rel_l2(eng.weight_grad(l).cpu().numpy(), a) for l, (a, b) in enumerate(orc.last_grads))
        gb = max(T._rel_l2(eng.bias_grad(l).cpu().numpy(), b) for l, (a, b) in enumerate(orc.last_grads))
        print(name, "vs bf16-rounded oracle step %d: loss %.2e gnorm %.2e sqp %.2e dW %.2e db %.2e" % (
            step, abs(loss - float(ro["loss"])) / float(ro["loss"]), abs(math.sqrt(gsq) - float(ro["grad_norm"])) / float(ro["grad_norm"]),
            abs(sqp - float(ro["sq_partial"])) / float(ro["sq_partial"]), gw, gb))
