import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/mui-deepautoencoder_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from golden_util import Golden
from codae.train import HipEmbeddingTrainer
g = Golden("embedding_square"); m = g.meta
sched = [(w.shape[1], w.shape[0], r) for (w, _), r in zip(g.params("init"), g.relu_flags())]
t = HipEmbeddingTrainer(sched, torch.tensor(g["data"]), torch.tensor(g["binary_masks"]).to(torch.uint8),
                        torch.tensor(g["mask_to_use"]).to(torch.int32), m["lr"], m["weight_decay"], clip=1.0,
                        max_batch=m["batch"], precision="f32", device="cuda:0")
t.load_params(g.params("init"))
calls = g.calls()
for i, (idx, run) in enumerate(calls[:8]):
    ii = torch.tensor(idx, dtype=torch.int32, device="cuda:0")
    if i < 5:
        t.train_batch(ii, run=run); torch.cuda.synchronize(); print("train", i, len(idx), t.engine.read_scalars(), flush=True)
    else:
        print("eval", i, len(idx), flush=True)
        y = t.eval_batch(ii, run=run, want_y=True); torch.cuda.synchronize(); print("  ok", t.engine.read_scalars(), float(y.sum()), flush=True)
        y = t.eval_batch(ii, run=run, want_y=False); torch.cuda.synchronize(); print("  ok no-y", t.engine.read_scalars(), flush=True)
