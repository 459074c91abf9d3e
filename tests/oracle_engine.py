"""CPU stand-in for DaeEngine's step_* interface, built on the oracle (test infrastructure).
Lets the shipped data-parallel driver (codae.train.DataParallel) run under gloo without a GPU."""
import numpy as np
import torch

from oracle import dae_oracle as O


class OracleEngine:
    def __init__(self, schedule, params):
        self.schedule = schedule
        self.L = len(schedule)
        self.relu = [r for _, _, r in schedule]
        self.w_off, self.b_off, off = [], [], 0
        for k, n, _ in schedule:
            self.w_off.append(off); off += (k * n + 63) // 64 * 64
        for k, n, _ in schedule:
            self.b_off.append(off); off += (n + 63) // 64 * 64
        self.n_param = off
        self.params = torch.zeros(off); self.grads = torch.zeros(off)
        self.m = torch.zeros(off); self.v = torch.zeros(off)
        for l, (w, b) in enumerate(params):
            self.weight(l).copy_(torch.tensor(w)); self.bias(l).copy_(torch.tensor(b))
        self.t = 0

    def _v(self, flat, l, bias):
        k, n, _ = self.schedule[l]
        return flat[self.b_off[l]:self.b_off[l] + n] if bias else flat[self.w_off[l]:self.w_off[l] + n * k].view(n, k)

    def weight(self, l): return self._v(self.params, l, False)
    def bias(self, l): return self._v(self.params, l, True)

    def _plist(self):
        return [(self.weight(l).numpy(), self.bias(l).numpy()) for l in range(self.L)]

    def step_forward_loss(self, batch, hyper):
        x, fmask = batch
        y, self.acts = O.forward(self._plist(), self.relu, O.corrupt(x, fmask), keep=True)
        rows = hyper["global_rows"]
        self.d = ((y - x) * np.float32(2.0 / (rows * x.shape[1]))).astype(np.float32)
        self.sq = float(((x - y) ** 2).sum())

    def step_backward(self, B, lo, hi):
        ps = self._plist()
        for l in range(hi - 1, lo - 1, -1):
            if self.relu[l]:
                self.d = self.d * (self.acts[l + 1] > 0)
            self._v(self.grads, l, False).copy_(torch.tensor(self.d.T @ self.acts[l]))
            self._v(self.grads, l, True).copy_(torch.tensor(self.d.sum(0)))
            if l > 0:
                self.d = (self.d @ ps[l][0]).astype(np.float32)

    # ---- sharded update counterparts (DataParallel(sharded=True)) ----
    step_count = 0

    def new_accumulator(self):
        return torch.zeros(1, dtype=torch.float64)

    def span_sumsq(self, lo, hi, acc):
        acc += float((self.grads[lo:hi].double() ** 2).sum())

    def step_update_span(self, hyper, lo, hi, total_sq):
        g = self.grads.numpy()[lo:hi].copy()
        total = np.float32(np.sqrt(np.float32(float(total_sq[0]))))
        g = g * np.float32(min(1.0, hyper["clip"] / (float(total) + 1e-6)))
        state = {"t": self.t, "m": [(self.m.numpy()[lo:hi], np.zeros(0, np.float32))], "v": [(self.v.numpy()[lo:hi], np.zeros(0, np.float32))]}
        new = O.adam_step([(self.params.numpy()[lo:hi], np.zeros(0, np.float32))], [(g, np.zeros(0, np.float32))], state,
                          hyper["lr"], hyper["wd"])
        self.params[lo:hi] = torch.tensor(new[0][0])
        self.m[lo:hi] = torch.tensor(state["m"][0][0]); self.v[lo:hi] = torch.tensor(state["v"][0][0])

    def replica_tensors(self):
        return [self.params]

    def after_replica_sync(self):
        self.t += 1                              # (one optimizer step done: Adam's t advances once for all spans)

    def step_update(self, hyper):
        self.t += 1
        g = self.grads.numpy().copy()
        total = np.sqrt((g.astype(np.float64) ** 2).sum())
        g = g * np.float32(min(1.0, hyper["clip"] / (total + 1e-6)))
        state = {"t": self.t - 1, "m": [(self.m.numpy(), np.zeros(0, np.float32))], "v": [(self.v.numpy(), np.zeros(0, np.float32))]}
        new = O.adam_step([(self.params.numpy(), np.zeros(0, np.float32))], [(g, np.zeros(0, np.float32))], state,
                          hyper["lr"], hyper["wd"])
        self.params.copy_(torch.tensor(new[0][0]))
        self.m.copy_(torch.tensor(state["m"][0][0])); self.v.copy_(torch.tensor(state["v"][0][0]))
        self.grad_norm = float(total)
