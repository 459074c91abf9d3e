"""Engine-level parity on a real MI355X: the reference training scripts' runs (golden fixtures
generated from the reference itself) replayed through the HIP path, in both integration modes:

  fused    : HipEmbeddingTrainer (codae_train_step: gather+corrupt, GEMM chain, loss, clip, Adam)
  drop-in  : codae.model classes + torch MSELoss / CombinedCriterion / clip_grad_norm_ / Adam,
             i.e. what the reference scripts execute when they import this package.

Tolerance: rtol 1e-3 / atol 1e-5 (BASELINE.json north_star) in fp32 mode.
"""
import math
import os

import numpy as np
import pytest

from golden_util import Golden, close, max_err
from replay import replay_abalone, replay_embedding

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


@pytest.fixture(params=["chain", "layers"])
def step_path(request, monkeypatch):
    """Narrow bf16 stacks (every width <= 512) have two implementations of codae_train_step: the persistent fused chain
    (one launch for gather + forward + loss + data gradients, one grouped launch for the weight gradients) and the
    per-layer GEMM launches that wide stacks always take (CODAE_NO_CHAIN=1, read at codae_create).  Tests that use this
    fixture run on both."""
    if request.param == "layers":
        monkeypatch.setenv("CODAE_NO_CHAIN", "1")
    else:
        monkeypatch.delenv("CODAE_NO_CHAIN", raising=False)
    return request.param


class FusedTrainerAdapter:
    """step/evaluate interface of tests/replay.py on top of HipEmbeddingTrainer."""

    def __init__(self, g, precision):
        from codae.train import HipEmbeddingTrainer
        m = g.meta
        sched = [(w.shape[1], w.shape[0], r) for (w, _), r in zip(g.params("init"), g.relu_flags())]
        self.t = HipEmbeddingTrainer(sched, torch.tensor(g["data"]), torch.tensor(g["binary_masks"]).to(torch.uint8),
                                     torch.tensor(g["mask_to_use"]).to(torch.int32), m["lr"], m["weight_decay"],
                                     clip=1.0, max_batch=m["batch"], precision=precision, device=DEV)
        self.t.load_params(g.params("init"))
        self.data = g["data"]
        self.pending = None

    def _idx(self, x):
        return self.pending

    def step(self, x, fmask):
        idx = torch.tensor(self.pending_idx, dtype=torch.int32, device=DEV)
        self.t.engine.zero_metric_sums()
        self.t.train_batch(idx, run=self.pending_run)
        sq, sqp, gsq, loss = self.t.engine.read_scalars()
        return {"loss": loss, "grad_norm": math.sqrt(gsq), "sq_full": sq, "sq_partial": sqp}

    def evaluate(self, x, fmask):
        idx = torch.tensor(self.pending_idx, dtype=torch.int32, device=DEV)
        self.t.engine.zero_metric_sums()
        y = self.t.eval_batch(idx, run=self.pending_run, want_y=True)
        sq, sqp, _, _ = self.t.engine.read_scalars()
        return {"y": y.cpu().numpy(), "sq_full": sq, "sq_partial": sqp}


def _replay_fused(g, adapter):
    """replay_embedding feeds x/fmask; the fused path wants indices: wrap the calls list."""
    calls = g.calls()
    it = iter(calls)
    orig_step, orig_eval = adapter.step, adapter.evaluate

    def step(x, fmask):
        adapter.pending_idx, adapter.pending_run = next(it)
        return orig_step(x, fmask)

    def evaluate(x, fmask):
        adapter.pending_idx, adapter.pending_run = next(it)
        return orig_eval(x, fmask)
    adapter.step, adapter.evaluate = step, evaluate
    return replay_embedding(g, adapter)


@pytest.mark.parametrize("name", ["embedding_square", "embedding_taper", "embedding_wide_square", "embedding_wide_taper"])
def test_fused_step_replays_reference_run_f32(name):
    g = Golden(name)
    ad = FusedTrainerAdapter(g, "f32")
    book, steps, _ = _replay_fused(g, ad)
    assert close(steps["loss"], g["step_loss"]), max_err(steps["loss"], g["step_loss"])
    assert close(steps["grad_norm"], g["step_grad_norm"]), max_err(steps["grad_norm"], g["step_grad_norm"])
    for k in ("ftl", "ptl", "fvl", "pvl", "rl"):
        assert close(book[k], g["book_" + k]), (k, book[k], g["book_" + k])
    eng = ad.t.engine
    for l, (gw, gb) in enumerate(g.params("final")):
        assert close(eng.weight(l).cpu().numpy(), gw), ("W", l)
        assert close(eng.bias(l).cpu().numpy(), gb), ("b", l)
    if "adam_m__0" not in g.z.files:            # (the 64-multiple fixtures carry no Adam moments: file size)
        return
    for l, ((mw, mb), (vw, vb)) in enumerate(zip(g.list("adam_m"), g.list("adam_v"))):
        assert close(eng._view(eng.adam_m, l, False).cpu().numpy(), mw, atol=1e-7)
        assert close(eng._view(eng.adam_m, l, True).cpu().numpy(), mb, atol=1e-7)
        assert close(eng._view(eng.adam_v, l, False).cpu().numpy(), vw, atol=1e-10)
        assert close(eng._view(eng.adam_v, l, True).cpu().numpy(), vb, atol=1e-10)


# ---- the bf16 (throughput, benchmarked) engine against REFERENCE runs -------------------------------------------------
# Stated tolerance of the bf16 mode (derived from tools/bf16_curve_dev.py on MI355X, which prints
# the deviations of exactly these replays: every bound below is ~2x the largest deviation measured, not wider):
# measured (r02, MI355X; wide_square / wide_taper): step loss 9.5e-4 / 3.9e-4, grad norm 1.6e-2 / 1.3e-2, book metrics
# <= 3.8e-4 / 8.6e-5, ranking loss 1.3e-3 / 8.4e-4, weight update 3.9e-2 / 2.4e-2, first-step gradients 0.15 / 0.06
BF16_STEP_LOSS_RTOL = 2e-3        # per-step training loss, every step of the run
BF16_STEP_GNORM_RTOL = 3.5e-2     # per-step clip_grad_norm_ total norm
BF16_BOOK_RTOL = 1e-3             # per-epoch ftl / ptl / fvl / pvl (RMSE-type metrics)
BF16_RL_ATOL = 3e-3               # per-epoch ranking loss (a mean of rank fractions in [0, 1])
BF16_UPDATE_REL_L2 = 0.08         # ||(p - p0) - (p_ref - p0)|| / ||p_ref - p0|| per weight matrix after the whole run
# first-step gradients, per tensor, relative L2 vs the reference's fp32 gradients.  bf16 rounding of the activations
# flips the ReLU mask of the ~0.3 % of units whose pre-activation is within rounding of zero, per layer, and the
# flipped terms are O(1) each: the error grows towards the first layer (10 layers: 0.4 % at the output layer, 15 % at
# layer 0).  That this is the arithmetic and not the kernels is pinned by test_fused_bf16_matches_bf16_rounded_oracle.
BF16_GRAD0_REL_L2 = {"embedding_wide_square": 0.30, "embedding_wide_taper": 0.12}


def _rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def bf16_replay_deviations(name):
    """Replay a reference run through the bf16 fused step; {quantity: largest deviation} (also used by
    tools/bf16_curve_dev.py to derive the tolerances above)."""
    g = Golden(name)
    ad = FusedTrainerAdapter(g, "bf16")
    book, steps, _ = _replay_fused(g, ad)
    ref_l, ref_n = g["step_loss"], g["step_grad_norm"]
    dev = {"steps": len(ref_l),
           "step_loss": float(np.max(np.abs(np.asarray(steps["loss"]) - ref_l) / np.abs(ref_l))),
           "step_grad_norm": float(np.max(np.abs(np.asarray(steps["grad_norm"]) - ref_n) / np.abs(ref_n)))}
    for k in ("ftl", "ptl", "fvl", "pvl"):
        dev["book_" + k] = float(np.max(np.abs(np.asarray(book[k], dtype=np.float64) - g["book_" + k]) / np.abs(g["book_" + k])))
    dev["book_rl_abs"] = float(np.max(np.abs(np.asarray(book["rl"], dtype=np.float64) - g["book_rl"])))
    eng = ad.t.engine
    upd_w, upd_b = [], []
    for l, ((w0, b0), (w1, b1)) in enumerate(zip(g.params("init"), g.params("final"))):
        upd_w.append(_rel_l2(eng.weight(l).cpu().numpy() - w0, w1 - w0))
        upd_b.append(_rel_l2(eng.bias(l).cpu().numpy() - b0, b1 - b0))
    dev["update_w"] = max(upd_w); dev["update_b"] = max(upd_b)
    dev["loss_first_last"] = (float(ref_l[0]), float(ref_l[-1]))
    return dev


@pytest.mark.parametrize("name", ["embedding_wide_square", "embedding_wide_taper"])
def test_fused_bf16_replays_reference_run(name, step_path):
    """The BENCHMARKED mode pinned to the reference: a whole run of the reference's own script (4 epochs, 36 optimizer
    steps + validation, widths multiples of 64) replayed through the bf16 engine; the loss curve is asserted per step."""
    d = bf16_replay_deviations(name)
    assert d["steps"] == 36
    assert d["loss_first_last"][1] < 0.9 * d["loss_first_last"][0]           # the curve moves: not a flat line compared
    assert d["step_loss"] <= BF16_STEP_LOSS_RTOL, d
    assert d["step_grad_norm"] <= BF16_STEP_GNORM_RTOL, d
    for k in ("ftl", "ptl", "fvl", "pvl"):
        assert d["book_" + k] <= BF16_BOOK_RTOL, (k, d)
    assert d["book_rl_abs"] <= BF16_RL_ATOL, d
    assert d["update_w"] <= BF16_UPDATE_REL_L2, d


@pytest.mark.parametrize("name", ["embedding_wide_square", "embedding_wide_taper"])
def test_fused_bf16_matches_bf16_rounded_oracle(name, step_path):
    """Pins the bf16 KERNELS (as opposed to the bf16 arithmetic): the oracle with its rounding hook restates the
    reference's algorithm with bf16 stores at the points where the engine rounds (input, weight shadow, activations,
    activation gradients; fp32 accumulation, fp32 master weights and Adam).  On the first step - identical parameters
    on both sides - every gradient tensor agrees to 1e-3 relative L2 (measured 2e-8 on the 6-layer stack, 7e-4 on the
    10-layer one: fp32 summation order, MFMA vs BLAS, now and then moves one bf16 rounding by an ulp), against 0.06-0.15
    vs the fp32 reference.  Later steps compare loss and gradient norm only: Adam normalises every element's step to
    ~lr, so an element whose gradient is within rounding of zero moves +lr on one side and -lr on the other, ReLU masks
    flip with it, and the two trajectories' GRADIENTS drift apart by 4-24 % rel. L2 within an epoch while their LOSS
    curves stay within 1.3e-4 (tools/bf16_curve_dev.py prints both) - the same sensitivity that makes bf16 sit 15 %
    from the fp32 gradients at layer 0."""
    from oracle import dae_oracle as O
    g = Golden(name)
    m = g.meta
    ad = FusedTrainerAdapter(g, "bf16")
    orc = O.EmbeddingTrainer(g.params("init"), g.relu_flags(), m["lr"], m["weight_decay"], quant=O.bf16_round)
    eng = ad.t.engine
    for step, (idx, run) in enumerate(g.calls()[:9]):                      # the first epoch's nine optimizer steps
        _, fmask = O.get_masks(g["binary_masks"], g["nb_missing_per_run"], g["mask_to_use"], 1, idx, run)
        ro = orc.step(g["data"][idx], fmask)
        eng.zero_metric_sums()
        ad.t.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=run)
        sq, sqp, gsq, loss = eng.read_scalars()
        first = step == 0
        assert abs(loss - float(ro["loss"])) <= (1e-6 if first else 4e-4) * float(ro["loss"]), (step, loss, ro["loss"])
        assert abs(math.sqrt(gsq) - float(ro["grad_norm"])) <= (1e-5 if first else 1e-2) * float(ro["grad_norm"]), (step, math.sqrt(gsq), ro["grad_norm"])
        assert abs(sqp - float(ro["sq_partial"])) <= (1e-6 if first else 6e-4) * float(ro["sq_partial"])
        if first:
            for l, (gw, gb) in enumerate(orc.last_grads):                  # this step's gradients, every tensor
                assert _rel_l2(eng.weight_grad(l).cpu().numpy(), gw) <= 2e-3, ("dW", l)
                assert _rel_l2(eng.bias_grad(l).cpu().numpy(), gb) <= 2e-3, ("db", l)


@pytest.mark.parametrize("name", ["embedding_square", "embedding_taper"])
def test_fused_bf16_on_widths_that_are_not_multiples_of_64(name):
    """The reference only requires io % embedding_size == 0; its stock fixtures here are 3 x 16 = 48 wide (square) and taper
    48 -> 40 -> 32 -> 24 -> 16 -> ... -> 48.  Round 2's bf16 engine refused such stacks (every width is a GEMM k extent of whole
    64-deep tiles) and they fell to the exact-fp32 engine; now the activation / activation-gradient buffers carry zero pad columns
    up to the next multiple of 64 and the weights stay unpadded (a row's over-read is the head of the next row, times zero).
    The reference run's first epoch through the bf16 step against the bf16-rounding oracle: first-step loss 1e-6, every gradient
    tensor to 2e-3 relative L2 (the kernel-level bound of the wide fixtures), later steps loss 4e-4 / grad-norm 1e-2."""
    from oracle import dae_oracle as O
    g = Golden(name)
    m = g.meta
    ad = FusedTrainerAdapter(g, "bf16")
    eng = ad.t.engine
    assert eng.precision == 1 and any(k % 64 or n % 64 for k, n, _ in eng.schedule)
    orc = O.EmbeddingTrainer(g.params("init"), g.relu_flags(), m["lr"], m["weight_decay"], quant=O.bf16_round)
    n_train = sum(1 for _ in g.calls()[:6])
    for step, (idx, run) in enumerate(g.calls()[:n_train]):
        _, fmask = O.get_masks(g["binary_masks"], g["nb_missing_per_run"], g["mask_to_use"], 1, idx, run)
        ro = orc.step(g["data"][idx], fmask)
        eng.zero_metric_sums()
        ad.t.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=run)
        sq, sqp, gsq, loss = eng.read_scalars()
        first = step == 0
        assert abs(loss - float(ro["loss"])) <= (1e-6 if first else 4e-4) * float(ro["loss"]), (step, loss, ro["loss"])
        assert abs(math.sqrt(gsq) - float(ro["grad_norm"])) <= (1e-5 if first else 1e-2) * float(ro["grad_norm"]), (step, math.sqrt(gsq), ro["grad_norm"])
        assert abs(sqp - float(ro["sq_partial"])) <= (1e-6 if first else 6e-4) * float(ro["sq_partial"])
        if first:
            for l, (gw, gb) in enumerate(orc.last_grads):
                assert _rel_l2(eng.weight_grad(l).cpu().numpy(), gw) <= 2e-3, ("dW", l)
                assert _rel_l2(eng.bias_grad(l).cpu().numpy(), gb) <= 2e-3, ("db", l)


@pytest.mark.parametrize("name", ["embedding_wide_square", "embedding_wide_taper"])
def test_fused_first_step_grads_bf16(name):
    g = Golden(name)
    ad = FusedTrainerAdapter(g, "bf16")
    idx, run = g.calls()[0]
    eng = ad.t.engine
    batch = ad.t._batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run)
    hyper = eng.hyper(g.meta["lr"], g.meta["weight_decay"], 1.0, global_rows=len(idx))
    eng.step_forward_loss(batch, hyper)
    eng.step_backward(len(idx), 0, eng.L)
    torch.cuda.synchronize()
    for l, (gw, gb) in enumerate(g.list("grad0")):
        assert _rel_l2(eng.weight_grad(l).cpu().numpy(), gw) <= BF16_GRAD0_REL_L2[name], ("dW", l)
        assert _rel_l2(eng.bias_grad(l).cpu().numpy(), gb) <= BF16_GRAD0_REL_L2[name], ("db", l)


@pytest.mark.parametrize("name", ["embedding_square", "embedding_wide_square", "embedding_wide_taper"])
def test_fused_first_step_grads_f32(name):
    g = Golden(name)
    ad = FusedTrainerAdapter(g, "f32")
    idx, run = g.calls()[0]
    eng = ad.t.engine
    batch = ad.t._batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run)
    hyper = eng.hyper(g.meta["lr"], g.meta["weight_decay"], 1.0, global_rows=len(idx))
    eng.step_forward_loss(batch, hyper)
    eng.step_backward(len(idx), 0, eng.L)
    torch.cuda.synchronize()
    for l, (gw, gb) in enumerate(g.list("grad0")):
        assert close(eng.weight_grad(l).cpu().numpy(), gw, atol=1e-8), ("dW", l)
        assert close(eng.bias_grad(l).cpu().numpy(), gb, atol=1e-8), ("db", l)


class DropInEmbeddingAdapter:
    """The reference script's loop body with this package's model + torch loss/optimizer."""

    def __init__(self, g):
        from codae.model import EmbeddingDenoisingAutoencoder
        m = g.meta
        torch.manual_seed(m["seed"])           # same seed as the golden run -> same Xavier draws
        self.model = EmbeddingDenoisingAutoencoder(m["S"] * m["E"], m["z"], m["E"], m["nb_input_layer"],
                                                   m["nb_output_layer"], False)
        self.init_matches = all(np.array_equal(p.detach().numpy(), g["init__" + n.replace(".", "__")])
                                for n, p in self.model.state_dict().items())
        self.model.to(DEV)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=m["lr"], weight_decay=m["weight_decay"])
        self.crit = torch.nn.MSELoss(reduction="mean")

    def step(self, x, fmask):
        x = torch.tensor(x, device=DEV); fmask = torch.tensor(fmask, device=DEV)
        c = self.model.corrupt(input_data=x, mask=fmask)
        y = self.model(c)
        loss = self.crit(x, y)
        self.opt.zero_grad()
        loss.backward()
        n = torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1)
        self.opt.step()
        se = ((x - y) ** 2).detach()
        return {"loss": float(loss), "grad_norm": float(n), "sq_full": float(se.sum()),
                "sq_partial": float(((1 - fmask) * se).sum()), "y": y.detach().cpu().numpy()}

    def evaluate(self, x, fmask):
        x = torch.tensor(x, device=DEV); fmask = torch.tensor(fmask, device=DEV)
        y = self.model(self.model.corrupt(input_data=x, mask=fmask))
        se = ((x - y) ** 2).detach()
        return {"y": y.detach().cpu().numpy(), "sq_full": float(se.sum()), "sq_partial": float(((1 - fmask) * se).sum())}


@pytest.mark.parametrize("name", ["embedding_square", "embedding_taper"])
def test_dropin_model_replays_reference_run(name):
    g = Golden(name)
    ad = DropInEmbeddingAdapter(g)
    assert ad.init_matches, "same torch seed must reproduce the reference's initial weights bit for bit"
    book, steps, _ = replay_embedding(g, ad)
    assert close(steps["loss"], g["step_loss"]), max_err(steps["loss"], g["step_loss"])
    assert close(steps["grad_norm"], g["step_grad_norm"])
    for k in ("ftl", "ptl", "fvl", "pvl", "rl"):
        assert close(book[k], g["book_" + k]), (k, book[k], g["book_" + k])
    sd = ad.model.state_dict()
    for n in g.names:
        assert close(sd[n].cpu().numpy(), g["final__" + n.replace(".", "__")]), n
    assert "codae.hip" in str(type(ad.model._engine).__module__)


class DropInAbaloneAdapter:
    def __init__(self, g):
        from codae.model import MixedVariableDenoisingAutoencoder
        from codae.tool import CombinedCriterion
        m = g.meta
        torch.manual_seed(m["seed"])
        self.model = MixedVariableDenoisingAutoencoder(m["arch"], 11, 11, torch.device(DEV), 2, 2, True)
        self.init_matches = all(np.array_equal(p.detach().numpy(), g["init__" + n.replace(".", "__")])
                                for n, p in self.model.state_dict().items())
        self.model.to(DEV)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=m["lr"], weight_decay=m["weight_decay"])
        self.crit = CombinedCriterion(arch=m["arch"], k_max=m["k_max"], device=torch.device(DEV),
                                      observation_mask=torch.tensor(g["type_mask"]), weight=m["weight"], reduction="mean")

    def step(self, x, fmask):
        x = torch.tensor(x, device=DEV); fmask = torch.tensor(fmask, device=DEV)
        y = self.model(self.model.corrupt(input_data=x, mask=fmask))
        loss = self.crit(x=x, y=y)
        self.opt.zero_grad()
        loss.backward()
        n = torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1)
        self.opt.step()
        return {"loss": float(loss), "grad_norm": float(n), "y": y.detach().cpu().numpy()}

    def evaluate(self, x, fmask):
        x = torch.tensor(x, device=DEV); fmask = torch.tensor(fmask, device=DEV)
        return {"y": self.model(self.model.corrupt(input_data=x, mask=fmask)).detach().cpu().numpy()}


def test_dropin_abalone_replays_reference_run():
    """11-wide layers: every GEMM runs on the exact-fp32 MFMA kernel with ragged tiles."""
    g = Golden("abalone_k2")
    ad = DropInAbaloneAdapter(g)
    assert ad.init_matches
    book, steps = replay_abalone(g, ad)
    assert close(steps["loss"], g["step_loss"]), max_err(steps["loss"], g["step_loss"])
    assert close(steps["grad_norm"], g["step_grad_norm"])
    for k in ("ftl", "ptl", "fvl", "pvl"):
        assert close(book[k], g["book_" + k]), k
    for k in ("ftl_per_k", "ptl_per_k", "fvl_per_k", "pvl_per_k"):
        assert close(np.stack(book[k]), g["book_" + k]), k
    sd = ad.model.state_dict()
    for n in g.names:
        assert close(sd[n].cpu().numpy(), g["final__" + n.replace(".", "__")]), n


def _oracle_vs_engine(precision, S, E, B, steps, tol):
    """Seeded synthetic run at a size the oracle finishes in seconds: fused HIP step vs oracle."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(42)
    N = 4 * B
    data = rng.random((N, io), dtype=np.float32)
    data = data / (data.max() - data.min())
    sched = O.layer_schedule(io, io, 4, 4, False, "embedding")
    params = O.init_params(sched, rng)
    arch = [{"size": E, "position": s * E} for s in range(S)]
    bm, nmr, _ = O.corrupter_tables(arch, 1)
    mtu = np.stack([rng.permutation(S) for _ in range(N)])
    lr, wd = 1e-3, 1e-4
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu).to(torch.int32),
                             lr, wd, 1.0, max_batch=B, precision=precision, device=DEV)
    tr.load_params(params)
    orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], lr, wd)
    for s in range(steps):
        idx = rng.permutation(N)[:B]
        _, fmask = O.get_masks(bm, nmr, mtu, 1, idx, 0)
        ro = orc.step(data[idx], fmask)
        tr.engine.zero_metric_sums()
        tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=0)
        sq, sqp, gsq, loss = tr.engine.read_scalars()
        assert abs(loss - float(ro["loss"])) <= tol * abs(float(ro["loss"])), (s, loss, ro["loss"])
        assert abs(math.sqrt(gsq) - float(ro["grad_norm"])) <= 5 * tol * float(ro["grad_norm"]), (s, math.sqrt(gsq), ro["grad_norm"])
        assert abs(sqp - float(ro["sq_partial"])) <= tol * float(ro["sq_partial"])
    return tr, orc


def test_fused_f32_vs_oracle_3x128_batch1024():
    """Three Adam steps (lr 1e-3) of the fp32 engine against the fp32 oracle: loss, gradient norm and the partial metric sum of
    every step to 1e-3 (in _oracle_vs_engine); the updated parameters at rtol 1e-3 / atol 1e-5 - for all but the elements whose
    gradient sits within fp32 rounding of zero.  Adam's first steps move EVERY element by lr in the direction of the sign of
    g + wd w, so two fp32 evaluations that disagree on such a sign end 2 lr apart on that element.  The engine's fp32-MFMA GEMMs
    round like numpy's sgemm and agree with the oracle on every one of the 1.47 M elements (CODAE_F32_GEMM=native: 0 outside,
    largest difference 3e-7); its default GEMMs (three bf16 planes per operand, gemm_f32x3.hip) are CLOSER TO THE TRUTH than the
    oracle - against a float64 evaluation of the first step the oracle gets 74 signs wrong, these GEMMs none
    (test_f32_first_step_gradients_against_float64 below: sgemm-style fp32 puts two pre-activations that sit within an ulp of
    zero on the other side of the ReLU than float64 does, the plane GEMMs' pre-activations are accurate enough to take
    float64's side) - and so land on the other side of the oracle on 0.4 % of the elements.
    Bound: at least 99 % of the elements inside rtol 1e-3 / atol 1e-5, every element inside 2 lr per step."""
    steps, lr = 3, 1e-3
    tr, orc = _oracle_vs_engine("f32", 3, 128, 1024, steps, 1e-3)
    outside = total = 0
    for l, (w, b) in enumerate(orc.params):
        got = tr.engine.weight(l).cpu().numpy()
        d = np.abs(got.astype(np.float64) - w)
        outside += int((d > 1e-5 + 1e-3 * np.abs(w)).sum()); total += w.size
        assert d.max() <= 2 * lr * steps + 1e-5, (l, d.max())
        assert close(tr.engine.bias(l).cpu().numpy(), b, atol=2 * lr * steps + 1e-5)
    assert outside <= 0.01 * total, (outside, total)


@pytest.mark.parametrize("S,E,B", [(3, 128, 1024), (3, 256, 500)])
def test_f32_first_step_gradients_against_float64(monkeypatch, S, E, B):
    """The same stack's first-step weight gradients against a float64 evaluation of the step (torch on the GPU), for the numpy
    fp32 oracle and for the fp32 engine on both of its GEMM kernels: relative L2 over all weight gradients and the number of
    elements of g + wd w with the wrong sign.  Measured: oracle 2.6e-4 / 74 wrong signs of 1 474 560; engine on the fp32-MFMA GEMMs
    the same 2.6e-4 / 74 (it reproduces numpy's rounding); engine on the bf16-plane GEMMs 3.7e-7 / 0.  The default engine must be at
    least as close to float64 as the oracle is, in both measures, and within 1e-5 outright.
    (3 x 256, batch 500: a ragged batch at a width where the ten weight gradients go out as ONE grouped launch of the plane kernel,
    gemm_f32x3_grouped_kernel - 36 tiles per layer.  Measured there: oracle 5.1e-5 / 39 wrong signs; fp32-MFMA GEMMs 3.1e-7 / 0 - its
    split-K sums no longer follow numpy's -; bf16-plane GEMMs 3.6e-7 / 0.)"""
    from codae import hip as H
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(42)
    N = 4 * B
    data = rng.random((N, io), dtype=np.float32)
    data = data / (data.max() - data.min())
    sched = O.layer_schedule(io, io, 4, 4, False, "embedding")
    params = O.init_params(sched, rng)
    bm, nmr, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = np.stack([rng.permutation(S) for _ in range(N)])
    idx = rng.permutation(N)[:B]
    _, fmask = O.get_masks(bm, nmr, mtu, 1, idx, 0)
    wd = 1e-4
    x = torch.tensor(data[idx], dtype=torch.float64, device=DEV)
    h = x * torch.tensor(fmask, dtype=torch.float64, device=DEV)
    Ws = [torch.tensor(w, dtype=torch.float64, device=DEV, requires_grad=True) for w, _ in params]
    bs = [torch.tensor(b, dtype=torch.float64, device=DEV, requires_grad=True) for _, b in params]
    for l, (_, _, relu) in enumerate(sched):
        h = h @ Ws[l].T + bs[l]
        if relu:
            h = torch.relu(h)
    ((h - x) ** 2).mean().backward()
    truth = [W.grad.cpu().numpy() for W in Ws]

    def distance(grads):
        wrong, num, den = 0, 0.0, 0.0
        for l, g in enumerate(grads):
            ga = g.astype(np.float64) + wd * params[l][0]
            ta = truth[l] + wd * params[l][0]
            wrong += int((np.sign(ga) != np.sign(ta)).sum())
            num += float(((g - truth[l]) ** 2).sum()); den += float((truth[l] ** 2).sum())
        return (num / den) ** 0.5, wrong
    orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], 1e-3, wd)
    orc.step(data[idx], fmask)
    o_rel, o_wrong = distance([gw for gw, _ in orc.last_grads])
    res = {}
    for mode in ("native", "x3"):
        monkeypatch.setenv("CODAE_F32_GEMM", mode)
        H.check(H.lib().codae_reload_env())
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu).to(torch.int32), 1e-3, wd, 1.0,
                                 max_batch=B, precision="f32", device=DEV)
        tr.load_params(params)
        tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=0)
        res[mode] = distance([tr.engine.weight_grad(l).cpu().numpy() for l in range(tr.engine.L)])
        del tr
    monkeypatch.delenv("CODAE_F32_GEMM")
    H.check(H.lib().codae_reload_env())
    print("first-step weight gradients vs float64 (rel L2, wrong signs): oracle %.3g / %d, fp32-MFMA GEMMs %.3g / %d, bf16-plane GEMMs %.3g / %d"
          % (o_rel, o_wrong, res["native"][0], res["native"][1], res["x3"][0], res["x3"][1]))
    assert res["native"][0] <= 1e-3 and res["native"][1] <= 2 * o_wrong + 10, (res, o_rel, o_wrong)
    assert res["x3"][0] <= 1e-5 and res["x3"][0] <= 2 * o_rel and res["x3"][1] <= o_wrong + 2, (res, o_rel, o_wrong)


@pytest.mark.parametrize("group_tile", ["auto", "0", "1", "2"])
def test_chain_step_equals_per_layer_step(monkeypatch, group_tile):
    """(group_tile: the grouped weight-gradient launch on its automatic tile, then forced to 128 x 128, 64 x 128, 64 x 64;
    the norm's sum g^2 comes from that launch's epilogue, scalars[2] below.)  The persistent fused chain against the per-layer launches on BASELINE config C2's shape (3 x 128, batch 1024; and a
    ragged batch of a tapered stack): both run the same MFMA instruction over the same k order, so every saved
    activation and every activation gradient must agree BIT FOR BIT (the whole workspaces are compared); weight
    gradients differ only by fp32 summation order (the per-layer path splits the batch reduction 8 ways, the grouped
    launch does not), bias gradients and the loss by how many rows a partial sum spans (16 vs 128)."""
    from codae.hip.engine import DaeEngine
    from codae import hip
    from oracle import dae_oracle as O
    if group_tile == "auto":
        monkeypatch.delenv("CODAE_GROUP_TILE", raising=False)
    else:
        monkeypatch.setenv("CODAE_GROUP_TILE", group_tile)
    cases = [(3, 128, 384, 4, 1024), (3, 64, 64, 2, 333)]
    if group_tile == "auto":
        # the chain's limits: widest panel (512), one sample (15 pad rows), 15 layers, the row limit itself (128 workgroups)
        cases += [(4, 128, 512, 2, 200), (3, 64, 192, 2, 1), (2, 64, 128, (7, 6), 100), (3, 64, 192, 2, 2048)]
    for (S, E, z, nl, B) in cases:
        io = S * E
        rng = np.random.default_rng(B)
        n_in, n_out = nl if isinstance(nl, tuple) else (nl, nl)
        sched = O.layer_schedule(io, z, n_in, n_out, False, "embedding")
        params = O.init_params(sched, rng)
        data = torch.tensor(rng.random((B + 50, io), dtype=np.float32), device=DEV)
        bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
        table = torch.tensor(bm).to(torch.uint8).to(DEV)
        mid = torch.tensor(rng.integers(0, S, B), dtype=torch.int32, device=DEV)
        rows = torch.tensor(rng.permutation(B + 50)[:B], dtype=torch.int32, device=DEV)
        out = []
        for chain in (True, False):
            if chain:
                monkeypatch.delenv("CODAE_NO_CHAIN", raising=False)
            else:
                monkeypatch.setenv("CODAE_NO_CHAIN", "1")
            eng = DaeEngine(sched, B, "bf16", DEV)
            assert eng.step_path(B) == ("chain" if chain else "layers"), (S, E, z, nl, B)
            eng.load_params(params)
            batch = eng.make_batch(data, rows, mid, table)
            eng.train_step(batch, eng.hyper(1e-3, 1e-4, clip=1.0, global_rows=B))
            torch.cuda.synchronize()
            out.append((eng.grads.clone(), eng.read_scalars(), eng.params.clone(), eng.b_off[0], eng.acts.clone(), eng.dacts.clone()))
        (ga, sa, pa, nw, aa, da), (gb, sb, pb, _, ab, db) = out
        assert torch.equal(aa, ab), "saved activations"
        assert torch.equal(da, db), "activation gradients"
        assert float(da.float().abs().max()) > 0
        assert float((ga[:nw] - gb[:nw]).abs().max()) <= 1e-5 * float(gb[:nw].abs().max())
        assert float((ga[nw:] - gb[nw:]).abs().max()) <= 1e-5 * float(gb[nw:].abs().max())
        for k in (0, 1, 3):
            assert abs(sa[k] - sb[k]) <= 1e-6 * abs(sb[k]), (k, sa, sb)
        assert abs(sa[2] - sb[2]) <= 1e-5 * sb[2]
        assert float((pa - pb).abs().max()) <= 2.1e-3          # (an Adam step moves an element by at most ~lr either way)
        assert float((pa - pb).abs().mean()) <= 1e-6


def test_fused_bf16_vs_oracle_3x128_batch1024(step_path):
    """bf16 operands, fp32 accumulate: loss / grad-norm within 2 % of the fp32 oracle (stated
    tolerance of the throughput mode); parameters move by lr-sized Adam steps either way."""
    tr, orc = _oracle_vs_engine("bf16", 3, 128, 1024, 3, 2e-2)
    lr, steps = 1e-3, 3
    for l, (w, b) in enumerate(orc.params):
        d = np.abs(tr.engine.weight(l).cpu().numpy() - w)
        # Adam normalises every element's step to ~lr, so an element whose gradient is within bf16
        # noise of zero can move the other way: at most 2*lr apart per step, and only a few do
        assert d.max() <= 2 * lr * steps + 1e-6, (l, d.max())
        assert np.median(d) < 2e-5, (l, np.median(d))
        assert (d > lr).mean() < 0.02, (l, (d > lr).mean())


def test_fused_bf16_ragged_batch(step_path):
    """partial last batch (B not a multiple of 64): zero-padded rows must not leak into gradients."""
    tr, orc = _oracle_vs_engine("bf16", 3, 64, 200, 2, 2e-2)


def test_engine_rejects_bad_shapes():
    from codae.hip import HipError
    from codae.hip.engine import DaeEngine
    with pytest.raises(HipError):
        DaeEngine([(11, 11, True), (11, 11, False)], 64, "bf16", DEV)      # widths not % 8 (16-byte rows)
    DaeEngine([(48, 40, True), (40, 48, False)], 64, "bf16", DEV)          # not % 64: fine since round 3 (padded row strides)
    eng = DaeEngine([(11, 11, True), (11, 11, False)], 64, "f32", DEV)
    with pytest.raises(HipError):
        eng.forward(torch.zeros(65, 11, device=DEV))                         # batch > max_batch
    with pytest.raises(HipError):
        eng.forward(torch.zeros(4, 12, device=DEV))                          # wrong width


def test_bucketed_allreduce_path_on_rccl_single_rank(monkeypatch):
    """One-rank RCCL group with the collectives forced on: the bucketed backward + all-reduce +
    update path must give exactly the plain fused step (sum over one rank is the identity)."""
    import torch.distributed as dist
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    monkeypatch.setenv("CODAE_DP_FORCE_ALLREDUCE", "1")
    monkeypatch.setenv("CODAE_NO_CHAIN", "1")        # (the bucketed step is per-layer: compare it with the per-layer fused step)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        S, E, B = 3, 64, 256
        io = S * E
        rng = np.random.default_rng(9)
        data = rng.random((2 * B, io), dtype=np.float32)
        sched = O.layer_schedule(io, io, 4, 4, False, "embedding")
        params = O.init_params(sched, rng)
        bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
        mtu = rng.integers(0, S, (2 * B, 1)).astype(np.int32)
        outs = []
        for distributed, sharded in ((False, False), (True, False), (True, True)):
            # clip at 100: the coefficient is exactly 1 on both paths (the fused step gathers sum g^2 in the slab reduces,
            # the data-parallel step in one pass over the reduced gradients: different fp32 summation orders)
            tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4,
                                     100.0, max_batch=B, precision="bf16", device=DEV, distributed=distributed, n_buckets=4,
                                     sharded_update=sharded)
            tr.load_params(params)
            for s in range(3):
                tr.train_batch(torch.arange(s * 64, s * 64 + B, dtype=torch.int32, device=DEV), run=0)
            outs.append((tr.engine.params.clone(), tr.engine.shadow.clone(), tr.engine.shadow_t.clone()))
            if distributed:
                assert tr.dp.always_reduce and len(tr.dp.buckets) == 4 and tr.dp.sharded == sharded
        # no float atomics anywhere in the step (bias gradients: per-tile partial sums added in a fixed order):
        # the bucketed path must reproduce the fused step bit for bit, bias block included
        for k in range(3):                     # fp32 parameters, bf16 shadow, transposed shadow
            assert torch.equal(outs[0][k], outs[1][k])
            # sharded update (reduce-scatter / span Adam / all-gather of the shadows; one rank = the whole vector): its
            # Adam runs in the flat kernel, the fused step's in the tiled one - the same formula, but the compiler
            # contracts a multiply-add differently in a handful of places (measured: 117 of 370,560 parameters differ,
            # by at most 4.7e-10 = one ulp); a bf16 shadow element may then round the other way once in a blue moon
            d = (outs[0][k].float() - outs[2][k].float()).abs()
            if k == 0:
                assert float(d.max()) <= 1e-8 and int((d > 0).sum()) <= d.numel() // 1000
            else:
                assert int((d > 0).sum()) <= 4
    finally:
        dist.destroy_process_group()


def test_library_owned_rccl_step_equals_the_torch_distributed_step(monkeypatch):
    """codae_train_step_dp - the library's own RCCL communicator, the bucket all-reduces issued on its own stream inside one call
    per step - against the same bucketed step through torch.distributed (CODAE_DP_FORCE_ALLREDUCE=1), one rank (all this build's
    boxes have): three steps at 3 x 256 / batch 2048, per-layer buckets: parameters, Adam moments, both bf16 shadows and the
    step's scalars bit-identical (a one-rank SUM all-reduce is the identity; everything else is the same launches)."""
    import torch.distributed as dist
    from codae.train import HipEmbeddingTrainer, init_rccl_process_group
    from oracle import dae_oracle as O
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("CODAE_DP_FORCE_ALLREDUCE", "1")
    S, E, B = 3, 256, 2048
    io = S * E
    rng = np.random.default_rng(5)
    data = rng.random((2 * B, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 3, 3, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (2 * B, 1)).astype(np.int32)
    order = [torch.tensor(rng.permutation(2 * B)[:B], dtype=torch.int32, device=DEV) for _ in range(3)]
    init_rccl_process_group(torch.device(DEV))
    try:
        outs = []
        for native in (False, True):
            tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                     max_batch=B, precision="bf16", device=DEV, distributed=True, native_dp=native)
            assert tr.dp.native == native and len(tr.dp.buckets) == len(sched)
            tr.load_params(params)
            for idx in order:
                tr.train_batch(idx, run=0)
            eng = tr.engine
            outs.append((eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.shadow.clone(), eng.shadow_t.clone(), eng.read_scalars()))
            del tr
        for a, b2 in zip(outs[0][:5], outs[1][:5]):
            assert torch.equal(a, b2)
        assert outs[0][5] == outs[1][5] and outs[0][5][3] > 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S,E,B", [(3, 128, 1000), (3, 320, 8192)], ids=["tile128x128", "tile256x192"])
def test_fused_loss_epilogue_matches_separate_loss_kernel(monkeypatch, S, E, B):
    """bf16 training step with the MSE loss folded into the last forward GEMM vs the same step with
    the stand-alone loss kernel (CODAE_NO_FUSED_LOSS=1): same dY up to bf16 rounding of y, so loss,
    metric sums and gradient norm must agree closely; ragged B exercises the zeroed pad rows."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(11)
    N = B + 100
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    idx = torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV)
    res = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("CODAE_NO_FUSED_LOSS", raising=False)
        else:
            monkeypatch.setenv("CODAE_NO_FUSED_LOSS", "1")
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                 max_batch=B, precision="bf16", device=DEV)
        tr.load_params(params)
        tr.train_batch(idx, run=0)
        sq, sqp, gsq, loss = tr.engine.read_scalars()
        res.append((sq, sqp, gsq ** 0.5, loss, tr.engine.bias_grad(tr.engine.L - 1).cpu().numpy().copy(),
                    tr.engine.weight_grad(0).cpu().numpy().copy()))
    a, b = res
    for k in range(4):
        assert abs(a[k] - b[k]) <= 2e-3 * abs(b[k]), (k, a[k], b[k])
    assert np.allclose(a[4], b[4], rtol=2e-2, atol=1e-7)          # last-layer bias gradient
    assert np.allclose(a[5], b[5], rtol=5e-2, atol=2e-6)          # first-layer weight gradient (through the whole chain)


def _c3_problem():
    """BASELINE config C3 exactly as bench.py builds it: embedding.yaml's topology at 3 x 512 -> 10 x
    Linear(1536, 1536), batch 8192, dataset / blanked slots from default_rng(1234 / 5678), Xavier weights, Adam lr 1e-5 wd 1e-4."""
    import bench
    from oracle import dae_oracle as O
    S, E, B = bench.CONFIGS["c3"]
    io = S * E
    sched = bench.square_schedule(io, bench.N_IN, bench.N_OUT)
    data, blank = bench.make_inputs(2 * B, io, S)
    params = O.init_params(sched, np.random.default_rng(0))
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    idx = np.random.default_rng(3).permutation(2 * B)[:B]
    return S, E, B, io, sched, data, blank, params, bm, idx


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_full_size_c3_step_matches_the_oracle(precision):
    """ONE whole optimizer step of the headline configuration (3 x 512, batch 8192, 10 layers: 1.12 TFLOP) through
    codae_train_step against the oracle's step on the same inputs - the loop body of script/train_dae_on_embedding.py:198-223
    of the reference at the size BASELINE.json's metric is quoted on (the golden fixtures stop at io 192).

    fp32 engine vs the fp32 oracle: loss, total gradient norm, both metric sums to 1e-5 (measured 8e-7 / 2e-8 / 8e-7 / 7e-7);
    every bias gradient to 1e-3 and four weight-gradient rows per layer (first, last, two inner) to 6e-3 relative L2 (the fp32
    oracle's own distance from the float64 step on these rows is 4.1e-3; the engine is ALSO held to 2.5e-3 of the float64 step) and
    elementwise rtol 1e-3 + 5e-3 of the row's largest entry (a C3 gradient entry is ~4e-7: BASELINE's atol 1e-5 alone would
    pass anything); UPDATED parameters at BASELINE's rtol 1e-3 / atol 1e-5; Adam's first moment to 1e-3 relative L2 per row.
    bf16 engine vs the oracle with its bf16 rounding hook (the same algorithm, rounded where the engine rounds): loss to
    1e-6, gradient norm 1e-4, gradient rows 4e-2 relative L2 (C3_STEP_BOUNDS has the measured values and the why)."""
    import math
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    S, E, B, io, sched, data, blank, params, bm, idx = _c3_problem()
    import bench
    fmask = bm[blank[idx]].astype(np.float32)
    orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], bench.LR, bench.WD, quant=O.bf16_round if precision == "bf16" else None)
    ref = orc.step(data[idx], fmask)
    tr = HipEmbeddingTrainer(sched, torch.from_numpy(data), torch.from_numpy(bm).to(torch.uint8),
                             torch.from_numpy(blank.reshape(-1, 1).astype(np.int32)), bench.LR, bench.WD, bench.CLIP, max_batch=B,
                             precision=precision, device=DEV)
    tr.load_params(params)
    eng = tr.engine
    assert eng.step_path(B) == "layers"
    eng.zero_metric_sums()
    tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=0)
    sq, sqp, gsq, loss = eng.read_scalars()
    tol = {"f32": dict(loss=1e-5, gnorm=1e-5, sums=1e-5, grad=1e-3), "bf16": dict(loss=1e-6, gnorm=1e-4, sums=1e-5, grad=2e-3)}[precision]
    assert abs(loss - float(ref["loss"])) <= tol["loss"] * float(ref["loss"]), (loss, ref["loss"])
    assert abs(math.sqrt(gsq) - float(ref["grad_norm"])) <= tol["gnorm"] * float(ref["grad_norm"]), (math.sqrt(gsq), ref["grad_norm"])
    # (the oracle's metric sums are np.sum over a float32 array, as the reference's: pairwise fp32 over 12.6 M terms)
    assert abs(sq - float(ref["sq_full"])) <= max(tol["sums"], 3e-5) * float(ref["sq_full"]), (sq, ref["sq_full"])
    assert abs(sqp - float(ref["sq_partial"])) <= max(tol["sums"], 3e-5) * float(ref["sq_partial"]), (sqp, ref["sq_partial"])
    rows = [0, 511, 1029, io - 1]
    worst = {"db": 0.0, "dW": 0.0, "dW_elem": 0.0, "m": 0.0, "W": 0.0, "b": 0.0}
    for l, (gw, gb) in enumerate(orc.last_grads):
        got_w = eng.weight_grad(l)[rows].cpu().numpy()
        worst["db"] = max(worst["db"], _rel_l2(eng.bias_grad(l).cpu().numpy(), gb))
        for k, r in enumerate(rows):
            worst["dW"] = max(worst["dW"], _rel_l2(got_w[k], gw[r]))
            # elementwise: |got - ref| - rtol |ref| relative to the row's largest entry (BASELINE's atol 1e-5 alone would pass
            # anything: a C3 gradient entry is ~4e-7)
            excess = np.abs(got_w[k] - gw[r]) - 1e-3 * np.abs(gw[r])
            worst["dW_elem"] = max(worst["dW_elem"], float(excess.max() / np.abs(gw[r]).max()))
            m = eng.adam_m.view(-1)[eng.w_off[l] + r * io: eng.w_off[l] + (r + 1) * io].cpu().numpy()
            worst["m"] = max(worst["m"], _rel_l2(m, orc.adam["m"][l][0][r]))
        w_new, b_new = orc.params[l]
        dw = np.abs(eng.weight(l)[rows].cpu().numpy() - w_new[rows]) - 1e-3 * np.abs(w_new[rows])
        dbb = np.abs(eng.bias(l).cpu().numpy() - b_new) - 1e-3 * np.abs(b_new)
        worst["W"] = max(worst["W"], float(dw.max())); worst["b"] = max(worst["b"], float(dbb.max()))
    if precision == "f32":
        # The oracle is fp32 numpy: on these rows ITS distance from the same step evaluated in float64 is 4.1e-3 (two fp32 summation
        # orders of a 10-layer 1536-wide ReLU stack flip pre-activations that sit within an ulp of zero), the engine's is 1.2e-3
        # on the bf16-plane GEMMs and 3.6e-3 on the fp32-MFMA ones (tools/abl/f32_truth.py).  So next to the oracle bound the engine
        # is held against the float64 step itself: every sampled weight-gradient row within 2.5e-3, every bias gradient within 3e-4.
        x64 = torch.tensor(data[idx], dtype=torch.float64, device=DEV)
        h64 = x64 * torch.tensor(fmask, dtype=torch.float64, device=DEV)
        W64 = [torch.tensor(w, dtype=torch.float64, device=DEV, requires_grad=True) for w, _ in params]
        b64 = [torch.tensor(b, dtype=torch.float64, device=DEV, requires_grad=True) for _, b in params]
        for l, (_, _, relu) in enumerate(sched):
            h64 = h64 @ W64[l].T + b64[l]
            if relu:
                h64 = torch.relu(h64)
        ((h64 - x64) ** 2).mean().backward()
        w64 = max(_rel_l2(eng.weight_grad(l)[rows].cpu().numpy()[k], W64[l].grad[r].cpu().numpy()) for l in range(eng.L) for k, r in enumerate(rows))
        d64 = max(_rel_l2(eng.bias_grad(l).cpu().numpy(), b64[l].grad.cpu().numpy()) for l in range(eng.L))
        o64 = max(_rel_l2(gw[r], W64[l].grad[r].cpu().numpy()) for l, (gw, _) in enumerate(orc.last_grads) for r in rows)
        print("C3 f32 step vs the float64 step: worst sampled dW row %.3g (the fp32 oracle's own: %.3g), worst db %.3g" % (w64, o64, d64))
        # (CODAE_F32_GEMM=native in the environment: the fp32-MFMA GEMMs sit where the fp32 oracle sits, 3.6e-3)
        assert w64 <= (6e-3 if os.environ.get("CODAE_F32_GEMM", "")[:1] == "n" else 2.5e-3) and d64 <= 3e-4, (w64, d64, o64)
    print("C3 %s step vs oracle: loss %.3g gnorm %.3g sq %.3g sqp %.3g rel; worst over layers / sampled rows: %s" % (
        precision, abs(loss - float(ref["loss"])) / float(ref["loss"]), abs(math.sqrt(gsq) - float(ref["grad_norm"])) / float(ref["grad_norm"]),
        abs(sq - float(ref["sq_full"])) / float(ref["sq_full"]), abs(sqp - float(ref["sq_partial"])) / float(ref["sq_partial"]), worst))
    bound = C3_STEP_BOUNDS[precision]
    for k, v in worst.items():
        assert v <= bound[k], (k, v, bound[k], worst)


# worst deviation of the C3-size step from the oracle, per quantity (measured values in the test's docstring / DESIGN.md section 3):
#   db / dW / m: relative L2 of a bias gradient / a sampled weight-gradient row / a row of Adam's first moment;
#   dW_elem: largest (|got - ref| - 1e-3 |ref|) / max|row|;  W / b: largest |got - ref| - 1e-3 |ref| of the UPDATED parameters
#   (BASELINE's atol: 1e-5)
# Measured (round 3): f32 db 7.7e-5, dW 3.5e-3 (1.34e-3 on the fp32-MFMA GEMMs, whose rounding happens to follow numpy's: against
# the float64 step the oracle's own rows are 4.1e-3 off, the engine's 1.2e-3 - the test holds it to that as well; layer 0 / 1:
# two fp32 summation orders of a 10-layer 1536-wide ReLU stack flip the mask of the pre-activations within an ulp of zero),
# dW_elem 2.9e-3, m 3.4e-4, W 0, b 2.4e-6; bf16 (vs the bf16-rounding oracle)
# dW 2.0e-2, dW_elem 3.0e-2, W 7.4e-6, b 1.2e-5 - Adam's FIRST step moves every element by lr = 1e-5 in the direction of its
# gradient's sign, so an element whose gradient is within rounding of zero lands 2 lr apart: bf16 W / b bound = 2.5e-5.
C3_STEP_BOUNDS = {"f32": {"db": 1e-3, "dW": 6e-3, "dW_elem": 6e-3, "m": 2e-3, "W": 1e-5, "b": 1e-5},
                  "bf16": {"db": 3e-2, "dW": 4e-2, "dW_elem": 1e-1, "m": 4e-2, "W": 2.5e-5, "b": 2.5e-5}}


def test_full_size_c3_gradient_is_additive_over_row_shards(monkeypatch):
    """BASELINE config C3 (10 x Linear(1536,1536), batch 8192, bf16) through a size-independent property: with the loss
    scaled by the GLOBAL batch, grad(full batch) = grad(first half) + grad(second half) — what the data-parallel
    sharding relies on.  Rows are independent in the forward, so the only differences are fp32 summation order in the
    weight-gradient GEMM and of the bias partial sums.

    Two backward schedules exist at this shape and both are exercised: the single-GPU step's (data-gradient chain, then every
    weight gradient in ONE grouped launch with unsplit K) and the data-parallel step's (per-layer split-K weight gradients on the
    side stream, issued bucket by bucket without joins: codae_step_backward_async).  Each is bit-reproducible; the bucketed
    no-join backward gives the joined per-layer backward's gradients EXACTLY, bias block included (no atomics); the two schedules
    agree to fp32 summation order."""
    from codae.hip.engine import DaeEngine
    S, E, B, L = 3, 512, 8192, 10
    io = S * E
    g = torch.Generator(device="cpu").manual_seed(21)
    relu = [True] * 4 + [False] + [True] * 4 + [False]
    lim = (6.0 / (2 * io)) ** 0.5
    init = [((torch.rand(io, io, generator=g) * 2 - 1) * lim, torch.zeros(io)) for _ in range(L)]
    data = torch.rand(B + 64, io, generator=g).to(DEV)
    table = torch.ones(S, io, dtype=torch.uint8)
    for s in range(S):
        table[s, s * E:(s + 1) * E] = 0
    table = table.to(DEV)
    mask_id = torch.randint(0, S, (B,), generator=g, dtype=torch.int32).to(DEV)
    rows = torch.randperm(B + 64, generator=g)[:B].to(torch.int32).to(DEV)

    def make_engine():
        eng = DaeEngine([(io, io, r) for r in relu], B, "bf16", DEV)
        eng.load_params(init)
        return eng, eng.hyper(1e-5, 1e-4, clip=1.0, global_rows=B)

    def grads_of(eng, hyper, lo, hi, buckets=None):
        batch = eng.make_batch(data, rows[lo:hi].contiguous(), mask_id[lo:hi].contiguous(), table)
        eng.step_forward_loss(batch, hyper)
        if buckets is None:
            eng.step_backward(hi - lo, 0, L)
        else:
            for blo, bhi in buckets:
                eng.step_backward(hi - lo, blo, bhi, join=False)
            eng.join()
        torch.cuda.synchronize()
        return eng.grads.clone()

    eng, hyper = make_engine()                                # the single-GPU schedule (grouped unsplit weight gradients)
    full = grads_of(eng, hyper, 0, B)
    nw = eng.b_off[0]                                   # weights first, then the bias block
    assert float(full[:nw].abs().max()) > 0 and bool(torch.isfinite(full).all())
    assert torch.equal(full, grads_of(eng, hyper, 0, B))      # the same bits on a second run
    bucketed = grads_of(eng, hyper, 0, B, buckets=[(6, 10), (3, 6), (1, 3), (0, 1)])
    assert torch.equal(bucketed, grads_of(eng, hyper, 0, B, buckets=[(6, 10), (3, 6), (1, 3), (0, 1)]))
    scale = float(full[:nw].abs().max())
    assert float((full[:nw] - bucketed[:nw]).abs().max()) <= 1e-5 * scale + 1e-3 * float((full[:nw] - 0).abs().mean())
    assert torch.equal(full[nw:], bucketed[nw:])              # (bias gradients: the same partial rows in the same order)
    halves = grads_of(eng, hyper, 0, B // 2) + grads_of(eng, hyper, B // 2, B)
    assert float((full[:nw] - halves[:nw]).abs().max()) <= 2e-3 * scale
    assert float((full[nw:] - halves[nw:]).abs().max()) <= 2e-3 * float(full[nw:].abs().max())
    del eng
    monkeypatch.setenv("CODAE_NO_DEFER_WGRAD", "1")           # (read at codae_create) round 2's joined per-layer backward
    eng2, hyper2 = make_engine()
    joined = grads_of(eng2, hyper2, 0, B)
    assert torch.equal(joined, bucketed)


def test_full_size_c5_step_properties():
    """BASELINE config C5 (6 slots x 1024 = io 6144, 10 x Linear(6144,6144), batch 16384, bf16) through size-independent
    properties: (1) the gradient of the full batch equals the sum of the gradients of its two halves when the loss is
    scaled by the global batch (what the 8-GPU sharding of this config rests on); (2) the same inputs give the same
    bits twice; (3) one optimizer step lowers the loss on the batch it was computed on; (4) the bf16 weight shadow and
    its transposed copy, written by the Adam pass, are exactly the rounded fp32 parameters."""
    from codae.hip.engine import DaeEngine
    S, E, B, L = 6, 1024, 16384, 10
    io = S * E
    g = torch.Generator(device="cpu").manual_seed(31)
    relu = [True] * 4 + [False] + [True] * 4 + [False]
    eng = DaeEngine([(io, io, r) for r in relu], B, "bf16", DEV)
    lim = (6.0 / (2 * io)) ** 0.5
    eng.load_params([((torch.rand(io, io, generator=g) * 2 - 1) * lim, torch.zeros(io)) for _ in range(L)])
    data = torch.rand(B + 64, io, generator=g).to(DEV)
    table = torch.ones(S, io, dtype=torch.uint8)
    for s in range(S):
        table[s, s * E:(s + 1) * E] = 0
    table = table.to(DEV)
    mask_id = torch.randint(0, S, (B,), generator=g, dtype=torch.int32).to(DEV)
    rows = torch.randperm(B + 64, generator=g)[:B].to(torch.int32).to(DEV)
    hyper = eng.hyper(1e-4, 1e-4, clip=1.0, global_rows=B)

    def grads_of(lo, hi):
        batch = eng.make_batch(data, rows[lo:hi].contiguous(), mask_id[lo:hi].contiguous(), table)
        eng.step_forward_loss(batch, hyper)
        eng.step_backward(hi - lo, 0, L)
        torch.cuda.synchronize()
        return eng.grads.clone()

    full = grads_of(0, B)
    nw = eng.b_off[0]
    assert float(full[:nw].abs().max()) > 0 and bool(torch.isfinite(full).all())
    assert torch.equal(full, grads_of(0, B))
    halves = grads_of(0, B // 2)
    halves += grads_of(B // 2, B)
    assert float((full[:nw] - halves[:nw]).abs().max()) <= 2e-3 * float(full[:nw].abs().max())
    assert float((full[nw:] - halves[nw:]).abs().max()) <= 2e-3 * float(full[nw:].abs().max())
    del halves
    batch = eng.make_batch(data, rows, mask_id, table)
    eng.train_step(batch, hyper)
    l0 = eng.read_scalars()[3]
    eng.train_step(batch, eng.hyper(1e-4, 1e-4, clip=1.0, global_rows=B))
    l1 = eng.read_scalars()[3]
    assert l1 < l0, (l0, l1)
    w3 = eng.weight(3)
    sh = eng.shadow[eng.w_off[3]:eng.w_off[3] + io * io].view(io, io)
    sht = eng.shadow_t[eng.w_off[3]:eng.w_off[3] + io * io].view(io, io)
    assert torch.equal(sh, w3.bfloat16()) and torch.equal(sht, w3.bfloat16().t())


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_step_is_bitwise_deterministic_run_to_run(precision):
    """(f32: the bf16-plane GEMMs and the grouped weight-gradient launch of the parity engine, gemm_f32x3.hip.)  VERDICT r1 weak #9: round 1's bias-gradient column sums used atomicAdd(float), so two runs of the product on the
    same inputs differed in the last bits.  Now every reduction has a fixed order: 5 unsynchronised steps, twice, from
    the same state -> identical parameters, Adam moments and loss, bit for bit (two-stream backward included)."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    S, E, B = 3, 256, 2048
    io = S * E
    rng = np.random.default_rng(3)
    N = 2 * B
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    order = [torch.tensor(rng.permutation(N)[:B - 37 * s], dtype=torch.int32, device=DEV) for s in range(5)]   # ragged too
    runs = []
    for _ in range(2):
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                 max_batch=B, precision=precision, device=DEV)
        tr.load_params(params)
        for s in range(5):
            tr.train_batch(order[s], run=0)
        runs.append((tr.engine.params.clone(), tr.engine.adam_m.clone(), tr.engine.adam_v.clone(), tr.engine.read_scalars()))
    for a, b in zip(runs[0][:3], runs[1][:3]):
        assert torch.equal(a, b)
    assert runs[0][3] == runs[1][3]


def test_two_stream_step_matches_single_stream_over_many_steps(monkeypatch):
    """Race check for the step's stream choreography (backward on two streams, per-layer dA buffers, alternating slab
    buffers, tail wgrad on the caller's stream, transposed shadow refreshed on the side stream): 40 steps at a size that
    takes the big-tile kernels must track the same run with everything on ONE stream (CODAE_SINGLE_STREAM=1, where no
    ordering can go wrong).  Without float atomics the two are the same arithmetic in the same order: identical bits."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    S, E, B, steps = 3, 256, 4096, 40
    io = S * E
    rng = np.random.default_rng(77)
    N = 2 * B
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    order = [torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV) for _ in range(steps)]
    runs = []
    # (the per-layer backward: what the data-parallel step and CODAE_NO_DEFER_WGRAD=1 run; the single-GPU default - every weight
    #  gradient in one grouped launch - has no second stream at all: tools/soak.py and the determinism tests cover it)
    monkeypatch.setenv("CODAE_NO_DEFER_WGRAD", "1")
    for single in (True, False):
        if single:
            monkeypatch.setenv("CODAE_SINGLE_STREAM", "1")
        else:
            monkeypatch.delenv("CODAE_SINGLE_STREAM", raising=False)
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                 max_batch=B, precision="bf16", device=DEV)
        tr.load_params(params)
        losses = []
        for s in range(steps):
            tr.train_batch(order[s], run=0)          # no host sync inside the loop: the streams run ahead freely
            if s % 8 == 7 or s == steps - 1:
                losses.append(tr.engine.read_scalars()[3])
        runs.append((losses, tr.engine.params.clone()))
    (la, pa), (lb, pb) = runs
    assert la[-1] < la[0]                                    # it trains
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb)


@pytest.mark.parametrize("S,E,B", [(3, 64, 128), (3, 256, 4096)], ids=["small", "bigtile"])
def test_graph_replay_matches_eager_step(S, E, B):
    """codae_train_step_graph (whole step, both backward streams, captured once and replayed with the step count in
    device memory) vs codae_train_step: same parameters after 6 steps up to float-atomics order; fresh row indices go
    through the persistent index buffer, and Adam's bias correction must follow the step count."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(5)
    N = 3 * B
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    order = [torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV) for _ in range(6)]
    out = []
    for graph in (False, True):
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                 max_batch=B, precision="bf16", device=DEV, use_graph=graph)
        tr.load_params(params)
        for s in range(6):
            tr.train_batch(order[s], run=0)
        out.append((tr.engine.params.clone(), tr.engine.read_scalars()))
    (pa, sa), (pb, sb) = out
    assert abs(sa[3] - sb[3]) <= 1e-5 * abs(sa[3]), (sa, sb)          # last loss
    d = (pa - pb).abs()
    assert float(d.mean()) <= 1e-5 and float(d.max()) <= 2 * 1e-3 * 6, (float(d.mean()), float(d.max()))


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_fused_step_with_two_blanked_slots_vs_oracle(precision, step_path):
    """k_max = 2 on the FUSED embedding step (`--nb_missing 2`; data_tool.py:186-226): the mask table then holds every
    1-subset and every 2-subset of the slots, and a sample's mask id for a run may blank one slot or two.  4 slots x 64:
    10 mask rows; each step uses another run.  Loss, grad-norm and BOTH metric sums (the partial one weighs exactly the
    blanked columns) against the oracle on the same batches."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    import random
    S, E, B, N = 4, 64, 300, 900
    io = S * E
    rng = np.random.default_rng(77)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    data = rng.random((N, io), dtype=np.float32)
    params = O.init_params(sched, rng)
    bm, nmr, per_k = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 2)
    assert bm.shape[0] == 10 and per_k == [4, 6] and int((bm == 0).sum(1).max()) == 2 * E
    mtu = O.corrupter_mask_to_use(N, bm.shape[0], random.Random(5))
    lr, wd = 1e-3, 1e-4
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu).to(torch.int32),
                             lr, wd, 1.0, max_batch=B, precision=precision, device=DEV)
    tr.load_params(params)
    orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], lr, wd)
    tol = 1e-3 if precision == "f32" else 2e-2
    two = 0
    sq_ref = sqp_ref = 0.0
    for run in (0, 3, 9):
        idx = rng.permutation(N)[:B]
        _, fmask = O.get_masks(bm, nmr, mtu, 2, idx, run)
        two += int(((fmask == 0).sum(1) == 2 * E).sum())
        ro = orc.step(data[idx], fmask)
        tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=run)
        sq, sqp, gsq, loss = tr.engine.read_scalars()
        assert abs(loss - float(ro["loss"])) <= tol * abs(float(ro["loss"])), (run, loss, ro["loss"])
        assert abs(math.sqrt(gsq) - float(ro["grad_norm"])) <= 5 * tol * float(ro["grad_norm"]), (run, math.sqrt(gsq), ro["grad_norm"])
        sq_ref += float(ro["sq_full"]); sqp_ref += float(ro["sq_partial"])
        assert abs(sq - sq_ref) <= tol * sq_ref and abs(sqp - sqp_ref) <= tol * sqp_ref, (run, sq, sq_ref, sqp, sqp_ref)
    assert two > B                          # (6 of the 10 mask rows blank two slots)


@pytest.mark.parametrize("S,E,B", [(3, 256, 4096), (3, 64, 128), (3, 512, 1024), (3, 512, 128), (3, 512, 8192)],
                         ids=["64x64-768wg", "64x64-3wg", "64x64-192wg", "stock-batch", "c3-pipelined"])
def test_training_forward_and_fused_loss_repeat_bit_for_bit(S, E, B):
    """The training forward (loss fused into the last GEMM) launched 60 times on the same batch: the loss, both metric sums,
    the dY workspace and every partial-sum row must come out identical every time.  (The 64 x 64 instantiation of the
    fused-loss kernel lost a thread's worth of one workgroup's sum (x-y)^2 in ~10 % of the launches until the compiler was
    kept from packing that accumulation: gemm_bf16.hip, tools/abl/loss_repeat.py.)"""
    import ctypes as C
    from codae import hip
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(B)
    N = 2 * B
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    idx = torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV)
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                             max_batch=B, precision="bf16", device=DEV)
    tr.load_params(params)
    eng = tr.engine
    eng.bias_parts.zero_()
    batch = tr._batch(idx, 0)
    hyper = eng.hyper(1e-3, 1e-4, 1.0, global_rows=B)
    seen = set()
    for i in range(60):
        eng.zero_metric_sums()
        hip.check(hip.lib().codae_step_forward_loss(eng._h, C.byref(eng.bufs), C.byref(batch), C.byref(hyper), None, hip.current_stream()))
        sq, sqp, _, loss = eng.read_scalars()
        seen.add((sq, sqp, loss, int(eng.dacts.view(torch.int16).to(torch.int64).sum()), float(eng.bias_parts.double().sum())))
    assert len(seen) == 1, sorted(seen)[:4]


@pytest.mark.parametrize("S,E,B", [(3, 512, 3000), (3, 256, 8192), (3, 512, 128)], ids=["mid-128x192", "mid-narrow", "small-64x64"])
def test_small_and_mid_tiles_give_the_bits_of_the_128x128_kernel(monkeypatch, S, E, B):
    """Three fused steps with the shape-dependent small-launch kernels (64 x 64 tiles with a 4-stage ring, the pipelined
    kernel on 128 x 192 tiles, unsplit weight gradients) against the same steps with all of that off
    (CODAE_NO_DEEP_SMALL=1: everything on the 128 x 128 one-barrier kernel).  Every kernel accumulates a tile's k range in
    the same order, so activations, activation gradients and weight gradients must be bit-identical; bias gradients and
    the metric sums differ only by how many rows a partial sum spans."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(B + E)
    N = B + 64
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    idx = torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV)
    outs = []
    for new in (True, False):
        if new:
            monkeypatch.delenv("CODAE_NO_DEEP_SMALL", raising=False)
        else:
            monkeypatch.setenv("CODAE_NO_DEEP_SMALL", "1")
        monkeypatch.setenv("CODAE_WGRAD_SPLITK", "1" if B <= 256 else "5")      # the same K split on both sides
        monkeypatch.setenv("CODAE_NO_DEFER_WGRAD", "1")                         # (per-layer weight gradients on both sides)
        monkeypatch.setenv("CODAE_NO_RELU_BITS", "1")      # (the workspace is compared whole: no 1-bit mask region that only one side writes)
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 100.0,
                                 max_batch=B, precision="bf16", device=DEV)
        tr.load_params(params)
        assert tr.engine.step_path(B) == "layers"
        tr.train_batch(idx, run=0)
        eng = tr.engine
        nw = eng.b_off[0]
        outs.append((eng.acts.clone(), eng.dacts.clone(), eng.grads[:nw].clone(), eng.grads[nw:].clone(), eng.read_scalars()))
    (aa, da, ga, ba, sa), (ab, db, gb, bb, sb) = outs
    assert torch.equal(da, db), "activation gradients"
    assert torch.equal(ga, gb), "weight gradients"
    assert torch.equal(aa, ab), "saved activations"
    assert float((ba - bb).abs().max()) <= 1e-5 * float(bb.abs().max())
    assert abs(sa[3] - sb[3]) <= 1e-6 * abs(sb[3])


@pytest.mark.parametrize("S,E,B", [(3, 512, 3000), (3, 320, 8192), (3, 24, 1500)], ids=["mid-128x192", "big-256x192", "narrow-padded"])
def test_one_bit_relu_mask_gives_the_bits_of_the_activation_mask(monkeypatch, S, E, B):
    """The data gradient's ReLU mask as one bit per element, written by the forward epilogue (round 3), against the mask taken
    from the saved activation (CODAE_NO_RELU_BITS=1): two fused steps - activation gradients, weight and bias gradients,
    updated parameters bit-identical (the bit IS `activation > 0` of the stored bf16 value)."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(B + E)
    N = B + 64
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 3, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
    idx = torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV)
    outs = []
    for bits in (True, False):
        if bits:
            monkeypatch.delenv("CODAE_NO_RELU_BITS", raising=False)
        else:
            monkeypatch.setenv("CODAE_NO_RELU_BITS", "1")
        monkeypatch.setenv("CODAE_NO_CHAIN", "1")
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                 max_batch=B, precision="bf16", device=DEV)
        tr.load_params(params)
        for _ in range(2):
            tr.train_batch(idx, run=0)
        eng = tr.engine
        outs.append((eng.dacts.clone(), eng.grads.clone(), eng.params.clone(), eng.read_scalars()))
    (da, ga, pa, sa), (db, gb, pb, sb) = outs
    assert float(ga.abs().max()) > 0
    assert torch.equal(da, db), "activation gradients"
    assert torch.equal(ga, gb), "gradients"
    assert torch.equal(pa, pb), "parameters"
    assert sa == sb


@pytest.mark.parametrize("B", [128, 500])
def test_exact_fp32_small_batch_split_k_matches_oracle_and_unsplit(monkeypatch, B):
    """Exact-fp32 engine at the reference's stock batch size on a wide stack (3 x 512): forward and data-gradient launches
    split K over workgroups and finish in a reduce that applies bias / ReLU / ReLU mask / column sums.  Two steps against
    the fp32 oracle at the parity tolerance, and against the same engine with the split off (CODAE_NO_DEEP_SMALL=1):
    identical arithmetic up to fp32 summation order."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    S, E = 3, 512
    io = S * E
    rng = np.random.default_rng(B)
    N = 2 * B
    data = rng.random((N, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
    params = O.init_params(sched, rng)
    bm, nmr, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = np.stack([rng.permutation(S) for _ in range(N)])
    lr, wd = 1e-3, 1e-4
    order = [rng.permutation(N)[:B] for _ in range(2)]
    outs = []
    for split in (True, False):
        if split:
            monkeypatch.delenv("CODAE_NO_DEEP_SMALL", raising=False)
        else:
            monkeypatch.setenv("CODAE_NO_DEEP_SMALL", "1")
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu).to(torch.int32),
                                 lr, wd, 1.0, max_batch=B, precision="f32", device=DEV)
        tr.load_params(params)
        orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], lr, wd)
        first = None
        for idx in order:
            _, fmask = O.get_masks(bm, nmr, mtu, 1, idx, 0)
            ro = orc.step(data[idx], fmask)
            tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=0)
            sq, sqp, gsq, loss = tr.engine.read_scalars()
            assert abs(loss - float(ro["loss"])) <= 1e-3 * abs(float(ro["loss"])), (split, loss, ro["loss"])
            assert abs(math.sqrt(gsq) - float(ro["grad_norm"])) <= 5e-3 * float(ro["grad_norm"]), (split, math.sqrt(gsq), ro["grad_norm"])
            if first is None:          # (after an update the two runs' parameters differ by Adam's +-lr flips of near-zero gradients)
                errs = [_rel_l2(tr.engine.weight_grad(l).cpu().numpy(), gw) for l, (gw, gb) in enumerate(orc.last_grads)]
                assert max(errs) <= 1e-3, (split, errs)                 # every layer's weight gradient against the oracle's (parity tolerance)
                first = (tr.engine.grads.clone(), tr.engine.acts.clone(), max(errs))
        outs.append(first)
    (ga, aa, ea), (gb, ab, eb) = outs
    # another summation order moves a pre-activation by ~1e-7; where one sits that close to zero its ReLU (and the mask of the
    # data gradient) may flip, so single gradient entries may differ by more than rounding: the split run must be as close
    # to the oracle as the unsplit one (both asserted above), and the two close to each other in the bulk
    assert ea <= 2 * eb + 1e-6
    d = (ga - gb).abs()
    assert float(d.max()) <= 5e-3 * float(gb.abs().max()) and float(d.mean()) <= 1e-3 * float(gb.abs().mean())
    assert float((aa.view(torch.float32) - ab.view(torch.float32)).abs().max()) <= 1e-4


def _fuzz_cases():
    rng = np.random.default_rng(2024)
    cases = []
    for _ in range(8):
        S = int(rng.integers(3, 6))            # (k_max = 1 needs at least 3 slots: Corrupter's own check)
        E = int(rng.choice([64, 128, 192, 320]))
        io = S * E
        z = int(rng.choice([io, max(64, (io // 2) // 64 * 64), 64]))
        B = int(rng.integers(65, 2600))
        cases.append((S, E, z, int(rng.integers(2, 5)), int(rng.integers(2, 5)), B))   # (1 layer per side: the reference itself raises)
    # widths that are multiples of 8 but not of 64 (the reference only asks io % embedding_size == 0): bf16 engine on padded row
    # strides; and one that is not a multiple of 8 (exact-fp32 engine)
    cases += [(3, 16, 48, 2, 2, 200), (3, 24, 40, 3, 2, 777), (5, 40, 104, 2, 3, 1500), (4, 72, 136, 4, 4, 2200), (3, 168, 504, 2, 2, 1111),
              (3, 20, 30, 2, 2, 300)]
    return cases


@pytest.mark.parametrize("S,E,z,nb_in,nb_out,B", _fuzz_cases())
def test_fused_random_topologies_vs_oracle(S, E, z, nb_in, nb_out, B, step_path):
    """Seeded random stacks (tapers to z, 2-4 hidden layers per side, ragged batches, every tile / split-K choice
    the dispatcher makes for them): two fused steps against the fp32 oracle - bf16 kernels when every width is a multiple
    of 8 (loss within 2 %, grad-norm 10 %), exact-fp32 kernels otherwise (1e-3 / 5e-3)."""
    from codae.train import HipEmbeddingTrainer
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(S * 1000 + E + B)
    sched = O.layer_schedule(io, z, nb_in, nb_out, False, "embedding")
    # widths that are not multiples of 8 (16-byte bf16 rows) exist only in the exact-fp32 mode (any shape): tighter tolerance there
    precision = "f32" if any(k % 8 or n % 8 for k, n, _ in sched) else "bf16"
    tol = 1e-3 if precision == "f32" else 2e-2
    N = 2 * B
    data = rng.random((N, io), dtype=np.float32)
    params = O.init_params(sched, rng)
    arch = [{"size": E, "position": s * E} for s in range(S)]
    bm, nmr, _ = O.corrupter_tables(arch, 1)
    mtu = np.stack([rng.permutation(S) for _ in range(N)])
    lr, wd = 1e-3, 1e-4
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu).to(torch.int32),
                             lr, wd, 1.0, max_batch=B, precision=precision, device=DEV)
    tr.load_params(params)
    orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], lr, wd)
    for s in range(2):
        idx = rng.permutation(N)[:B]
        _, fmask = O.get_masks(bm, nmr, mtu, 1, idx, 0)
        ro = orc.step(data[idx], fmask)
        tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=DEV), run=0)
        sq, sqp, gsq, loss = tr.engine.read_scalars()
        assert abs(loss - float(ro["loss"])) <= tol * abs(float(ro["loss"])), (precision, s, loss, ro["loss"])
        assert abs(math.sqrt(gsq) - float(ro["grad_norm"])) <= 5 * tol * float(ro["grad_norm"]), (precision, s, math.sqrt(gsq), ro["grad_norm"])
