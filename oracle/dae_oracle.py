"""CPU oracle for the CODAE denoising-autoencoder training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package
(`mui-deepautoencoder_amd/`) may import, call, link or execute this module.
It is used by `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py`, always as the checker / baseline, never as the product path.

What it is: a plain numpy fp32 restatement of the algorithm of the reference
(victordeleau/MUI-DeepAutoEncoder, mounted read-only at /root/reference), one
function per row of SURVEY.md section 8(a).  Every function cites the reference
file:line it follows.  The arithmetic the reference delegates to PyTorch
(nn.Linear, ReLU, MSELoss, NLLLoss, log_softmax, clip_grad_norm_, optim.Adam;
pinned pytorch=1.5.0 in env.yml:60, 2.10.0 in this image) is restated from the
published formulas of those ops.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself run in the build container: `tests/golden/make_golden.py` drives the two
reference training scripts end to end on seeded synthetic inputs and stores what
they produced in `tests/golden/*.npz`; `tests/test_oracle_golden.py` replays
those runs through this module and compares.

Summation order: all reductions are numpy fp32 (pairwise) unless a function
says float64; matmul is numpy's (BLAS sgemm).  Parity is therefore to
tolerance (rtol 1e-3 / atol 1e-5, BASELINE.json north_star), not bitwise.

`quant`: forward / backward / EmbeddingTrainer take an optional rounding hook.  With quant=None (the default, and
what every golden replay uses) they are the reference's fp32 math.  With quant=bf16_round they restate WHERE the
bf16 (throughput) engine rounds - corrupted input, weight shadow, every stored activation and activation gradient
to bf16, every accumulation in fp32, master weights and Adam in fp32 - so the HIP kernels can be pinned tightly to
"the reference's algorithm evaluated in that arithmetic", separately from how far that arithmetic itself sits from
the fp32 reference run (tests/test_gpu_parity.py).
"""

import itertools
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# a1: layer-size schedule and initialisation
# --------------------------------------------------------------------------

def layer_schedule(io_size, z_size, nb_input_layer, nb_output_layer,
                   steep_layer_size, kind):
    """List of (in_features, out_features, relu_after) for every Linear.

    kind == "embedding": codae/model/embedding_denoising_autoencoder.py:49-126
        (increment = floor(delta / nb_layer), :52-54; last decoder Linear takes
        the previous layer's width, :126).
    kind == "mixed": codae/model/mixed_variable_denoising_autoencoder.py:45-122
        (increment = ceil(...), :48-50; last decoder Linear is io->io, :122).

    The flag name is inverted in the reference: increments are non-zero only
    when steep_layer_size is False (embedding_...py:51).
    """
    assert kind in ("embedding", "mixed")
    rnd = math.floor if kind == "embedding" else math.ceil
    inc_in = inc_out = 0
    if not steep_layer_size:
        delta = io_size - z_size
        inc_in = rnd(delta / nb_input_layer)
        inc_out = rnd(delta / nb_output_layer)

    layers = [(io_size, io_size, True)]
    last_out = None
    for i in range(1, nb_input_layer):
        if steep_layer_size:
            layers.append((io_size, io_size, True))
        else:
            a = max(io_size - (i - 1) * inc_in, z_size)
            b = max(io_size - i * inc_in, z_size)
            layers.append((a, b, True))
            last_out = b
    if steep_layer_size:
        layers.append((io_size, z_size, False))
    else:
        if last_out is None:
            # embedding_...py:89 reads a name that was never bound
            raise UnboundLocalError("next_layer_output_size")
        layers.append((last_out, z_size, False))

    last_out = None
    for i in range(nb_output_layer):
        if steep_layer_size:
            layers.append((z_size if i == 0 else io_size, io_size, True))
        else:
            a = min(z_size + i * inc_out, io_size)
            b = min(z_size + (i + 1) * inc_out, io_size)
            layers.append((a, b, True))
            last_out = b
    if kind == "embedding":
        if last_out is None:
            # embedding_...py:126 with steep_layer_size=True
            raise UnboundLocalError("next_layer_output_size")
        layers.append((last_out, io_size, False))
    else:
        layers.append((io_size, io_size, False))
    return layers


def xavier_bound(fan_in, fan_out):
    """torch.nn.init.xavier_uniform_ bound, gain 1 (embedding_...py:197)."""
    return math.sqrt(6.0 / (fan_in + fan_out))


def init_params(schedule, rng):
    """Xavier-uniform weights [out,in], zero bias (embedding_...py:188-211).

    Uses a numpy Generator; the reference draws from torch's global RNG, so
    parity tests feed captured initial weights instead of calling this.
    """
    params = []
    for (k, n, _) in schedule:
        a = xavier_bound(k, n)
        w = rng.uniform(-a, a, size=(n, k)).astype(F32)
        params.append((w, np.zeros(n, dtype=F32)))
    return params


# --------------------------------------------------------------------------
# a9: Corrupter (structured whole-variable blanking masks)
# --------------------------------------------------------------------------

def corrupter_tables(arch, k_max):
    """binary_masks[nb_run, io], nb_missing_per_run[nb_run], nb_corruption_per_k.

    codae/tool/data_tool.py:186-219.  Row order: all 1-subsets in
    itertools.combinations order, then all 2-subsets, ...
    """
    n_var = len(arch)
    if (k_max < 0) | (k_max > n_var - 1):          # data_tool.py:184
        raise Exception("Invalid k_max number. k_max > 0 && k_max < nb_predictor - 1")
    io_size = sum(v["size"] for v in arch)
    rows, per_run, per_k = [], [], []
    for k in range(1, k_max + 1):
        subsets = list(itertools.combinations(range(n_var), k))
        per_k.append(len(subsets))
        for subset in subsets:
            m = np.ones(io_size, dtype=F32)
            for idx in subset:
                p, s = arch[idx]["position"], arch[idx]["size"]
                m[p:p + s] = 0
            rows.append(m)
            per_run.append(k)
    return np.stack(rows), np.asarray(per_run, dtype=np.int64), per_k


def corrupter_mask_to_use(nb_observation, nb_run, pyrandom):
    """Per-observation permutation of mask ids (data_tool.py:222-226).

    `pyrandom` is a `random.Random` (or the `random` module): the reference
    calls random.sample(list(range(nb_run)), nb_run) once per observation.
    """
    ids = list(range(nb_run))
    return np.asarray([pyrandom.sample(ids, nb_run) for _ in range(nb_observation)],
                      dtype=np.int64)


def get_masks(binary_masks, nb_missing_per_run, mask_to_use, k_max, batch_indices, run):
    """(masks[k_max] each [B,io], fmask [B,io]) — data_tool.py:239-262."""
    ids = mask_to_use[np.asarray(batch_indices, dtype=np.int64), run]
    fm = binary_masks[ids]
    ks = nb_missing_per_run[ids]
    masks = [fm * (ks == k + 1)[:, None].astype(F32) for k in range(k_max)]
    fmask = masks[0].copy()
    for m in masks[1:]:
        fmask = fmask + m
    return masks, fmask


# --------------------------------------------------------------------------
# a2/a3: corrupt + forward
# --------------------------------------------------------------------------

def corrupt(x, mask):
    """input.clone() * mask (embedding_...py:226-239, mixed_...py:247-262)."""
    return (x * mask).astype(F32)


def bf16_round(a):
    """fp32 -> bf16 (round to nearest even) -> fp32: the value a bf16 store keeps."""
    u = np.ascontiguousarray(a, dtype=F32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(F32)


def forward(params, relu_flags, x, keep=False, quant=None):
    """y = Linear/ReLU chain (embedding_...py:137-185): h = relu?(h W^T + b).
    quant: operands (input, weights, every hidden activation) pass through it; the last layer's y stays fp32."""
    h = x.astype(F32)
    if quant is not None:
        h = quant(h)
    acts = [h]
    last = len(params) - 1
    for l, ((w, b), relu) in enumerate(zip(params, relu_flags)):
        h = h @ (w if quant is None else quant(w)).T + b
        if relu:
            h = np.maximum(h, 0)
        h = h.astype(F32)
        if quant is not None and l != last:
            h = quant(h)
        acts.append(h)
    return (h, acts) if keep else h


# --------------------------------------------------------------------------
# a4: MSE loss (embedding script) and its gradient
# --------------------------------------------------------------------------

def mse_mean(x, y):
    """torch.nn.MSELoss('mean')(input=x, target=y): unmasked mean over all
    B*io elements (script/train_dae_on_embedding.py:173,206)."""
    d = (x - y).astype(F32)
    return F32(np.mean(d * d, dtype=F32))


def mse_mean_grad_y(x, y):
    """dL/dy of mse_mean: 2 (y - x) / (B * io)."""
    return ((y - x) * F32(2.0 / x.size)).astype(F32)


# --------------------------------------------------------------------------
# a5: CombinedCriterion (abalone training loss + monitor)
# --------------------------------------------------------------------------

def _log_softmax(z):
    m = z.max(axis=1, keepdims=True)
    e = z - m
    return (e - np.log(np.exp(e).sum(axis=1, keepdims=True))).astype(F32)


def combined_mean(arch, weight, x, y):
    """CombinedCriterion(reduction='mean').__call__ (metering.py:155-180).

    regression: w_i * sqrt(mean (x-y)^2) over the variable's [B,size] slice;
    classification: w_i * mean_B NLL(log_softmax(y_slice), argmax(x_slice));
    summed and divided by len(arch).
    """
    total = F32(0)
    for i, v in enumerate(arch):
        p, s = v["position"], v["size"]
        xs, ys = x[:, p:p + s], y[:, p:p + s]
        if v["type"] == "regression":
            li = np.sqrt(np.mean((xs - ys) ** 2, dtype=F32))
        else:
            t = xs.argmax(axis=1)
            lsm = _log_softmax(ys)
            li = -np.mean(lsm[np.arange(len(t)), t], dtype=F32)
        total = total + F32(li) * F32(weight[i])
    return F32(total / len(arch))


def combined_mean_grad_y(arch, weight, x, y):
    """Gradient of combined_mean w.r.t. y (autograd of metering.py:155-180)."""
    g = np.zeros_like(y, dtype=F32)
    B = x.shape[0]
    n_var = len(arch)
    for i, v in enumerate(arch):
        p, s = v["position"], v["size"]
        xs, ys = x[:, p:p + s], y[:, p:p + s]
        c = F32(weight[i]) / F32(n_var)
        if v["type"] == "regression":
            rmse = np.sqrt(np.mean((xs - ys) ** 2, dtype=F32))
            g[:, p:p + s] = c * (ys - xs) / (F32(B * s) * rmse)
        else:
            t = xs.argmax(axis=1)
            sm = np.exp(_log_softmax(ys))
            sm[np.arange(B), t] -= 1
            g[:, p:p + s] = c * sm / F32(B)
    return g.astype(F32)


def combined_full(arch, x, y):
    """CombinedCriterion(reduction='none') (metering.py:131-152): [B, n_var];
    regression -> squared error (size-1 variables), classification -> NLL."""
    out = np.zeros((x.shape[0], len(arch)), dtype=F32)
    for i, v in enumerate(arch):
        p, s = v["position"], v["size"]
        xs, ys = x[:, p:p + s], y[:, p:p + s]
        if v["type"] == "regression":
            out[:, i:i + 1] = (xs - ys) ** 2
        else:
            t = xs.argmax(axis=1)
            out[:, i] = -_log_softmax(ys)[np.arange(len(t)), t]
    return out


def mask_transformation(observation_mask, n_loss):
    """get_mask_transformation (data_tool.py:16-43): io -> variable 0/1 matrix;
    a run of zeros (one-hot block) contributes a single 1 at its first column."""
    T = np.zeros((len(observation_mask), n_loss), dtype=F32)
    b, c = True, 0
    for i in range(len(observation_mask)):
        if observation_mask[i] == 1:
            b = True
            T[i, c] = 1
            c += 1
        elif b:
            b = False
            T[i, c] = 1
            c += 1
    return T


def get_per_k(loss, masks, T):
    """CombinedCriterion.get_per_k (metering.py:187-197): float64 [k_max, n_var]."""
    io = T.shape[0]
    out = np.zeros((len(masks), T.shape[1]))
    for i, m in enumerate(masks):
        mm = np.matmul(m, np.ones((io, io)))
        mm[mm > 1] = 1
        out[i, :] = np.sum(np.matmul(mm, T) * loss, axis=0)
    return out


def get_partial(loss, fmask, T):
    """CombinedCriterion.get_partial (metering.py:200-204)."""
    return (1 - np.matmul(fmask, T)) * loss


def normalizer_undo(data, scale, dmin):
    """Normalizer.undo (data_tool.py:80-90)."""
    return (data * scale + dmin).astype(F32)


# --------------------------------------------------------------------------
# a6: backward of the Linear/ReLU chain
# --------------------------------------------------------------------------

def backward(params, relu_flags, acts, dy, quant=None):
    """Gradients [(dW, db)] of the chain (autograd of embedding_...py:137-185).

    acts[l] is the input of layer l, acts[l+1] its (post-ReLU) output.
    dA = dH * [h>0] uses the post-activation (ReLU(inplace=True), :64).
    quant: every activation gradient is stored through it (the last layer's bias gradient sums the unrounded dy).
    """
    grads = [None] * len(params)
    d = dy.astype(F32)
    db_top = d.sum(axis=0, dtype=F32)
    if quant is not None:
        d = quant(d)
    for l in range(len(params) - 1, -1, -1):
        w, _ = params[l]
        if relu_flags[l]:
            d = (d * (acts[l + 1] > 0)).astype(F32)
        db = db_top if (quant is not None and l == len(params) - 1) else d.sum(axis=0, dtype=F32)
        grads[l] = ((d.T @ acts[l]).astype(F32), db)
        if l > 0:
            d = (d @ (w if quant is None else quant(w))).astype(F32)
            if quant is not None:
                d = quant(d)
    return grads


# --------------------------------------------------------------------------
# a7/a8: clip_grad_norm_ + Adam
# --------------------------------------------------------------------------

def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_(params, 1) (train_dae_on_embedding.py:213):
    total = ||(||g_i||)||_2 ; g *= min(1, max_norm / (total + 1e-6)).
    Returns (clipped grads, total_norm)."""
    sq = F32(0)
    for gw, gb in grads:
        sq = sq + F32(np.sum(gw.astype(F32) ** 2, dtype=F32)) + F32(np.sum(gb ** 2, dtype=F32))
    total = F32(np.sqrt(sq))
    coef = F32(min(1.0, float(max_norm) / (float(total) + 1e-6)))
    return [((gw * coef).astype(F32), (gb * coef).astype(F32)) for gw, gb in grads], total


def adam_init(params):
    return {"t": 0,
            "m": [(np.zeros_like(w), np.zeros_like(b)) for w, b in params],
            "v": [(np.zeros_like(w), np.zeros_like(b)) for w, b in params]}


def adam_step(params, grads, state, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam.step, amsgrad off, L2-coupled decay
    (train_dae_on_embedding.py:160-163,215).  Returns new params; state in place."""
    state["t"] += 1
    t = state["t"]
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    step_size = F32(lr / bc1)
    inv_sqrt_bc2 = F32(1.0 / math.sqrt(bc2))
    new = []
    for l, ((w, b), (gw, gb)) in enumerate(zip(params, grads)):
        outs = []
        ml, vl = list(state["m"][l]), list(state["v"][l])
        for j, (p, g) in enumerate(((w, gw), (b, gb))):
            g = (g + F32(weight_decay) * p).astype(F32)
            ml[j] = (F32(beta1) * ml[j] + F32(1 - beta1) * g).astype(F32)
            vl[j] = (F32(beta2) * vl[j] + F32(1 - beta2) * g * g).astype(F32)
            denom = np.sqrt(vl[j]) * inv_sqrt_bc2 + F32(eps)
            outs.append((p - step_size * ml[j] / denom).astype(F32))
        state["m"][l], state["v"][l] = tuple(ml), tuple(vl)
        new.append(tuple(outs))
    return new


# --------------------------------------------------------------------------
# datasets (a10 input layout)
# --------------------------------------------------------------------------

def concatenated_embedding_dataset(embeddings, used_category):
    """ConcatenatedEmbeddingDataset.__init__
    (codae/dataset/concatenated_embedding_dataset.py:28-101).
    Returns dict(data[N,S*E] scaled by (max-min) with no min shift,
    data_per_category list of raw [N,E], arch, index)."""
    index = [k for k, v in embeddings.items() if all(c in v for c in used_category)]
    E = len(embeddings[index[0]][used_category[0]])
    per_cat = [np.asarray([embeddings[i][c] for i in index], dtype=F32) for c in used_category]
    data = np.concatenate(per_cat, axis=1).astype(F32)
    scale = F32(data.max() - data.min()).item()
    data = (data / F32(scale)).astype(F32)
    arch, pos = [], 0
    for name in used_category:
        arch.append({"name": name, "lambda": 1, "size": E, "type": "regression", "position": pos})
        pos += E
    return {"data": data, "data_per_category": per_cat, "arch": arch, "index": index,
            "embedding_size": E, "nb_used_category": len(used_category),
            "nb_predictor": E * len(used_category), "scale": scale}


def mixed_variable_dataset(columns, names, is_numeric):
    """MixedVariableDataset.__init__ (codae/dataset/mixed_variable_dataset.py:21-86).

    columns: list of 1-D arrays/lists (one per dataframe column), names,
    is_numeric[i]: dtype float64/int64 -> one regression column, else one-hot
    in first-seen label order."""
    N = len(columns[0])
    arch, pos = [], 0
    for name, col, num in zip(names, columns, is_numeric):
        if num:
            size, typ = 1, "regression"
        else:
            size, typ = len(set(col)), "classification"
        arch.append({"name": name, "lambda": 1, "size": size, "type": typ, "position": pos})
        pos += size
    data = np.zeros((N, pos), dtype=np.float64)
    type_mask = np.zeros(pos, dtype=F32)
    for v, col, num in zip(arch, columns, is_numeric):
        p, s = v["position"], v["size"]
        if num:
            data[:, p] = np.asarray(col, dtype=np.float64)
            type_mask[p:p + s] = 1
        else:
            seen = {}
            for i, lab in enumerate(col):
                if lab not in seen:
                    seen[lab] = len(seen)
                data[i, p + seen[lab]] = 1
    return {"data": data.astype(F32), "arch": arch, "type_mask": type_mask,
            "io_size": pos, "nb_predictor": len(columns)}


# --------------------------------------------------------------------------
# f1: RankingLoss (validation metric)
# --------------------------------------------------------------------------

def ranking_loss(prediction, fmask, indices, data_per_category, embedding_size,
                 validation_indices):
    """RankingLoss.get (codae/tool/metering.py:46-79), k=1 only."""
    S = len(data_per_category)
    getter = np.zeros(S * embedding_size, dtype=F32)
    for s in range(S):
        getter[s * embedding_size] = s
    val = np.asarray(validation_indices, dtype=np.int64)
    total = 0.0
    for i, idx in enumerate(indices):
        c = int(np.dot((1 - fmask[i]).astype(F32), getter))
        q = prediction[i, c * embedding_size:(c + 1) * embedding_size].astype(F32)
        d = data_per_category[c]
        num = (d * q[None, :]).sum(axis=1, dtype=F32)
        den = np.maximum(np.sqrt((d * d).sum(axis=1, dtype=F32)), F32(1e-8)) * \
            max(F32(np.sqrt((q * q).sum(dtype=F32))), F32(1e-8))
        s = (num / den).astype(F32)
        rank = int(np.sum(s[idx] > s[val]))
        total += 1 - (rank / (len(val) - 1))
    return total


# --------------------------------------------------------------------------
# the two hot loops, replayed from captured batch orders
# --------------------------------------------------------------------------

class EmbeddingTrainer:
    """State + one step of the inner loop of script/train_dae_on_embedding.py:194-223."""

    def __init__(self, params, relu_flags, lr, weight_decay, clip=True, quant=None):
        self.params = [(w.astype(F32).copy(), b.astype(F32).copy()) for w, b in params]
        self.relu = list(relu_flags)
        self.lr, self.wd, self.clip = lr, weight_decay, clip
        self.adam = adam_init(self.params)
        self.quant = quant
        self.last_grads = None

    def step(self, x, fmask):
        """Returns dict(loss, grad_norm, sq_full, sq_partial, y)."""
        c = corrupt(x, fmask)                                     # :200
        y, acts = forward(self.params, self.relu, c, keep=True, quant=self.quant)   # :203
        loss = mse_mean(x, y)                                     # :206
        grads = backward(self.params, self.relu, acts, mse_mean_grad_y(x, y), quant=self.quant)  # :210
        self.last_grads = grads
        gnorm = None
        if self.clip:
            grads, gnorm = clip_grad_norm(grads, 1.0)             # :213
        self.params = adam_step(self.params, grads, self.adam, self.lr, self.wd)  # :215
        se = ((x - y) ** 2).astype(F32)                           # :218
        return {"loss": loss, "grad_norm": gnorm, "y": y,
                "sq_full": F32(np.sum(se)),                       # :220
                "sq_partial": F32(np.sum((1 - fmask) * se))}      # :223

    def evaluate(self, x, fmask):
        """Validation body (:245-258), no parameter update."""
        y = forward(self.params, self.relu, corrupt(x, fmask), quant=self.quant)
        se = ((x - y) ** 2).astype(F32)
        return {"y": y, "sq_full": F32(np.sum(se)), "sq_partial": F32(np.sum((1 - fmask) * se))}


class MixedTrainer:
    """One step of the inner loop of script/train_dae_on_abalone.py:202-236."""

    def __init__(self, params, relu_flags, arch, weight, lr, weight_decay, clip=True):
        self.params = [(w.astype(F32).copy(), b.astype(F32).copy()) for w, b in params]
        self.relu = list(relu_flags)
        self.arch, self.weight = arch, weight
        self.lr, self.wd, self.clip = lr, weight_decay, clip
        self.adam = adam_init(self.params)

    def step(self, x, fmask):
        c = corrupt(x, fmask)                                     # :209
        y, acts = forward(self.params, self.relu, c, keep=True)   # :212
        loss = combined_mean(self.arch, self.weight, x, y)        # :215
        dy = combined_mean_grad_y(self.arch, self.weight, x, y)
        grads = backward(self.params, self.relu, acts, dy)        # :219
        gnorm = None
        if self.clip:
            grads, gnorm = clip_grad_norm(grads, 1.0)             # :222
        self.params = adam_step(self.params, grads, self.adam, self.lr, self.wd)  # :224
        return {"loss": loss, "grad_norm": gnorm, "y": y}

    def evaluate(self, x, fmask):
        return {"y": forward(self.params, self.relu, corrupt(x, fmask))}
