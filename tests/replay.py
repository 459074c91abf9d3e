"""Replay the two reference training scripts from a golden fixture's captured
inputs (initial weights, mask tables, exact batch order) through any trainer
object exposing step(x, fmask) / evaluate(x, fmask).  The epoch bookkeeping
below restates script/train_dae_on_embedding.py:187-275 and
script/train_dae_on_abalone.py:189-332 with the metric arithmetic taken from
the oracle (test infrastructure)."""
import math

import numpy as np

from oracle import dae_oracle as O


def replay_embedding(g, trainer):
    m = g.meta
    data = g["data"]
    bm, mtu, nmr = g["binary_masks"], g["mask_to_use"], g["nb_missing_per_run"]
    per_cat = list(g["data_per_category"])
    val_idx = g["validation_indices"]
    nb_train, N = m["nb_train"], m["N"]
    nb_val = N - nb_train
    nb_pred, S, E = m["S"] * m["E"], m["S"], m["E"]
    nt = math.ceil(nb_train / m["batch"])
    nv = math.ceil(nb_val / m["batch"])
    calls = g.calls()
    assert len(calls) == m["epochs"] * (nt + nv)
    book = {k: [] for k in ("ftl", "ptl", "fvl", "pvl", "rl")}
    steps = {"loss": [], "grad_norm": []}
    first = None
    c = 0
    for _ in range(m["epochs"]):
        ftl = ptl = np.float32(0)
        for _ in range(nt):
            idx, run = calls[c]; c += 1
            _, fmask = O.get_masks(bm, nmr, mtu, m["k_max"], idx, run)
            r = trainer.step(data[idx], fmask)
            if first is None:
                first = r
            steps["loss"].append(float(r["loss"])); steps["grad_norm"].append(float(r["grad_norm"]))
            ftl += np.float32(r["sq_full"]); ptl += np.float32(r["sq_partial"])
        book["ftl"].append(np.sqrt(ftl / (nb_pred * nb_train)))
        book["ptl"].append(np.sqrt(ptl / (nb_train * nb_pred / S)))
        fvl = pvl = np.float32(0); rl = 0.0
        for _ in range(nv):
            idx, run = calls[c]; c += 1
            _, fmask = O.get_masks(bm, nmr, mtu, m["k_max"], idx, run)
            r = trainer.evaluate(data[idx], fmask)
            fvl += np.float32(r["sq_full"]); pvl += np.float32(r["sq_partial"])
            if "rank" in r:
                rl += float(r["rank"])
            else:
                rl += O.ranking_loss(np.asarray(r["y"]), fmask, idx, per_cat, E, val_idx)
        book["fvl"].append(np.sqrt(fvl / (nb_pred * nb_val)))
        book["pvl"].append(np.sqrt(pvl / (nb_val * nb_pred / S)))
        book["rl"].append(rl / nb_val)
    return book, steps, first


def replay_abalone(g, trainer):
    m = g.meta
    arch = m["arch"]
    data = g["data"]
    bm, mtu, nmr = g["binary_masks"], g["mask_to_use"], g["nb_missing_per_run"]
    per_k = [int(v) for v in g["nb_corruption_per_k"]]
    nb_run = sum(per_k)
    T = O.mask_transformation(g["type_mask"], len(arch))
    nmin, nscale = g["norm_min"], g["norm_scale"]
    nb_train, N = m["nb_train"], m["N"]
    nb_val = N - nb_train
    nb_pred = len(arch)
    nt = math.ceil(nb_train / m["batch"])
    nv = math.ceil(nb_val / m["batch"])
    calls = g.calls()
    assert len(calls) == m["epochs"] * nb_run * (nt + nv)
    book = {k: [] for k in ("ftl", "ptl", "fvl", "pvl", "ftl_per_k", "ptl_per_k", "fvl_per_k", "pvl_per_k")}
    steps = {"loss": [], "grad_norm": []}
    c = 0

    def monitor(x, y, masks, fmask, acc):
        x = x.copy(); y = np.asarray(y).copy()
        x[:, 3:] = O.normalizer_undo(x[:, 3:], nscale, nmin)          # abalone.py:227
        y[:, 3:] = O.normalizer_undo(y[:, 3:], nscale, nmin)          # abalone.py:228
        loss = O.combined_full(arch, x, y)                            # :231
        acc["f"] += np.sum(loss)
        acc["f_k"] += O.get_per_k(loss, masks, T)                     # :233
        part = O.get_partial(loss, fmask, T)                          # :234
        acc["p"] += np.sum(part)
        acc["p_k"] += O.get_per_k(part, masks, T)                     # :236

    def finish(acc, n):
        f_k, p_k = acc["f_k"], acc["p_k"]
        for i in range(len(per_k)):
            f_k[i, :] = f_k[i, :] / (n * sum(per_k[:i + 1]))
            p_k[i, :] = p_k[i, :] / (n * sum(per_k[:i + 1]) / nb_pred)
        f = acc["f"] / (sum(per_k) * n)
        p = acc["p"] / (sum(per_k) * n / nb_pred)
        f_k[:, 1:] = np.sqrt(f_k[:, 1:]); p_k[:, 1:] = np.sqrt(p_k[:, 1:])
        return np.sqrt(f), np.sqrt(p), f_k, p_k

    for _ in range(m["epochs"]):
        acc = {"f": 0, "p": 0, "f_k": np.zeros((m["k_max"], nb_pred)), "p_k": np.zeros((m["k_max"], nb_pred))}
        for run in range(nb_run):
            for _ in range(nt):
                idx, r_ = calls[c]; c += 1
                assert r_ == run
                masks, fmask = O.get_masks(bm, nmr, mtu, m["k_max"], idx, run)
                x = data[idx]
                r = trainer.step(x, fmask)
                steps["loss"].append(float(r["loss"])); steps["grad_norm"].append(float(r["grad_norm"]))
                monitor(x, r["y"], masks, fmask, acc)
        f, p, f_k, p_k = finish(acc, nb_train)
        book["ftl"].append(f); book["ptl"].append(p); book["ftl_per_k"].append(f_k); book["ptl_per_k"].append(p_k)
        acc = {"f": 0, "p": 0, "f_k": np.zeros((m["k_max"], nb_pred)), "p_k": np.zeros((m["k_max"], nb_pred))}
        for run in range(nb_run):
            for _ in range(nv):
                idx, r_ = calls[c]; c += 1
                masks, fmask = O.get_masks(bm, nmr, mtu, m["k_max"], idx, run)
                x = data[idx]
                r = trainer.evaluate(x, fmask)
                monitor(x, r["y"], masks, fmask, acc)
        f, p, f_k, p_k = finish(acc, nb_val)
        book["fvl"].append(f); book["pvl"].append(p); book["fvl_per_k"].append(f_k); book["pvl_per_k"].append(p_k)
    return book, steps
