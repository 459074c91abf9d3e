#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only).  It
executes the reference's own training scripts end to end with `runpy`
(script/train_dae_on_embedding.py, script/train_dae_on_abalone.py) on seeded
synthetic inputs of the same schema as the real ones (the real abalone.data and
embedding JSON are not in the image) and records, through
a few capture hooks that do not touch RNG consumption, everything a replay
needs:

  inputs   : dataset matrix, initial state_dict, mask tables, the exact
             sequence of (batch_indices, run) the DataLoader produced
  outputs  : per-step loss and clip_grad_norm_ total norm, first-step grads,
             final parameters, Adam exp_avg / exp_avg_sq, the per-epoch `book`
             metrics the script logged

Nothing of the reference's source is stored: fixtures are plain arrays.

Usage:  python tests/golden/make_golden.py            (writes *.npz next to it)
"""
import json
import os
import random
import runpy
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_reference():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # codae/model/__init__.py imports an empty stub that needs torchvision
    # (absent here): register an empty module so the import succeeds.
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)


class Capture:
    """Hooks around reference/torch entry points; all pass-through."""

    def __init__(self, model_cls_path):
        import torch
        import codae.tool.data_tool as dt
        self.torch = torch
        self.calls = []        # (phase-agnostic) list of (indices tuple, run)
        self.losses = []
        self.norms = []
        self.first_grads = None
        self.init_state = None
        self._undo = []

        cap = self
        orig_get = dt.Corrupter.get_masks

        def get_masks(self_, batch_indices, run):
            cap.calls.append((tuple(int(i) for i in batch_indices), int(run)))
            return orig_get(self_, batch_indices, run)
        dt.Corrupter.get_masks = get_masks
        self._undo.append(lambda: setattr(dt.Corrupter, "get_masks", orig_get))

        orig_bw = torch.Tensor.backward

        def backward(self_, *a, **k):
            cap.losses.append(float(self_.item()))
            return orig_bw(self_, *a, **k)
        torch.Tensor.backward = backward
        self._undo.append(lambda: setattr(torch.Tensor, "backward", orig_bw))

        orig_clip = torch.nn.utils.clip_grad_norm_

        def clip(parameters, max_norm, *a, **k):
            parameters = list(parameters)
            if cap.first_grads is None:
                cap.first_grads = [p.grad.detach().clone().numpy() for p in parameters]
            n = orig_clip(parameters, max_norm, *a, **k)
            cap.norms.append(float(n))
            return n
        torch.nn.utils.clip_grad_norm_ = clip
        self._undo.append(lambda: setattr(torch.nn.utils, "clip_grad_norm_", orig_clip))

        mod_name, cls_name = model_cls_path
        mod = __import__(mod_name, fromlist=[cls_name])
        cls = getattr(mod, cls_name)
        orig_init = cls.__init__

        def init(self_, *a, **k):
            orig_init(self_, *a, **k)
            cap.init_state = {n: t.detach().clone().numpy() for n, t in self_.state_dict().items()}
        cls.__init__ = init
        self._undo.append(lambda: setattr(cls, "__init__", orig_init))

    def close(self):
        for u in reversed(self._undo):
            u()


def _seed_all(s):
    import torch
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def _pack_calls(calls):
    """ragged list of index tuples -> flat int64 + offsets + runs."""
    flat = np.asarray([i for c, _ in calls for i in c], dtype=np.int64)
    off = np.cumsum([0] + [len(c) for c, _ in calls]).astype(np.int64)
    runs = np.asarray([r for _, r in calls], dtype=np.int64)
    return flat, off, runs


def _state_arrays(prefix, d):
    return {prefix + k.replace(".", "__"): v for k, v in d.items()}


def make_embedding(name, seed, N, S, E, z, nb_in, nb_out, batch, epochs, lr, wd, store_adam=True):
    import torch
    import yaml
    cats = ["top", "bottom", "shoe", "bag", "hat", "scarf"][:S]
    rng = np.random.default_rng(seed)
    emb = {}
    for i in range(N):
        emb["obs%05d" % i] = {c: [float(v) for v in rng.standard_normal(E).astype(np.float32) * 0.5 + 0.25]
                              for c in cats}
    # observations lacking a used category must be filtered out (dataset :28-38)
    for i in range(7):
        emb["incomplete%02d" % i] = {cats[0]: [0.0] * E}
    cfg = {"MODEL": {"Z_SIZE": z, "BATCH_SIZE": batch, "NB_INPUT_LAYER": nb_in,
                     "NB_OUTPUT_LAYER": nb_out, "STEEP_LAYER_SIZE": False, "EPOCH": epochs,
                     "LEARNING_RATE": lr, "WEIGHT_DECAY": wd, "NB_CORRUPTED": 1, "TRUNK_GRAD": True},
           "DATASET": {"NAME": "EMBEDDING", "USED_CATEGORY": cats, "EMBEDDING_SIZE": E,
                       "SHUFFLE": True, "SPLIT": [0.7, 0.3]},
           "SEED": 27493045}
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.makedirs("log")
        with open("emb.json", "w") as f:
            json.dump(emb, f)
        with open("cfg.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        cap = Capture(("codae.model.embedding_denoising_autoencoder", "EmbeddingDenoisingAutoencoder"))
        _seed_all(seed)
        argv = sys.argv
        sys.argv = ["train_dae_on_embedding.py", "--embedding_path", "emb.json",
                    "--output_path", "out", "--config", "cfg.yaml"]
        try:
            g = runpy.run_path(os.path.join(REF, "script/train_dae_on_embedding.py"), run_name="__main__")
        finally:
            sys.argv = argv
            cap.close()
            os.chdir(HERE)
    model, opt, corr, ds, book = g["model"], g["optimizer"], g["corrupter"], g["dataset"], g["book"]
    flat, off, runs = _pack_calls(cap.calls)
    out = {
        "meta": np.asarray(json.dumps({
            "kind": "embedding", "seed": seed, "N": int(ds.nb_observation), "S": S, "E": E, "z": z,
            "nb_input_layer": nb_in, "nb_output_layer": nb_out, "batch": batch, "epochs": epochs,
            "lr": lr, "weight_decay": wd, "k_max": 1, "categories": cats,
            "nb_train": len(g["train_indices"]), "scale": ds.scale,
            "param_names": [n for n, _ in model.named_parameters()]})),
        "data": ds.data.numpy(),
        "raw_first_rows": np.asarray([emb["obs%05d" % i][cats[0]] for i in range(4)], dtype=np.float32),
        "data_per_category": np.stack([ds.data_per_category[c].numpy() for c in range(S)]),
        "train_indices": np.asarray(g["train_indices"], dtype=np.int64),
        "validation_indices": np.asarray(g["validation_indices"], dtype=np.int64),
        "binary_masks": corr.binary_masks.numpy(),
        "mask_to_use": corr.mask_to_use.numpy(),
        "nb_missing_per_run": np.asarray(corr.nb_missing_per_run, dtype=np.int64),
        "calls_flat": flat, "calls_off": off, "calls_run": runs,
        "step_loss": np.asarray(cap.losses, dtype=np.float64),
        "step_grad_norm": np.asarray(cap.norms, dtype=np.float64),
        "book_ftl": np.asarray(book["ftl"], dtype=np.float64),
        "book_ptl": np.asarray(book["ptl"], dtype=np.float64),
        "book_fvl": np.asarray(book["fvl"], dtype=np.float64),
        "book_pvl": np.asarray(book["pvl"], dtype=np.float64),
        "book_rl": np.asarray(book["rl"], dtype=np.float64),
    }
    out.update(_state_arrays("init__", cap.init_state))
    out.update(_state_arrays("final__", {n: t.detach().numpy() for n, t in model.state_dict().items()}))
    for i, gr in enumerate(cap.first_grads):
        out["grad0__%d" % i] = gr
    st = opt.state_dict()["state"]
    if store_adam:       # (the wide fixtures keep the files small: Adam moments are pinned by the 48-wide ones)
        for i in sorted(st):
            out["adam_m__%d" % i] = st[i]["exp_avg"].numpy()
            out["adam_v__%d" % i] = st[i]["exp_avg_sq"].numpy()
    out["adam_step"] = np.asarray(float(st[0]["step"]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "steps", len(cap.losses), "ftl", book["ftl"], "rl", book["rl"])


def make_abalone(name, seed, N, batch, epochs, lr, wd, nb_missing):
    import torch
    import yaml
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(N + 1):                      # first line is eaten as header (abalone.py:83-85)
        sex = "MFI"[int(rng.integers(0, 3))]
        fl = rng.random(7) * np.array([0.8, 0.65, 0.3, 2.8, 1.5, 0.76, 1.0]) + 0.01
        rings = int(rng.integers(1, 30))
        lines.append(",".join([sex] + ["%.4f" % v for v in fl] + [str(rings)]))
    cfg = {"MODEL": {"Z_SIZE": 11, "BATCH_SIZE": batch, "NB_INPUT_LAYER": 2, "NB_OUTPUT_LAYER": 2,
                     "STEEP_LAYER_SIZE": True, "EPOCH": epochs, "LEARNING_RATE": lr,
                     "WEIGHT_DECAY": wd, "TRUNK_GRAD": True},
           "DATASET": {"NAME": "abalone", "SHUFFLE": True, "SPLIT": [0.7, 0.3]},
           "SEED": 27123045, "EVALUATION": {"MODE": "TR"},
           "PLOT": {"TRAINING_ERROR_PER_K": False, "VALIDATION_ERROR_PER_K": False,
                    "FULL_ERROR": False, "PARTIAL_ERROR": False}}
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.makedirs("log")
        os.makedirs("data")
        with open("data/abalone.data", "w") as f:
            f.write("\n".join(lines) + "\n")
        with open("cfg.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        cap = Capture(("codae.model.mixed_variable_denoising_autoencoder", "MixedVariableDenoisingAutoencoder"))
        _seed_all(seed)
        argv = sys.argv
        sys.argv = ["train_dae_on_abalone.py", "--dataset_path", "data", "--output_path", "out",
                    "--config", "cfg.yaml", "--nb_missing", str(nb_missing)]
        try:
            g = runpy.run_path(os.path.join(REF, "script/train_dae_on_abalone.py"), run_name="__main__")
        finally:
            sys.argv = argv
            cap.close()
            os.chdir(HERE)
    model, opt, corr, ds, book, nz = (g["model"], g["optimizer"], g["corrupter"], g["dataset"],
                                      g["book"], g["tensor_normazer"])
    flat, off, runs = _pack_calls(cap.calls)
    arch = [{k: (int(v) if k in ("size", "position", "lambda") else str(v)) for k, v in a.items()}
            for a in ds.arch]
    out = {
        "meta": np.asarray(json.dumps({
            "kind": "abalone", "seed": seed, "N": int(ds.nb_observation), "batch": batch, "epochs": epochs,
            "lr": lr, "weight_decay": wd, "k_max": nb_missing, "arch": arch,
            "nb_train": len(g["train_indices"]), "weight": [0.4, 1, 1, 1, 1, 1, 1, 1, 1],
            "csv_first_lines": lines[:3],
            "param_names": [n for n, _ in model.named_parameters()]})),
        "data": ds.data.numpy(), "type_mask": ds.type_mask.numpy(),
        "norm_min": nz.min.numpy(), "norm_scale": nz.scale.numpy(),
        "train_indices": np.asarray(g["train_indices"], dtype=np.int64),
        "validation_indices": np.asarray(g["validation_indices"], dtype=np.int64),
        "binary_masks": corr.binary_masks.numpy(),
        "mask_to_use": corr.mask_to_use.numpy(),
        "nb_missing_per_run": np.asarray(corr.nb_missing_per_run, dtype=np.int64),
        "nb_corruption_per_k": np.asarray(corr.nb_corruption_per_k, dtype=np.int64),
        "calls_flat": flat, "calls_off": off, "calls_run": runs,
        "step_loss": np.asarray(cap.losses, dtype=np.float64),
        "step_grad_norm": np.asarray(cap.norms, dtype=np.float64),
        "book_ftl": np.asarray(book["ftl"], dtype=np.float64),
        "book_ptl": np.asarray(book["ptl"], dtype=np.float64),
        "book_fvl": np.asarray(book["fvl"], dtype=np.float64),
        "book_pvl": np.asarray(book["pvl"], dtype=np.float64),
        "book_ftl_per_k": np.stack(book["ftl_per_k"]), "book_ptl_per_k": np.stack(book["ptl_per_k"]),
        "book_fvl_per_k": np.stack(book["fvl_per_k"]), "book_pvl_per_k": np.stack(book["pvl_per_k"]),
    }
    out.update(_state_arrays("init__", cap.init_state))
    out.update(_state_arrays("final__", {n: t.detach().numpy() for n, t in model.state_dict().items()}))
    for i, gr in enumerate(cap.first_grads):
        out["grad0__%d" % i] = gr
    st = opt.state_dict()["state"]
    for i in sorted(st):
        out["adam_m__%d" % i] = st[i]["exp_avg"].numpy()
        out["adam_v__%d" % i] = st[i]["exp_avg_sq"].numpy()
    out["adam_step"] = np.asarray(float(st[0]["step"]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "steps", len(cap.losses), "ftl", book["ftl"])


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures can only be regenerated in the build container")
    _install_reference()
    # square stack (embedding.yaml topology scaled down): 10 x Linear(48,48)
    make_embedding("embedding_square", seed=1234, N=400, S=3, E=16, z=48, nb_in=4, nb_out=4,
                   batch=64, epochs=2, lr=1e-3, wd=1e-4)
    # tapered stack (z < io): 48->48->40->32->24->16 | 16->24->32->40->48->48
    make_embedding("embedding_taper", seed=4321, N=300, S=3, E=16, z=16, nb_in=4, nb_out=4,
                   batch=50, epochs=2, lr=1e-3, wd=1e-4)
    # Widths that are multiples of 64, so the bf16 (throughput) engine can replay a REFERENCE run too:
    # embedding.yaml's topology (4+4, z = io) at 3 slots x 64: 10 x Linear(192,192), 4 epochs of 9 steps
    make_embedding("embedding_wide_square", seed=2468, N=400, S=3, E=64, z=192, nb_in=4, nb_out=4,
                   batch=32, epochs=4, lr=1e-3, wd=1e-4, store_adam=False)
    # taper 192->192->128->64 | 64->128->192->192
    make_embedding("embedding_wide_taper", seed=8642, N=400, S=3, E=64, z=64, nb_in=2, nb_out=2,
                   batch=32, epochs=4, lr=1e-3, wd=1e-4, store_adam=False)
    # abalone schema, --nb_missing 2 (45 augmentation runs per epoch)
    make_abalone("abalone_k2", seed=99, N=300, batch=64, epochs=1, lr=5e-4, wd=1e-6, nb_missing=2)
