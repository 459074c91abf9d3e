"""Batch assembly helpers of `codae.tool` (codae/tool/data_tool.py:16-162 of the reference)."""
import hashlib
import json
import os

import numpy as np
import torch

from ..dataset import ConcatenatedEmbeddingDataset


def get_mask_transformation(observation_mask, loss_mask):
    """io-position -> variable 0/1 matrix T [len(observation_mask), len(loss_mask)]
    (data_tool.py:16-43).  A regression position owns a column; a run of non-regression
    positions (a one-hot block) owns one column, set on its first position only."""
    obs = [int(v == 1) for v in observation_mask]
    T = torch.zeros((len(obs), len(loss_mask)))
    col = 0
    prev_regression = True
    for i, is_regression in enumerate(obs):
        if is_regression or prev_regression:
            T[i, col] = 1
            col += 1
        prev_regression = bool(is_regression)
    return T


class Normalizer:
    """min-max (de)normalisation with a fitted sklearn MinMaxScaler's parameters
    (data_tool.py:46-90)."""

    def __init__(self, normalizer, device, normalization_type="min_max"):
        self.normalization_type = normalization_type
        self.device = device
        self.min = torch.Tensor(normalizer.data_min_).to(device)
        self.max = torch.Tensor(normalizer.data_max_).to(device)
        self.scale = torch.Tensor(normalizer.data_range_).to(device)

    def do(self, data):
        return (data - self.min) / self.scale

    def undo(self, data):
        return (data * self.scale) + self.min


def collate_embedding(batch):
    """[(row, idx)] -> (stacked rows, tuple of idx) (data_tool.py:96-103)."""
    rows, indices = zip(*batch)
    return torch.stack(rows), indices


def simple_collate(batch):
    return torch.stack(batch)


def load_dataset_of_embeddings(embedding_path, config, cache_dir="tmp/"):
    """JSON `{obs_id: {category: [float]}}` -> ConcatenatedEmbeddingDataset
    (data_tool.py:114-162).  The reference caches a pickle of the dataset object keyed by
    the file's ctime; this build keeps the key but stores plain arrays (.npz), never a pickle."""
    used = config["DATASET"]["USED_CATEGORY"]
    key = hashlib.sha1(str(os.stat(embedding_path)[9]).encode('utf-8')).hexdigest()
    cache = os.path.join(cache_dir, key + "_" + hashlib.sha1("|".join(used).encode()).hexdigest()[:8] + "_dataset.npz")
    if os.path.exists(cache):
        z = np.load(cache, allow_pickle=False)
        ids = [str(s) for s in z["index"]]
        blocks = z["blocks"]
        emb = {k: {c: blocks[n][i] for n, c in enumerate(used)} for i, k in enumerate(ids)}
        return ConcatenatedEmbeddingDataset(embeddings=emb, used_category=used)
    try:
        with open(embedding_path, 'r') as f:
            embeddings = json.load(f)
    except Exception:
        raise Exception("Error while reading embedding json file.")
    dataset = ConcatenatedEmbeddingDataset(embeddings=embeddings, used_category=used)
    os.makedirs(cache_dir, exist_ok=True)
    tmp = os.path.join(cache_dir, "new_dataset_tmp.npz")
    np.savez(tmp, index=np.asarray(dataset.index),
             blocks=np.stack([dataset.data_per_category[n].cpu().numpy() for n in range(len(used))]))
    os.replace(tmp, cache)
    return dataset
