"""ConcatenatedEmbeddingDataset: `{obs_id: {category: [float]*E}}` -> data[N, S*E] fp32.

Counterpart of codae/dataset/concatenated_embedding_dataset.py:9-143 (same constructor,
attributes, `__getitem__ -> (row, idx)` and `.to(device)`), built with whole-array numpy
operations instead of N x S torch.cat calls.  The matrix is what stays resident in HBM and what
the gather kernel reads (codae_batch.data).
"""
import numpy as np
import torch
from torch.utils.data.dataset import Dataset


class ConcatenatedEmbeddingDataset(Dataset):

    def __init__(self, embeddings, used_category, transform=None):
        self.embeddings = embeddings
        self.transform = transform
        self.used_category = used_category
        self.nb_used_category = len(used_category)

        # keep observations that have every used category (reference :28-38)
        self.index = [k for k, v in embeddings.items() if all(c in v for c in used_category)]
        self.filtered_embeddings = {k: embeddings[k] for k in self.index}
        self.nb_observation = len(self.index)
        self.embedding_size = len(self.filtered_embeddings[self.index[0]][used_category[0]])

        blocks = [np.asarray([self.filtered_embeddings[k][c] for k in self.index], dtype=np.float32)
                  .reshape(self.nb_observation, self.embedding_size) for c in used_category]
        # raw per-slot matrices stay unscaled (reference :62-63; RankingLoss compares against them)
        self.data_per_category = {n: torch.from_numpy(b.copy()) for n, b in enumerate(blocks)}
        data = torch.from_numpy(np.concatenate(blocks, axis=1))

        # global (max - min) scaling with no shift (reference :69-74)
        self.min = data.min()
        self.max = data.max()
        self.scale = (self.max - self.min).item()
        self.data = data / self.scale

        self.arch = []
        self.io_size = 0
        for name in used_category:
            self.arch.append({"name": name, "lambda": 1, "size": self.embedding_size,
                              "type": "regression", "position": self.io_size})
            self.io_size += self.embedding_size
        self.type_mask = torch.ones((self.io_size))
        self.nb_predictor = self.embedding_size * self.nb_used_category

    def __len__(self):
        return self.nb_observation

    def __getitem__(self, idx):
        if self.transform is not None:
            return self.transform(self.data[idx]), idx
        return self.data[idx], idx

    def to(self, device):
        self.data = self.data.to(device)
        for i in range(len(self.data_per_category)):
            self.data_per_category[i] = self.data_per_category[i].to(device)
