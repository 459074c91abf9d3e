"""MixedVariableDataset: pandas frame -> data[N, io] fp32 (numeric column -> one regression
column, other columns -> one-hot in first-seen label order).

Counterpart of codae/dataset/mixed_variable_dataset.py:7-141, vectorised per column.
"""
import numpy as np
import torch
from torch.utils.data.dataset import Dataset


class MixedVariableDataset(Dataset):

    def __init__(self, pd_dataset):
        self.pd_dataset = pd_dataset
        self.variable_names = pd_dataset.columns
        self.nb_predictor = len(pd_dataset.columns)
        self.nb_observation = len(pd_dataset)
        self.io_size = 0
        self.arch = []
        for i, column in enumerate(pd_dataset):
            dtype = pd_dataset.dtypes.iloc[i]
            numeric = (dtype == "float64") or (dtype == "int64")
            size = 1 if numeric else int(pd_dataset[column].nunique())
            self.arch.append({"name": column, "lambda": 1, "size": size,
                              "type": "regression" if numeric else "classification",
                              "position": self.io_size})
            self.io_size += size

        self.type_mask = torch.zeros((self.io_size))
        data = np.zeros((self.nb_observation, self.io_size), dtype=np.float64)
        self.map = {}
        for v in self.arch:
            p, s = v["position"], v["size"]
            col = pd_dataset[v["name"]].to_numpy()
            if v["type"] == "regression":
                self.type_mask[p:p + s] = 1
                data[:, p] = col.astype(np.float64)
            else:
                labels = {}
                for lab in col:               # first-seen order (reference :64-72)
                    if lab not in labels:
                        labels[lab] = len(labels)
                self.map[v["name"]] = dict(labels, COUNT=len(labels))
                codes = np.fromiter((labels[lab] for lab in col), dtype=np.int64, count=len(col))
                data[np.arange(len(col)), p + codes] = 1
        self.data = torch.Tensor(data)

    def __len__(self):
        return self.nb_observation

    def __getitem__(self, idx):
        return self.data[idx], idx

    def _categorical_to_OHE(self, label, max):
        out = np.zeros(max)
        out[label] = 1
        return out

    def to(self, device):
        self.data = self.data.to(device)
        self.type_mask = self.type_mask.to(device)

    def cosine_similarity(self, query, indices=None):
        pass
