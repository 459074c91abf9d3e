"""Corrupter: structured whole-variable blanking masks (codae/tool/data_tool.py:165-262).

Same constructor, attributes and `get_masks` return types as the reference.  The tables are
kept in the compact form the kernels consume — a uint8 `binary_masks` table, the k of every
mask, and an int32 `mask_to_use[N, nb_run]` — resident on `device`; `get_masks` expands a
batch with one kernel (codae_expand_masks) instead of a per-observation Python loop with a
host->device row copy each, and the fused training step never expands at all: it takes
`mask_ids()`.
"""
import itertools
import random

import torch

from ..hip import HipError
from ..hip import check as _check, current_stream as _stream, lib as _hip_lib, ptr as _ptr


class Corrupter:

    def __init__(self, nb_observation, arch, k_max, device):
        self.nb_observation = nb_observation
        self.arch = arch
        self.k_max = k_max
        self.device = torch.device(device) if not isinstance(device, torch.device) else device

        if (k_max < 0) | (k_max > len(self.arch) - 1):       # reference :184
            raise Exception("Invalid k_max number. k_max > 0 && k_max < nb_predictor - 1")

        self.io_size = sum(v["size"] for v in self.arch)
        self.nb_predictor = len(self.arch)

        # one row per k-subset of variables, k = 1..k_max, itertools order (reference :193-209)
        rows, self.nb_missing_per_run, self.nb_corruption_per_k = [], [], []
        for k in range(1, k_max + 1):
            subsets = list(itertools.combinations(range(self.nb_predictor), k))
            self.nb_corruption_per_k.append(len(subsets))
            for subset in subsets:
                row = torch.ones((self.io_size))
                for v in subset:
                    p = self.arch[v]["position"]
                    row[p:p + self.arch[v]["size"]] = 0
                rows.append(row)
                self.nb_missing_per_run.append(k)
        self.nb_run = sum(self.nb_corruption_per_k)
        self.binary_masks = torch.stack(rows) if rows else torch.zeros((0, self.io_size))

        # per-observation random order of the masks; consumes Python's `random` exactly like
        # the reference (:222-226) so a seeded run reproduces its mask assignment
        self.corrupted_index = list(range(self.nb_run))
        self.mask_to_use = torch.LongTensor(
            [random.sample(self.corrupted_index, self.nb_run) for _ in range(nb_observation)]
        ).reshape(nb_observation, self.nb_run)

        self.nb_subset_per_variable = []
        for i in range(1, self.k_max + 1):
            k_subset = 1
            for j in range(1, i):
                k_subset *= (self.nb_predictor - j) / 2
            self.nb_subset_per_variable.append(k_subset)

        # compact device-resident tables for the kernels
        self.mask_table_u8 = self.binary_masks.to(torch.uint8).contiguous().to(self.device)
        self.k_of_mask_i32 = torch.tensor(self.nb_missing_per_run, dtype=torch.int32, device=self.device)
        self.mask_to_use_i32 = self.mask_to_use.to(torch.int32).contiguous().to(self.device)

    def mask_ids(self, batch_indices, run):
        """int32 [B] mask-table row per sample = mask_to_use[idx][run] (reference :255-258)."""
        if not torch.is_tensor(batch_indices):
            batch_indices = torch.as_tensor(list(batch_indices), dtype=torch.long)
        idx = batch_indices.to(device=self.device, dtype=torch.long)
        return self.mask_to_use_i32[idx, run].contiguous()

    def get_masks(self, batch_indices, run):
        """(list of k_max tensors [B, io], their sum) on `device` (reference :239-262)."""
        B = len(batch_indices)
        ids = self.mask_ids(batch_indices, run)
        if self.device.type == "cuda":
            masks = torch.empty((self.k_max, B, self.io_size), dtype=torch.float32, device=self.device)
            fmask = torch.empty((B, self.io_size), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _check(_hip_lib().codae_expand_masks(_ptr(ids), _ptr(self.mask_table_u8), _ptr(self.k_of_mask_i32), B,
                                                     self.io_size, self.k_max, _ptr(masks), _ptr(fmask), _stream()))
            return [masks[k] for k in range(self.k_max)], fmask
        # host tensors requested (device == cpu): plain table lookups, no arithmetic
        ids = ids.long()
        fmask = self.binary_masks[ids]
        ks = self.k_of_mask_i32.long()[ids]
        masks = [fmask * (ks == k + 1).unsqueeze(1) for k in range(self.k_max)]
        return masks, fmask
