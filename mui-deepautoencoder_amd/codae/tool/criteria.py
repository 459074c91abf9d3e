"""Criteria and metrics of `codae.tool` (codae/tool/metering.py:24-204 of the reference).

CombinedCriterion (the abalone loss) and RankingLoss (the validation-only rank metric) keep the
reference's call signatures and numerics.  With tensors on a HIP device they run as kernels of
libcodae_hip.so (criteria.hip: codae_combined_loss_fwd_bwd / _full, codae_ranking_loss); with host
tensors (which upstream also accepts) they are whole-tensor torch expressions, no per-sample loops.
"""
import ctypes as C

import numpy as np
import torch

from ..hip import check as _check, current_stream as _stream, lib as _hip_lib, ptr as _ptr
from .batching import get_mask_transformation


class _CombinedMeanFn(torch.autograd.Function):
    """loss = CombinedCriterion mean loss; d loss / d y from the same kernel pass."""

    @staticmethod
    def forward(ctx, y, x, crit):
        loss, dy = crit._hip_mean(x, y.detach())
        ctx.save_for_backward(dy)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        return g * dy, None, None


def get_rmse(x, y):
    return np.sqrt(np.mean((x - y) ** 2))


class RankingLoss:
    """Cosine-rank of the reconstructed slot among the validation inventory (metering.py:29-79)."""

    def __init__(self, dataset, validation_indices, device):
        self.dataset = dataset
        self.device = device
        self.category_getter = torch.zeros((self.dataset.nb_predictor), device=self.device)
        for i in range(self.dataset.nb_used_category):
            self.category_getter[i * self.dataset.embedding_size] = i
        self.validation_indices = validation_indices
        self._val = None

    def _get_inventory(self, dev):
        ds = self.dataset
        S, E = ds.nb_used_category, ds.embedding_size
        if getattr(self, "_inv", None) is None or self._inv.device != dev:
            self._inv = torch.stack([ds.data_per_category[c].to(dev, torch.float32) for c in range(S)]).contiguous()
            self._inv_norm = torch.empty(self._inv.shape[:2], dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _check(_hip_lib().codae_row_norms(_ptr(self._inv), S * self._inv.shape[1], E, _ptr(self._inv_norm), _stream()))
            self._val_i32 = torch.as_tensor(list(self.validation_indices), dtype=torch.int32, device=dev)
            self._out = torch.zeros(1, dtype=torch.float64, device=dev)

    def _get_hip(self, prediction, fmask, indices):
        ds = self.dataset
        dev = prediction.device
        S, E = ds.nb_used_category, ds.embedding_size
        self._get_inventory(dev)
        idx = torch.as_tensor(list(indices), dtype=torch.int32, device=dev)
        pred = prediction.detach().to(torch.float32).contiguous()
        fm = fmask.to(device=dev, dtype=torch.float32).contiguous()
        self._out.zero_()
        with torch.cuda.device(dev):
            _check(_hip_lib().codae_ranking_loss(_ptr(pred), _ptr(fm), _ptr(idx), pred.shape[0], pred.shape[1], S, E,
                                                 _ptr(self._inv), _ptr(self._inv_norm), self._inv.shape[1],
                                                 _ptr(self._val_i32), len(self.validation_indices), _ptr(self._out),
                                                 _stream()))
        return float(self._out.item())

    # ---- device-resident form: one batched call per validation batch, one read-back per epoch ---------
    def _device_tables(self, dev):
        self._get_inventory(dev)
        if getattr(self, "_inv_val", None) is None or self._inv_val.device != dev:
            S, E = self.dataset.nb_used_category, self.dataset.embedding_size
            V = len(self.validation_indices)
            self._inv_val = torch.empty((S, V, E), dtype=torch.float32, device=dev)
            self._inv_val_norm = torch.empty((S, V), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _check(_hip_lib().codae_gather_inventory_rows(_ptr(self._inv), self._inv.shape[1], E, S, _ptr(self._val_i32), V,
                                                              _ptr(self._inv_val), _stream()))
                _check(_hip_lib().codae_row_norms(_ptr(self._inv_val), S * V, E, _ptr(self._inv_val_norm), _stream()))
            pos = torch.full((self._inv.shape[1],), -1, dtype=torch.int32)
            pos[torch.as_tensor(list(self.validation_indices), dtype=torch.long)] = torch.arange(V, dtype=torch.int32)
            self._val_pos = pos.to(dev)
            # rows of a slot's validation inventory that hold the same bytes share a group id: the reference never counts an exact
            # copy of the sample's own row (equal similarities out of one cosine_similarity call), so neither may we (codae_hip.h)
            groups = [torch.unique(self._inv_val[c], dim=0, return_inverse=True)[1] for c in range(S)]
            grp = torch.stack(groups).to(torch.int32).contiguous()
            self._val_group = grp if int(grp.max()) + 1 < V else None          # (no duplicates: nothing to skip)
            self._acc = torch.zeros(1, dtype=torch.float64, device=dev)
            self._work = None

    def add(self, prediction, row_idx, corrupter, run=0, chunk=4096):
        """Accumulate RankingLoss.get(prediction, fmask, indices) of one validation batch ON THE DEVICE: `row_idx` int32
        dataset rows (device), masks taken from `corrupter`'s device tables for `run`.  No host synchronisation; read the
        epoch's sum with total()."""
        dev = prediction.device
        self._device_tables(dev)
        S, E = self.dataset.nb_used_category, self.dataset.embedding_size
        V = len(self.validation_indices)
        pred = prediction.detach()
        if pred.dtype != torch.float32 or not pred.is_contiguous():
            pred = pred.to(torch.float32).contiguous()
        B = pred.shape[0]
        chunk = min(int(chunk), V)
        if self._work is None or self._work.numel() < B * chunk or self._rows.numel() < 8 * B:
            self._work = torch.empty(B * chunk, dtype=torch.float32, device=dev)
            self._rows = torch.empty(8 * B, dtype=torch.int32, device=dev)
            self._perm = torch.empty(S * B + S, dtype=torch.int32, device=dev)
            self._q = torch.empty(B * E, dtype=torch.float32, device=dev)
        m2u = corrupter.mask_to_use_i32
        with torch.cuda.device(dev):
            _check(_hip_lib().codae_ranking_loss_batched(
                _ptr(pred), B, pred.shape[1], S, E, _ptr(row_idx), None, _ptr(m2u), m2u.shape[1], int(run),
                _ptr(corrupter.mask_table_u8), _ptr(self._inv), _ptr(self._inv_norm), self._inv.shape[1],
                _ptr(self._inv_val), _ptr(self._inv_val_norm), _ptr(self._val_pos),
                _ptr(self._val_group) if self._val_group is not None else None, V, _ptr(self._work), chunk, _ptr(self._rows),
                _ptr(self._perm), _ptr(self._q), _ptr(self._acc), _stream()))
        self._keep = (pred, row_idx)             # (alive until the stream has consumed them)

    def total(self, reset=True):
        """Sum of the batches add()ed since the last reset (one device-to-host read)."""
        v = float(self._acc.item())
        if reset:
            self._acc.zero_()
        return v

    def get(self, prediction, fmask, indices):
        if prediction.device.type == "cuda":
            return self._get_hip(prediction, fmask, indices)
        E = self.dataset.embedding_size
        dev = prediction.device
        if self._val is None or self._val.device != dev:
            self._val = torch.as_tensor(list(self.validation_indices), dtype=torch.long, device=dev)
        idx = torch.as_tensor(list(indices), dtype=torch.long, device=dev)
        # blanked slot of every sample: dot(1 - fmask, category_getter) (metering.py:56), k = 1 only
        slot = torch.matmul(1 - fmask, self.category_getter.to(dev)).long()
        pred = prediction.detach()
        total = 0.0
        for c in torch.unique(slot).tolist():
            rows = (slot == c).nonzero(as_tuple=True)[0]
            inv = self.dataset.data_per_category[c].to(dev)                      # [N, E] unscaled
            q = pred[rows, c * E:(c + 1) * E]                                    # [b, E]
            inv_n = inv / inv.norm(dim=1, keepdim=True).clamp_min(1e-8)
            q_n = q / q.norm(dim=1, keepdim=True).clamp_min(1e-8)
            s = q_n @ inv_n.t()                                                  # [b, N] cosine
            own = s.gather(1, idx[rows].unsqueeze(1))                            # s[idx]
            rank = (own > s[:, self._val]).sum(dim=1)                            # strict > over validation ids
            total += float((1 - rank.double() / (len(self.validation_indices) - 1)).sum())
        return total


class CombinedCriterion:
    """Per-variable RMSE / NLL loss following dataset.arch (metering.py:82-204)."""

    def __init__(self, arch, k_max, device, observation_mask, weight=None, reduction="none"):
        if k_max < 0 | k_max >= len(arch):      # chained comparison kept as upstream (never fires)
            raise Exception("Error: maximum number of corrupted index [k_max] must be (> 0) && (< len(arch)).")
        self.arch = arch
        self.k_max = k_max
        self.device = device
        self.reduction = reduction
        self.weight = torch.ones((1, len(self.arch))) if weight is None else torch.Tensor(weight)
        self.observation_mask = observation_mask
        self.io_size = len(self.observation_mask)
        # upstream tests type == "continuous", which no dataset emits: only the length is used
        self.loss_mask = [1 if v["type"] == "continuous" else 0 for v in arch]
        self.mask_transformation = get_mask_transformation(
            observation_mask=self.observation_mask, loss_mask=self.loss_mask).cpu().numpy()
        self.MSE_criterion = torch.nn.MSELoss(reduction=self.reduction)
        self.CE_criterion = torch.nn.NLLLoss(reduction=self.reduction)

    def __call__(self, x, y, as_numpy=False):
        if self.reduction == "mean":
            return self._mean_loss(x, y, as_numpy)
        if self.reduction == "none":
            return self._full_loss(x, y, as_numpy)
        raise Exception("Unknown reduction type.")

    # ---- HIP path (tensors on a HIP device) -------------------------------------------------
    def _tables(self, dev):
        t = getattr(self, "_dev_tables", None)
        if t is None or t[0].device != dev:
            i32 = dict(dtype=torch.int32, device=dev)
            w = self.weight.reshape(-1).to(torch.float32)
            if w.numel() != len(self.arch):
                raise Exception("CombinedCriterion: one weight per variable is required on the HIP path")
            t = (torch.tensor([v["position"] for v in self.arch], **i32),
                 torch.tensor([v["size"] for v in self.arch], **i32),
                 torch.tensor([0 if v["type"] == "regression" else 1 for v in self.arch], **i32),
                 w.to(dev).contiguous(),
                 torch.zeros(len(self.arch), dtype=torch.float64, device=dev))
            self._dev_tables = t
        return t

    def _hip_mean(self, x, y):
        dev = y.device
        pos, size, typ, w, acc = self._tables(dev)
        x = x.detach().to(torch.float32).contiguous()
        y = y.to(torch.float32).contiguous()
        dy = torch.empty_like(y)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _check(_hip_lib().codae_combined_loss_fwd_bwd(_ptr(x), _ptr(y), y.shape[0], y.shape[1], len(self.arch), _ptr(pos),
                                                          _ptr(size), _ptr(typ), _ptr(w), _ptr(acc), _ptr(dy), _ptr(loss),
                                                          _stream()))
        return loss[0].to(torch.float32), dy

    def _hip_full(self, x, y):
        dev = y.device
        pos, size, typ, _, _ = self._tables(dev)
        x = x.detach().to(torch.float32).contiguous()
        y = y.detach().to(torch.float32).contiguous()
        out = torch.empty((y.shape[0], len(self.arch)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _check(_hip_lib().codae_combined_loss_full(_ptr(x), _ptr(y), y.shape[0], y.shape[1], len(self.arch), _ptr(pos),
                                                       _ptr(size), _ptr(typ), _ptr(out), _stream()))
        return out

    # ---- device-resident per-step accounting of the abalone sweep -----------------------------------
    def accumulate(self, x, y, mask_ids, corrupter, normalizer=None, first_scaled_column=0):
        """What script/train_dae_on_abalone.py:227-236 of the reference does on the host after every step - the monitor
        criterion of the de-normalised batch, `get_partial`, two `get_per_k`s and four running sums - as ONE launch that
        adds into fp64 tables on the device (codae_monitor_accumulate).  `mask_ids`: int32 device tensor = the Corrupter's
        mask-table row per sample (`corrupter.mask_ids(batch_indices, run)`); `normalizer`: a tool.Normalizer whose `undo` the
        script applies to columns >= first_scaled_column of x and y (None: none).  Nothing is copied to the host; read the
        epoch's tables with `accumulated()`."""
        dev = y.device
        pos, size, typ, _, _ = self._tables(dev)
        nv, k_max = len(self.arch), self.k_max
        if getattr(self, "_mon_acc", None) is None or self._mon_acc.device != dev:
            self._mon_acc = torch.zeros(2 + 2 * k_max * nv, dtype=torch.float64, device=dev)
            self._mon_undo = None
        if normalizer is not None and self._mon_undo is None:
            io = x.shape[1]
            sc = torch.ones(io, dtype=torch.float32, device=dev)
            mn = torch.zeros(io, dtype=torch.float32, device=dev)
            sc[first_scaled_column:] = normalizer.scale.to(dev, torch.float32)
            mn[first_scaled_column:] = normalizer.min.to(dev, torch.float32)
            self._mon_undo = (sc.contiguous(), mn.contiguous())
        x = x.detach().to(torch.float32).contiguous()
        y = y.detach().to(torch.float32).contiguous()
        us, um = self._mon_undo if normalizer is not None else (None, None)
        with torch.cuda.device(dev):
            _check(_hip_lib().codae_monitor_accumulate(_ptr(x), _ptr(y), y.shape[0], y.shape[1], nv, _ptr(pos), _ptr(size), _ptr(typ),
                                                       _ptr(us) if us is not None else None, _ptr(um) if um is not None else None,
                                                       _ptr(mask_ids), _ptr(corrupter.mask_table_u8), _ptr(corrupter.k_of_mask_i32),
                                                       k_max, _ptr(self._mon_acc), _stream()))
        self._mon_keep = (x, y, mask_ids)          # (alive until the stream has consumed them)

    def accumulated(self, reset=True):
        """(f, p, f_k [k_max, n_var], p_k [k_max, n_var]) summed over the accumulate() calls since the last reset: float64
        numpy, one device-to-host copy."""
        nv, k_max = len(self.arch), self.k_max
        a = self._mon_acc.cpu().numpy().copy()
        if reset:
            self._mon_acc.zero_()
        return float(a[0]), float(a[1]), a[2:2 + k_max * nv].reshape(k_max, nv), a[2 + k_max * nv:].reshape(k_max, nv)

    def _per_variable(self, x, y, v):
        p, s = v["position"], v["size"]
        xs, ys = x[:, p:p + s], y[:, p:p + s]
        if v["type"] == "regression":
            return self.MSE_criterion(input=xs, target=ys)
        return self.CE_criterion(input=torch.log_softmax(ys, dim=1), target=xs.max(dim=1)[1])

    def _full_loss(self, x, y, as_numpy=False):
        if y.device.type == "cuda":
            loss = self._hip_full(x, y).cpu()                   # upstream returns a host tensor (:133)
            return loss.numpy() if as_numpy else loss
        loss = torch.zeros((x.size()[0], len(self.arch)))       # on the host, as upstream (:133)
        for i, v in enumerate(self.arch):
            li = self._per_variable(x, y, v)
            if v["type"] == "regression":
                loss[:, i:i + 1] = li
            else:
                loss[:, i] = li
        if as_numpy:
            return loss.clone().cpu().detach().numpy()
        return loss

    def _mean_loss(self, x, y, as_numpy=False):
        if y.device.type == "cuda" and not as_numpy:
            return _CombinedMeanFn.apply(y, x, self)
        loss = []
        for v in self.arch:
            li = self._per_variable(x, y, v)
            loss.append(torch.sqrt(li) if v["type"] == "regression" else li)
        if as_numpy:
            out = sum(loss) / len(self.arch)
            return out.clone().cpu().detach().numpy()
        loss = [loss[i] * self.weight[i] for i in range(len(loss))]
        return sum(loss) / len(self.arch)

    def get_per_k(self, loss, masks):
        """Per-k column sums of `loss` over the rows whose mask blanks k variables
        (metering.py:187-197: matmul(mask, ones) clipped to 1 is a row indicator; times T it is
        that indicator spread over the variables)."""
        out = np.zeros((self.k_max, len(self.arch)))
        spread = self.mask_transformation.sum(axis=0).astype(np.float64)      # 1 per variable
        for i, mask in enumerate(masks):
            hit = (mask.sum(dim=1) > 0).cpu().numpy().astype(np.float64)
            out[i, :] = np.sum(hit[:, None] * spread[None, :] * loss, axis=0)
        return out

    def get_partial(self, loss, mask):
        return (1 - np.matmul(mask.cpu().numpy(), self.mask_transformation)) * loss
