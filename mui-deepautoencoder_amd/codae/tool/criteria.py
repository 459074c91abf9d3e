"""Criteria and metrics of `codae.tool` (codae/tool/metering.py:24-204 of the reference).

CombinedCriterion and RankingLoss keep the reference's call signatures and numerics.  They are
the abalone loss and the validation-only rank metric — rows "next" of the scope table (SURVEY.md
section 8f) — and are expressed here as whole-tensor torch operations on the tensors' own device
(no per-sample Python loops); the MSE training loss of the embedding path is the fused HIP
kernel (codae_mse_loss_fwd_bwd / codae_step_forward_loss).
"""
import numpy as np
import torch

from .batching import get_mask_transformation


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

    def get(self, prediction, fmask, indices):
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

    def _per_variable(self, x, y, v):
        p, s = v["position"], v["size"]
        xs, ys = x[:, p:p + s], y[:, p:p + s]
        if v["type"] == "regression":
            return self.MSE_criterion(input=xs, target=ys)
        return self.CE_criterion(input=torch.log_softmax(ys, dim=1), target=xs.max(dim=1)[1])

    def _full_loss(self, x, y, as_numpy=False):
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
