"""RankingLoss on a C3-shaped validation set (3 x 512, 131072 observations, 20 % validation, batches of 8192): the
wave-per-item kernel (RankingLoss.get: whole inventory re-read per sample, mask expansion and a host sync per batch)
against the batched GEMM form (RankingLoss.add / total).  Prints ms per validation batch and per validation epoch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae.tool import RankingLoss
S, E, N, B = 3, 512, 131072, 8192
V = N // 5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


class DS:
    nb_predictor, nb_used_category, embedding_size = S * E, S, E
    data_per_category = {c: torch.randn(N, E, generator=g) for c in range(S)}


table = torch.ones(S, S * E, dtype=torch.uint8)
for c in range(S):
    table[c, c * E:(c + 1) * E] = 0


class Corr:
    mask_table_u8 = table.to(dev)
    mask_to_use_i32 = torch.randint(0, S, (N, 1), generator=g, dtype=torch.int32).to(dev)


val = torch.randperm(N, generator=g)[:V].tolist()
rl = RankingLoss(DS(), val, device=dev)
idx = torch.tensor(val[:B], dtype=torch.int32, device=dev)
pred = torch.randn(B, S * E, generator=g).to(dev)
fmask = table.to(dev)[Corr.mask_to_use_i32[idx.long(), 0].long()].float()
n_batches = (V + B - 1) // B

rl.add(pred, idx, Corr, run=0); rl.total()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    rl.add(pred, idx, Corr, run=0)
got = rl.total() / 3
t_gemm = (time.perf_counter() - t0) / 3
torch.cuda.synchronize(); t0 = time.perf_counter()
ref = rl.get(pred, fmask, idx.tolist())
t_item = time.perf_counter() - t0
print("batched GEMM form : %8.2f ms per batch of %d, %8.1f ms per validation epoch (%d batches)" % (t_gemm * 1e3, B, t_gemm * 1e3 * n_batches, n_batches))
print("wave-per-item form: %8.2f ms per batch of %d, %8.1f ms per validation epoch" % (t_item * 1e3, B, t_item * 1e3 * n_batches))
print("sum over the batch: %.3f vs %.3f (difference %.4f = %.1f flipped comparisons)" % (got, ref, got - ref, abs(got - ref) * (V - 1)))
