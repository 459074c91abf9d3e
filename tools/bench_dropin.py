"""The reference's OWN training loop (script/train_dae_on_embedding.py:194-215 of the reference: corrupt -> model(x) -> MSELoss
-> backward -> clip_grad_norm_ -> torch.optim.Adam.step) run against this build's drop-in `codae.model` class on the
MI355X, i.e. what a user gets who only swaps the package: every Linear/ReLU of forward and backward goes through
libcodae_hip.so (one autograd Function), loss / clip / Adam stay torch ops.  Prints ms per step next to the fused
`codae_train_step` number bench.py reports for the same shape.

Usage: python tools/bench_dropin.py [slots] [embedding] [batch] [precision]   (default 3 512 8192 bf16)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
S, E, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (3, 512, 8192)
prec = sys.argv[4] if len(sys.argv) >= 5 else "bf16"
os.environ["CODAE_PRECISION"] = prec
import torch
from codae.model import EmbeddingDenoisingAutoencoder

dev = torch.device("cuda:0")
io = S * E
torch.manual_seed(0)
model = EmbeddingDenoisingAutoencoder(io_size=io, z_size=io, embedding_size=E, nb_input_layer=4, nb_output_layer=4, steep_layer_size=False).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-5, weight_decay=1e-4)
crit = torch.nn.MSELoss(reduction="mean")
data = torch.rand(4 * B, io, device=dev)
table = torch.ones(S, io, device=dev)
for s in range(S):
    table[s, s * E:(s + 1) * E] = 0
ids = torch.randint(0, S, (4 * B,), device=dev)


def step(i):
    x = data[(i % 4) * B:(i % 4 + 1) * B]
    mask = table[ids[(i % 4) * B:(i % 4 + 1) * B]]
    opt.zero_grad()
    c = model.corrupt(input_data=x, mask=mask)
    y = model(c)
    loss = crit(x, y)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
    opt.step()
    return loss


for i in range(5):
    step(i)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for i in range(n):
    loss = step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("drop-in class, reference loop, %d x %d batch %d, %s: %.3f ms/step = %.2f M samples/s (loss %.5f)" % (S, E, B, prec, dt * 1e3, B / dt / 1e6, float(loss.detach())))
