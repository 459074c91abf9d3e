"""What the reference's own stack does on the same MI355X: the C3 training step as plain torch GPU ops
(index_select batch, mask multiply, nn.Linear x 10 + ReLU, MSELoss, clip_grad_norm_, Adam), in fp32 (the reference's
dtype) and under bf16 autocast, timed like bench.py.  The reference's per-batch Python work (Corrupter.get_masks loop,
DataLoader collate; SURVEY.md 8a: 0.2-0.6 s per batch at B = 8192) is NOT included: this is its arithmetic only.
Usage: python tools/bench_torch_gpu_step.py [steps]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S, E, B = 3, 512, 8192
io = S * E
dev = torch.device("cuda:0")
data_np, blank = bench.make_inputs(16 * B, io, S)
data = torch.from_numpy(data_np).to(dev)
table = torch.ones(S, io, device=dev)
for s in range(S): table[s, s * E:(s + 1) * E] = 0
mask_id = torch.from_numpy(blank).to(dev).long()
sched = bench.square_schedule(io, 4, 4)
g = torch.Generator(device="cpu").manual_seed(1)
idx = [torch.randperm(16 * B, generator=g)[:B].to(dev) for _ in range(steps + 5)]

def build():
    torch.manual_seed(0)
    layers = []
    for k, n, relu in sched:
        lin = torch.nn.Linear(k, n); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.zeros_(lin.bias)
        layers.append(lin)
        if relu: layers.append(torch.nn.ReLU(inplace=True))
    return torch.nn.Sequential(*layers).to(dev)

for mode in ("fp32", "bf16 autocast", "bf16 autocast + fused Adam"):
    model = build()
    opt = torch.optim.Adam(model.parameters(), lr=bench.LR, weight_decay=bench.WD, fused=("fused" in mode))
    crit = torch.nn.MSELoss(reduction="mean")
    def step(i):
        x = data.index_select(0, i)
        c = x * table.index_select(0, mask_id.index_select(0, i))
        opt.zero_grad(set_to_none=True)
        if mode == "fp32":
            y = model(c)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model(c)
        loss = crit(x, y.float())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), bench.CLIP)
        opt.step()
    for w in range(5): step(idx[w])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps): step(idx[5 + s_])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("torch GPU step, %-28s %8.3f ms/step  %10.0f samples/s" % (mode, dt * 1e3, B / dt))
