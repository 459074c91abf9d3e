import time, torch
n = 131072
ind = torch.arange(n)
def t(f, k=20):
    f(); t0 = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t0) / k * 1e3
print("threads", torch.get_num_threads())
print("randperm ms", t(lambda: torch.randperm(n)))
print("random_ scalar ms", t(lambda: torch.empty((), dtype=torch.int64).random_()))
p = torch.randperm(n)
print("index ms", t(lambda: ind[p]))
print("to int32 ms", t(lambda: p.to(torch.int32)))
dev = torch.device("cuda:0")
print("to device int32 ms", t(lambda: p.to(device=dev, dtype=torch.int32)))
print("to device int64 ms", t(lambda: p.to(device=dev)))
x = p[:8192]
print("slice to device ms", t(lambda: x.to(device=dev, dtype=torch.int32)))
torch.set_num_threads(1)
print("1 thread: randperm ms", t(lambda: torch.randperm(n)), "index ms", t(lambda: ind[p]))
