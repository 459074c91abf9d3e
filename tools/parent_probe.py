import subprocess, sys, json, os
mode = sys.argv[1]
import torch
if mode in ("avail", "alloc", "work"):
    print("avail", torch.cuda.is_available(), flush=True)
if mode in ("alloc", "work"):
    x = torch.zeros(1 << 20, device="cuda:0"); torch.cuda.synchronize()
if mode == "work":
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        y = x * 2
    torch.cuda.synchronize()
out = subprocess.run([sys.executable, "bench.py", "--no-f32-parity", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"], capture_output=True, text=True)
for l in out.stdout.splitlines():
    if l.startswith("{"):
        d = json.loads(l); print(mode, d["ms_per_step"], d["host_enqueue_ms_per_step"], flush=True)
if out.returncode: print(out.stderr[-500:])
