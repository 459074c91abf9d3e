import subprocess, sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + sys.argv[1:], capture_output=True, text=True, timeout=600, cwd=ROOT)
for l in out.stdout.splitlines():
    if l.startswith("{"):
        d = json.loads(l); print("probe2", sys.argv[1:], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["host_enqueue_done_ms_first_steps"], d["ramp_up_step_ms"], flush=True)
