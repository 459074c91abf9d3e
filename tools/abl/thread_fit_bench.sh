#!/bin/bash
# bench.py's short form (what tests/test_gpu_bench.py and a cold driver run look like) with torch's intra-op pool at its default
# width and capped at the container's CPU quota; prints ms/step, the ramp-up steps and the CFS throttle counters around each run.
# Usage: bash tools/abl/thread_fit_bench.sh   (from the repo root, on a GPU box)
stat() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for fit in 1 0 1 0 1 0; do
  before=$(stat)
  if [ $fit = 1 ]; then line=$(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-parity 2>/dev/null | tail -1)
  else line=$(BENCH_NO_THREAD_FIT=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-parity 2>/dev/null | tail -1); fi
  echo "fit=$fit $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step %.3f  ramp %s  retimed %s  threads %s" % (d["ms_per_step"], d["ramp_up_step_ms"], d.get("retimed"), d.get("host_threads")))')"
  echo "   cpu.stat before: $before after: $(stat)"
done
