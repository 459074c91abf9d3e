#!/bin/bash
# per-class kernel times of the fp32 (parity mode) C3 step: python bench.py --precision f32, by_kernel table only
python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
print("ms/step %.3f  samples/s %.0f  loss %.6f" % (d["ms_per_step"], d["value"], d["final_loss"]))
for k,v in d["roofline"]["by_kernel"].items(): print("  %-12s %7.1f us x %2d = %.3f ms/step" % (k, 1e3*v["mean_ms"], v["launches"]//max(1,v["sampled_steps"]), v["ms_per_step"]))
'
