import json, sys
for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print(p, round(d["ms_per_step"], 4), "host", round(d["host_enqueue_ms_per_step"], 3),
          {k: round(v["mean_ms"] * 1e3, 1) for k, v in d["roofline"]["by_kernel"].items()})
