# two bench lines + the FETCH_SIZE / WRITE_SIZE passes of the C3 step: HBM bytes per launch of the GEMM classes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_wg; mkdir -p $O
cd $R; for i in 1 2; do python bench.py --no-f32-parity --no-cpu-baseline > $O/bench_$i.json 2>&1; done
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-f32-parity > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-f32-parity > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/hbm_traffic.py $F $W 3x512_b8192_bf16 $O/hbm_traffic.json | grep -E "wgrad|fwd|dgrad"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); by=d["roofline"]["by_kernel"]; print(f, round(d["ms_per_step"],4), {k:round(v["mean_ms"]*1e3,1) for k,v in by.items()})
PY
