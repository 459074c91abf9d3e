for n in 1 2 3 4; do
CODAE_HIP_LIB=$PWD/tools/abl/libcodae_ch$n.so timeout -k 10 200 python bench.py --config c2 --no-f32-parity > gpurun_out/chain_abl$n.log 2>&1
python - <<PY
import json;d=json.loads(open("gpurun_out/chain_abl$n.log").read().strip().splitlines()[-1]);print("abl $n", d["ms_per_step"],{k:round(v["mean_ms"]*1e3,1) for k,v in d["roofline"]["by_kernel"].items()})
PY
done
