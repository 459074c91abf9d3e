# chain vs per-layer step time over the batch size (3 x 128 stack): where CHAIN_MAX_ROWS should sit
for B in 2048 4096 8192; do
  for nc in 0 1; do
    if [ $nc = 1 ]; then export CODAE_NO_CHAIN=1; else unset CODAE_NO_CHAIN; fi
    timeout -k 10 200 python bench.py --config c2 --batch $B --no-f32-parity --no-cpu-baseline --no-kernel-events > gpurun_out/sweep_${B}_$nc.log 2>&1
    python - <<PY
import json;d=json.loads(open("gpurun_out/sweep_${B}_$nc.log").read().strip().splitlines()[-1]);print("B=$B no_chain=$nc ms/step", round(d["ms_per_step"],4))
PY
  done
done
