#!/bin/bash
# A/B of one environment setting on the whole bf16 C3 step, alternating runs on one box.  Usage: bash tools/abl/ab_env.sh VAR=value [VAR2=value2 ...]
for i in 1 2 3; do for setting in "" "$@"; do
  line=$(env $setting python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-parity 2>/dev/null | tail -1)
  echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline']['by_kernel']; print('%-28s ms %.4f  fwd %.2f dgrad %.2f wgrad %.2f loss %.2f' % ('${setting:-default}', d['ms_per_step'], 1e3*b['gemm_fwd']['mean_ms'], 1e3*b['gemm_dgrad']['mean_ms'], 1e3*b['gemm_wgrad']['mean_ms'], 1e3*b['loss']['mean_ms']))"
done; done
