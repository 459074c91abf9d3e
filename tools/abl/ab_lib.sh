#!/bin/bash
# A/B of library builds on one box: bench_gemm at the C3 shape and the whole bf16 step, alternating.  Usage: bash tools/abl/ab_lib.sh libA.so libB.so ...
for lib in default "$@"; do
  if [ $lib = default ]; then unset CODAE_HIP_LIB; else export CODAE_HIP_LIB=$PWD/$lib; fi
  echo "== $lib"; python tools/bench_gemm.py 8192 1536 1536 x 2>&1 | grep -v amdgpu | tail -3
done
for i in 1 2 3; do for lib in default "$@"; do
  if [ $lib = default ]; then unset CODAE_HIP_LIB; else export CODAE_HIP_LIB=$PWD/$lib; fi
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline']['by_kernel']; print('$lib ms %.4f  fwd %.2f dgrad %.2f wgrad %.2f loss %.2f' % (d['ms_per_step'], 1e3*b['gemm_fwd']['mean_ms'], 1e3*b['gemm_dgrad']['mean_ms'], 1e3*b['gemm_wgrad']['mean_ms'], 1e3*b['loss']['mean_ms']))"
done; done
