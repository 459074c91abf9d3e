#!/bin/bash
# one-rank data-parallel rehearsal (collectives forced) with the shipped library and with other builds, alternating.  Usage: bash tools/abl/ab_dp.sh lib.so ...
for i in 1 2 3; do for lib in default "$@"; do
  if [ $lib = default ]; then unset CODAE_HIP_LIB; else export CODAE_HIP_LIB=$PWD/$lib; fi
  BENCH_FORCE_DIST=1 CODAE_DP_FORCE_ALLREDUCE=1 MASTER_ADDR=127.0.0.1 python bench.py --no-cpu-baseline --no-f32-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline']['by_kernel']; print('$lib ms %.4f enqueue %.3f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']), {k: round(v['ms_per_step'],3) for k,v in b.items() if k.startswith('gemm') or k=='slab_reduce'})"
done; done
