for B in 128 512 1024 2048 4096; do
  for mode in auto x3 native; do
    if [ $mode = auto ]; then unset CODAE_F32_GEMM; else export CODAE_F32_GEMM=$mode; fi
    python bench.py --precision f32 --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $B mode $mode: %.3f ms/step' % d['ms_per_step'])"
  done
done
