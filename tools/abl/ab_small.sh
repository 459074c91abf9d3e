for i in 1 2 3; do for lib in default tools/abl/_libs/libpt.so; do
  if [ $lib = default ]; then unset CODAE_HIP_LIB; else export CODAE_HIP_LIB=$PWD/$lib; fi
  a=$(python bench.py --config c3 --batch 128 --no-f32-parity --no-cpu-baseline --no-kernel-events --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python bench.py --config c3 --batch 1024 --no-f32-parity --no-cpu-baseline --no-kernel-events --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  c=$(python bench.py --config c2 --no-f32-parity --no-cpu-baseline --no-kernel-events --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$lib  b128 $a  b1024 $b  c2 $c"
done; done
