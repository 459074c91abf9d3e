# forward-form GEMM (N = K = 1536, then 384) at several batch sizes: 128 x 128 tiles (CODAE_SMALL_TILE_MAX=0) vs 64 x 64 (=100000)
for shape in "256 1536 1536" "512 1536 1536" "1024 1536 1536" "2048 1536 1536" "4096 1536 1536" "1024 384 384" "4096 384 384" "8192 384 384"; do
  for mx in 0 100000; do
    CODAE_SMALL_TILE_MAX=$mx timeout -k 10 100 python tools/bench_gemm.py $shape s > gpurun_out/sweep_tmp.log 2>&1
    echo "M N K = $shape  small_tile_max=$mx  $(grep '^fwd' gpurun_out/sweep_tmp.log)"
  done
done
