set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${PROF_TAG:-r03a}
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py > $O/bench_c3.json.log 2>$O/bench_c3.err
timeout -k 10 300 python bench.py --config c2 > $O/bench_c2.json.log 2>$O/bench_c2.err
timeout -k 10 400 python bench.py --config c5 --no-f32-parity > $O/bench_c5.json.log 2>$O/bench_c5.err
timeout -k 10 300 python bench.py --config c3 --batch 128 --no-f32-parity --no-cpu-baseline --no-kernel-events --steps 200 --warmup 20 > $O/bench_stock_b128.json.log 2>$O/bench_stock_b128.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-parity > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-f32-parity > $O/pmc_mfma.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-f32-parity > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-f32-parity > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); M=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/hbm_traffic.py $F $W 3x512_b8192_bf16 $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1 || true
python tools/pmc_mfma_summary.py $M > $O/pmc_mfma_summary.txt 2>&1 || true
cat $O/pmc_mfma_summary.txt $O/hbm_traffic.txt
find $O -name "*.csv" | head -30
du -sh $O
