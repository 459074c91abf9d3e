#!/bin/bash
# Profiles of the fp32 (parity) engine at C3: kernel stats and MFMA-busy / clock counters of `bench.py --precision f32`, the kernel
# accuracy / timing table and the float64 truth comparison.  Usage (GPU box, repo root): bash tools/prof_f32.sh ; outputs in gpurun_out/prof_f32
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_f32; mkdir -p $O
python tools/abl/f32x3_check.py > $O/f32x3_check.txt 2>&1
python tools/abl/f32_truth.py 2>&1 | grep -v "amdgpu\|UserWarning\|detach\|print(" > $O/f32_truth.txt
python bench.py --precision f32 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_f32.json.log 2>/dev/null
CODAE_F32_GEMM=native python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3_f32_native.json.log 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --precision f32 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py --precision f32 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/pmc.log 2>&1
cd $R
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); M=$(find $O/pmc -name "*counter_collection.csv" | head -1)
cp "$S" $O/kernel_stats_bench_c3_f32.csv
python tools/pmc_mfma_summary.py "$M" > $O/pmc_mfma_summary_f32.txt 2>&1
cat $O/pmc_mfma_summary_f32.txt; head -12 $O/kernel_stats_bench_c3_f32.csv | cut -c1-160; tail -4 $O/f32_truth.txt
