#!/bin/bash
# SQ counters of the forward form of gemm_f32x3 (8192 x 1536 x 1536), one pass per counter group.  Usage (GPU box, repo root): bash tools/abl/pmc_x3.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_x3; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/abl/f32x3_time.py > $O/g$i.log 2>&1
  F=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_f32x3" in r["Kernel_Name"]:
        kind = "fwd (KC,KC)" if "true, true" in r["Kernel_Name"] or "Lb1ELb1E" in r["Kernel_Name"] else "dgrad (KC,KS)"
        agg[kind][(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for kind, d in agg.items():
    per = collections.defaultdict(list)
    for (disp, name), vals in d.items(): per[name].append(sum(vals))
    print(kind, " ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(per.items())))
PY
done
