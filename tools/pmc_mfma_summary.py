"""Per-kernel-class MFMA utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass.

  clock   = GRBM_GUI_ACTIVE / 8 XCDs / duration            (MI355X_MICROARCH.md, DVFS give-back: reads high on short dispatches)
  busy    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): the share of all SIMD-cycles in which the matrix pipe
            was busy (the counter is MFMA-busy SIMD-cycles summed over the chip)
  achieved = algorithmic flops / duration

Usage: python tools/pmc_mfma_summary.py <counter_collection.csv> [M N K [GEMMs per grouped wgrad launch]]   (GEMM shape for the TFLOP/s column)"""
import csv, sys, collections

path = sys.argv[1]
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (8192, 1536, 1536)
rows = collections.defaultdict(dict)
meta = {}
for r in csv.DictReader(open(path)):
    d = r["Dispatch_Id"]
    rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[d] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


N_GROUPED = int(sys.argv[5]) if len(sys.argv) > 5 else 10      # GEMMs inside one grouped weight-gradient launch (C3: 10 layers)


def classify(name):
    if "gemm_f32x3_grouped" in name:
        return "gemm_wgrad_x%d" % N_GROUPED       # (fp32 engine: every layer's weight gradient in one launch of the bf16-plane kernel)
    if "gemm_f32x3_kernel" in name:        # fp32 products from three bf16 planes per operand: <A k-contiguous, B k-contiguous>
        a = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",") if "<" in name else ["?", "?"]
        return {("true", "true"): "gemm_f32x3_fwd", ("true", "false"): "gemm_f32x3_dgrad", ("false", "false"): "gemm_f32x3_wgrad"}.get(tuple(a[:2]), "gemm_f32x3")
    if "gemm_bf16_pipe_grouped" in name:
        return "gemm_wgrad_x%d" % N_GROUPED
    if "gemm_bf16_pipe_kernel" in name:
        a = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
        if a[7] == "true":
            return "gemm_wgrad"
        return {"1": "gemm_fwd", "2": "gemm_dgrad", "3": "loss_gemm"}.get(a[9], "gemm_other")
    for key, cls in (("clip_adam_tiled", "adam"), ("reduce_slabs", "slab_reduce"), ("gather_corrupt", "gather"), ("chain_step", "chain"),
                     ("gemm_bf16_grouped", "wgrad_grouped"), ("gemm_f32", "gemm_f32")):
        if key in name:
            return cls
    return None


agg = collections.defaultdict(list)
for d, c in rows.items():
    cls = classify(meta[d][0])
    if cls is None or "GRBM_GUI_ACTIVE" not in c:
        continue
    dur_ns = meta[d][1]
    clk = c["GRBM_GUI_ACTIVE"] / 8.0 / dur_ns                   # GHz
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    # SQ_VALU_MFMA_BUSY_CYCLES = MFMA-busy SIMD-cycles summed over the chip (checked: 8192 x 1536 x 1536 = 2.36 M
    # v_mfma_f32_16x16x32_bf16 x 16 cycles = 37.7 M, the counter reads 37.8 M) -> fraction of 1024 SIMDs x cycles
    agg[cls].append((dur_ns, clk, mfma / (1024.0 * cycles), c.get("SQ_BUSY_CU_CYCLES", 0.0) / (256.0 * cycles)))
fl = 2.0 * M * N * K
print("%-14s %6s %9s %8s %10s %10s %9s" % ("class", "n", "avg us", "GHz", "MFMA busy", "CU busy", "TFLOP/s"))
for cls, v in sorted(agg.items()):
    med = sorted(x[0] for x in v)[len(v) // 2]
    v = [x for x in v if x[0] <= 3 * med]          # (the first dispatch of a kernel carries its code-object load: 917 us once)
    n = len(v)
    dur = sum(x[0] for x in v) / n / 1e3
    print("%-14s %6d %9.1f %8.2f %9.1f%% %9.1f%% %9s" % (cls, n, dur, sum(x[1] for x in v) / n, 100 * sum(x[2] for x in v) / n, 100 * sum(x[3] for x in v) / n,
                                                   ("%.0f" % ((N_GROUPED if cls.startswith("gemm_wgrad_x") else 1) * fl / (dur * 1e-6) / 1e12))
                                                   if cls.startswith("gemm") or cls == "loss_gemm" else "-"))
