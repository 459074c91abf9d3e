#!/usr/bin/env python3
"""Build-time guard on the gfx950 code of the bf16 GEMM kernels (run by __graft_entry__.build()).

Why: the k-strided operand forms issue their transposed LDS reads (ds_read_b64_tr_b16) as inline asm, hidden from the
compiler's s_waitcnt bookkeeping, and wait for them with hand-placed `s_waitcnt lgkmcnt`.  In round 1 a change that let
the compiler's waitcnt pass see those waits made it move one relative to the asm reads: MFMAs consumed fragments that had
not landed and a GPU test aborted (negative result (a)).  Nothing but a runtime test would catch a
toolchain bump doing the same, so this script disassembles the code object that SHIPS (llvm-objdump on the fat binary
inside libcodae_hip.so) and checks, for every bf16 GEMM kernel:

  1. dataflow: walking each K loop (twice, for the wrap-around), no instruction reads a register written by a
     ds_read_b64_tr_b16 before an `s_waitcnt lgkmcnt(N)` has retired that read (LDS operations return in order, so
     lgkmcnt(N) retires all but the N youngest);
  2. no `s_waitcnt vmcnt(0)` inside a K loop of the phase-pipelined kernel (the LDS-DMA prefetch queue must never be
     drained there: counted vmcnt only);
  3. the k-strided instantiations do contain transposed reads (the check is not vacuous), no GEMM kernel touches scratch;
  4. (EVERY kernel of the library, not only the GEMMs) no packed-fp32 VALU op - v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 -
     whose LO result takes the HI half of src1 or src2 (op_sel bit 1 or 2 set on a VGPR source).  On this MI355X pool that
     form returns the lo result of lanes 48-63 as if the re-routed half were ZERO whenever the SIMD's matrix pipe is busy
     (MFMAs of the same wave or of its neighbours; never without them; op_sel on src0 and op_sel_hi = 0 are exact):
     tools/abl/pk_fma_opsel_repro.hip shows it stand-alone, DESIGN.md section 5d has the numbers.  The compiler forms these
     instructions by itself when it SLP-packs a scalar reduction chain (`sq += d * d` in the fused-loss epilogue became
     v_pk_fma_f32 ... op_sel:[0,0,1]: one workgroup's loss sum came out one term short in 4-28 % of the launches).

Usage: python tools/check_isa.py [path/to/libcodae_hip.so]     exit code 0 = all kernels pass
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "mui-deepautoencoder_amd", "codae", "hip", "libcodae_hip.so")

REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))")


def regs_of(operand):
    """set of ('v'|'a', index) named by one operand string."""
    out = set()
    for m in REG.finditer(operand):
        kind = m.group(1)
        if m.group(2) is not None:
            out.update((kind, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((kind, int(m.group(4))))
    return out


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="codae_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
        if not objs:
            raise RuntimeError("no gfx950 code object inside %s" % lib)
        text = ""
        for o in objs:
            text += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", o], check=True,
                                   capture_output=True, text=True).stdout
        return text
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    """{mangled: 'kernel<a, b, ...>'} from the Itanium-mangled integer / bool template arguments (no c++filt needed)."""
    out = {}
    for n in names:
        m = re.search(r"(gemm_bf16(?:_pipe|_snake)?_kernel)I((?:L[ib]\d+E)+)E", n)
        mg = re.search(r"gemm_bf16_grouped_kernelILi(\d+)ELi(\d+)EE", n)
        if m:
            out[n] = "%s<%s>" % (m.group(1), ", ".join(re.findall(r"L[ib](\d+)E", m.group(2))))
        elif "gemm_bf16_pipe_grouped_kernel" in n:
            # every weight gradient of a step in one launch: the 256 x 192 pipelined tile, both operands k-strided, fp32 out
            out[n] = "gemm_bf16_pipe_kernel<256, 192, 4, 2, 4, 1, 1, 1, 0, 0>"
        elif mg:
            # grouped weight gradients of narrow stacks: the one-barrier tile, 2 x 2 waves, both operands k-strided, fp32 out
            out[n] = "gemm_bf16_kernel<%s, %s, 2, 2, 1, 1, 1, 0, 2>" % (mg.group(1), mg.group(2))
        else:
            out[n] = n
    return out


def functions(text):
    """{mangled name: [(address, mnemonic, operand string)]}"""
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9a-fA-F]+):(.*)$", line)
        if m:
            args = m.group(2)
            t = re.search(r"<[^>]*\+0x[0-9a-fA-F]+>", m.group(4))      # branch target, printed behind the raw bytes
            if t:
                args += " " + t.group(0)
            cur.append((int(m.group(3), 16), m.group(1), args))
    return funcs


def loops_with_mfma(insns):
    """[(first index, last index)] of innermost backward-branch regions that contain MFMAs."""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(insns)}
    base = insns[0][0] if insns else 0
    out = []
    for i, (a, op, args) in enumerate(insns):
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
        if not m:
            continue
        tgt = base + int(m.group(1), 16)
        j = addr_to_idx.get(tgt)
        if j is not None and j <= i and any(o.startswith("v_mfma") for _, o, _ in insns[j:i + 1]):
            out.append((j, i))
    # innermost only; a region that also stores to global memory is an epilogue loop laid out around tail code, not a K loop
    inner = [r for r in out if not any(o != r and r[0] <= o[0] and o[1] <= r[1] for o in out)]
    return [r for r in inner if not any(o.startswith(("global_store", "buffer_store")) for _, o, _ in insns[r[0]:r[1] + 1])]


PK_F32 = ("v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32")


def packed_opsel_errors(insns):
    """rule 4: packed fp32 ops whose lo result is routed from the hi half of src1 / src2 (a VGPR pair)."""
    errs = []
    for a, o, ar in insns:
        if o not in PK_F32:
            continue
        m = re.search(r"op_sel:\[([01,]+)\]", ar)
        if not m:
            continue
        sel = [int(x) for x in m.group(1).split(",")]
        ops = [x.strip() for x in re.split(r",(?![^\[]*\])", ar.split(" op_sel")[0].split(" neg_")[0])]
        srcs = ops[1:]                                   # ops[0] is the destination
        for i in range(1, len(sel)):
            if sel[i] and i < len(srcs) and srcs[i].startswith("v"):
                errs.append("%s at 0x%x: lo result takes the hi half of src%d (%s): wrong in lanes 48-63 beside MFMAs (DESIGN.md 5d)"
                            % (o, a, i, ar))
    return errs


def check_kernel(name, pretty, insns):
    errs = []
    is_snake = "gemm_bf16_snake_kernel" in pretty
    is_pipe = "gemm_bf16_pipe_kernel" in pretty or is_snake          # (phase-pipelined: counted vmcnt only in the K loop)
    args = [a.strip() for a in pretty[pretty.index("<") + 1:pretty.rindex(">")].split(",")]
    if is_snake:
        a_mode, b_mode = int(args[3]), int(args[4])
    else:
        a_mode, b_mode = (int(args[5]), int(args[6])) if is_pipe else (int(args[4]), int(args[5]))
    dbg = int(args[8]) if (is_pipe and not is_snake and len(args) > 8) else 0
    k_strided = a_mode == 1 or b_mode == 1
    n_tr = sum(1 for _, o, _ in insns if o == "ds_read_b64_tr_b16")
    if k_strided and n_tr == 0:
        errs.append("k-strided instantiation without ds_read_b64_tr_b16")
    if any(o.startswith("scratch_") for _, o, _ in insns):
        errs.append("uses scratch (register spill)")
    loops = loops_with_mfma(insns)
    if not loops and not dbg:
        errs.append("no K loop with MFMAs found")
    # check 1 runs over every K loop twice (wrap-around) and once over the whole kernel in address order (prologue, peeled
    # last K-tile, epilogue: straight-line code between the loops)
    walks = [insns[lo:hi + 1] * 2 for lo, hi in loops] + [insns]
    # the 4-stage instantiations of the one-barrier kernel (last template argument 4) keep three K-tiles in flight too
    deep = (not is_pipe) and len(args) >= 9 and args[8] == "4"
    for lo, hi in loops:
        if (is_pipe and not dbg) or deep:
            for a, o, ar in insns[lo:hi + 1]:
                if o == "s_waitcnt" and re.search(r"vmcnt\(0\)", ar):
                    errs.append("s_waitcnt vmcnt(0) inside the K loop at 0x%x" % a)
    for walk in walks:
        if not k_strided:
            continue
        lds_queue = []           # outstanding LGKM operations in issue order: set of destination registers each
        for a, o, ar in walk:
            ops = [x.strip() for x in ar.split(",")] if ar else []
            if o == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", ar)
                if m:
                    keep = int(m.group(1))
                    lds_queue = lds_queue[len(lds_queue) - keep:] if keep else []
                continue
            if o.startswith("ds_") or o.startswith("s_load") or o.startswith("s_buffer_load"):
                dst = regs_of(ops[0]) if (o.startswith("ds_read") and ops) else set()
                pending = set().union(*[q for q in lds_queue if q["tr"]] and [q["regs"] for q in lds_queue if q["tr"]] or [set()])
                srcs = set().union(*[regs_of(x) for x in (ops[1:] if o.startswith("ds_read") else ops)]) if ops else set()
                if srcs & pending:
                    errs.append("%s at 0x%x reads a transposed-read register before its wait" % (o, a))
                lds_queue.append({"regs": dst, "tr": o == "ds_read_b64_tr_b16"})
                continue
            pending = set()
            for q in lds_queue:
                if q["tr"]:
                    pending |= q["regs"]
            if not pending:
                continue
            # every register the instruction names except a pure destination of a non-accumulating op is a read; being
            # conservative (treat all named registers as read) only risks false alarms, never a missed hazard
            used = set().union(*[regs_of(x) for x in ops]) if ops else set()
            if used & pending:
                errs.append("%s at 0x%x touches %s before the s_waitcnt lgkmcnt that covers its ds_read_b64_tr_b16"
                            % (o, a, sorted(used & pending)[:4]))
                break
    return errs


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    text = disassemble(lib)
    funcs = functions(text)
    gemm = [n for n in funcs if "gemm_bf16" in n and "kernel" in n]
    pretty = demangle(gemm)
    n_ks, bad = 0, 0
    n_pk = 0
    for n in sorted(funcs):                      # rule 4: every function of the code object
        n_pk += sum(1 for _, o, _ in funcs[n] if o in PK_F32)
        errs = packed_opsel_errors(funcs[n])
        if errs:
            bad += 1
            for e in errs[:6]:
                print("FAIL %s: %s" % (n[:110], e))
    for n in sorted(gemm):
        p = pretty[n]
        if "<" not in p:
            continue
        errs = check_kernel(n, p, funcs[n])
        short = p[p.index("gemm_bf16"):p.rindex(">") + 1]
        if "gemm_bf16_pipe_kernel" in p:
            a = [x.strip() for x in p[p.index("<") + 1:p.rindex(">")].split(",")]
            n_ks += int(a[5] == "1" or a[6] == "1")
        if "gemm_bf16_snake_kernel" in p:
            a = [x.strip() for x in p[p.index("<") + 1:p.rindex(">")].split(",")]
            n_ks += int(a[3] == "1" or a[4] == "1")
        if errs:
            bad += 1
            for e in errs:
                print("FAIL %s: %s" % (short, e))
    print("check_isa: %d bf16 GEMM kernels (%d k-strided pipelined instantiations), %d functions / %d packed-fp32 ops scanned "
          "for hi-half routing, %d failing" % (len(gemm), n_ks, len(funcs), n_pk, bad))
    if n_ks == 0:
        print("FAIL: no k-strided pipelined instantiation found - the guard would be vacuous")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
