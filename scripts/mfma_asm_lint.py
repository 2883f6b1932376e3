"""ISA lint for the inline-asm MFMAs of the fused kernels (fused_kernel.h, NMFX_G1_ASM).

hipcc's hazard recogniser does not look into inline asm: to it an asm `v_mfma_f32_32x32x2_f32 v[a:b], ...` is an instruction whose result is there at once.  A 16-pass MFMA
delivers it 18 wait states later.  The kernel keeps its own readers away (mfma_settle), but the COMPILER may put a register copy, a spill or a re-homing `v_mov` of the
accumulator tuple right behind such an MFMA -- it did, in the chain kernels (functors 7 / 8 / ...: `v_mov_b64 v[48:49], v[32:33]` between the 127th and the 128th MFMA of
a chain, to free v[18:33] for the next tile's partial-S loads), and every K > 256 test failed on the hardware with code that read right (round 6).  This script finds that
in the BUILT objects: for every `v_mfma` with a VGPR destination it walks the following instructions until 18 wait states have passed (s_nop N = N + 1, an MFMA = its 16
passes, anything else = 1) and reports any instruction that touches a register of the destination tuple -- except the next MFMA of the same chain (same destination and
SrcC: back-to-back accumulation needs no wait).

    python scripts/mfma_asm_lint.py [objdir] [--kernel-filter substr]      exit 1 when something is found
tests/test_abi_and_host.py runs it over the shipped objects."""
import glob
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
NEED = 18


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def lint_code_object(co_bytes, kfilter=None):
    with tempfile.NamedTemporaryFile(suffix=".co") as t:
        t.write(co_bytes)
        t.flush()
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", t.name], capture_output=True, text=True).stdout
    return lint_text(dis, kfilter)


def lint_text(dis, kfilter=None):
    """the walk itself, on llvm-objdump text -> (findings, number of VGPR-destination MFMAs seen)"""
    findings, kernel, n_mfma = [], None, 0
    pending = []   # [dst regs (set), dst operand text, wait states since, mnemonic line]
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel, pending = m.group(1), []
            continue
        if kernel is None or (kfilter and kfilter not in kernel):
            continue
        ins = line.split("//")[0].strip()
        if not ins or ins.endswith(":"):
            continue
        parts = ins.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        touched = regs(ops)
        is_mfma = mn.startswith("v_mfma")
        for p in pending:
            hit = touched & p[0]
            if hit:
                chain = is_mfma and ops.split(",")[0].strip() == p[1] and ops.split(",")[-1].strip().split()[0] == p[1]
                if not chain and p[2] < NEED:
                    findings.append((kernel, p[3], ins, p[2]))
        step = 16 if is_mfma else (int(ops.split()[0], 0) + 1 if mn == "s_nop" else 1)
        for p in pending:
            p[2] += step
        pending = [p for p in pending if p[2] < NEED]
        if mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = []   # what follows an unconditional jump in the layout is another block (conditional branches: the fall-through is followed, the taken side is not)
            continue
        if is_mfma:
            dst = ops.split(",")[0].strip()
            if dst.startswith("v"):
                n_mfma += 1
                pending.append([regs(dst), dst, 0, ins])
    return findings, n_mfma


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kfilter = sys.argv[sys.argv.index("--kernel-filter") + 1] if "--kernel-filter" in sys.argv else None
    if kfilter and kfilter in args:
        args.remove(kfilter)
    target = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmf_toolbox_amd", "csrc", "_obj")
    files = sorted(glob.glob(os.path.join(target, "*.o"))) if os.path.isdir(target) else [target]
    total, nm = [], 0
    for f in files:
        if os.path.isdir(target) and "fused" not in os.path.basename(f):
            continue
        for co in code_objects(f):
            fnd, n = lint_code_object(co, kfilter)
            total += fnd
            nm += n
    seen = set()
    for k, a, b, ws in total:
        key = (k, a.split(",")[0], b.split()[0])
        if key in seen:
            continue
        seen.add(key)
        print("%s\n    %s\n    -> %s   (%d wait states later, %d needed)" % (k, a, b, ws, NEED))
    print("%d VGPR-destination MFMAs checked, %d finding(s) in %d kernel(s)" % (nm, len(total), len({k for k, *_ in total})))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
