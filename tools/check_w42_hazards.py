#!/usr/bin/env python3
"""Build-time lint for srk_conv_w42.hip: its MFMAs are inline assembly, which the compiler's hazard recogniser does not see.
gfx950 needs two wait states between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB; this script compiles the
file to assembly and checks that no v_mfma in the wino42 kernels reads a register that one of the two preceding instructions
wrote (VALU destinations only: loads are covered by s_waitcnt, which the compiler does insert for inline-asm operands).
Exit status 1 on a violation.  Usage: python tools/check_w42_hazards.py [--keep]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "super-resolution_amd", "csrc", "srk_conv_w42.hip")


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    out = os.path.join(tempfile.mkdtemp(), "w42.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC), "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
           SRC, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        return 2
    bad = total = 0
    inside = False
    prev = []          # the last two real instructions: (mnemonic, dst registers)
    for line in open(out):
        t = line.strip()
        if re.match(r"_ZN.*wino42_kernel.*:", t):
            inside, prev = True, []
            continue
        if not inside or not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            if inside and t.startswith(".Lfunc_end"):
                inside = False
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if mn.startswith("v_mfma"):
            total += 1
            src = regs(ops[1]) | regs(ops[2])
            for k, (pm, pd) in enumerate(reversed(prev)):
                if pm.startswith("v_") and not pm.startswith("v_mfma") and (pd & src):
                    # an s_nop N in between provides N + 1 wait states
                    bad += 1
                    print("HAZARD: %s reads v%s written %d instruction(s) earlier by %s" % (t, sorted(pd & src), k + 1, pm))
        waits = 0
        if mn == "s_nop":
            waits = int(ops[0]) + 1
        dst = regs(ops[0]) if (ops and mn.startswith("v_") and not mn.startswith("v_cmp")) else set()
        if waits >= 2:
            prev = []
        else:
            prev = (prev + [(mn, dst)])[-2:]
    print("checked %d v_mfma instructions in the wino42 kernels: %d hazard(s)" % (total, bad))
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
