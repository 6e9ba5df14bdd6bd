#!/usr/bin/env python3
"""Build-time lint for srk_conv_w42.hip, run on the OBJECT THAT IS LINKED INTO libsrk.so (build/srk_conv_w42.o: the device code is
extracted with llvm-objdump --offloading and disassembled), so the flags checked are the flags shipped.

1. The kernel's MFMAs are inline assembly, which the compiler's hazard recogniser does not see.  gfx950 needs two wait states
   between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB: no v_mfma in the wino42 kernels may read a register
   that one of the two preceding instructions wrote (VALU destinations only: loads are covered by s_waitcnt, which the compiler
   does insert for inline-asm operands).  The window is reset at branches and at symbols.
2. The stage barrier's hand-written counted `s_waitcnt vmcnt(N)` encodes a count of vector-memory operations in flight.  Weights
   through the LDS ring (W42_LDSW=1, the default): N = 6, the six youngest are the weight DMA instructions of one channel pair
   (buffer_load_dwordx4 ... lds) and the next older one is the last halo piece of the next chunk.  Weights into registers
   (W42_LDSW=0): N = 12 register loads of one phase (buffer_load_dwordx2) behind a halo DMA piece.  Either way "all but the N
   youngest" retires exactly every DMA piece of the next chunk; a schedule edit that changes the count fails here instead of
   racing on the GPU.  (The check cannot tell a weight DMA from a halo piece by its mnemonic: it relies on the count.)
Exit status 1 on a violation.  Usage: python tools/check_w42_hazards.py [object file]"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "super-resolution_amd", "csrc", "build", "srk_conv_w42.o")
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, "w42.o")
    shutil.copy(obj, local)
    r = subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True)
    dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
    if r.returncode or not dev:
        sys.stderr.write(r.stdout + r.stderr + "\nno device code object in %s\n" % obj)
        return None
    r = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, dev[0])], capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        return None
    return r.stdout


def check_w22():
    """srk_wgrad_w22.hip, row-owner form: its MFMAs are builtins (the compiler sees them), but its packed operand transform is inline
    assembly, whose results the hazard recogniser cannot classify; the wino24 kernel's MFMAs are inline assembly as well.  Same rule as 1.:
    no v_mfma of these kernels may read a register that one of the two preceding instructions wrote."""
    obj = os.path.join(ROOT, "super-resolution_amd", "csrc", "build", "srk_wgrad_w22.o")
    if not os.path.exists(obj):
        sys.stderr.write("%s not found: build the library first\n" % obj)
        return 2
    text = disassemble(obj)
    if text is None:
        return 2
    bad = total = 0
    inside, prev = False, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            inside, prev = ("wino22_kernel" in m.group(1) or "wino24_kernel" in m.group(1)), []
            continue
        if not inside:
            continue
        t = line.split("//")[0].strip()
        if not t:
            continue
        parts = t.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if mn.startswith("s_cbranch") or mn == "s_branch" or mn == "s_endpgm":
            prev = []
            continue
        if mn.startswith("v_mfma"):
            total += 1
            src = regs(ops[1]) | regs(ops[2])
            for k, (pm, pd) in enumerate(reversed(prev)):
                if pm.startswith("v_") and not pm.startswith("v_mfma") and (pd & src):
                    bad += 1
                    print("HAZARD (wino22 / wino24): %s reads v%s written %d instruction(s) earlier by %s" % (t, sorted(pd & src), k + 1, pm))
        waits = int(ops[0]) + 1 if mn == "s_nop" else 0
        dst = regs(ops[0]) if (ops and mn.startswith("v_") and not mn.startswith("v_cmp")) else set()
        prev = [] if waits >= 2 else (prev + [(mn, dst)])[-2:]
    print("checked %d v_mfma instructions in the wino22 / wino24 kernels of super-resolution_amd/csrc/build/srk_wgrad_w22.o: %d violation(s)" % (total, bad))
    return 1 if bad or total == 0 else 0


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else OBJ
    if not os.path.exists(obj):
        sys.stderr.write("%s not found: build the library first (make -C super-resolution_amd/csrc)\n" % obj)
        return 2
    text = disassemble(obj)
    if text is None:
        return 2
    if len(sys.argv) <= 1:
        rc22 = check_w22()
        if rc22:
            return rc22
    bad = total = waits_checked = 0
    inside = False
    prev = []          # the last two real instructions: (mnemonic, dst registers)
    vmem = []          # vector-memory instructions of the current kernel in program order
    prev_text = ""
    last_wait = None   # the s_waitcnt ... vmcnt(N) directly in front of the current instruction, if any
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            inside = "wino42_kernel" in m.group(1) or "wino42_chain_kernel" in m.group(1)
            prev, vmem, last_wait = [], [], None
            continue
        if not inside:
            continue
        t = line.split("//")[0].strip()
        if not t:
            continue
        parts = t.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if mn.startswith("s_cbranch") or mn == "s_branch" or mn == "s_endpgm":
            prev = []
            continue
        if mn.startswith("buffer_") or mn.startswith("global_") or mn.startswith("flat_"):
            vmem.append(t)
        if mn == "s_barrier" and last_wait is not None:
            # the stage barrier: the hand-written counted wait sits directly in front of it (the compiler's own vmcnt waits, in
            # front of the first use of a loaded register, are not followed by a barrier)
            waits_checked += 1
            n = int(re.search(r"vmcnt\((\d+)\)", last_wait).group(1))
            young, older = vmem[-n:] if n else [], vmem[-n - 1:len(vmem) - n]
            regs_form = all(v.startswith("buffer_load_dwordx2") and " lds" not in v for v in young)          # W42_LDSW=0: 12 register loads
            ring_form = n == 6 and all(v.startswith("buffer_load_dwordx4") and " lds" in v for v in young)      # W42_LDSW=1: 6 ring DMA instructions
            ok = len(young) == n and (regs_form or ring_form) and len(older) == 1 and " lds" in older[0]
            if n and not ok:
                bad += 1
                print("VMCNT: `%s` + s_barrier does not sit behind [halo DMA piece, %d weight loads / ring DMA instructions]; last vector-memory ops: %s"
                      % (last_wait, n, [v.split()[0] + (" lds" if " lds" in v else "") for v in vmem[-n - 1:]]))
        # (counted, vmcnt only; a counted wait right behind a `vmcnt(0)` is the compiler's own, weaker, and means nothing)
        counted = re.fullmatch(r"s_waitcnt vmcnt\((\d+)\)", t) and not t.endswith("(0)") and prev_text != "s_waitcnt vmcnt(0)"
        last_wait = t if counted else None
        prev_text = t
        if mn.startswith("v_mfma"):
            total += 1
            src = regs(ops[1]) | regs(ops[2])
            for k, (pm, pd) in enumerate(reversed(prev)):
                if pm.startswith("v_") and not pm.startswith("v_mfma") and (pd & src):
                    bad += 1
                    print("HAZARD: %s reads v%s written %d instruction(s) earlier by %s" % (t, sorted(pd & src), k + 1, pm))
        waits = 0
        if mn == "s_nop":
            waits = int(ops[0]) + 1          # an s_nop N provides N + 1 wait states
        dst = regs(ops[0]) if (ops and mn.startswith("v_") and not mn.startswith("v_cmp")) else set()
        if waits >= 2:
            prev = []
        else:
            prev = (prev + [(mn, dst)])[-2:]
    print("checked %d v_mfma instructions and %d counted vmcnt waits in the wino42 kernels of %s: %d violation(s)"
          % (total, waits_checked, os.path.relpath(obj, ROOT), bad))
    return 1 if bad or total == 0 or waits_checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
