#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel matrix-pipe and wave-state figures (profiles/r02*_mfma.*).

    python tools/mfma_from_pmc.py gpurun_out/r02f/pmc_mfma gpurun_out/r02f/pmc_sq profiles/r02f_mfma

Per kernel (summed over its dispatches):
  MfmaUtil      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/XCCs x SIMDs): rocprofv3's own MfmaUtil expression with the gfx950
                  facts of MI355X_MICROARCH.md filled in (1024 SIMDs; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs);
                  BUSY_CYCLES counts shader cycles, 64 per v_mfma_f32_32x32x2_f32.
  MOPS_F32*512  = executed fp32 MFMA FLOPs (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512), / duration = executed TFLOP/s
  clock         = GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches shorter than ~0.3 ms: guide, DVFS give-back)
  clock_slope   = least-squares slope of GRBM_GUI_ACTIVE/8 against duration over the kernel's dispatches (needs >= 3 distinct durations): the
                  shader clock DURING the kernel, free of the fixed ~10 us of GUI_ACTIVE that surrounds a short dispatch;
                  MfmaUtil_at_slope = BUSY_CYCLES / (1024 x duration x clock_slope), the pipe occupancy at the clock the kernel really ran at
  MOPS_F16*512  = executed 16-bit MFMA FLOPs (SQ_INSTS_VALU_MFMA_MOPS_F16 x 512) when that counter was collected
  wave states   = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES; MFMA/VALU co-execution
"""
import collections, csv, glob, json, os, re, sys


def load(d):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            agg[k]["_n"] += 1
            agg[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            PER[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), float(r["Counter_Value"]) / 8))
    return agg


PER = collections.defaultdict(list)


def slope_clock(name):
    """GHz from the regression of GUI_ACTIVE/8 (cycles) on duration (ns) over one kernel's dispatches; None when they do not spread."""
    pts = PER.get(name, [])
    if len(pts) < 3:
        return None
    mx = sum(p[0] for p in pts) / len(pts); my = sum(p[1] for p in pts) / len(pts)
    sxx = sum((p[0] - mx) ** 2 for p in pts)
    if sxx < len(pts) * (0.15 * mx) ** 2:
        return None
    return sum((p[0] - mx) * (p[1] - my) for p in pts) / sxx


def short(name):
    m = re.search(r"((?:conv3x3|wgrad)_[a-z0-9_]+_kernel(?:<[^>]*>)?|pack_kernel)", name)
    if not m:
        return None
    t = re.search(r"_kernelI(.*?)EEv", name)                                   # un-demangled template arguments (16-bit kernels)
    if t and "<" not in m.group(1):
        a = t.group(1).replace("DF16_", "fp16,").replace("DF16b", "bf16,").replace("Lb0E", "0,").replace("Lb1E", "1,")
        a = re.sub(r"Li(\d+)E", r"\1,", a)
        return m.group(1) + "<" + a.rstrip(",") + ">"
    return m.group(1)


def main():
    dm, ds, out = sys.argv[1:4]
    M, S = load(dm), load(ds)
    XCC, SIMDS = 8, 1024
    res, rows = {}, []
    for name, a in M.items():
        s = short(name)
        if not s or a["_ns"] < 2e5:
            continue
        dur = a["_ns"] * 1e-9
        gui = a["GRBM_GUI_ACTIVE"] / XCC
        util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS) if gui else 0.0
        flops = (a["SQ_INSTS_VALU_MFMA_MOPS_F32"] + a.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)) * 512
        e = {"launches": int(a["_n"]), "avg_us_under_pmc": dur / a["_n"] * 1e6, "MfmaUtil": util, "executed_f32_mfma_tflops": flops / dur / 1e12,
             "clock_GHz": gui / dur * 1e-9}
        if a.get("SQ_INSTS_VALU_MFMA_MOPS_F16"):
            e["executed_mfma_tflops_f16"] = a["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512 / dur / 1e12
        ck = slope_clock(name)
        if ck:
            e["clock_slope_GHz"] = ck
            e["MfmaUtil_at_slope_clock"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * dur * ck * 1e9)
        b = S.get(name)
        if b and b.get("SQ_WAVE_CYCLES"):
            wc = b["SQ_WAVE_CYCLES"]
            e.update(wait_any=b["SQ_WAIT_ANY"] / wc, wait_inst_any=b["SQ_WAIT_INST_ANY"] / wc, active_inst_any=b["SQ_ACTIVE_INST_ANY"] / wc,
                     valu_mfma_coexec_over_mfma_busy=(b["SQ_VALU_MFMA_COEXEC_CYCLES"] / a["SQ_VALU_MFMA_BUSY_CYCLES"]) if a["SQ_VALU_MFMA_BUSY_CYCLES"] else None)
        res[s] = e
        rows.append((s, e))
    rows.sort(key=lambda t: -t[1]["avg_us_under_pmc"] * t[1]["launches"])
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as f:
        f.write("kernel | launches | avg us (under PMC) | MfmaUtil | executed MFMA TFLOP/s | clock GHz (GUI/duration) | clock GHz (slope) | MfmaUtil at slope clock | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY (of wave cycles)\n")
        for s, e in rows:
            f.write("%s | %d | %.1f | %.3f | %.1f | %.2f | %s | %s | %s | %s | %s\n" % (
                s, e["launches"], e["avg_us_under_pmc"], e["MfmaUtil"], e["executed_f32_mfma_tflops"], e["clock_GHz"],
                *["%.3f" % e[k] if k in e else "-" for k in ("clock_slope_GHz", "MfmaUtil_at_slope_clock", "wait_any", "wait_inst_any", "active_inst_any")]))
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
