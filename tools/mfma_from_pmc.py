#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel matrix-pipe and wave-state figures (profiles/r02*_mfma.*).

    python tools/mfma_from_pmc.py gpurun_out/r02f/pmc_mfma gpurun_out/r02f/pmc_sq profiles/r02f_mfma

Per kernel (summed over its dispatches):
  MfmaUtil      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/XCCs x SIMDs): rocprofv3's own MfmaUtil expression with the gfx950
                  facts of MI355X_MICROARCH.md filled in (1024 SIMDs; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs);
                  BUSY_CYCLES counts shader cycles, 64 per v_mfma_f32_32x32x2_f32.
  MOPS_F32*512  = executed fp32 MFMA FLOPs (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512), / duration = executed TFLOP/s
  clock         = GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches shorter than ~0.3 ms: guide, DVFS give-back)
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
    return agg


def short(name):
    m = re.search(r"((?:conv3x3|wgrad)_[a-z0-9_]+_kernel(?:<[^>]*>)?|pack_kernel)", name)
    return m.group(1) if m else None


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
        flops = a["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
        e = {"launches": int(a["_n"]), "avg_us_under_pmc": dur / a["_n"] * 1e6, "MfmaUtil": util, "executed_f32_mfma_tflops": flops / dur / 1e12,
             "clock_GHz": gui / dur * 1e-9}
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
        f.write("kernel | launches | avg us (under PMC) | MfmaUtil | executed f32 MFMA TFLOP/s | clock GHz | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY (of wave cycles)\n")
        for s, e in rows:
            f.write("%s | %d | %.1f | %.3f | %.1f | %.2f | %s | %s | %s\n" % (
                s, e["launches"], e["avg_us_under_pmc"], e["MfmaUtil"], e["executed_f32_mfma_tflops"], e["clock_GHz"],
                *["%.3f" % e[k] if k in e else "-" for k in ("wait_any", "wait_inst_any", "active_inst_any")]))
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
