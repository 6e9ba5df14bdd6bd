#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json.

    python tools/traffic_from_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_traffic

Per kernel name: launches, mean HBM bytes per launch.  Units/corrections as MI355X_MICROARCH.md (HBM): the counters
are in KB; on gfx950 FETCH_SIZE reports exactly 1/2 of a wide (16 B/lane) coalesced read stream, so it is doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import collections, csv, glob, json, os, re, sys


def load(d, counter):
    f = (glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv")))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def short(name):
    m = re.search(r"((?:conv3x3|wgrad)_[a-z0-9_]+_kernel(?:<[^>]*>)?|pack_kernel)", name)
    return m.group(1) if m else None


def main():
    fd, wd, out = sys.argv[1:4]
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res, table = {}, []
    for name in F:
        s = short(name)
        if not s:
            continue
        n, fs = F[name]
        nw, ws = W.get(name, [0, 0.0])
        fetch = 2.0 * fs * 1024 / n           # gfx950: x2 (see docstring)
        write = ws * 1024 / max(nw, 1)
        if s not in res or n > res[s]["launches"]:      # (un-demangled template variants share a short name: keep the dominant one)
            res[s] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                      "hbm_bytes_per_launch": fetch + write, "correction": "FETCH_SIZE x2 (gfx950), KB units"}
        table.append((s, n, fetch / 1e6, write / 1e6))
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as f:
        f.write("kernel | launches | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch (WRITE_SIZE)\n")
        for t in sorted(table, key=lambda t: -t[1] * (t[2] + t[3])):
            f.write("%s | %d | %.2f | %.2f\n" % t)
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
