# what binds the 16-bit chain kernel's epilogue: variants with the stores dropped (exp1), the elementwise work dropped (exp2), both (exp3)  [results wrong]
set -o pipefail
out=gpurun_out/r04o
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > /dev/null 2>&1
for r in 1 2; do for v in base exp1 exp2 exp3; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  echo "== h16 $v round $r" >> $out/ab_h16.txt
  TIMING_ONLY=1 FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/ab_h16.txt || { tail -15 $out/ab_h16.txt; exit 1; }
  python3 bench.py --workload c4 --steps 10 --warmup 4 --no-alt --no-cpu-baseline 2> $out/bench_${v}_$r.err | tail -1 > $out/bench_c4_${v}_$r.json
done; done
unset SRK_LIB_PATH
cat $out/ab_h16.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04o/bench_c4_*.json")):
    try:
        j = json.loads(open(f).read()); r = j["roofline"]
        print(f.split("/")[-1], round(j["ms_per_step"], 2), r["avg_us"], r["frac"], {k[:28]: v["ms"] for k, v in list(r["by_kernel"].items())[:2]})
    except Exception as e:
        print(f, "unreadable", e)
PY
