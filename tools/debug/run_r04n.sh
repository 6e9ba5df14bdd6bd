# sign bits of the 16-bit chain epilogue from packed integer min / max (default) vs convert back + compare (variant signsf): tests, then interleaved A/B
set -o pipefail
out=gpurun_out/r04n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in signsf base; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  timeout -k 10 500 python3 -m pytest tests/test_h16_gpu.py -q > $out/pytest_$v.log 2>&1; echo "pytest $v rc=$?"; tail -25 $out/pytest_$v.log | cut -c1-300
done
for r in 1 2 3; do for v in base signsf; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  echo "== h16 $v round $r" >> $out/ab_h16.txt
  FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/ab_h16.txt || { tail -15 $out/ab_h16.txt; exit 1; }
  python3 bench.py --workload c4 --steps 10 --warmup 4 --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > $out/bench_c4_${v}_$r.json
done; done
unset SRK_LIB_PATH
cat $out/ab_h16.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04n/bench_c4_*.json")):
    j = json.loads(open(f).read()); r = j["roofline"]
    print(f.split("/")[-1], round(j["ms_per_step"], 2), r["avg_us"], r["frac"], {k[:28]: v["ms"] for k, v in list(r["by_kernel"].items())[:2]})
PY
