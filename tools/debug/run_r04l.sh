# 16-bit chain epilogue without scratch reloads: full-line vs half-line stores, interleaved A/B; stamps; tests
set -o pipefail
out=gpurun_out/r04l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > /dev/null 2>&1      # (warm the box: the first process pays page-in)
for r in 1 2 3; do for v in base halflines; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  echo "== h16 $v round $r" >> $out/ab_h16.txt
  FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/ab_h16.txt || { tail -15 $out/ab_h16.txt; exit 1; }
  python3 bench.py --workload c4 --steps 10 --warmup 4 --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > $out/bench_c4_${v}_$r.json
done; done
unset SRK_LIB_PATH
cat $out/ab_h16.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04l/bench_c4_*.json")):
    j = json.loads(open(f).read()); r = j["roofline"]
    print(f.split("/")[-1], round(j["ms_per_step"], 2), r["avg_us"], r["frac"], {k[:28]: v["ms"] for k, v in list(r["by_kernel"].items())[:2]})
PY
for b in 0 1; do
  echo "== M16 BWD=$b" >> $out/stamps.txt
  BWD=$b SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so timeout -k 10 200 python3 tools/stamp_h16_chain.py >> $out/stamps.txt 2>&1 || { tail -20 $out/stamps.txt; exit 1; }
done
grep -v amdgpu $out/stamps.txt
timeout -k 10 600 python3 -m pytest tests/test_h16_gpu.py tests/test_chain_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
