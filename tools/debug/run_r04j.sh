# full-line stores of the 16-bit chain epilogue (DPP exchange): correctness, timing, HBM traffic; fp32 start-skew sweep under protocol v4
set -o pipefail
out=gpurun_out/r04j
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
FMT=7 REPS=3 timeout -k 10 200 python3 tools/debug/chain_check.py > $out/chain_check_h16.txt 2>&1 || { tail -15 $out/chain_check_h16.txt; exit 1; }
tail -4 $out/chain_check_h16.txt
FMT=8 REPS=2 NO_TIMING=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -2
timeout -k 10 600 python3 -m pytest tests/test_h16_gpu.py -x -q > $out/pytest_h16.log 2>&1 || { tail -40 $out/pytest_h16.log; exit 1; }
tail -2 $out/pytest_h16.log
C="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4.json 2> $out/bench_c4.err || { tail -20 $out/bench_c4.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch_c4 -o f --output-format csv -- $C > /dev/null 2> $out/pmc_fetch_c4.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write_c4 -o w --output-format csv -- $C > /dev/null 2> $out/pmc_write_c4.err || exit 1
python3 tools/traffic_from_pmc.py $out/pmc_fetch_c4 $out/pmc_write_c4 $out/traffic_c4 | head -6
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r04j/bench_c4.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("c4", j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"], {k: v["ms"] for k, v in r["by_kernel"].items()})
PY
for g in 4 8; do for ns in 0 3000 5000 8000 12000; do
  echo "== w42 N=32 skew $ns ns x $g groups" >> $out/skew_w42.txt
  SRK_W42_CHAIN_SKEW_NS=$ns SRK_W42_CHAIN_SKEW_GROUPS=$g FMT=6 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/skew_w42.txt
done; done
for ns in 0 3000 6000; do
  echo "== w42 N=16 skew $ns ns x 4 groups" >> $out/skew_w42.txt
  SRK_W42_CHAIN_SKEW_NS=$ns FMT=6 N=16 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/skew_w42.txt
done
cat $out/skew_w42.txt
