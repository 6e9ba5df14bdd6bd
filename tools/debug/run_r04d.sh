# round 4, fourth GPU call: start-skew sweep of the chain kernels, tests, bias-path diagnosis, c4 A/B
set -o pipefail
out=gpurun_out/r04d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for g in 4 8; do for ns in 0 400 800 1200 1800 2500; do
  echo "== h16 M16 skew $ns ns x $g groups" >> $out/skew.txt
  SRK_H16_CHAIN_SKEW_NS=$ns SRK_H16_CHAIN_SKEW_GROUPS=$g FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/skew.txt || { tail -5 $out/skew.txt; exit 1; }
done; done
for ns in 0 800 1500 2500 4000; do
  echo "== w42 N=32 skew $ns ns x 4 groups" >> $out/skew.txt
  SRK_W42_CHAIN_SKEW_NS=$ns FMT=6 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/skew.txt || { tail -5 $out/skew.txt; exit 1; }
done
for ns in 0 1500 3000; do
  echo "== w42 N=16 skew $ns ns x 4 groups" >> $out/skew.txt
  SRK_W42_CHAIN_SKEW_NS=$ns FMT=6 N=16 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/skew.txt || { tail -5 $out/skew.txt; exit 1; }
done
cat $out/skew.txt
timeout -k 10 900 python3 -m pytest tests/test_h16_gpu.py tests/test_chain_gpu.py tests/test_models_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 600 python3 tools/debug/bias_path.py > $out/bias_path.txt 2>&1 || { tail -20 $out/bias_path.txt; exit 1; }
cat $out/bias_path.txt
python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4_m16.json 2> $out/bench_c4_m16.err || { tail -20 $out/bench_c4_m16.err; exit 1; }
SRK_H16_CHAIN_M16=0 python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4_m32.json 2> $out/bench_c4_m32.err || { tail -20 $out/bench_c4_m32.err; exit 1; }
python3 - <<'PY'
import json
for n in ("m16", "m32"):
    j = json.loads(open(f"gpurun_out/r04d/bench_c4_{n}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(n, j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"], {k: v["ms"] for k, v in r["by_kernel"].items()})
PY
