# protocol v4 (bounded flag waits + drain, no census): timing, protocol tests, default bench, holder bench, whole suite
set -o pipefail
out=gpurun_out/r04i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for n in 32 16; do
  echo "== w42 v4 N=$n" >> $out/chain_check.txt
  FMT=6 N=$n REPS=2 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -3 >> $out/chain_check.txt || { tail -5 $out/chain_check.txt; exit 1; }
done
echo "== h16 M16" >> $out/chain_check.txt
FMT=7 REPS=2 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -3 >> $out/chain_check.txt || { tail -15 $out/chain_check.txt; exit 1; }
cat $out/chain_check.txt
timeout -k 10 600 python3 -m pytest tests/test_chain_gpu.py -x -q > $out/pytest_chain.log 2>&1 || { tail -40 $out/pytest_chain.log; exit 1; }
tail -2 $out/pytest_chain.log
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r04i/bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("gan", j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"])
for k, v in j.get("configs", {}).items():
    print(k, v["ms_per_step"], v["roofline"]["kernel"], v["roofline"]["avg_us"], v["roofline"]["frac"])
PY
timeout -k 10 600 python3 tools/debug/holder_bench.py > $out/holder_bench.txt 2>&1 || { tail -20 $out/holder_bench.txt; exit 1; }
cat $out/holder_bench.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
