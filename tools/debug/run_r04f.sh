# round 4, sixth GPU call: census v2 (separate go word), bias through the accumulators; then the whole GPU suite and the default bench
set -o pipefail
out=gpurun_out/r04f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for n in 32 16; do
  echo "== w42 census v2 N=$n" >> $out/chain_check.txt
  FMT=6 N=$n REPS=2 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -3 >> $out/chain_check.txt || { tail -5 $out/chain_check.txt; exit 1; }
done
echo "== h16 M16 (bias through the accumulators)" >> $out/chain_check.txt
FMT=7 REPS=3 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -13 >> $out/chain_check.txt || { tail -15 $out/chain_check.txt; exit 1; }
echo "== h16 M16 bf16" >> $out/chain_check.txt
FMT=8 REPS=2 NO_TIMING=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -3 >> $out/chain_check.txt || { tail -15 $out/chain_check.txt; exit 1; }
cat $out/chain_check.txt
for b in 0 1; do
  echo "== M16=1 BWD=$b" >> $out/stamps.txt
  BWD=$b SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so timeout -k 10 200 python3 tools/stamp_h16_chain.py >> $out/stamps.txt 2>&1 || { tail -20 $out/stamps.txt; exit 1; }
done
cat $out/stamps.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r04f/bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("gan", j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"])
for k, v in j.get("configs", {}).items():
    print(k, v["ms_per_step"], v["roofline"]["kernel"], v["roofline"]["avg_us"], v["roofline"]["frac"])
PY
