# round 4, second GPU call: the 16x16x32 form of the 16-bit chain kernel (correctness, A/B against the 32x32x16 form), bias-path diagnosis
set -o pipefail
out=gpurun_out/r04b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== chain_check M16=1" > $out/chain_check.txt
FMT=7 REPS=4 timeout -k 10 300 python3 tools/debug/chain_check.py >> $out/chain_check.txt 2>&1 || { tail -20 $out/chain_check.txt; exit 1; }
echo "== chain_check M16=0" >> $out/chain_check.txt
SRK_H16_CHAIN_M16=0 FMT=7 REPS=2 timeout -k 10 300 python3 tools/debug/chain_check.py >> $out/chain_check.txt 2>&1 || { tail -20 $out/chain_check.txt; exit 1; }
echo "== chain_check bf16 M16=1" >> $out/chain_check.txt
FMT=8 REPS=2 NO_TIMING=1 timeout -k 10 300 python3 tools/debug/chain_check.py >> $out/chain_check.txt 2>&1 || { tail -20 $out/chain_check.txt; exit 1; }
cat $out/chain_check.txt
timeout -k 10 900 python3 -m pytest tests/test_h16_gpu.py tests/test_chain_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 600 python3 tools/debug/bias_path.py > $out/bias_path.txt 2>&1 || { tail -20 $out/bias_path.txt; exit 1; }
cat $out/bias_path.txt
python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4_m16.json 2> $out/bench_c4_m16.err || { tail -20 $out/bench_c4_m16.err; exit 1; }
SRK_H16_CHAIN_M16=0 python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4_m32.json 2> $out/bench_c4_m32.err || { tail -20 $out/bench_c4_m32.err; exit 1; }
python3 - <<'PY'
import json
for n in ("m16", "m32"):
    j = json.loads(open(f"gpurun_out/r04b/bench_c4_{n}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(n, j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"], {k: v["ms"] for k, v in r["by_kernel"].items()})
PY
