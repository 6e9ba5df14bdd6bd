# round 4, third GPU call: stamps of the 16x16x32 chain form, tests, bias-path diagnosis, c4 A/B
set -o pipefail
out=gpurun_out/r04c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for m in 1 0; do for b in 0 1; do
  echo "== M16=$m BWD=$b" >> $out/stamps.txt
  SRK_H16_CHAIN_M16=$m BWD=$b SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so timeout -k 10 200 python3 tools/stamp_h16_chain.py >> $out/stamps.txt 2>&1 || { tail -20 $out/stamps.txt; exit 1; }
done; done
cat $out/stamps.txt
timeout -k 10 900 python3 -m pytest tests/test_h16_gpu.py tests/test_chain_gpu.py tests/test_models_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 600 python3 tools/debug/bias_path.py > $out/bias_path.txt 2>&1 || { tail -20 $out/bias_path.txt; exit 1; }
cat $out/bias_path.txt
python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4_m16.json 2> $out/bench_c4_m16.err || { tail -20 $out/bench_c4_m16.err; exit 1; }
SRK_H16_CHAIN_M16=0 python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline > $out/bench_c4_m32.json 2> $out/bench_c4_m32.err || { tail -20 $out/bench_c4_m32.err; exit 1; }
python3 - <<'PY'
import json
for n in ("m16", "m32"):
    j = json.loads(open(f"gpurun_out/r04c/bench_c4_{n}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(n, j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"], {k: v["ms"] for k, v in r["by_kernel"].items()})
PY
