# round 4, fifth GPU call: what the census costs the fp32 chain kernels (variants), epilogue set-up stamp of the 16-bit chain kernel, tests, bias path, c4 A/B
set -o pipefail
out=gpurun_out/r04e
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in base nocensus arriveonly; do for n in 32 16; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_w42_$v.so; fi
  echo "== w42 $v N=$n" >> $out/census_ab.txt
  FMT=6 N=$n REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/census_ab.txt || { tail -5 $out/census_ab.txt; exit 1; }
done; done
unset SRK_LIB_PATH
cat $out/census_ab.txt
for b in 0 1; do
  echo "== M16=1 BWD=$b" >> $out/stamps.txt
  BWD=$b SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so timeout -k 10 200 python3 tools/stamp_h16_chain.py >> $out/stamps.txt 2>&1 || { tail -20 $out/stamps.txt; exit 1; }
done
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
    j = json.loads(open(f"gpurun_out/r04e/bench_c4_{n}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(n, j["ms_per_step"], r["kernel"], r["avg_us"], r["frac"], {k: v["ms"] for k, v in r["by_kernel"].items()})
PY
