# 16-bit chain: argument struct out of scratch (traffic), full-line vs half-line stores (A/B); fp32 start skew in the whole step
set -o pipefail
out=gpurun_out/r04k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in base halflines; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  echo "== h16 $v" >> $out/ab_h16.txt
  FMT=7 REPS=2 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -3 >> $out/ab_h16.txt || { tail -15 $out/ab_h16.txt; exit 1; }
  python3 bench.py --workload c4 --steps 8 --warmup 3 --no-alt --no-cpu-baseline 2> $out/bench_c4_$v.err | tail -1 > $out/bench_c4_$v.json || { tail -20 $out/bench_c4_$v.err; exit 1; }
done
unset SRK_LIB_PATH
cat $out/ab_h16.txt
C="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch_c4 -o f --output-format csv -- $C > /dev/null 2> $out/pmc_fetch_c4.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write_c4 -o w --output-format csv -- $C > /dev/null 2> $out/pmc_write_c4.err || exit 1
python3 tools/traffic_from_pmc.py $out/pmc_fetch_c4 $out/pmc_write_c4 $out/traffic_c4 | head -5
for r in 1 2; do for ns in 0 4000; do
  SRK_W42_CHAIN_SKEW_NS=$ns python3 bench.py --steps 20 --warmup 5 --no-configs --no-alt --no-cpu-baseline --no-kernel-timing 2> /dev/null | tail -1 > $out/bench_gan_skew${ns}_$r.json
  SRK_W42_CHAIN_SKEW_NS=$ns python3 bench.py --workload g_only --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-kernel-timing 2> /dev/null | tail -1 > $out/bench_g_skew${ns}_$r.json
done; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04k/bench_*.json")):
    try:
        j = json.loads(open(f).read())
        r = j.get("roofline") or {}
        print(f.split("/")[-1], round(j["ms_per_step"], 2), r.get("avg_us"), r.get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
