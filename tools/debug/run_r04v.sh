# c4: per-iteration times (is there a host stall?), chain forms on / off; then the bench line three times
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r04v
WORKLOAD=c4 ITERS=40 python3 tools/debug/step_times.py 2>&1 | tail -2
SRK_H16_CHAIN=0 WORKLOAD=c4 ITERS=25 python3 tools/debug/step_times.py 2>&1 | tail -2
for r in 1 2 3; do python3 bench.py --workload c4 --steps 10 --warmup 4 --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r04v/bench_c4_$r.json; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04v/bench_c4_*.json")):
    j = json.loads(open(f).read()); r = j["roofline"]
    print(f.split("/")[-1], round(j["ms_per_step"], 2), r["avg_us"], r["frac"], r.get("probed_step_conv_ms_over_avg_step_ms"))
PY
