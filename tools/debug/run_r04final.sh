# what the driver runs at round end, on the final tree: the GPU suite, smoke, the bench line
set -o pipefail
out=gpurun_out/r04final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r04final/bench.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("gan", round(j["ms_per_step"], 2), j["value"], r["frac"], r["avg_us"], j["cpu_baseline"]["value"])
for k, v in j["configs"].items():
    print(k, round(v["ms_per_step"], 2), v["roofline"]["frac"], v["roofline"].get("avg_us"), v["roofline"].get("probed_step_conv_ms_over_avg_step_ms"))
PY
