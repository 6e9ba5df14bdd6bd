# wino24 weight gradient as the default: whole GPU suite, then the headline step A/B against wino22 (SRK_WGRAD_W22_FORM=1), interleaved
set -o pipefail
out=gpurun_out/r04u
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for r in 1 2 3; do for f in 2 1; do
  SRK_WGRAD_W22_FORM=$f python3 bench.py --steps 20 --warmup 5 --no-configs --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > $out/bench_gan_form${f}_$r.json
done; done
for f in 2 1; do SRK_WGRAD_W22_FORM=$f python3 bench.py --workload g_only --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > $out/bench_gonly_form${f}.json; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04u/bench_*.json")):
    j = json.loads(open(f).read()); r = j["roofline"]
    print(f.split("/")[-1], round(j["ms_per_step"], 2), r["avg_us"], r["frac"], {k[:34]: v["ms"] for k, v in list(r["by_kernel"].items())[:2]})
PY
