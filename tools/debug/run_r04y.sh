# per-shape kernel times of the headline step (SRK_KT_DETAIL=1): the HR tail and the discriminator layers
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r04y
SRK_KT_DETAIL=1 python3 bench.py --steps 6 --warmup 3 --no-configs --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r04y/bench_detail.json
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r04y/bench_detail.json").read())
bk = j["roofline"]["by_kernel"]
rows = sorted(bk.items(), key=lambda kv: -kv[1]["ms"])
for k, v in rows[:60]:
    if "chain" in k or "wino24" in k: continue
    print(f"{v['ms']:7.3f} ms  x{v['launches']:3d}  {k[:150]}")
PY
