#!/bin/bash
# the two forms of the 2-D Winograd weight gradient in the training step (same box): headline gan step and the warm-up (g_only) step
for f in 1 0 1 0; do
  for wl in gan g_only; do
    echo "== SRK_WGRAD_W22_FORM=$f $wl: $(SRK_WGRAD_W22_FORM=$f timeout -k 10 200 python bench.py --workload $wl --steps 12 --warmup 4 --no-configs --no-alt --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); bk=d["roofline"].get("by_kernel",{}); print(round(d["ms_per_step"],2), "ms;", {k:(v["launches"], round(v["ms"],2)) for k,v in bk.items() if "wino22" in k})')"
  done
done
