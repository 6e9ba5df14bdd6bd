#!/bin/bash
# Row-owner form of the 2-D Winograd weight gradient vs the tile-owner form: dense-block batch at N = 32 (same box, same process order).
for f in 1 0 1 0; do
  echo "== SRK_WGRAD_W22_FORM=$f"; SRK_WGRAD_W22_FORM=$f N=32 ITERS=60 timeout -k 10 120 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
done
