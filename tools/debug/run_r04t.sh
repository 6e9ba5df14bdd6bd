# wino24 FAST pieces: correctness (LDS poisoned; ragged shapes take the general form), wgrad tests, timing against the general form (SRK_WGRAD_W24_FAST=0)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r04t
timeout -k 10 300 python3 tools/debug/w24_dev.py > gpurun_out/r04t/dev_1.txt 2>&1; grep -c "form" gpurun_out/r04t/dev_1.txt; grep "agree\|MISMATCH\|nan\|batched" gpurun_out/r04t/dev_1.txt
timeout -k 10 500 python3 -m pytest tests/test_conv_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
for r in 1 2; do for f in 1 0; do echo "== fast=$f: $(SRK_WGRAD_W24_FAST=$f N=32 ITERS=60 timeout -k 10 120 python tools/bench_wgrad.py 2>&1 | grep batched)"; done; done
