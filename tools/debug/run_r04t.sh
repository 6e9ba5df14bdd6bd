# wino24 weight gradient: correctness (LDS poisoned before every launch; twice), the wgrad tests
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r04t
for r in 1 2; do timeout -k 10 300 python3 tools/debug/w24_dev.py > gpurun_out/r04t/dev_$r.txt 2>&1; grep -c "form" gpurun_out/r04t/dev_$r.txt; grep "agree\|MISMATCH\|nan\|batched" gpurun_out/r04t/dev_$r.txt; done
timeout -k 10 500 python3 -m pytest tests/test_conv_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
