#!/bin/bash
# 16-row form: correctness vs five launches + timing (batch 16 dense block), then the warm-up step
N=16 FMT=6 REPS=3 timeout -k 10 200 python tools/debug/chain_check.py 2>&1 | tail -3
timeout -k 10 200 python bench.py --workload g_only --steps 12 --warmup 4 --no-alt --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('g_only', d['ms_per_step'], d['roofline']['frac'])"
