#!/bin/bash
# A/B of library builds on one GPU box: tools/ab.sh base p1 base p1   (variant X = super-resolution_amd/csrc/build_var/libsrk_X.so)
for v in "$@"; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  timeout -k 10 200 python bench.py --steps ${STEPS:-8} --warmup 3 --no-alt --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
bk=d['roofline']['by_kernel']
print('%-8s %8.2f ms/step  dominant %7.2f us  |' % (os.environ['V'], d['ms_per_step'], d['roofline']['avg_us']), ' '.join('%s=%.1f' % (k.split('_kernel')[0][-12:], v['ms']) for k,v in list(bk.items())[:3]))
"
done
