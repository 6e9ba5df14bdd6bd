#!/bin/bash
# A/B of environment switches on one GPU box: tools/ab_env.sh "SRK_WINO4_NH=2" "SRK_WINO4_NH=1" ...
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --steps ${STEPS:-8} --warmup 3 --no-alt --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | V="$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
bk=d['roofline']['by_kernel']
print('%-24s %8.2f ms/step  dominant %7.2f us  |' % (os.environ['V'], d['ms_per_step'], d['roofline']['avg_us']), ' '.join('%s=%.1f' % (k.split('_kernel')[0][-12:], v['ms']) for k,v in list(bk.items())[:3]))
"
done
