#!/bin/bash
# Collects what profiles/<tag>_* is made from, on the GPU box:  bash tools/profile_round.sh r02k
# kernel trace + stats of the default bench, four separate PMC passes (never combined with other trace domains), bench lines.
set -o pipefail
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline"
P="python3 bench.py --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
echo "bench done" 
rocprofv3 --kernel-trace --stats -d $out/prof -o p --output-format csv -- $B > $out/bench_profiled.json 2> $out/prof.err || exit 1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- $P > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- $P > /dev/null 2> $out/pmc_write.err || exit 1
echo "traffic passes done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- $P > /dev/null 2> $out/pmc_mfma.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace -d $out/pmc_sq -o s --output-format csv -- $P > /dev/null 2> $out/pmc_sq.err || exit 1
echo "mfma passes done"
python3 bench.py --workload g_only > $out/bench_g_only.json 2> $out/bench_g_only.err
python3 bench.py --workload c4 > $out/bench_c4.json 2> $out/bench_c4.err
ls $out
