set -o pipefail
out=gpurun_out/r02l_g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P="python3 bench.py --workload g_only --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- $P > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- $P > /dev/null 2> $out/pmc_write.err || exit 1
rocprofv3 --kernel-trace --stats -d $out/prof -o p --output-format csv -- python3 bench.py --workload g_only --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $out/bench_profiled.json 2> $out/prof.err || exit 1
ls $out
