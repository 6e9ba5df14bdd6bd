# rocprofv3 evidence for round 3 (run on the GPU box from the repo root):
#   kernel-trace stats of the default bench command and of the c4 workload, HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of c4
set -o pipefail
out=gpurun_out/r03prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C4="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o p --output-format csv -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $out/bench_c4_profiled.json 2> $out/prof_c4.err || exit 1
echo c4-stats-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch_c4 -o f --output-format csv -- $C4 > /dev/null 2> $out/pmc_fetch_c4.err || exit 1
echo c4-fetch-done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write_c4 -o w --output-format csv -- $C4 > /dev/null 2> $out/pmc_write_c4.err || exit 1
echo c4-write-done
rocprofv3 --kernel-trace --stats -d $out/prof_gan -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled.json 2> $out/prof_gan.err || exit 1
echo gan-stats-done
ls $out $out/prof_c4
