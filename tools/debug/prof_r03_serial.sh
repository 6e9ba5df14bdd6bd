# kernel-trace stats WITHOUT stream overlap (weight gradients / discriminators on the main stream): per-kernel durations that are
# not stretched by a kernel running beside them -- what bench.py's event probe measures (it switches the overlaps off for its step)
set -o pipefail
out=gpurun_out/r03prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export SRK_OVERLAP_WGRAD=0 SRK_D_OVERLAP=0 SRK_D_STREAMS=0
rocprofv3 --kernel-trace --stats -d $out/prof_c4_serial -o p --output-format csv -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $out/bench_c4_profiled_serial.json 2> $out/prof_c4_serial.err || exit 1
rocprofv3 --kernel-trace --stats -d $out/prof_gan_serial -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled_serial.json 2> $out/prof_gan_serial.err || exit 1
head -4 $out/prof_c4_serial/p_kernel_stats.csv | cut -c1-160
head -4 $out/prof_gan_serial/p_kernel_stats.csv | cut -c1-160
