# rocprofv3 evidence for the chain forms, headline workload (gan): kernel-trace stats without stream overlap (what bench.py's probe
# measures), the default (overlapped) schedule, HBM traffic of its kernels
set -o pipefail
out=gpurun_out/r03zp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
GAN="python3 bench.py --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-configs --no-kernel-timing"
SRK_OVERLAP_WGRAD=0 SRK_D_OVERLAP=0 SRK_D_STREAMS=0 rocprofv3 --kernel-trace --stats -d $out/prof_gan_serial -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled_serial.json 2> $out/prof_gan_serial.err || exit 1
echo gan-serial-done
rocprofv3 --kernel-trace --stats -d $out/prof_gan -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled.json 2> $out/prof_gan.err || exit 1
echo gan-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch_gan -o f --output-format csv -- $GAN > /dev/null 2> $out/pmc_fetch_gan.err || exit 1
echo fetch-done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write_gan -o w --output-format csv -- $GAN > /dev/null 2> $out/pmc_write_gan.err || exit 1
echo write-done
head -5 $out/prof_gan_serial/p_kernel_stats.csv | cut -c1-170
