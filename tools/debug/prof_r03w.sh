# end of round 3 (row-owner weight gradient): kernel-trace stats of the gan workload without stream overlap, the matrix-pipe counters,
# the HBM traffic passes (each --pmc pass on its own, kernel-trace only), then the default bench line
set -o pipefail
out=gpurun_out/r03w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P="python3 bench.py --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-configs --no-kernel-timing"
SRK_OVERLAP_WGRAD=0 SRK_D_OVERLAP=0 SRK_D_STREAMS=0 rocprofv3 --kernel-trace --stats -d $out/prof_gan_serial -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled_serial.json 2> $out/prof_gan_serial.err || exit 1
echo gan-serial-done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- $P > /dev/null 2> $out/pmc_mfma.err || exit 1
echo mfma-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- $P > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- $P > /dev/null 2> $out/pmc_write.err || exit 1
echo traffic-done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo default-done
find $out -name "*.csv" | head -20
