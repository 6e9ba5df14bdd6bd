# rocprofv3 evidence for the chain form (round 3, second half): kernel-trace stats of the c4 workload (serial schedule: one stream, so
# that kernel durations are not inflated by co-running launches), HBM traffic of its kernels, then the default bench line
set -o pipefail
out=gpurun_out/r03final_c4
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C4="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
SRK_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d $out/prof_c4_serial -o p --output-format csv -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $out/bench_c4_profiled_serial.json 2> $out/prof_c4_serial.err || exit 1
echo c4-serial-stats-done
rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o p --output-format csv -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $out/bench_c4_profiled.json 2> $out/prof_c4.err || exit 1
echo c4-stats-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch_c4 -o f --output-format csv -- $C4 > /dev/null 2> $out/pmc_fetch_c4.err || exit 1
echo c4-fetch-done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write_c4 -o w --output-format csv -- $C4 > /dev/null 2> $out/pmc_write_c4.err || exit 1
echo c4-write-done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo default-done
ls $out $out/prof_c4
