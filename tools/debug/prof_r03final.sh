# final state of round 3: kernel-trace stats (no stream overlap) of the c4 and gan workloads, then the default bench line
set -o pipefail
out=gpurun_out/r03final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SRK_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d $out/prof_c4_serial -o p --output-format csv -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_c4_profiled_serial.json 2> $out/prof_c4_serial.err || exit 1
echo c4-done
SRK_OVERLAP_WGRAD=0 SRK_D_OVERLAP=0 SRK_D_STREAMS=0 rocprofv3 --kernel-trace --stats -d $out/prof_gan_serial -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled_serial.json 2> $out/prof_gan_serial.err || exit 1
echo gan-done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo default-done
