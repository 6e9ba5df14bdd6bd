# round 4 profiles: kernel-trace stats of the gan and c4 workloads without stream overlap, matrix-pipe counters and HBM traffic of both
# (each --pmc pass on its own, kernel-trace only; the program directly behind --), then the default bench line
set -o pipefail
out=gpurun_out/r04p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
G="python3 bench.py --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-configs --no-kernel-timing"
C="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
SRK_OVERLAP_WGRAD=0 SRK_D_OVERLAP=0 SRK_D_STREAMS=0 rocprofv3 --kernel-trace --stats -d $out/prof_gan_serial -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_gan_profiled_serial.json 2> $out/prof_gan_serial.err || { tail -5 $out/prof_gan_serial.err; exit 1; }
echo gan-serial-done
SRK_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d $out/prof_c4_serial -o p --output-format csv -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-configs > $out/bench_c4_profiled_serial.json 2> $out/prof_c4_serial.err || { tail -5 $out/prof_c4_serial.err; exit 1; }
echo c4-serial-done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- $G > /dev/null 2> $out/pmc_mfma.err || { tail -5 $out/pmc_mfma.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc_sq -o s --output-format csv -- $G > /dev/null 2> $out/pmc_sq.err || { tail -5 $out/pmc_sq.err; exit 1; }
echo gan-mfma-done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES --kernel-trace -d $out/pmc_mfma_c4 -o m --output-format csv -- $C > /dev/null 2> $out/pmc_mfma_c4.err || { tail -5 $out/pmc_mfma_c4.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc_sq_c4 -o s --output-format csv -- $C > /dev/null 2> $out/pmc_sq_c4.err || { tail -5 $out/pmc_sq_c4.err; exit 1; }
echo c4-mfma-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- $G > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- $G > /dev/null 2> $out/pmc_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch_c4 -o f --output-format csv -- $C > /dev/null 2> $out/pmc_fetch_c4.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write_c4 -o w --output-format csv -- $C > /dev/null 2> $out/pmc_write_c4.err || exit 1
echo traffic-done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo default-done
timeout -k 10 300 python3 tools/debug/soak.py > $out/soak.txt 2>&1 || { tail -20 $out/soak.txt; exit 1; }
cat $out/soak.txt
