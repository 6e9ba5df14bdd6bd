# matrix-pipe and wave-state counters of the c4 workload (16-bit kernels): two --pmc passes, kernel-trace only
set -o pipefail
out=gpurun_out/r03prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C4="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $out/pmc_mfma_c4 -o m --output-format csv -- $C4 > /dev/null 2> $out/pmc_mfma_c4.err || { tail -5 $out/pmc_mfma_c4.err; exit 1; }
echo mfma-done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc_sq_c4 -o s --output-format csv -- $C4 > /dev/null 2> $out/pmc_sq_c4.err || { tail -5 $out/pmc_sq_c4.err; exit 1; }
echo sq-done
ls $out/pmc_mfma_c4 $out/pmc_sq_c4
