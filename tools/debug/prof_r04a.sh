# round 4, first GPU call: GPU tests, default bench line, matrix-pipe counters of the c4 step (its own --pmc pass, kernel-trace only)
set -o pipefail
out=gpurun_out/r04a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
echo default-done
C4="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES --kernel-trace -d $out/pmc_mfma_c4 -o m --output-format csv -- $C4 > /dev/null 2> $out/pmc_mfma_c4.err || { tail -5 $out/pmc_mfma_c4.err; exit 1; }
echo mfma-done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc_sq_c4 -o s --output-format csv -- $C4 > /dev/null 2> $out/pmc_sq_c4.err || { tail -5 $out/pmc_sq_c4.err; exit 1; }
echo sq-done
find $out -name "*.csv" | head
