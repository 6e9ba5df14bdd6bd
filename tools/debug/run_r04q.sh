# 16x16x32 chain kernel: conflict-free X swizzle under the real ds_read_b128 lane groups (default) vs the one-conv kernels' swizzle (variant signsf = previous commit): tests, A/B, LDS counters
set -o pipefail
out=gpurun_out/r04q
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
FMT=7 REPS=2 NO_TIMING=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -3 || exit 1
FMT=8 REPS=2 NO_TIMING=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | tail -2 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_h16_gpu.py tests/test_chain_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for r in 1 2 3; do for v in base signsf; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  echo "== h16 $v round $r" >> $out/ab_h16.txt
  TIMING_ONLY=1 FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/ab_h16.txt || { tail -15 $out/ab_h16.txt; exit 1; }
  python3 bench.py --workload c4 --steps 10 --warmup 4 --no-alt --no-cpu-baseline 2> /dev/null | tail -1 > $out/bench_c4_${v}_$r.json
done; done
unset SRK_LIB_PATH
cat $out/ab_h16.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04q/bench_c4_*.json")):
    j = json.loads(open(f).read()); r = j["roofline"]
    print(f.split("/")[-1], round(j["ms_per_step"], 2), r["avg_us"], r["frac"], {k[:28]: v["ms"] for k, v in list(r["by_kernel"].items())[:2]})
PY
C="python3 bench.py --workload c4 --steps 3 --warmup 2 --no-alt --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES --kernel-trace -d $out/pmc_mfma_c4 -o m --output-format csv -- $C > /dev/null 2> $out/pmc_mfma_c4.err || { tail -5 $out/pmc_mfma_c4.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc_sq_c4 -o s --output-format csv -- $C > /dev/null 2> $out/pmc_sq_c4.err || { tail -5 $out/pmc_sq_c4.err; exit 1; }
python3 tools/mfma_from_pmc.py $out/pmc_mfma_c4 $out/pmc_sq_c4 $out/mfma_c4 | head -4
