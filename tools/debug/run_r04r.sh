# what the 16x16x32 chain kernel's main loop waits for: timing-only variants (no global loads behind the first stage / no fragment reads / no flag waits)  [results wrong]
set -o pipefail
out=gpurun_out/r04r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TIMING_ONLY=1 FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py > /dev/null 2>&1
for r in 1 2; do for v in base ml1 ml2 ml3 ml4 ml7; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_h16_$v.so; fi
  echo "== h16 $v round $r" >> $out/ab_h16.txt
  TIMING_ONLY=1 FMT=7 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/ab_h16.txt || { tail -15 $out/ab_h16.txt; exit 1; }
done; done
unset SRK_LIB_PATH
cat $out/ab_h16.txt
