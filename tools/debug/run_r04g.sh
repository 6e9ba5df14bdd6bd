# bisect of the census cost in the fp32 chain kernels (timing-only variants of srk_conv_w42.hip)
set -o pipefail
out=gpurun_out/r04g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in base nocensus arriveonly c3 c4 c5; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_w42_$v.so; fi
  echo "== w42 $v N=16" >> $out/census_bisect.txt
  FMT=6 N=16 REPS=1 timeout -k 10 200 python3 tools/debug/chain_check.py 2>&1 | grep "block at" >> $out/census_bisect.txt || { tail -5 $out/census_bisect.txt; exit 1; }
done
cat $out/census_bisect.txt
