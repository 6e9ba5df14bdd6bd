# same-box A/B of conv library variants on warm / cold inputs:  bash tools/debug/ab_cold.sh base prev b3 base prev b3
for v in "$@"; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/bench_cold.py
done
