#!/bin/bash
# weights through the LDS ring (base) vs into registers (build_var/libsrk_w42regsw.so: -DW42_LDSW=0): dense block at batch 32 and 16
for v in base w42regsw base w42regsw; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  for n in 32 16; do
    echo "== $v N=$n"; N=$n FMT=6 REPS=2 timeout -k 10 200 python tools/debug/chain_check.py 2>&1 | tail -3
  done
done
