#!/bin/bash
# same-box A/B of 16-bit chain kernel variants (super-resolution_amd/csrc/build_var/libsrk_X.so)
for v in base "$@"; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  echo "== $v"
  FMT=7 REPS=${REPS:-4} timeout -k 10 200 python tools/debug/chain_check.py 2>&1 | tail -3
done
