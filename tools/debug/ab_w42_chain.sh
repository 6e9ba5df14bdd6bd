#!/bin/bash
# same-box A/B of the wino42 chain options (variant libs in super-resolution_amd/csrc/build_var/libsrk_wPQ.so: P = W42_CH_PREFETCH, Q = W42_CH_PUBLISH)
for v in base "$@"; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  echo "== $v"
  FMT=6 REPS=2 timeout -k 10 200 python tools/debug/chain_check.py 2>&1 | tail -3
done
