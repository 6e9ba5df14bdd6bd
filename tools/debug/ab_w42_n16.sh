#!/bin/bash
# same-box A/B of srk_conv_w42.hip builds on the 16-row form (batch 16: the warm-up / G-only workload)
for v in "$@"; do
  if [ "$v" = base ]; then unset SRK_LIB_PATH; else export SRK_LIB_PATH=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  echo "== $v"
  N=16 FMT=6 REPS=2 timeout -k 10 200 python tools/debug/chain_check.py 2>&1 | grep "block at"
done
