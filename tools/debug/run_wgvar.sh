#!/bin/bash
# same-box timing of srk_wgrad_w22.hip builds (tools/debug/build_w22_var.sh): dense-block batch at N = 32
for v in "$@"; do
  if [ $v = base ]; then lib=super-resolution_amd/libsrk.so; else lib=super-resolution_amd/csrc/build_var/libsrk_wg_$v.so; fi
  echo "== $v: $(SRK_LIB_PATH=$lib N=32 ITERS=60 timeout -k 10 120 python tools/bench_wgrad.py 2>&1 | grep batched)"
done
