for v in base; do
  if [ $v = base ]; then lib=super-resolution_amd/libsrk.so; else lib=super-resolution_amd/csrc/build_var/libsrk_wg_$v.so; fi
  echo "== $v"; SRK_LIB_PATH=$lib N=32 timeout -k 10 120 python tools/bench_wgrad.py 2>&1 | grep batched
  echo "== $v, 1-D kernel"; SRK_WGRAD_WINO22=0 SRK_LIB_PATH=$lib N=32 timeout -k 10 120 python tools/bench_wgrad.py 2>&1 | grep batched
done
