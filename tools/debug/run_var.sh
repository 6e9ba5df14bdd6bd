mkdir -p gpurun_out/r02i
timeout -k 10 200 python -m pytest tests/test_conv_gpu.py -q -k "wino" 2>&1 | tail -2
timeout -k 10 200 python tools/bench_fmt.py > gpurun_out/r02i/bench_fmt3.log 2>&1
for v in stamp NO_LB NO_DMA NO_DS; do
  if [ $v = stamp ]; then lib=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so; else lib=super-resolution_amd/csrc/build_var/libsrk_$v.so; fi
  echo "== $v" >> gpurun_out/r02i/var.log
  SRK_LIB_PATH=$lib timeout -k 10 120 python tools/stamp_w42.py 2>&1 | grep "Cin=320:\|wave 0 cycles\|main loop:" | tail -3 >> gpurun_out/r02i/var.log
done
cat gpurun_out/r02i/bench_fmt3.log gpurun_out/r02i/var.log
