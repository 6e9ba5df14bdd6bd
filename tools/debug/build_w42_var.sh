#!/bin/bash
# tools/debug/build_w42_var.sh NAME -DW42_X=1 ...  ->  super-resolution_amd/csrc/build_var/libsrk_w42_NAME.so (the shipped objects + this build of srk_conv_w42.hip)
set -e
C=super-resolution_amd/csrc; name=$1; shift
mkdir -p $C/build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -I$C -fno-slp-vectorize "$@" \
  -c $C/srk_conv_w42.hip -o $C/build_var/w42_$name.o
objs=$(ls $C/build/*.o | grep -v srk_conv_w42.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build_var/libsrk_w42_$name.so $objs $C/build_var/w42_$name.o
echo built $C/build_var/libsrk_w42_$name.so
