#!/bin/bash
# tools/debug/build_w22_var.sh NAME -DW22_XS0=9 ...  ->  super-resolution_amd/csrc/build_var/libsrk_wg_NAME.so (the shipped objects + this build of srk_wgrad_w22.hip)
set -e
C=super-resolution_amd/csrc; name=$1; shift
mkdir -p $C/build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -I$C -fno-slp-vectorize "$@" \
  -c $C/srk_wgrad_w22.hip -o $C/build_var/w22_$name.o
objs=$(ls $C/build/*.o | grep -v srk_wgrad_w22.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build_var/libsrk_wg_$name.so $objs $C/build_var/w22_$name.o
echo built $C/build_var/libsrk_wg_$name.so
