#!/bin/bash
# tools/debug/build_h16_var.sh NAME -DH16_X=1 ...  ->  super-resolution_amd/csrc/build_var/libsrk_h16_NAME.so (the shipped objects + this build of srk_conv_h16.hip)
set -e
C=super-resolution_amd/csrc; name=$1; shift
mkdir -p $C/build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++20 -Iinclude -I$C "$@" -c $C/srk_conv_h16.hip -o $C/build_var/h16_$name.o
objs=$(ls $C/build/*.o | grep -v srk_conv_h16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build_var/libsrk_h16_$name.so $objs $C/build_var/h16_$name.o
echo built $C/build_var/libsrk_h16_$name.so
