#!/bin/bash
# register / spill figures of the 16-bit conv kernels + where the chain kernel's scratch accesses sit relative to its MFMAs
# (M = v_mfma, L / S = scratch load / store, | = s_barrier)
set -e
B=/root/repo/super-resolution_amd/csrc/build
O=$B/srk_conv_h16.o
cd /tmp && rm -rf h16dis && mkdir h16dis && cd h16dis
(cd $B && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading srk_conv_h16.o > /dev/null) && mv $B/srk_conv_h16.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 dev.o && rm -f $B/srk_conv_h16.o.0.host-*
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.o | grep -E "\.name:|vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed" | paste - - - - - | sed 's/  */ /g' | grep -i "${1:-chain}" | cut -c1-250
/opt/rocm/lib/llvm/bin/llvm-objdump -d dev.o > all.s
awk '/^[0-9a-f]+ <.*chain_kernelIDF16_E/ {p=1} /^[0-9a-f]+ <.*chain_kernelIDF16bE/ {p=0} p' all.s > chain.s
grep -n "v_mfma\|scratch_\|s_barrier" chain.s | awk '{ if ($0 ~ /v_mfma/) t="M"; else if ($0 ~ /scratch_load/) t="L"; else if ($0 ~ /scratch_store/) t="S"; else t="|"; printf "%s", t } END {print ""}' | fold -w 150
