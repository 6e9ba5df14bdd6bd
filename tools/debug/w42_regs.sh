#!/bin/bash
# register / spill figures of the wino42 kernels + where the chain kernel's scratch accesses sit relative to its MFMAs
# (m = one phase of 12 v_mfma, M = a single one, L / S = scratch load / store, | = s_barrier)
set -e
B=/root/repo/super-resolution_amd/csrc/build
cd /tmp && rm -rf w42dis && mkdir w42dis && cd w42dis
(cd $B && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading srk_conv_w42.o > /dev/null) && mv $B/srk_conv_w42.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 dev.o && rm -f $B/srk_conv_w42.o.0.host-*
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.o | grep -E "\.name:|vgpr_count|vgpr_spill|sgpr_spill|agpr_count|private_segment_fixed" | paste - - - - - - | sed 's/  */ /g' | cut -c1-230
/opt/rocm/lib/llvm/bin/llvm-objdump -d dev.o > all.s
awk '/^[0-9a-f]+ <.*wino42_chain_kernelILi2E/ {p=1; next} /^[0-9a-f]+ </ {p=0} p' all.s > chain2.s
grep -n "v_mfma\|scratch_\|s_barrier" chain2.s | awk '{ if ($0 ~ /v_mfma/) t="M"; else if ($0 ~ /scratch_load/) t="L"; else if ($0 ~ /scratch_store/) t="S"; else t="|"; printf "%s", t } END {print ""}' | sed 's/MMMMMMMMMMMM/m/g' | fold -w 160
echo "scalar flag loads: $(grep -c 's_load_dword.*glc' chain2.s)"
