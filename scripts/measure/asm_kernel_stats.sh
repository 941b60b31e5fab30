#!/bin/bash
# static picture of one kernel of the tree's hpk_kernels.hip: registers, spills, SGPR spill traffic (v_readlane / v_writelane), barriers
# usage: asm_kernel_stats.sh [mangled-name substring] [EXTRA flags]   (default: the single-pair weight-input stencil)
K=${1:-hpk_stencil_sILb0ELb1ELb0EEE}
make -C /root/repo/hicpeaks_amd/csrc asm EXTRA="$2" 2>&1 | grep -E "error" 
cd /root/repo/build/asm && S=hpk_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
N=$(grep -oE "^_ZN12_GLOBAL__N_1[0-9]+$K[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
awk -v n="$N:" 'index($0,n)==1{p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $S > /tmp/kern.s
echo "$N: $(wc -l < /tmp/kern.s) lines, readlane $(grep -c v_readlane /tmp/kern.s) writelane $(grep -c v_writelane /tmp/kern.s) barriers $(grep -c s_barrier /tmp/kern.s) scratch $(grep -c scratch_ /tmp/kern.s)"
awk -v n="$N" 'index($0, n ".num_vgpr"){p=1} p&&/^; (NumVgprs|TotalNumSgprs|ScratchSize|codeLenInByte)/{print} /^; Occupancy/{if(p)exit}' $S
