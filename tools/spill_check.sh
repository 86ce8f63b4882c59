#!/bin/bash
# spill_check.sh [kernel-mangled-prefix]: device-only compile of the working tree; prints VGPRs / scratch and every scratch instruction (with the
# s_waitcnt in front of it) of one kernel -- a spill of a prefetched register inside the particle loop waits for the whole prefetch (vmcnt(0)).
cd "$(dirname "$0")/../sparkl_amd/csrc"
K=${1:-_ZN3spk7k_g2p2gILin1ELb0ELb1ELi1E}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize $EXTRA -S --cuda-device-only -o /tmp/spill_k.s sparkl_hip.hip 2>&1 | grep -E "error" | head
awk -v k="$K" 'index($1,k)==1 && $1 ~ /:$/{p=1} p{print} p&&/^\.Lfunc_end/{exit}' /tmp/spill_k.s > /tmp/spill_kern.s
echo "lines $(wc -l < /tmp/spill_kern.s)  valu $(grep -cE '^\s+v_' /tmp/spill_kern.s)  lds $(grep -cE '^\s+ds_' /tmp/spill_kern.s)"
grep -A60 "amdhsa_kernel $K" /tmp/spill_k.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" | tr -s '\t ' ' ' | tr '\n' ';'; echo
grep -n -B1 "scratch_" /tmp/spill_kern.s | grep -E "scratch_|waitcnt" | cut -c1-110
