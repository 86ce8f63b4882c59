#!/bin/bash
# isa_hist.sh <kernel-substring> [N]: resource usage + instruction histogram of one kernel (device-only compile)
cd "$(dirname "$0")/../sparkl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize $EXTRA -S --cuda-device-only -o /tmp/k.s sparkl_hip.hip 2>/dev/null
for sym in $(grep -E "^_ZN3spk[0-9]+$1.*:" /tmp/k.s | cut -d: -f1); do
echo "== $sym"
awk -v s="$sym:" '$1==s{p=1} p{print} p&&/^\.Lfunc_end/{exit}' /tmp/k.s > /tmp/kern.s
echo "lines $(wc -l < /tmp/kern.s)"
grep -A40 "amdhsa_kernel $sym" /tmp/k.s | grep -E "next_free_vgpr|next_free_sgpr|group_segment|accum_offset|private_segment_fixed" | tr -s '\t ' ' ' | tr '\n' ';'; echo
grep -oE "^\s+[a-z_0-9]+" /tmp/kern.s | sort | uniq -c | sort -rn | head -${2:-30} | awk '{printf "%s:%s  ", $2, $1} END{print ""}'
done
