#!/bin/bash
# build_rev.sh <git-rev> <name> [EXTRA flags]: build sparkl_amd/lib/<name>.so from the kernel sources of a git revision (A/B baselines)
set -e
cd "$(dirname "$0")/.."
rev=$1; name=$2; shift 2
d=/tmp/spk_rev_$name; rm -rf $d; mkdir -p $d/sparkl_amd/csrc $d/include
for f in $(git ls-tree --name-only $rev sparkl_amd/csrc/ | grep -E "\.(h|hip)$") include/sparkl_hip.h; do git show $rev:$f > $d/$f; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result "$@" -o sparkl_amd/lib/$name.so $d/sparkl_amd/csrc/sparkl_hip.hip
echo built sparkl_amd/lib/$name.so from $rev
