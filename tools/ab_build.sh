#!/bin/bash
# ab_build.sh <name> [extra hipcc flags]: build sparkl_amd/lib/ab_<name>.so from the working tree with extra flags (A/B variants;
# the .so files are git-ignored but travel to the GPU box).  Select one with SPARKL_HIP_LIB=sparkl_amd/lib/ab_<name>.so.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result "$@" \
    -o sparkl_amd/lib/ab_$name.so sparkl_amd/csrc/sparkl_hip.hip
echo built sparkl_amd/lib/ab_$name.so "$@"
