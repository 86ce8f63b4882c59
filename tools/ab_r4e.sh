#!/bin/bash
# round 4: full GPU suite on the current tree, slab soak with the global balance (no drift / 6 m/s drift)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest.txt 2>&1; tail -4 gpurun_out/r4e/pytest.txt
(echo "# tools/debug/soak_slabs.py 4 [drift]: global balance on the block-plane histogram (round 4)"; python tools/debug/soak_slabs.py 4; python tools/debug/soak_slabs.py 4 6) > gpurun_out/r4e/soak_slabs.txt 2>&1
cat gpurun_out/r4e/soak_slabs.txt
