#!/bin/bash
# 2-rank slab bench on ONE GPU (gloo staging): exercises bench.py's N>1 path end to end
cd $GRAFT_REPO_ROOT
SPARKL_BENCH_BACKEND=gloo SPARKL_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 4 --particles 1e6 --no-cpu-baseline --no-late-phase 2>&1 | grep -v "socket.cpp\|Gloo\|amdgpu.ids" | grep -v "^\[W\|^  \|^=\|^$" | tail -30
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-late-phase 2>/dev/null | cut -c1-400
