#!/bin/bash
# r06_slabs.sh: the slab-loop measurements of round 6 on the GPU box (8 x-slabs of one device): peer-direct vs staged exchange, on the scaled column of
# VERDICT r05 (1.25 M particles per slab) and on the geometry of real strong scaling (the 10 M column cut 8 ways) -> gpurun_out/r06/slab8_*.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
filt() { grep -E "one pipeline|substeps" | sed -E "s/per slab \[.*validator/validator/" | cut -c1-330; }
{
echo "== scaled column, 1.25 M per slab, peer-direct exchange"; python tools/probe.py slabs --world 8 --scene column --per-slab 1.25e6 --substeps 600 --report 200 --single 2>&1 | filt
echo "== scaled column, 1.25 M per slab, staged exchange (SPARKL_HIP_DIRECT=0)"; SPARKL_HIP_DIRECT=0 python tools/probe.py slabs --world 8 --scene column --per-slab 1.25e6 --substeps 600 --report 200 --single 2>&1 | filt
echo "== the 10 M column cut 8 ways (192 x 250 x 200: slabs 3 node blocks thick), peer-direct"; python tools/probe.py slabs --world 8 --scene column --strong --substeps 600 --report 200 --single 2>&1 | filt
echo "== the same, staged"; SPARKL_HIP_DIRECT=0 python tools/probe.py slabs --world 8 --scene column --strong --substeps 400 --report 200 --single 2>&1 | filt
} > gpurun_out/r06/slab8_b.txt
cat gpurun_out/r06/slab8_b.txt
