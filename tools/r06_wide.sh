#!/bin/bash
# r06_wide.sh: A/B of the wide (512-thread) k_g2p2g on short launches, one box: the 10 M column cut 8 ways on one device, the 1 M neo-Hookean cube,
# (the variant is not in the tree: apply tools/wide_workgroup_experiment.patch first -- profiles/r06_kernel_experiments.txt, 6)
# and the parity tests that run the Drucker-Prager / neo-Hookean views with the wide kernel forced on every launch.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
filt() { grep -E "one pipeline|substeps|FIXED" | sed -E "s/per slab \[.*validator/validator/" | cut -c1-330; }
{
for w in 0 4096; do
  echo "== SPARKL_HIP_WIDE_BLOCKS=$w: the 10 M column cut 8 ways, peer-direct"
  SPARKL_HIP_WIDE_BLOCKS=$w python tools/probe.py slabs --world 8 --scene column --strong --substeps 600 --report 200 --single 2>&1 | filt
  echo "== SPARKL_HIP_WIDE_BLOCKS=$w: neo1m"
  SPARKL_HIP_WIDE_BLOCKS=$w python bench.py --config neo1m --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))"
done
echo "== parity with the wide kernel on every launch"
SPARKL_HIP_WIDE_BLOCKS=100000000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -q -m gpu -x 2>&1 | tail -5
} > gpurun_out/r06/wide_ab.txt 2>&1
cat gpurun_out/r06/wide_ab.txt
