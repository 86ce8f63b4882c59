#!/bin/bash
# A/B harness: for every lib variant given, check P2G parity on a small scene and time the 10M bench.
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export SPARKL_HIP_LIB=$PWD/sparkl_amd/lib/$lib
  echo "=== $lib ${SPARKL_HIP_NO_FUSE:+(no fuse)}"
  python tools/tools_diag_p2g.py 2>&1 | head -1
  python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-late-phase 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/step %.3f  value %.3e  kernels %s  step_frac %.3f  roofline %s %.3f' % (d['ms_per_step'], d['value'], {k: round(v, 3) for k, v in d['kernels_avg_launch_ms'].items()}, d['step_frac_of_hbm_peak'], d['roofline']['kernel'], d['roofline']['frac']))"
done
