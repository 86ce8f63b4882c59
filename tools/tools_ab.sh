#!/bin/bash
# A/B harness: for every lib variant given, check P2G parity on a small scene and time the 10M bench.
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export SPARKL_HIP_LIB=$PWD/sparkl_amd/lib/$lib
  echo "=== $lib"
  python tools/tools_diag_p2g.py 2>&1 | head -1
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/step %.3f  value %.3e  kernels %s  frac %.3f' % (d['ms_per_step'], d['value'], {k: round(v, 3) for k, v in d['kernels_ms_per_step'].items()}, d['step_frac_of_hbm_peak']))"
done
