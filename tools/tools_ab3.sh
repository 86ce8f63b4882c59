#!/bin/bash
# timing-only A/B over lib variants (results of SPK_EXP_* builds are physically wrong on purpose)
cd $GRAFT_REPO_ROOT
export SPARKL_HIP_LIB_AB=1
for lib in "$@"; do
  export SPARKL_HIP_LIB=$PWD/sparkl_amd/lib/$lib
  echo "=== $lib"
  python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-late-phase 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/step %.3f  kernels %s' % (d['ms_per_step'], {k: round(v, 3) for k, v in d['kernels_avg_launch_ms'].items()}))"
done
