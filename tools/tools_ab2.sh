#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "--- with kernel clock"; python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-late-phase 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], d.get('kernels_ms_per_step'))"
echo "--- without"; python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-late-phase --no-kernel-clock 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'])"
