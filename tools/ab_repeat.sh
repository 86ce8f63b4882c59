#!/bin/bash
# ab_repeat.sh N spec...: N interleaved repeats of tools/ab_bench.py (same arguments), to average out box / clock noise
N=$1; shift
for i in $(seq $N); do python tools/ab_bench.py --steps 100 --warmup 20 "$@" 2>&1 | grep -v amdgpu.ids; done
