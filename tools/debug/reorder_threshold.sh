#!/bin/bash
# late / deep phase of the 10M collapse against the in-place reorder threshold (run heads per particle; library default 0.6)
cd $GRAFT_REPO_ROOT
for h in ${HEADS:-0.6 1.5 3.0 6.0}; do
  SPARKL_BENCH_REORDER_HEADS=$h timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-copy-peak > /tmp/b_$h.json 2>/dev/null
  python - <<PY
import json
o=json.load(open("/tmp/b_$h.json"))
l,d=o["late_collapse"],o["deep_collapse"]
print("heads/particle $h: late %.4f (g2p2g %.4f reorder %.2f ms) deep %.4f (g2p2g %.4f reorder %.2f ms resorts %d)"%(l["ms_per_step"], l["kernels_avg_launch_ms"]["g2p2g"], l["reorder_ms_total"], d["ms_per_step"], d["kernels_avg_launch_ms"]["g2p2g"], d["reorder_ms_total"], d["resorts"]))
PY
done
