#!/bin/bash
# ab_r4.sh <case>: what round 4 ran on the GPU box (gpurun -- bash tools/ab_r4.sh <case>); results under profiles/r04*.  Cases b - d need the
# library variants of that experiment (tools/ab_build.sh / tools/build_rev.sh, see profiles/r04_kernel_experiments.txt); kept as the record of the commands.
cd $GRAFT_REPO_ROOT
case "$1" in
mset_occupancy)
# round 4, first A/B on one box: model-specialised k_g2p2g (MSET) and the occupancy variants
mkdir -p gpurun_out/r4a
hipcc --offload-arch=gfx950 -O3 -o /tmp/pl tools/microbench/permlane_swap.hip 2>/dev/null && /tmp/pl > gpurun_out/r4a/permlane.txt 2>&1
for rep in 1 2; do
python tools/ab_bench.py base=,SPARKL_HIP_MSET=0 mset1= mb3=sparkl_amd/lib/ab_mb3.so mb2=sparkl_amd/lib/ab_mb2.so >> gpurun_out/r4a/ab.txt 2>&1
done
cat gpurun_out/r4a/permlane.txt gpurun_out/r4a/ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.txt 2>&1; tail -5 gpurun_out/r4a/pytest.txt
;;
pair_return_map)
# round 4: the pair return map (MSET=2) against the model-specialised kernel (MSET=1), three prefetch positions; parity of the pair path
mkdir -p gpurun_out/r4b
SPARKL_HIP_MSET=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sand or sheared or column or rebin or deferred" > gpurun_out/r4b/pytest_pair.txt 2>&1; tail -5 gpurun_out/r4b/pytest_pair.txt
for rep in 1 2; do
python tools/ab_bench.py mset1=,SPARKL_HIP_MSET=1 pair=,SPARKL_HIP_MSET=2 pair_pf3=sparkl_amd/lib/ab_pf3.so,SPARKL_HIP_MSET=2 pair_pf1=sparkl_amd/lib/ab_pf1.so,SPARKL_HIP_MSET=2 >> gpurun_out/r4b/ab.txt 2>&1
done
cat gpurun_out/r4b/ab.txt
;;
diet_counters)
# round 4: instruction diet of k_g2p2g (FMA chains, det F - 1 from the return map, descriptor rows as pairs, 32-bit block offsets, dead-lane init)
mkdir -p gpurun_out/r4c
for rep in 1 2; do
python tools/ab_bench.py v0=sparkl_amd/lib/ab_v0.so diet= nomerge=sparkl_amd/lib/ab_nomerge.so >> gpurun_out/r4c/ab.txt 2>&1
done
cat gpurun_out/r4c/ab.txt
bash tools/pmc.sh v0 sparkl_amd/lib/ab_v0.so > gpurun_out/r4c/pmc.txt 2>&1
bash tools/pmc.sh diet "" >> gpurun_out/r4c/pmc.txt 2>&1
bash tools/pmc.sh nomerge sparkl_amd/lib/ab_nomerge.so >> gpurun_out/r4c/pmc.txt 2>&1
cat gpurun_out/r4c/pmc.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4c/pytest.txt 2>&1; tail -3 gpurun_out/r4c/pytest.txt
;;
diet_parts)
# round 4: which parts of the diet pay: descriptor rows as pairs (default) vs nine float rows (rows9), merge of split halves off (nomerge)
mkdir -p gpurun_out/r4d
for rep in 1 2; do
python tools/ab_bench.py --no-cold-window v0=sparkl_amd/lib/ab_v0.so diet= rows9=sparkl_amd/lib/ab_rows9.so rows9nm=sparkl_amd/lib/ab_rows9nm.so nomerge=sparkl_amd/lib/ab_nomerge.so >> gpurun_out/r4d/ab.txt 2>&1
done
cat gpurun_out/r4d/ab.txt
timeout 600 python -m pytest tests/test_gpu_slab.py -m gpu -x -q -k "comm_init_all or aborts or world_of_one" > gpurun_out/r4d/pytest.txt 2>&1; tail -5 gpurun_out/r4d/pytest.txt
SPARKL_HIP_LIB=$PWD/sparkl_amd/lib/ab_rows9nm.so SPARKL_HIP_LIB_AB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sand or sheared or rebin" > gpurun_out/r4d/pytest_nm.txt 2>&1; tail -3 gpurun_out/r4d/pytest_nm.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late-phase > gpurun_out/r4d/bench_driver.json 2>gpurun_out/r4d/bench_driver.err; python -c "
import json; o=json.loads(open('gpurun_out/r4d/bench_driver.json').read().strip().splitlines()[-1]); print('driver', o['ms_per_step'], o['step_frac_of_hbm_peak'], 'cold', o.get('cold_window'))"
;;
suite_and_soak)
# round 4: full GPU suite on the current tree, slab soak with the global balance (no drift / 6 m/s drift)
mkdir -p gpurun_out/r4e
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest.txt 2>&1; tail -4 gpurun_out/r4e/pytest.txt
(echo "# tools/probe.py slabs --world 4 [drift]: global balance on the block-plane histogram (round 4)"; python tools/probe.py slabs --world 4; python tools/probe.py slabs --world 4 --drift 6) > gpurun_out/r4e/soak_slabs.txt 2>&1
cat gpurun_out/r4e/soak_slabs.txt
;;
suite)
mkdir -p gpurun_out/r4f
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r4f/pytest.txt 2>&1; tail -30 gpurun_out/r4f/pytest.txt
;;
suite_tail)
mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests/test_gpu_full_size_parity.py -m gpu -x -q -k "flowing or fluid" > gpurun_out/r4g/pytest1.txt 2>&1; tail -12 gpurun_out/r4g/pytest1.txt
timeout 1500 python -m pytest tests/test_gpu_slab.py tests/test_golden_trajectories.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4g/pytest2.txt 2>&1; tail -12 gpurun_out/r4g/pytest2.txt
;;
final)
# the round's record: full GPU suite (parity.jsonl), bench lines of the three single-GPU configurations (default and driver-style), rocprofv3 passes
mkdir -p gpurun_out/r04; rm -f gpurun_out/parity.jsonl
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest.txt 2>&1; tail -3 gpurun_out/r04/pytest.txt
cp gpurun_out/parity.jsonl gpurun_out/r04/r04_parity.jsonl
python bench.py > gpurun_out/r04/r04_bench_sand.json 2> gpurun_out/r04/bench_sand.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/r04_bench_sand_driver.json 2>> gpurun_out/r04/bench_sand.err
for c in neo1m fluid20m; do
python bench.py --config $c --no-cpu-baseline > gpurun_out/r04/r04_bench_${c%%[0-9]*}.json 2> gpurun_out/r04/bench_$c.err
python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/r04_bench_${c%%[0-9]*}_driver.json 2>> gpurun_out/r04/bench_$c.err
done
bash tools/prof.sh r04 sand10m 32 > /dev/null 2>&1; cp gpurun_out/prof_r04_sand10m/kernel_stats.csv gpurun_out/r04/r04_sand10m_kernel_stats.csv; cp gpurun_out/prof_r04_sand10m/pmc_summary.txt gpurun_out/r04/r04_sand10m_pmc_summary.txt
bash tools/prof.sh r04 fluid20m 32 > /dev/null 2>&1; cp gpurun_out/prof_r04_fluid20m/kernel_stats.csv gpurun_out/r04/r04_fluid20m_kernel_stats.csv; cp gpurun_out/prof_r04_fluid20m/pmc_summary.txt gpurun_out/r04/r04_fluid20m_pmc_summary.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/r04_bench_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r = o.get("roofline") or {}
    line = f"{f}: {o['ms_per_step']:.4f} ms step {o['step_frac_of_hbm_peak']:.3f} kernel {r.get('kernel')} {r.get('frac', 0):.3f}"
    for k in ("cold_window", "late_collapse", "deep_collapse"):
        if o.get(k): line += f" | {k} {o[k]['ms_per_step']:.4f} ({o[k].get('step_frac_of_hbm_peak', 0):.3f})"
    print(line)
PY
;;
*) echo "cases: mset_occupancy pair_return_map diet_counters diet_parts suite_and_soak suite suite_tail final" ;;
esac
