#!/bin/bash
# r06_bench8.sh: bench.py's DEFAULT N = 8 path exactly as the driver launches it (torch.distributed.run, one process per rank), on the ONE GPU of the box:
# the two substitutions a one-GPU box needs are the host-staged transport behind the RCCL function table (RCCL refuses two ranks on one device) and all ranks on
# device 0.  Eight processes map each other's exchange arenas through hipIpc (56 mappings), every grid update waits for eight mails.  NOT a scaling number.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
export SPARKL_HIP_TRANSPORT=host SPARKL_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 SPARKL_HIP_COMM_TIMEOUT_S=60
for n in 8 4 2; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 50 --warmup 10 --no-copy-peak --clock-wake-ms 0 \
   > gpurun_out/r06/bench_${n}procs_one_device.json 2> gpurun_out/r06/bench_${n}procs_one_device.err
echo "rc $? (n = $n)"; python tools/r06_bench_brief.py gpurun_out/r06/bench_${n}procs_one_device.json; python - <<PY
import json
o = json.load(open("gpurun_out/r06/bench_${n}procs_one_device.json"))
print("n_gpus", o["n_gpus"], "ranks", o["ranks"], "scaling", o["scaling"], "particles_total", o["config"]["particles_total"], "direct", o.get("direct_exchange"), "halo_ms", o.get("halo_ms_per_step"))
PY
tail -2 gpurun_out/r06/bench_${n}procs_one_device.err
done
