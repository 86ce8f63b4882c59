"""One RANK of a two-process slab job on ONE GPU (tests/test_gpu_slab_procs.py starts it as a fresh child process).

The process runs the library's own group loop (sparkl_hip_group_substeps / sparkl_hip_group_step: the C++ code the N > 1 bench path and a
multi-GPU host run) with a communicator of `world` ranks over the host-staged test transport (SPARKL_HIP_TRANSPORT=host,
sparkl_amd/csrc/host_transport.h): bootstrap from a unique id, overlap calibration all-reduce, collective re-bins with migration and slab
balance, adaptive dt all-reduce, and the failure path when a peer dies.

usage: slab_proc_worker.py RANK WORLD UID_FILE KIND NSTEPS OUT_DIR [die_after=K] [adaptive=FRAMES] [balance=0|1]
exit codes: 0 ok, 3 the group call failed (message on stderr), 17 this rank was told to die."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPARKL_HIP_TRANSPORT"] = "host"     # before the library loads its communication table

import numpy as np  # noqa: E402

from sparkl_amd._lib import SparklHipError  # noqa: E402
from sparkl_amd.particles import ParticleComponents  # noqa: E402
from sparkl_amd.pipeline import MpmPipeline  # noqa: E402


def main():
    rank, world, uid_file, kind, nsteps, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    opts = dict(a.split("=", 1) for a in sys.argv[7:])
    die_after = int(opts.get("die_after", -1))
    adaptive = int(opts.get("adaptive", 0))
    from tests.test_gpu_slab import _ownership, _scene
    sc, dt = _scene(kind)
    cuts, all_gids = _ownership(sc, world)
    gids = all_gids[rank]
    id_to_caller = np.concatenate(all_gids)
    comps = ParticleComponents(0)
    for k in ParticleComponents.__slots__:
        setattr(comps, k, np.ascontiguousarray(getattr(sc.particles, k)[gids]))
    # bootstrap: rank 0 creates the id, the others wait for the file (what MPI / torch.distributed would broadcast)
    if rank == 0:
        uid = MpmPipeline.comm_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                sys.exit("rank %d: no unique id after 120 s" % rank)
            time.sleep(0.01)
        uid = open(uid_file, "rb").read()
    pipe = MpmPipeline(sc.cell_width, device_id=0, domain_log2_bytes=sc.domain_log2_bytes)
    pipe.set_slab(0, cuts[rank], cuts[rank + 1], id_offset=int(sum(len(g) for g in all_gids[:rank])))
    pipe.slab_balance(opts.get("balance", "1") != "0")
    pipe.comm_init(uid, rank, world)
    assert pipe.comm_allreduce_sum(rank + 1) == world * (world + 1) // 2
    pipe.set_models(sc.models)
    pipe.set_colliders(sc.colliders)
    pipe.upload_particles(comps)
    nsub, sim = 0, 0.0
    try:
        if die_after >= 0:
            if die_after > 0:
                MpmPipeline.group_substeps([pipe], sc.params, sc.gravity, dt, die_after)
            if rank == world - 1:
                os._exit(17)                       # no close, no abort: a killed rank
            MpmPipeline.group_substeps([pipe], sc.params, sc.gravity, dt, max(nsteps - die_after, 1))
        elif adaptive:
            for _ in range(adaptive):
                t, n = MpmPipeline.group_step([pipe], sc.params, sc.gravity)
                sim += t
                nsub += n
        else:
            MpmPipeline.group_substeps([pipe], sc.params, sc.gravity, dt, nsteps)
            nsub = nsteps
    except SparklHipError as e:
        print("rank %d: group call failed: %s" % (rank, e), file=sys.stderr, flush=True)
        os._exit(3)
    got = pipe.read_particles()
    local_ids = pipe.read_sorted_particle_ids()
    ax, lo, hi, moves = pipe.slab_get()
    from tests.parity import save_components
    save_components(os.path.join(out_dir, f"rank{rank}.npz"), got, pos=got.position["point"], vel=got.velocity["vector"], F=got.volume["deformation_gradient"],
             C=got.velocity_gradient["m"], failed=got.status["failed"], caller=id_to_caller[local_ids], rebins=pipe.group_rebins(), n0=len(gids), n1=len(got),
             nsub=nsub, sim=sim, overlap=np.array(pipe.overlap_info()), direct=np.array([int(pipe.direct_info()["ready"]), pipe.direct_info()["epochs"]]), direct_fine=int(pipe.direct_info()["arena_fine_grained"]), direct_handshakes=pipe.direct_info()["handshakes"], direct_chosen=str(pipe.direct_info()["chosen"]), direct_timed_us=np.array(list(pipe.direct_info()["timed_us_per_substep"].values())), cut_lo=lo, cut_hi=hi, cut_moves=moves)
    pipe.close()


if __name__ == "__main__":
    main()
