// HIP kernels of the MI355X MPM hot path (gfx950, wave64).  See DESIGN.md for the data layout.
//
// Particles live in HBM as eight float4 streams, physically ordered by (home block, cell):
//   Q0 = (x, y, z, F0)  Q1 = (F1..F4)  Q2 = (F5..F8)        F column-major, stored as F - I; (x, y, z) RELATIVE TO THE HOME BLOCK, in cells (cell_frac)
//   Q3 = (vx, vy, vz, C0) Q4 = (C1..C4) Q5 = (C5..C8)       C = velocity_gradient (APIC affine)
//   D  = (plastic_hardening, log_vol_gain, plastic_def_grad_det, psi_pos)
//   K  = (mass, volume0, elastic_hardening, flags)           read-only on the hot path
// so that P2G streams exactly 7 x 16 B = 112 B per particle with dwordx4 loads.
//
// The sparse grid is a list of 4x4x4 node blocks (Morton ordered) with a 27-entry neighbour table;
// every block that holds particles owns an 8x8x8 "tile" (its 4^3 nodes, the +2 B-spline support and a
// one-cell drift margin on each side).  P2G accumulates a tile in LDS and writes it once (no global
// atomics); grid_update gathers the <= 8 overlapping tiles of each node; G2P stages the 8^3 node
// velocities back into LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "probe_hooks.h"

namespace spk {

constexpr int TILE_W = 8;
constexpr int TILE_N = TILE_W * TILE_W * TILE_W;  // 512 nodes
constexpr int BLK_THREADS = 256;
// The f64 accumulation tile of the scatter kernels lives in LDS with padded z-planes: logical node n = x + 8 y + 64 z sits at
// acc_slot(n) = n + 8 z - 4 (z & 1), i.e. the planes start at 0, 68, 144, 212, 288, ... doubles.  A wave's scatter tasks hit,
// in ONE ds_add_f64, the same (sx, sy) node of 8 neighbouring cells (4 in x, 2 in y) in 3 consecutive z-planes: with a 64-double
// plane stride the three planes fall on the same LDS banks (512 B = 2 x the bank period), with this layout the 24 addresses
// occupy 24 different bank pairs whichever plane the run starts in (residues 0, 4, 16, 20 mod 32 doubles, cyclically).
constexpr int ACC_N = 576;

enum : uint32_t {
    FLAG_FAILED = 1u << 0,
    FLAG_STATIC = 1u << 1,
    FLAG_KINEMATIC = 1u << 2,
    FLAG_INVALID = 1u << 3,       // outside the SPGrid domain: frozen, scatters nothing
    FLAG_PHASE_NOT_ONE = 1u << 4, // phase lives in the side array
    FLAG_REMOVED = 1u << 5,       // migrated to a neighbour slab; dropped at the next re-bin
    FLAG_MODEL_SHIFT = 8
};

__device__ __forceinline__ int acc_slot(int n) { const int p = n >> 6; return n + 8 * p - 4 * (p & 1); }
// acc_slot of the (0, 0) node of support plane sz of a particle whose associated cell has tile index `base` (x, y <= 5, so base + sx + 8 sy
// never carries into the plane bits): the nine nodes of the plane are then this + sx + 8 sy -- constant offsets the ds_add_f64 encodes itself,
// where acc_slot(base + off) per node costs six integer instructions, 27 times per deposit
__device__ __forceinline__ int acc_plane_base(int base, int sz) { const int p = (base >> 6) + sz; return base + 64 * sz + 8 * p - 4 * (p & 1); }

struct Ctl {
    uint32_t abort_flag[2];  // parity double-buffer: kernels of substep s read [s&1], G2P of s writes [(s+1)&1]
    uint32_t steps_done;     // substeps whose G2P ran
    uint32_t dt_bits;        // atomicMin target of the dt estimator (float bits, dt >= 0)
    uint32_t dt_nan;
    uint32_t num_invalid;    // particles that left the domain since the last re-bin
    // Re-bin-on-disorder policy.  Between re-bins particles keep their slots, so lanes that used to share a cell drift
    // apart, the scatter's DPP pre-reduction finds shorter runs and the LDS atomics multiply (k_g2p2g went 0.57 -> 1.0 ms
    // over ~300 substeps of the collapsing column).  Every scatter counts its run heads; once the heads in excess of the
    // count right after the last re-bin, summed over the substeps since, pass `excess_limit` (= the cost of a re-bin in
    // heads), the abort chain is raised exactly like for a margin violation and the host re-bins.
    uint32_t heads[2];       // run heads of the scatter of substep t, at [t & 1]
    uint32_t heads_base;     // heads of the first scatter after the last re-bin (0 = not measured yet)
    float excess_acc;
    float excess_limit;      // 0 = policy off
    uint32_t disorder_aborts;
    uint32_t last_heads;
    // cheaper remedy, tried first: k_block_reorder re-sorts every particle block by current cell in place (no host
    // round trip, block membership unchanged).  Requested once the excess passes `reorder_limit`.
    float reorder_limit;     // 0 = off
    uint32_t reorder_request;
    uint32_t reorders;
    // adaptive step, look-ahead: k_dt_reduce found a bound other than the one the already enqueued substeps were planned
    // with and stopped them through the abort chain (the abort is NOT a re-bin request; the host clears both)
    uint32_t dt_mismatch;
    // velocity bound h / (|v| + 6 sqrt(3) (h^2/4) |C| / h) of the particles that left the domain (float bits, min): the reference keeps
    // them in the set with their last v and C, so `adaptive_timestep_length` (timestep_estimator.rs:25-35) still sees them, while no
    // kernel here visits them any more.  Rebuilt at every re-bin, extended by G2P when a particle dies; k_dt_reduce folds it in.
    uint32_t dead_dt_bits;
    // peer-direct halo exchange (PeerBox below): a kernel gave up waiting for a peer's mail or halo records (1 = mail, 2 = halo): the peer is gone.
    // The abort chain is raised with it, the host turns it into an error (sparkl_hip_group_substeps)
    uint32_t direct_error;
    // Two words of PINNED HOST memory the device can write (hipHostMalloc, mapped): [0] is set to 1 by whatever RAISES the abort chain (margin
    // violation, disorder policy, dt mismatch), [1] mirrors steps_done.  The host enqueues a fixed-dt batch ahead of the GPU; with these it
    // stays only a few substeps ahead (polling [1], no events, no synchronisation) and stops enqueueing the moment [0] is raised -- the rest
    // of a 128-substep batch used to run as ~60 no-op launches behind every re-bin request of the flowing phases.  NULL = not used.
    uint32_t* host_words;
};
// The host words are written through a GLOBAL-address-space pointer.  `ctl->host_words` loaded from memory is a generic pointer, a store
// through it is a FLAT instruction, and behind a path that may have issued one the compiler can no longer count the vector-memory
// operations in flight: every s_waitcnt vmcnt(N) of the kernels that inline this -- the prologue's "small loads first, particle loads stay
// in flight" among them -- became vmcnt(0): +20 % on the 1 M cube, measured (profiles/r05_kernel_experiments.txt).
typedef __attribute__((address_space(1))) uint32_t* global_u32p;
// raise the abort chain for the substeps from parity `np` on (and tell the host, see Ctl::host_words)
SPK_DEV void raise_abort(Ctl* ctl, int np) {
    ctl->abort_flag[np] = 1u;
    const global_u32p hw = (global_u32p)ctl->host_words;
    if (hw) hw[0] = 1u;
}
SPK_DEV void note_step(Ctl* ctl) {
    const uint32_t v = atomicAdd(&ctl->steps_done, 1u) + 1u;
    const global_u32p hw = (global_u32p)ctl->host_words;
    if (hw) hw[1] = v;
}

// ---- peer-direct halo exchange (round 6) -------------------------------------------------------------------------------------------------
// The group loop of rounds 3 - 5 spent, per slab and substep, one 4-byte all-reduce (or a single-thread kernel), a pack kernel, two copies (or an
// ncclSend / Recv pair) and a merge kernel in series in front of the grid update: 28 - 60 us of launch latencies for ~1 MB.  When every rank can
// map its neighbours' memory -- members of one process (the same device, or peer access between devices: the reference's own shape,
// cuda_mpm_pipeline.rs:458-472), or hipIpc mappings between the processes of one node -- the exchange is ONE kernel per slab and substep:
//   k_halo_push   (behind the particle kernel of substep s - 1)  posts this member's re-bin flag of substep s into EVERY member's mailbox, packs the
//                 partial sums of the shared node planes and stores them straight into the neighbours' halo buffers of parity s & 1, at the node-block
//                 slots negotiated at the re-bin (k_halo_keys / k_halo_slots); the last workgroup publishes "all records of epoch e have landed";
//   k_grid_update (direct)  waits for the mail of all members (the MAX of the flags is the agreed flag of the substep: every member skips the same
//                 substeps, what the all-reduce was for) and for the neighbours' epoch words, then reads the halo like before.
// Everything a peer writes lives in one fine-grained allocation per member (the "arena": PeerBox + two halo buffers).  Mail and epoch words are
// double-buffered by epoch parity: a fast peer is at most one substep ahead (its push of epoch e + 1 sits behind its grid update of epoch e, which
// waited for THIS member's mail of epoch e), never two.  Payload: plain stores, per wave s_waitcnt vmcnt(0), workgroup barrier, one lane's system-scope
// (write-through, system scope: halo_store_wt), per wave s_waitcnt vmcnt(0), workgroup barrier, then the arrival counter; flag words: relaxed
// system-scope atomics; the reader polls relaxed and reads the records past the caches (halo_load_wt) -- the "sc1 payload + drained flag / sc1
// loads" form of MI355X_MICROARCH.md (inter-workgroup visibility), at system scope and on fine-grained memory because the writer is another device.
// A peer that never answers: the wait gives up after `timeout_ticks` of the 100 MHz wall clock, raises the abort chain and Ctl::direct_error.
constexpr int MAX_DIRECT = 64;
struct PeerBox {
    unsigned long long mail[2][MAX_DIRECT];   // [epoch & 1][writer]: (epoch << 32) | re-bin flag of that member for that substep
    unsigned int halo_epoch[2][2];            // [epoch & 1][side]: written by the neighbour on `side` once all of its records of that epoch have landed
    unsigned int pad[12];
};
typedef __attribute__((address_space(1))) unsigned long long* global_u64p;
struct DirectWait {
    const PeerBox* box;        // this member's box, NULL = not a direct launch
    int n_mail;                // members in the job
    unsigned int epoch;
    int wait_side[2];          // a neighbour on that side pushes records
    unsigned long long timeout_ticks;
    // 1: a one-wave k_direct_agree in front of this launch has already waited (mail AND records) and left the agreed flag in the control block: no
    // wave of this launch spins.  Set when members of different processes share a physical device (test setups, partitioned GPUs): a grid update
    // whose every wave polls fills the device with its own spinning waves, and the kernel of the OTHER process that would post the mail never gets a
    // slot -- measured as a 20 s stall on the first substep of two 5 M-particle ranks on one MI355X.  With one rank per device the wait stays
    // inside the grid update (one launch less per substep).
    int pre_agreed;
};
struct DirectPush {
    const int* list[2];        // halo columns of each cut (k_halo_list)
    const int2* dst[2];        // per column: node-block slot of the UPPER block (.x) and of the LOWER block (.y) in the neighbour's block list, -1 = not there
    int n[2];
    float4* remote_halo[2];    // the neighbour's halo buffer of this epoch's parity
    unsigned int* remote_epoch[2];   // &neighbour_box->halo_epoch[epoch & 1][its side that faces this member]
    unsigned long long* mail[MAX_DIRECT];   // &box_of_member_k->mail[epoch & 1][this member's index], k < n_mail
    int n_mail;
    unsigned int epoch;
    unsigned int* arrive;      // arrival counter of this launch's workgroups: plain device memory of this member (zero between launches)
};
#define SPK_RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
// the agreed re-bin flag of the substep: MAX over the mail of all members; false = timed out
SPK_DEV bool direct_wait_mail(const DirectWait& W, int lane, bool& aborted) {
    const unsigned long long t0 = wall_clock64();
    const int par = W.epoch & 1u;
    unsigned long long v = (unsigned long long)W.epoch << 32;
    for (unsigned spin = 0;; ++spin) {
        if (lane < W.n_mail) v = __hip_atomic_load((global_u64p)&W.box->mail[par][lane], SPK_RLX_SYS);
        if (__all((unsigned int)(v >> 32) == W.epoch)) break;
        if ((spin & 63u) == 63u && wall_clock64() - t0 > W.timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    aborted = __any((unsigned int)v != 0u);
    return true;
}
SPK_DEV bool direct_wait_halo(const DirectWait& W, int lane) {
    const unsigned long long t0 = wall_clock64();
    const int par = W.epoch & 1u;
    for (unsigned spin = 0;; ++spin) {
        unsigned int e = W.epoch;
        if (lane < 2 && W.wait_side[lane]) e = __hip_atomic_load((global_u32p)&W.box->halo_epoch[par][lane], SPK_RLX_SYS);
        if (__all(e == W.epoch)) break;
        if ((spin & 63u) == 63u && wall_clock64() - t0 > W.timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return true;   // the records are read with halo_load_wt: no fence
}
// The halo payload travels WRITE-THROUGH and is read past the caches: system scope on either side (global_store / global_load ... sc0 sc1; the load as
// two 8-byte relaxed atomics, the store as below).  No release fence in the sender (a buffer_wbl2 per workgroup) and, above all, no acquire fence in
// the receiver: a buffer_inv per wave of the grid update serialised on the L2 and cost more than the whole staged exchange had (round 6, first cut).
// The store as ONE 16-byte instruction with the same scope bits the two 8-byte atomics compile to (global_store_dwordx2 ... sc0 sc1): half the write
// requests on the link for the same bytes (a record is 16 B; the lanes of a cut along x hit every fourth record of a block, so nothing wider coalesces).
// Nobody reads a record before the sender's epoch word, which is stored behind s_waitcnt vmcnt(0) of every storing wave: tearing cannot be observed.
typedef float spk_f4v __attribute__((ext_vector_type(4)));
SPK_DEV void halo_store_wt(float4* p, const float4& v) {
    const spk_f4v r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(r) : "memory");
}
SPK_DEV float4 halo_load_wt(const float4* p) {
    const global_u64p q = (global_u64p)p;
    const unsigned long long a = __hip_atomic_load(q, SPK_RLX_SYS), b = __hip_atomic_load(q + 1, SPK_RLX_SYS);
    return make_float4(__uint_as_float((unsigned int)a), __uint_as_float((unsigned int)(a >> 32)), __uint_as_float((unsigned int)b), __uint_as_float((unsigned int)(b >> 32)));
}
SPK_DEV void direct_give_up(Ctl* ctl, int parity, unsigned int code) {
    ctl->direct_error = code;
    raise_abort(ctl, parity);
    raise_abort(ctl, (parity + 1) & 1);
}
struct ParticleArrays {
    float4* q[6];
    float4* d;
    float4* k;
    float4* kinvel;   // (kx, ky, kz, -)
    float4* aux;      // (phase, radius0, nacc_alpha, bits of the particle's caller/global id): cold, only moved by re-bins
    float4* frac;     // (crack_propagation_factor, crack_threshold, -, -): ParticleFracture, NULL unless the caller uploaded it
    // Mirror of the flags words (k[i].w) as a dense 4-byte stream, or NULL.  Built by the re-bin (k_permute) when every binned particle carries the
    // same (mass, volume0, elastic hardening) -- a lattice-sampled body, the benchmark's column -- and the models are the fused kernel's Drucker-Prager
    // view: that kernel (MSET = 4) then reads 4 bytes per particle where the k stream costs 16, the three constants travel in StepParams.  Kept in
    // step by everything that touches a binned particle's flags between two re-bins: the G2P stores (newly failed) and the in-place block reorder.
    uint32_t* fl;
};
SPK_DEV uint32_t particle_id(const ParticleArrays& P, int i) { return reinterpret_cast<const uint32_t*>(&P.aux[i])[3]; }

struct DevCuboid;
struct DevHeightField;
struct DevTriMesh;
struct DevPolyline;
struct GridArrays {
    const int4* blk_origin;    // cell coordinate of node (0,0,0) of the block, w = tile index or -1
    const int* blk_pfirst;
    const int* blk_pcount;
    const int* blk_nbr;        // [B][27], index of neighbour block or -1
    const int* blk_tnbr;       // [B][27], tile (= particle block) of that neighbour block or -1
    const int* pblk;           // [PB] block index of each tile owner
    const int4* pb_info;       // [PB][2]: (origin.xyz, block index), (pfirst, pcount, -, -): one hop instead of three
    float4* tiles;             // [PB][512] (mass, px, py, pz); a core-only tile keeps its 6^3 inner nodes compactly in front
    int* tile_mode;            // [PB] which margin planes the tile stores beside its 6^3 core (6 bits, see tile_write): 0 = core only, 63 = all 8^3; 64 = scalar mass tile
    // [PB][2] run heads of the block's last fused scatter, and of its first one after the last re-bin / reorder of the block (0 = not
    // taken yet): k_block_reorder only touches the blocks whose runs have fragmented (the static bulk of a collapse never does)
    int2* pb_heads;
    float4* node_vel;          // [B][64]  (vx, vy, vz, mass)
    float4* node_mom;          // [B][64]  (px, py, pz, -) debug capture or NULL
    const float4* halo_in;     // [B][64]  (mass, px, py, pz) received from neighbour slabs, or NULL
    int halo_axis, halo_lo, halo_hi;   // slab cuts (block planes): only the node blocks of planes lo - 1, lo, hi - 1, hi receive
    // Launch over a subset (slab groups overlap the halo exchange with the interior blocks: the cut-adjacent particle / node blocks run
    // on a second stream behind the exchange).  NULL = every block.  `primary`: this launch does the once-per-substep bookkeeping
    // (steps_done, disorder policy); exactly one launch of a kernel per substep has it.
    const int* pb_list; int pb_count;
    const int* nb_list; int nb_count;
    int primary;
    int pb_plain;              // 1: pb_list is a launch order, indexed by blockIdx.x itself (k_order_keys); 2: the same list read back to front; 0: a subset in Morton order, indexed through xcd_remap
    uint32_t* dt_partial;      // [gridDim] per-workgroup minimum of the dt bound (float bits), reduced by k_dt_reduce
    // Projection cache of the grid update (the reference's GPU path keeps `projection_status` / `projection_scaled_dir` in its grid nodes,
    // src_kernels/cuda/grid_update.rs:111-146): [heightfield | trimesh collider][B][64] (proj.xyz, status) of the node's projection on that
    // collider -- colliders are static between two sparkl_hip_colliders_set_* calls, node blocks between two re-bins, and both clear the
    // cache.  status (bits of .w): 0 not computed, 1 nothing within reach, 2 outside, 3 inside.  NULL = off (SPARKL_HIP_PROJ_CACHE=0).
    float4* node_proj;
    int* node_proj_blk;        // [collider][B]: 0 not computed, 1 no node of the block is within reach of the collider (the wave skips it without touching node_proj), 2 some are
    const DevCuboid* cuboids;  // the collider arrays, for enable_boundary_particle_projection (G2P tail)
    const DevHeightField* heightfields;
    const DevTriMesh* trimeshes;
    const DevPolyline* polylines;
    int num_polylines;
    int num_blocks;
    int num_pblocks;
};

struct StepParams {
    float h;
    float inv_d;      // 4 / h^2
    float inv_h;      // 1 / h (the host's IEEE quotient: the same bits as the device's division, without its twelve instructions per workgroup)
    float dt;
    float dt_h;       // dt / h: advection of the block-relative positions (cells per unit of velocity)
    float gx, gy, gz;
    int boundary_handling;
    uint32_t dofs;
    int parity;
    int dom;          // valid cell coords: 0 <= c, c + 5 < dom
    int num_cuboids;
    int num_heightfields;
    int num_trimeshes;
    int project_particles;   // SolverParameters::enable_boundary_particle_projection
    int collider_handling;   // some collider carries its own grid_boundary_handling (SparklColliderOptions): the grid update may not skip on boundary_handling == None
    int penalty;             // some collider has a penalty_stiffness: the RARE G2P kernels apply the force
    float ku_mass, ku_vol0, ku_eh;   // the (mass, volume0, elastic hardening) every binned particle shares when the launch runs on ParticleArrays::fl
    int jp_one;              // every particle was uploaded with plastic_deformation_gradient_det == 1 and every model is a plain fluid (nothing changes it): k_vol_p2g does not read it
    int want_dt;      // G2P also reduces the next substep's dt bound into ctl->dt_bits (adaptive stepping)
    int num_models;   // entries of the model table (<= MAX_LDS_MODELS: the particle kernels stage it in LDS)
    float dt_next;    // k_g2p2g: the step length the fused scatter (P2G of the NEXT substep) is built for; = dt for fixed-dt batches
    // k_g2p2g inside a fixed-dt batch: the next fused launch takes v and C from the grid, never from the particle, so v and C
    // (q[3..5]: 48 of the 112 B the kernel writes per particle) are only stored by the launch that ends the batch.  If the batch is cut
    // short (re-bin request), k_regather_c rebuilds them from the node velocities of the last substep that ran.
    int defer_c;
    // standalone P2G kernels (DPP scatter: their heads are lane-GROUP heads, another measure than the fused run-task scatter's distinct cells):
    // 1 = this launch feeds the disorder policy (Ctl::heads, pb_heads) -- the schedules that scatter with it every substep; 0 = it is the
    // incidental first scatter of a fused batch and stays out of the books (the policy then only ever sees k_g2p2g's measure)
    int count_heads;
    // k_grid_update behind k_vol_p2g (fluid volume sweep): the node masses are the ones the sweep's mass pass already merged into node_vel.w
    // (same particles, same weights), so k_vol_p2g scatters the momentum only and the tiles carry m = 0
    int mass_from_sweep;
};

// `handling`, `penalty`: SparklColliderOptions (per collider: -1 = SolverParameters::boundary_handling; 0 = no penalty force)
struct DevCuboid {
    float he[3];
    float q[4];
    float t[3];
    float friction;
    int handling;
    float penalty;
};
struct DevHeightField {
    const float* heights;   // nrows x ncols, column-major
    int nrows, ncols;
    float scale[3];
    float q[4];
    float t[3];
    float friction;
    int below_is_solid;
    float ymin, ymax;       // range of heights * scale[1] (local frame): a point farther above than the search reach cannot be near the surface
    const float2* coarse;   // [(nrows + 2) / 4][(ncols + 2) / 4] (min, max) of heights * scale[1] over blocks of 4 x 4 heightfield cells, column-major like `heights`
    int crows, ccols;
    int handling;
    float penalty;
    int flip_interior;      // with below_is_solid: inside = ABOVE the surface (gpu_collider.rs:82-88)
};

// Triangle mesh: 30 floats per triangle (a, b, c | pseudo-normals of a, b, c | of the edges ab, ac, bc | face normal; the
// normals are zero for a mesh that is not oriented) and a uniform grid over the bounding box: cell -> triangles whose box,
// grown by the search reach, touches the cell.
struct DevTriMesh {
    const float* tris;
    const int* cell_start;   // [ncells + 1]
    const int* cell_tris;
    float lo[3], hi[3];      // bounding box grown by the reach
    float inv_cell;
    int dims[3];
    float q[4];
    float t[3];
    float friction;
    int oriented, flip_interior;
    int handling;
    float penalty;
};

// Polyline collider (src_kernels/gpu_collider.rs:99-201, cuda_collider_set.rs:102-117): a CLOSED counter-clockwise polygon in the local x-y plane,
// extruded along local z; vertex i is joined to vertex (i + 1) % n.
struct DevPolyline {
    const float* verts;     // 3 floats per vertex (local frame)
    int n;
    float q[4];
    float t[3];
    float friction;
    int handling;
    float penalty;
    int flip_interior;
};

SPK_DEV int xcd_remap(int bid, int n);
// particle block of this workgroup: the whole list or the launch's subset
SPK_DEV int launch_pblock(const GridArrays& G) {
    if (G.pb_list) return G.pb_plain == 1 ? G.pb_list[blockIdx.x] : (G.pb_plain == 2 ? G.pb_list[G.pb_count - 1 - (int)blockIdx.x] : G.pb_list[xcd_remap(blockIdx.x, G.pb_count)]);
    return xcd_remap(blockIdx.x, G.num_pblocks);
}
// Observed placement is workgroup b -> XCD b % 8: give each XCD a contiguous run of Morton-ordered
// blocks so neighbouring tiles / node blocks share an L2.  Speed only, never correctness.
SPK_DEV int xcd_remap(int bid, int n) {
    const int per = n >> 3;
    if (per == 0) return bid;
    const int main_n = per << 3;
    if (bid >= main_n) return bid;
    return (bid & 7) * per + (bid >> 3);
}

// ---- block-relative positions (round 5) ----------------------------------------------------------------------------------------------
// A particle that belongs to a particle block (slot < n_binned) stores its position RELATIVE TO THE ORIGIN OF ITS HOME BLOCK, IN CELLS:
//   u = x / h - origin        (origin = cell coordinate of node (0, 0, 0) of the block; u in [-0.5, 5.5) while the particle is inside its tile).
// The associated cell (sp_grid.rs:216-226) is then  rint(u) - 1  in tile coordinates -- no multiply by 1/h, no tie test against the exact
// quotient, no subtraction of the origin --, the B-spline argument is  fx = u - (rint(u) - 1)  in [0.5, 1.5] (kernel.rs:63-73), exact up to
// the final + 1, and the coordinate keeps 2^-24 .. 2^-21 of a cell where a world coordinate of 10 m at h = 0.2 keeps 2^-18: the f32 position
// storage was the one place where this path was only as good as the reference's arithmetic.  The tail of the particle
// arrays (slots >= n_binned: fresh uploads, appended particles, emigrants, particles that left the domain) is stored in WORLD coordinates;
// the re-bin converts in both directions (k_compute_keys, k_permute), the read-back kernels convert to world f32 with one rounding
// (world_from_rel).  The C ABI is unchanged: world f32 in, world f32 out.
// A cell tie (u exactly k + 0.5) goes to the even cell where the reference's f32::round goes away from zero: the weights are continuous
// across the tie (the far node's weight is exactly 0 there), so which of the two cells is "associated" changes nothing but the rounding.
SPK_DEV void cell_frac_r(const float u[3], const float r[3] /* rint(u) */, float fx[3], int lc[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        fx[a] = (u[a] - r[a]) + 1.0f;   // u - r is exact
        lc[a] = (int)r[a];              // tile coordinate of the associated cell: rint(u) - 1 (block coordinate) + 1 (the tile's margin cell)
    }
}
SPK_DEV void cell_frac(const float u[3], float fx[3], int lc[3]) {
    const float r[3] = {rintf(u[0]), rintf(u[1]), rintf(u[2])};
    cell_frac_r(u, r, fx, lc);
}
// world coordinate of a block-relative one and back, one rounding each (cold paths only: re-bin, read-back, collider projection of particles)
SPK_DEV float world_from_rel(float u, int org, float h) { return (float)(((double)org + (double)u) * (double)h); }
SPK_DEV float rel_from_world(float x, int org, float h) { return (float)((double)x / (double)h - (double)org); }
// round(x / h) of a WORLD coordinate as the reference computes it (sp_grid.rs:217: f32::round of the f32 quotient, half away from zero)
SPK_DEV float round_cell_world(float x, float h) { return roundf(x / h); }
// Home block of slot i < n_binned: the particle blocks tile [0, n_binned) in order (pb_info[2 j + 1].x = first slot of block j).
SPK_DEV int slot_pblock(const int4* __restrict__ pb_info, int npb, int i) {
    int lo = 0, hi = npb - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pb_info[2 * mid + 1].x <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// world position of slot i, whichever part of the arrays it sits in
struct BinnedView { const int4* pb_info; int npb; int n_binned; float h; };
SPK_DEV void slot_world_pos(const BinnedView& B, const float4& q0, int i, float out[3]) {
    if (i < B.n_binned) {
        const int4 org = B.pb_info[2 * slot_pblock(B.pb_info, B.npb, i)];
        out[0] = world_from_rel(q0.x, org.x, B.h); out[1] = world_from_rel(q0.y, org.y, B.h); out[2] = world_from_rel(q0.z, org.z, B.h);
    } else { out[0] = q0.x; out[1] = q0.y; out[2] = q0.z; }
}
// Is the associated cell of a block-relative position inside the home block's tile AND inside the SPGrid domain?
//   tile: rint(u) in [0, 5];   domain (sp_grid.rs, particle_set.rs:129-136): 1 <= origin + rint(u) < dom - 4
// Every block whose tile lies inside the domain (origin >= 1 and origin + 5 < dom - 4 on every axis: all of them but those on the border
// of the grid) only needs the tile test, on the integer cell the scatter needs anyway: three unsigned compares and one NaN test of the
// coordinate sum (v_cvt_i32_f32 turns NaN into 0, which would pass).  `border` (wave-uniform, one SGPR) sends the others to the full test.
// (a plain int handed on BY VALUE: as a one-byte struct passed by reference through the particle lambdas the flag was not scalarised -- the
// compiler parked it in 256 B of LDS, one byte per thread, and read it back every iteration)
typedef int TileRange;
SPK_DEV TileRange tile_range(const int4& org, int dom) {
    const bool border = org.x < 1 || org.y < 1 || org.z < 1 || org.x + 5 >= dom - 4 || org.y + 5 >= dom - 4 || org.z + 5 >= dom - 4;
    return __builtin_amdgcn_readfirstlane(border ? 1 : 0);
}

// DPP row shifts inside a 16-lane row: lane i reads lane i + N (zero beyond the row end).
template <int N>
SPK_DEV float dpp_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x100 + N, 0xF, 0xF, true));
}
template <int N>
SPK_DEV int dpp_shl_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }
typedef float f2v __attribute__((ext_vector_type(2)));
SPK_DEV int dpp_shr1_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xF, 0xF, false); }
SPK_DEV float dpp_shr1_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true)); }   // lane i reads lane i - 1 of its row (0 at the row start)

// The model table is read-only for the lifetime of a launch.  Reading it through per-lane global pointers costs a vector load per
// parameter, each followed by s_waitcnt vmcnt(0) -- which, loads returning in order, also waits for every particle load issued
// before it: the software prefetch of the next batch was being waited for at the first model parameter the compiler touched.
// The particle kernels therefore copy the table into LDS in their prologue (the barrier that follows the tile staging covers
// it) and read parameters with ds_read (lgkmcnt): no interplay with the vector memory counter.  Tables longer than
// MAX_LDS_MODELS are read from global memory as before (template parameter LDS_MODELS = false).
// Cold per-particle words (phase of a damaged particle, kinematic velocity) sit behind a flag.  `flag ? *p : default` makes the
// compiler wait for the load -- with s_waitcnt vmcnt(0), i.e. for the whole prefetch in flight -- at the first USE of the value
// in the common path too.  rare_load keeps load and wait inside a wave-uniform branch that the common case skips.
SPK_DEV float rare_load(bool cond, const float* p, float dflt) {
    float r = dflt;
    SPK_HOT(!__any(cond));
    if (__any(cond)) {
        float t = dflt;
        if (cond) t = *p;
        asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(t));   // consumed here: the wait stays in this block
    }
    return r;
}
constexpr int MAX_LDS_MODELS = 16;
constexpr int MODEL_WORDS = (int)(sizeof(DevModel) / 4);
static_assert(sizeof(DevModel) % 4 == 0, "");
// The same copy in two halves for a prologue that cannot afford the round trip in the middle: the loads go out first (ahead of the caller's particle
// loads: vector loads return in order), the LDS writes follow where stage_models used to stand.  Words beyond the first 512 (tables of 16 models) take the loop.
struct ModelWords { uint32_t w0, w1; };
SPK_DEV ModelWords stage_models_load(const DevModel* __restrict__ models, int num_models, int tid) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(models);
    const int n = min(num_models, MAX_LDS_MODELS) * MODEL_WORDS;
    ModelWords m = {0u, 0u};
    if (tid < n) m.w0 = src[tid];
    if (tid + BLK_THREADS < n) m.w1 = src[tid + BLK_THREADS];
    return m;
}
SPK_DEV void stage_models_store(DevModel* s_models, const DevModel* __restrict__ models, int num_models, int tid, const ModelWords& m) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(models);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_models);
    const int n = min(num_models, MAX_LDS_MODELS) * MODEL_WORDS;
    if (tid < n) dst[tid] = m.w0;
    if (tid + BLK_THREADS < n) dst[tid + BLK_THREADS] = m.w1;
    for (int w = tid + 2 * BLK_THREADS; w < n; w += BLK_THREADS) dst[w] = src[w];
}
SPK_DEV void stage_models(DevModel* s_models, const DevModel* __restrict__ models, int num_models, int tid) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(models);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_models);
    const int n = min(num_models, MAX_LDS_MODELS) * MODEL_WORDS;
    for (int w = tid; w < n; w += BLK_THREADS) dst[w] = src[w];
}

// ------------------------------------------------------------------------------------------- P2G
// One workgroup per particle block.  LDS tile: 4 SoA planes of 512 floats.
// one reduction step over the four chains c0..c3 (asm operands %0..%3), mask operand M
#define SPK_FMAC_DPP(SH, M)                                                                      \
    "v_fmac_f32_dpp %0, %0, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"    \
    "v_fmac_f32_dpp %1, %1, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"    \
    "v_fmac_f32_dpp %2, %2, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"    \
    "v_fmac_f32_dpp %3, %3, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

#define SPK_FMAC_DPP3(SH, M)                                                                     \
    "v_fmac_f32_dpp %0, %0, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"    \
    "v_fmac_f32_dpp %1, %1, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"    \
    "v_fmac_f32_dpp %2, %2, " M " row_shl:" SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

// What one particle scatters: node s of its 3x3x3 support receives  w_s * (mass ; qv + sx A0 + sy A1 + sz A2).
struct Scatter {
    float fx[3];   // position in units of h relative to the first support node, in [0.5, 1.5]: the B-spline weights are eval_all(fx[a])
    float val_m;
    float qv[3];   // affine * d0 + m v
    M3 A;          // affine * h
    int base;      // tile index of the associated cell (also the cell key of the lane)
    int shell;     // faces of the one-cell drift margin the particle's support reaches: bit 2a = low side of axis a (tile coordinate 0), 2a + 1 = high (7)
};

// particle_to_grid.rs:42-59: stress, affine, weights.  `org` = cell origin of the home block.
// `T` (fused kernel only): the corotated stress the Drucker-Prager return map of this very thread formed from the decomposition
// of F it held (corotated_tau_e) -- no second decomposition.
// `Fe` = F - I, as the particle stores it.  `fx`, `lc`: cell_frac of the particle's block-relative position.
template <class MO>
SPK_DEV Scatter p2g_prepare(const MO& mo, uint32_t flags, const float fx[3], const int lc[3], const float v[3], const M3& C, const M3& Fe, float mass, float vol0, float eh,
                            float phase, float fluid_jp, float h, float inv_d, float dt, const SymStress* T = nullptr) {
    Scatter sc;
    M3 F = Fe;
    F.m[0] += 1.f; F.m[4] += 1.f; F.m[8] += 1.f;
    float d0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        sc.fx[a] = fx[a];
        d0[a] = -fx[a] * h;   // first support node - particle (particle.rs:110-120), world units
    }
    sc.base = lc[0] + TILE_W * lc[1] + TILE_W * TILE_W * lc[2];
    sc.shell = (lc[0] < 1 ? 1 : 0) | (lc[0] > 4 ? 2 : 0) | (lc[1] < 1 ? 4 : 0) | (lc[1] > 4 ? 8 : 0) | (lc[2] < 1 ? 16 : 0) | (lc[2] > 4 ? 32 : 0);
    M3 tau = m3_zero();
    SPK_HOT(!(flags & FLAG_FAILED) && T != nullptr && T->valid && phase == 1.f);
    if (!(flags & FLAG_FAILED)) {
        if (mo.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR) {
            if (T != nullptr && T->valid && phase == 1.f) tau = sym_to_m3(*T);
            else
                tau = phase == 1.f ? corotated_kirchhoff_fe(mo, eh, Fe) : corotated_kirchhoff(mo, phase, eh, F);
        } else if (mo.ctag == SPARKL_CONSTITUTIVE_NEO_HOOKEAN) tau = neo_hookean_kirchhoff(mo, phase, eh, F);
        else if (mo.ctag == SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH) tau = monaghan_kirchhoff(mo, mass, vol0, F.m[0], fluid_jp, C);
    }
    // affine = m C - (V0 inv_d dt) tau   (particle_to_grid.rs:58-59)
    const float ks = vol0 * inv_d * dt;
    M3 aff;
#pragma unroll
    for (int e = 0; e < 9; ++e) aff.m[e] = mass * C.m[e] - ks * tau.m[e];
#pragma unroll
    for (int r = 0; r < 3; ++r) sc.qv[r] = aff.m[r] * d0[0] + aff.m[r + 3] * d0[1] + aff.m[r + 6] * d0[2] + mass * v[r];
#pragma unroll
    for (int e = 0; e < 9; ++e) sc.A.m[e] = aff.m[e] * h;
    sc.val_m = mass;
    return sc;
}
SPK_DEV void scatter_clear(Scatter& sc) {
    sc.val_m = 0.f; sc.qv[0] = sc.qv[1] = sc.qv[2] = 0.f; sc.A = m3_zero(); sc.base = 0; sc.shell = 0;
    sc.fx[0] = sc.fx[1] = sc.fx[2] = 1.f;
}
// the 3 x 3 B-spline weights of a scatter descriptor (QuadraticKernel::eval_all per axis)
struct ScatterWeights { float w[3][3]; };
SPK_DEV ScatterWeights scatter_weights(const Scatter& sc) {
    ScatterWeights W;
#pragma unroll
    for (int a = 0; a < 3; ++a) eval_all(sc.fx[a], W.w[a]);
    return W;
}

// Accumulate one batch of scatters (one per lane; every lane of the wave must call this) into the f64 LDS tile.
//
// Measured on gfx950: ds_add_f32 retires ~1 lane per 2.6 cycles (169 cycles for a full wave), ds_add_f64 is pipelined
// (7.6 cycles per wave instruction) -- so the tile is f64: 20x cheaper atomics *and* one rounding to f32 at write-out.
// REDUCE_STEPS = log2 of the lane-group size over which same-cell particles are pre-summed with DPP before the LDS
// atomics (0: none, 2: groups of 4, 3: of 8, 4: of 16).  Particles are cell-sorted inside a block, so runs of lanes
// share a cell key: segmented suffix sums over runs inside aligned groups of G lanes, all VALU, then one atomic per
// run head.  Cross-lane ops are hoisted out of every `&&` / `||`: they must execute with all lanes enabled.
// Returns the number of run heads of the wave (wave-uniform): the disorder measure of the re-bin policy.
template <int REDUCE_STEPS>
SPK_DEV int p2g_scatter(double (*tile)[ACC_N], const Scatter& sc, bool live, int tid) {
    int nheads = 0;
    if (!__any(live)) return 0;   // wave-uniform: the tail waves of a partly filled batch skip the ~650 instructions below
    const ScatterWeights SW = scatter_weights(sc);
    if (REDUCE_STEPS > 0) {
        constexpr int G = 1 << REDUCE_STEPS;
        const int lane = tid & 63;
        const int r = lane & (G - 1);
        const int cell_key = live ? sc.base : -1;
        const int key_prev = dpp_shr1_i(cell_key, -2), key_next = dpp_shl_i<1>(cell_key);
        const bool head = (r == 0) | (key_prev != cell_key);
        nheads = (int)__popcll(__ballot(live & head));
        // f_o = 1.0 when lanes L .. L+o form one contiguous same-key run inside the group
        const int i1 = ((r + 1 < G) & (key_next == cell_key)) ? 1 : 0;
        const int i2 = (REDUCE_STEPS >= 2) ? (i1 & dpp_shl_i<1>(i1) & (r + 2 < G ? 1 : 0)) : 0;
        const int i4 = (REDUCE_STEPS >= 3) ? (i2 & dpp_shl_i<2>(i2) & (r + 4 < G ? 1 : 0)) : 0;
        const int i8 = (REDUCE_STEPS >= 4) ? (i4 & dpp_shl_i<4>(i4) & (r + 8 < G ? 1 : 0)) : 0;
        const float f1 = (float)i1, f2 = (float)i2, f4 = (float)i4, f8 = (float)i8;
        // Node values as two packed pairs (m, px), (py, pz): v_pk_mul/add_f32 retire two f32 per lane at ~1.15x the cost of
        // one (tools/microbench/pkfma.hip), and this kernel is VALU-bound.
        const f2v X01 = {0.f, sc.A.m[0]}, X23 = {sc.A.m[1], sc.A.m[2]};
#pragma unroll
        for (int sz = 0; sz < 3; ++sz)
#pragma unroll
            for (int sy = 0; sy < 3; ++sy) {
                const float wyz = SW.w[1][sy] * SW.w[2][sz];
                f2v P01 = {sc.val_m, sc.qv[0] + sy * sc.A.m[3] + sz * sc.A.m[6]};
                f2v P23 = {sc.qv[1] + sy * sc.A.m[4] + sz * sc.A.m[7], sc.qv[2] + sy * sc.A.m[5] + sz * sc.A.m[8]};
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    if (sx > 0) { P01 += X01; P23 += X23; }
                    const float w = SW.w[0][sx] * wyz;
                    const f2v c01 = P01 * w, c23 = P23 * w;
                    float c0 = c01.x, c1 = c01.y, c2 = c23.x, c3 = c23.y;
                    // c += f_o * row_shl:o(c), o = 1, 2, 4[, 8]: one v_fmac_f32_dpp each.  hipcc emits v_mov_b32_dpp +
                    // v_fmac (+ s_nop) for the builtin form, so this is hand-written; the four independent chains are
                    // interleaved so that a DPP read of c_k is >= 3 VALU ops after its write (DPP needs 2 wait states),
                    // and the leading s_nop covers the compiler's last write.
                    if (REDUCE_STEPS == 2) {
                        asm volatile("s_nop 1\n" SPK_FMAC_DPP("1", "%4") SPK_FMAC_DPP("2", "%5") : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2));
                    } else if (REDUCE_STEPS == 3) {
                        asm volatile("s_nop 1\n" SPK_FMAC_DPP("1", "%4") SPK_FMAC_DPP("2", "%5") SPK_FMAC_DPP("4", "%6")
                                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2), "v"(f4));
                    } else if (REDUCE_STEPS == 4) {
                        asm volatile("s_nop 1\n" SPK_FMAC_DPP("1", "%4") SPK_FMAC_DPP("2", "%5") SPK_FMAC_DPP("4", "%6") SPK_FMAC_DPP("8", "%7")
                                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2), "v"(f4), "v"(f8));
                    }
                    if (live && head) {
                        const int n = acc_plane_base(sc.base, sz) + sx + TILE_W * sy;
                        atomicAdd(&tile[0][n], (double)c0);
                        atomicAdd(&tile[1][n], (double)c1);
                        atomicAdd(&tile[2][n], (double)c2);
                        atomicAdd(&tile[3][n], (double)c3);
                    }
                }
            }
    } else if (live) {
#pragma unroll
        for (int sz = 0; sz < 3; ++sz)
#pragma unroll
            for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const float w = SW.w[0][sx] * SW.w[1][sy] * SW.w[2][sz];
                    const int n = acc_plane_base(sc.base, sz) + sx + TILE_W * sy;
                    atomicAdd(&tile[0][n], (double)(w * sc.val_m));
                    atomicAdd(&tile[1][n], (double)(w * (sc.qv[0] + sx * sc.A.m[0] + sy * sc.A.m[3] + sz * sc.A.m[6])));
                    atomicAdd(&tile[2][n], (double)(w * (sc.qv[1] + sx * sc.A.m[1] + sy * sc.A.m[4] + sz * sc.A.m[7])));
                    atomicAdd(&tile[3][n], (double)(w * (sc.qv[2] + sx * sc.A.m[2] + sy * sc.A.m[5] + sz * sc.A.m[8])));
                }
    }
    return nheads;
}


// ---- reduce-scatter over a 16-lane row (round 6) -----------------------------------------------------------------------------------------
// When all 16 lanes of a row sit in ONE cell (the fluid configurations: 64 particles per cell, two per lane) the segmented suffix sums above make
// every lane carry every node value through four DPP steps although one lane's result is used.  Here four nodes travel together and the row SPLITS
// the work: lanes 0-7 keep nodes {0, 1} and lanes 8-15 nodes {2, 3} while adding their partner's share (row_ror:8), then quads {0, 2} keep the
// first and quads {1, 3} the second of their two (row_shl:4 / row_shr:4), then the four lanes of a quad add up (two quad_perm steps): quad q ends
// with the row's total of node q.  The destination classes are written with the DPP bank mask (a bank = a quad of the row), so nothing is selected
// or moved: 8 v_add_f32_dpp per value triple of four nodes where the suffix sums take 16, and ONE deposit (conversion to f64 + ds_add_f64) per four
// nodes, issued by four lanes of the row at once, where they take four.  X[k * 3 + c]: value c (of three that travel together) of node k; out: X[c].
// (operand order inside a step: the instruction that READS a register's other half comes before the one that overwrites it; a DPP read sits at least
// two instructions behind the VALU write of its source)
#define SPK_RS_A1(D) "v_add_f32_dpp " D ", " D ", " D " row_ror:8 row_mask:0xf bank_mask:0x3\n"
#define SPK_RS_A2(D, S) "v_add_f32_dpp " D ", " S ", " S " row_ror:8 row_mask:0xf bank_mask:0xc\n"
#define SPK_RS_B1(D) "v_add_f32_dpp " D ", " D ", " D " row_shl:4 row_mask:0xf bank_mask:0x5\n"
#define SPK_RS_B2(D, S) "v_add_f32_dpp " D ", " S ", " S " row_shr:4 row_mask:0xf bank_mask:0xa\n"
#define SPK_RS_C1(D) "v_add_f32_dpp " D ", " D ", " D " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define SPK_RS_C2(D) "v_add_f32_dpp " D ", " D ", " D " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
SPK_DEV void row_reduce_scatter4x3(float (&X)[12]) {
    asm volatile("s_nop 1\n"
                 SPK_RS_A1("%0") SPK_RS_A1("%1") SPK_RS_A1("%2") SPK_RS_A1("%3") SPK_RS_A1("%4") SPK_RS_A1("%5")
                 SPK_RS_A2("%0", "%6") SPK_RS_A2("%1", "%7") SPK_RS_A2("%2", "%8") SPK_RS_A2("%3", "%9") SPK_RS_A2("%4", "%10") SPK_RS_A2("%5", "%11")
                 SPK_RS_B1("%0") SPK_RS_B1("%1") SPK_RS_B1("%2")
                 SPK_RS_B2("%0", "%3") SPK_RS_B2("%1", "%4") SPK_RS_B2("%2", "%5")
                 SPK_RS_C1("%0") SPK_RS_C1("%1") SPK_RS_C1("%2")
                 SPK_RS_C2("%0") SPK_RS_C2("%1") SPK_RS_C2("%2")
                 : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5])
                 : "v"(X[6]), "v"(X[7]), "v"(X[8]), "v"(X[9]), "v"(X[10]), "v"(X[11]));
}
// (sx, sy, sz) of support node idx = sx + 3 sy + 9 sz, idx < 28 (per lane: the node a quad ends up with depends on the quad)
SPK_DEV void support_node_of(int idx, int& sx, int& sy, int& sz) {
    sz = (idx * 57) >> 9;
    const int rem = idx - 9 * sz;
    sy = (rem * 11) >> 5;
    sx = rem - 3 * sy;
}

// Two particles per lane (slots 2j, 2j + 1: neighbours in the cell-sorted order, so usually the same cell).  Their node
// values are added in the lane first, which halves the DPP reduction, the f64 conversions and the LDS atomic
// instructions per particle.  A pair that straddles two cells -- the usual case once cells hold odd counts: the boundary of a cell falls between
// the two slots of a lane -- hands its second particle to the NEXT lane when that lane's first particle sits in the same cell (one more
// v_fmac_f32_dpp row_shr:1 per value, behind a wave-uniform test); only what is left (a cell with a single particle between two others, the
// last lane of a group) deposits on its own.
// WITH_MASS = false (k_vol_p2g): the momentum only -- three DPP chains, f64 conversions and atomics per node instead of four; tile[0] stays 0.
template <int REDUCE_STEPS, bool WITH_MASS = true>
SPK_DEV int p2g_scatter_pair(double (*tile)[ACC_N], const Scatter& sa, const Scatter& sb, bool live_a, bool live_b, int tid) {
    static_assert(REDUCE_STEPS >= 2 && REDUCE_STEPS <= 4, "");
    if (!__any(live_a | live_b)) return 0;
    constexpr int G = 1 << REDUCE_STEPS;
    const int lane = tid & 63;
    const int r = lane & (G - 1);
    const int key = live_a ? sa.base : -1;
    const bool same = live_a & live_b & (sb.base == sa.base);
    const int key_prev = dpp_shr1_i(key, -2), key_next = dpp_shl_i<1>(key);
    const bool join = live_b & !same & (r + 1 < G) & (key_next == sb.base);   // the next lane's first particle sits in b's cell (key_next >= 0: it is live)
    const bool lone_b = live_b & !same & !join;
    const bool head = (r == 0) | (key_prev != key);
    const int nheads = (int)__popcll(__ballot(live_a & head)) + (int)__popcll(__ballot(lone_b));
    if constexpr (REDUCE_STEPS == 4 && !WITH_MASS) {
        // every row of the wave inside one cell, both particles of every lane in it (wave-uniform): the row reduce-scatter (see row_reduce_scatter4x3)
        if (__all(same & ((r == 0) | (key_prev == key)))) {
            const ScatterWeights WA = scatter_weights(sa), WB = scatter_weights(sb);
            const int quad = (lane >> 2) & 3;
            float X[12];
#pragma unroll
            for (int sz = 0; sz < 3; ++sz)
#pragma unroll
                for (int sy = 0; sy < 3; ++sy) {
                    const float wa_yz = WA.w[1][sy] * WA.w[2][sz], wb_yz = WB.w[1][sy] * WB.w[2][sz];
                    float pa[3] = {sa.qv[0] + sy * sa.A.m[3] + sz * sa.A.m[6], sa.qv[1] + sy * sa.A.m[4] + sz * sa.A.m[7], sa.qv[2] + sy * sa.A.m[5] + sz * sa.A.m[8]};
                    float pb[3] = {sb.qv[0] + sy * sb.A.m[3] + sz * sb.A.m[6], sb.qv[1] + sy * sb.A.m[4] + sz * sb.A.m[7], sb.qv[2] + sy * sb.A.m[5] + sz * sb.A.m[8]};
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        if (sx > 0) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) { pa[c] += sa.A.m[c]; pb[c] += sb.A.m[c]; }
                        }
                        const float wa = WA.w[0][sx] * wa_yz, wb = WB.w[0][sx] * wb_yz;
                        const int idx = sx + 3 * sy + 9 * sz, k = idx & 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) X[3 * k + c] = pa[c] * wa + pb[c] * wb;
                        if (k == 3 || idx == 26) {
                            if (idx == 26) { X[9] = 0.f; X[10] = 0.f; X[11] = 0.f; }   // (27 nodes: the last group carries three)
                            row_reduce_scatter4x3(X);
                            const int node = (idx & ~3) + quad;
                            if ((lane & 3) == 0 && node < 27) {
                                int nx, ny, nz;
                                support_node_of(node, nx, ny, nz);
                                const int n = acc_plane_base(sa.base, nz) + nx + TILE_W * ny;
                                atomicAdd(&tile[1][n], (double)X[0]);
                                atomicAdd(&tile[2][n], (double)X[1]);
                                atomicAdd(&tile[3][n], (double)X[2]);
                            }
                        }
                    }
                }
            return nheads;
        }
    }
    const bool any_lone = __any(lone_b), any_join = __any(join);
    const float gj = dpp_shr1_f(join ? 1.f : 0.f);   // 1: the previous lane's second particle joins this lane's run
    const int i1 = ((r + 1 < G) & (key_next == key)) ? 1 : 0;
    const int i2 = i1 & dpp_shl_i<1>(i1) & (r + 2 < G ? 1 : 0);
    const int i4 = (REDUCE_STEPS >= 3) ? (i2 & dpp_shl_i<2>(i2) & (r + 4 < G ? 1 : 0)) : 0;
    const int i8 = (REDUCE_STEPS >= 4) ? (i4 & dpp_shl_i<4>(i4) & (r + 8 < G ? 1 : 0)) : 0;
    const float f1 = (float)i1, f2 = (float)i2, f4 = (float)i4, f8 = (float)i8;
    const float fs = same ? 1.f : 0.f;
    const ScatterWeights WA = scatter_weights(sa), WB = scatter_weights(sb);
    const f2v XA01 = {0.f, sa.A.m[0]}, XA23 = {sa.A.m[1], sa.A.m[2]}, XB01 = {0.f, sb.A.m[0]}, XB23 = {sb.A.m[1], sb.A.m[2]};
#pragma unroll
    for (int sz = 0; sz < 3; ++sz)
#pragma unroll
        for (int sy = 0; sy < 3; ++sy) {
            const float wa_yz = WA.w[1][sy] * WA.w[2][sz], wb_yz = WB.w[1][sy] * WB.w[2][sz];
            f2v PA01 = {sa.val_m, sa.qv[0] + sy * sa.A.m[3] + sz * sa.A.m[6]};
            f2v PA23 = {sa.qv[1] + sy * sa.A.m[4] + sz * sa.A.m[7], sa.qv[2] + sy * sa.A.m[5] + sz * sa.A.m[8]};
            f2v PB01 = {sb.val_m, sb.qv[0] + sy * sb.A.m[3] + sz * sb.A.m[6]};
            f2v PB23 = {sb.qv[1] + sy * sb.A.m[4] + sz * sb.A.m[7], sb.qv[2] + sy * sb.A.m[5] + sz * sb.A.m[8]};
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                if (sx > 0) { PA01 += XA01; PA23 += XA23; PB01 += XB01; PB23 += XB23; }
                const float wa = WA.w[0][sx] * wa_yz, wb = WB.w[0][sx] * wb_yz;
                const f2v b01 = PB01 * wb, b23 = PB23 * wb;
                const f2v c01 = PA01 * wa + b01 * fs, c23 = PA23 * wa + b23 * fs;
                float c0 = c01.x, c1 = c01.y, c2 = c23.x, c3 = c23.y;
                if (any_join) {   // wave-uniform
                    const float b0 = b01.x, b1 = b01.y, b2 = b23.x, b3 = b23.y;
                    if (WITH_MASS) asm volatile("s_nop 1\nv_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(c0) : "v"(b0), "v"(gj));
                    asm volatile("s_nop 1\n"
                                 "v_fmac_f32_dpp %0, %3, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                 "v_fmac_f32_dpp %1, %4, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                 "v_fmac_f32_dpp %2, %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                 : "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b1), "v"(b2), "v"(b3), "v"(gj));
                }
                if (!WITH_MASS) {   // (a dependent v_fmac_f32_dpp three instructions behind its producer: the two wait states DPP needs)
                    if (REDUCE_STEPS == 2) {
                        asm volatile("s_nop 1\n" SPK_FMAC_DPP3("1", "%3") SPK_FMAC_DPP3("2", "%4") : "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2));
                    } else if (REDUCE_STEPS == 3) {
                        asm volatile("s_nop 1\n" SPK_FMAC_DPP3("1", "%3") SPK_FMAC_DPP3("2", "%4") SPK_FMAC_DPP3("4", "%5") : "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2), "v"(f4));
                    } else {
                        asm volatile("s_nop 1\n" SPK_FMAC_DPP3("1", "%3") SPK_FMAC_DPP3("2", "%4") SPK_FMAC_DPP3("4", "%5") SPK_FMAC_DPP3("8", "%6")
                                     : "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2), "v"(f4), "v"(f8));
                    }
                } else if (REDUCE_STEPS == 2) {
                    asm volatile("s_nop 1\n" SPK_FMAC_DPP("1", "%4") SPK_FMAC_DPP("2", "%5") : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2));
                } else if (REDUCE_STEPS == 3) {
                    asm volatile("s_nop 1\n" SPK_FMAC_DPP("1", "%4") SPK_FMAC_DPP("2", "%5") SPK_FMAC_DPP("4", "%6")
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2), "v"(f4));
                } else {
                    asm volatile("s_nop 1\n" SPK_FMAC_DPP("1", "%4") SPK_FMAC_DPP("2", "%5") SPK_FMAC_DPP("4", "%6") SPK_FMAC_DPP("8", "%7")
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(f1), "v"(f2), "v"(f4), "v"(f8));
                }
                if (live_a && head) {
                    const int n = acc_plane_base(sa.base, sz) + sx + TILE_W * sy;
                    if (WITH_MASS) atomicAdd(&tile[0][n], (double)c0);
                    atomicAdd(&tile[1][n], (double)c1);
                    atomicAdd(&tile[2][n], (double)c2);
                    atomicAdd(&tile[3][n], (double)c3);
                }
                if (any_lone) {   // wave-uniform
                    if (lone_b) {
                        const int n = acc_plane_base(sb.base, sz) + sx + TILE_W * sy;
                        if (WITH_MASS) atomicAdd(&tile[0][n], (double)b01.x);
                        atomicAdd(&tile[1][n], (double)b01.y);
                        atomicAdd(&tile[2][n], (double)b23.x);
                        atomicAdd(&tile[3][n], (double)b23.y);
                    }
                }
            }
        }
    return nheads;
}

// ---- run-task scatter -----------------------------------------------------------------------------------------------
// The DPP scheme above makes every lane carry all 27 x 4 node values of its particle through 4 reduction steps (432
// v_fmac_f32_dpp per particle, each ~1.7 plain VALU ops) although only the run heads' results are used.  Here the wave
// turns the work around after the per-particle phase: the scatter descriptors stay in the registers of the lanes that
// computed them, the wave enumerates its runs of same-cell lanes (ballot + a 64-entry run list in LDS), and lane t becomes
// TASK t = (run, z-plane of the 3x3x3 support[, half of the run's particles]).  A task walks its particles, fetching their
// descriptors from the owning lanes with ds_bpermute (16 per particle), accumulates its 9 nodes x (m, px, py, pz) in
// registers with plain (packed) FMAs and deposits them with 36 f64 LDS atomics.  Per 64 particles of the lattice phase
// (8 runs of 8): 60 busy lanes x 4 particles x 9 nodes instead of 64 lanes x 27 nodes x 4 DPP steps; 36 instead of 108
// ds_add_f64 wave-instructions; no f32 -> f64 conversion per particle.  Ragged runs (the flowing phase) only change the
// task list: more runs -> more tasks -> more rounds of 64 tasks.
SPK_DEV float bperm_f(int byte_addr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(x))); }
// maximum over the wave as a scalar: 4 DPP steps inside the 16-lane rows, then the four row results through readlane
// (a __shfl_xor butterfly costs six dependent LDS round trips)
SPK_DEV int wave_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
SPK_DEV int wave_shr1_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false); }

// `s_run`: this wave's 64-entry chunk list in LDS.  Every lane of the wave must call this (cross-lane operations inside).
// Returns the number of live runs of the wave (wave-uniform): the disorder measure of the re-order policy.
//
// Balance.  A task walks its particles one per iteration and the wave iterates as long as its longest task, so the chunk a task
// gets must be short: a run longer than L particles is split into two halves -- two chunks, neighbours in the list, whose sums
// are added with one v_fmac_f32_dpp per value before ONE lane deposits (two lanes adding to the same address serialise in the
// LDS).  L is the smallest of 4, 5, 6, 8, 12 whose chunks still fit the rounds of 64 tasks the runs need anyway (wave-uniform):
// the lattice (8 runs of 8) splits every run (48 tasks, 4 iterations), the flowing column (10-25 ragged runs of 1-12) splits the
// long ones only (5-6 iterations where whole-run tasks took 8-12).  Task t = (plane z, chunk c) = z * C + c.
#define SPK_FMAC_WSHL(R) "v_fmac_f32_dpp " R ", " R ", %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// Grouping.  A task's chunk is a range of RANKS: the wave's live particles in the order (first lane of their cell, lane).  While the
// lanes are still sorted by cell (after a re-bin or a block reorder) rank = lane and a run is a range of lanes.  Once particles have
// crossed cells the members of one cell sit in several separate lane ranges; listed as separate runs they would cost a task triple
// each and deposit to the SAME nodes in the same ds_add_f64 (same-address atomics serialise in the LDS: 0.07 ms of the launch at
// substep 700 of the 10M column, tools/ablate.py).  So an unsorted wave (wave-uniform test) groups its lanes by cell first -- one
// scalar iteration per distinct cell: readlane the key, ballot the matches -- and moves every descriptor to the lane of its rank (seven
// ds_permute_b32; the affine part is simply written to its rank's LDS column), after which the task phase is the same for both.
SPK_DEV float perm_f(int byte_addr, float x) { return __int_as_float(__builtin_amdgcn_ds_permute(byte_addr, __float_as_int(x))); }
SPK_DEV int p2g_scatter_runs(double (*tile)[ACC_N], const Scatter& sc, bool live, int tid, int* s_run, float (*s_A)[64]) {
    if (!__any(live)) return 0;   // wave-uniform
    const int lane = tid & 63;
    const int key = live ? sc.base : 0x7fffffff;               // dead lanes sort behind the live ones and are never listed
    const int prev = wave_shr1_i(key, -2);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool perm = __any(prev > key);                       // wave-uniform
    bool lhead;   // first lane of its cell
    int len;      // particles of the cell (meaningful in head lanes)
    int rank;     // of this lane's particle
    if (!perm) {
        const bool head = (lane == 0) | (prev != key);
        const unsigned long long hb = __ballot(head);
        const unsigned long long above = (lane == 63) ? 0ull : (hb >> (lane + 1));
        len = above ? (int)__ffsll((long long)above) : (64 - lane);  // distance to the next head
        lhead = head & live;
        rank = lane;
    } else {
        const unsigned long long lv = __ballot(live);
        unsigned long long todo = lv, mine = 0ull;
        int off = 0;
        rank = 0;
        while (todo) {   // scalar loop: cells in the order of their first lane
            const int K = __builtin_amdgcn_readlane(key, (int)__ffsll((long long)todo) - 1);
            const unsigned long long m = __ballot(key == K);
            if (key == K) { mine = m; rank = off; }
            off += (int)__popcll(m);
            todo &= ~m;
        }
        lhead = live && (mine & lt) == 0ull;
        len = (int)__popcll(mine);
        rank += (int)__popcll(mine & lt);
        if (!live) rank = off + (int)__popcll(~lv & lt);   // dead lanes take the ranks behind: the move is a permutation
    }
    // the descriptor of rank r: seven values in lane r's registers (fetched by the tasks with ds_bpermute), the affine part in column r
    // of `s_A` (9 KB per workgroup instead of 9 more registers: the task phase is where the kernel's register pressure peaks, and
    // those 9 pay for carrying the whole next batch in flight)
    float d_fx0 = sc.fx[0], d_fx1 = sc.fx[1], d_fx2 = sc.fx[2], d_m = sc.val_m, d_q0 = sc.qv[0], d_q1 = sc.qv[1], d_q2 = sc.qv[2];
    if (perm) {
        const int ra = rank << 2;
        d_fx0 = perm_f(ra, d_fx0); d_fx1 = perm_f(ra, d_fx1); d_fx2 = perm_f(ra, d_fx2); d_m = perm_f(ra, d_m);
        d_q0 = perm_f(ra, d_q0); d_q1 = perm_f(ra, d_q1); d_q2 = perm_f(ra, d_q2);
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) s_A[e][rank] = sc.A.m[e];
    const unsigned long long hl = __ballot(lhead);
    const int nruns = (int)__popcll(hl);
    // split threshold: the smallest L whose chunks fit the task slots of the rounds the runs need anyway
    const int slots = ((3 * nruns + 63) >> 6) << 6;
    unsigned long long sb = 0ull;
    {
        const unsigned long long b4 = __ballot(lhead & (len > 4)), b5 = __ballot(lhead & (len > 5)), b6 = __ballot(lhead & (len > 6)),
                                 b8 = __ballot(lhead & (len > 8)), b12 = __ballot(lhead & (len > 12));
        const int room = slots / 3 - nruns;   // chunks that may be added
        sb = (int)__popcll(b4) <= room ? b4 : (int)__popcll(b5) <= room ? b5 : (int)__popcll(b6) <= room ? b6 : (int)__popcll(b8) <= room ? b8 : (int)__popcll(b12) <= room ? b12 : 0ull;
    }
    const int C = nruns + (int)__popcll(sb);
    if (lhead) {   // entry: first rank | particles << 8 | half flags | tile cell << 18
        const int c = (int)__popcll(hl & lt) + (int)__popcll(sb & lt);
        const bool split = (sb >> lane) & 1ull;
        const int n0 = split ? (len + 1) >> 1 : len;
        s_run[c] = rank | (n0 << 8) | (split ? 0x10000 : 0) | (sc.base << 18);              // bit 16: the next chunk is this run's second half
        if (split) s_run[c + 1] = (rank + n0) | ((len - n0) << 8) | 0x20000 | (sc.base << 18); // bit 17: second half
    }
    __builtin_amdgcn_wave_barrier();   // LDS operations of one wave execute in order: the lists are visible to its lanes
    const int ntasks = 3 * C;
    const bool any_split = sb != 0ull;
    for (int t0 = 0; t0 < ntasks; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < ntasks;
        const int z = (t >= C ? 1 : 0) + (t >= 2 * C ? 1 : 0);
        const int c = valid ? t - z * C : 0;
        const int info = s_run[c];
        const int start = info & 63;
        const int cnt = valid ? (info >> 8) & 127 : 0;
        const int first = start;
        const int run_base = (info >> 18) & 511;
        // trip count of the wave (scalar): the longest task
        const int maxcnt = wave_max_i32(cnt);
        const float fz = (float)z;
        // this task's z-weight is  za + zb (x - zg)^2  (eval_all: 0.5 (1.5 - x)^2 | 0.75 - (x - 1)^2 | 0.5 (x - 0.5)^2)
        const float za = z == 1 ? 0.75f : 0.f, zb = z == 1 ? -1.f : 0.5f, zg = 1.5f - 0.5f * fz;
        f2v acc01[9], acc23[9];
        // one particle of the task: its 9 nodes x (m, px, py, pz).  FIRST: the sums start from this particle's products (no zero
        // fill of the 36 accumulators, no add to zero)
        auto visit = [&](int j, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool on = j < cnt;
            const int here = on ? first + j : start;                            // an idle iteration re-reads a lane of its own run
            const int addr = here << 2;
            // stage 1: the plane's base value (m ; qv + z A2) and z-weight
            const float xz = bperm_f(addr, d_fx2) - zg;
            const float wz = on ? za + zb * (xz * xz) : 0.f;
            const float m = bperm_f(addr, d_m);
            const float q0 = bperm_f(addr, d_q0), q1 = bperm_f(addr, d_q1), q2 = bperm_f(addr, d_q2);
            const float a6 = s_A[6][here], a7 = s_A[7][here], a8 = s_A[8][here];
            f2v P01 = {m, q0 + fz * a6}, P23 = {q1 + fz * a7, q2 + fz * a8};
            // stage 2: the in-plane weights and increments; node (sx, sy) receives  wx wy wz * (m ; qv + sx A0 + sy A1 + z A2)
            float wy[3], wx[3];
            eval_all(bperm_f(addr, d_fx1), wy);
            eval_all(bperm_f(addr, d_fx0), wx);
            const f2v Y01 = {0.f, s_A[3][here]}, Y23 = {s_A[4][here], s_A[5][here]};
            const f2v X01 = {0.f, s_A[0][here]}, X23 = {s_A[1][here], s_A[2][here]};
#pragma unroll
            for (int sy = 0; sy < 3; ++sy) {
                if (sy > 0) { P01 += Y01; P23 += Y23; }
                const float wyz = wy[sy] * wz;
                f2v Q01 = P01, Q23 = P23;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    if (sx > 0) { Q01 += X01; Q23 += X23; }
                    const float w = wx[sx] * wyz;
                    if (FIRST) { acc01[3 * sy + sx] = Q01 * w; acc23[3 * sy + sx] = Q23 * w; }
                    else { acc01[3 * sy + sx] += Q01 * w; acc23[3 * sy + sx] += Q23 * w; }
                }
            }
        };
        visit(0, std::true_type{});
        for (int j = 1; j < maxcnt; ++j) visit(j, std::false_type{});
        // the halves of a split run sit in neighbouring lanes: add the second half's sums into the first, so that one lane
        // deposits per (run, plane).  A pair cut by the end of a round (first half in lane 63) deposits on both sides: different
        // instructions, no same-address conflict.
        bool deposit = cnt > 0;
        if (any_split) {   // wave-uniform
            const bool paired = (info & 0x10000) && valid && lane != 63;
            // (f = 0 times a non-finite neighbour sum would be NaN: sums are finite whenever the particle state is -- a particle whose F or
            // position has blown up is failed / invalid and scatters nothing -- and a select instead costs two more VALU per value, x 36)
            const float f = paired ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float a0 = acc01[k].x, a1 = acc01[k].y, a2 = acc23[k].x, a3 = acc23[k].y;
                asm volatile("s_nop 1\n" SPK_FMAC_WSHL("%0") SPK_FMAC_WSHL("%1") SPK_FMAC_WSHL("%2") SPK_FMAC_WSHL("%3") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
                acc01[k].x = a0; acc01[k].y = a1; acc23[k].x = a2; acc23[k].y = a3;
            }
            deposit = deposit && (!(info & 0x20000) || lane == 0);
        }
        if (deposit) {
            const int p = (run_base >> 6) + z;
            const int n0 = run_base + TILE_W * TILE_W * z + 8 * p - 4 * (p & 1);      // acc_slot of the plane's (0, 0) node
#pragma unroll
            for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const int n = n0 + sx + TILE_W * sy;
                    atomicAdd(&tile[0][n], (double)acc01[3 * sy + sx].x);
                    atomicAdd(&tile[1][n], (double)acc01[3 * sy + sx].y);
                    atomicAdd(&tile[2][n], (double)acc23[3 * sy + sx].x);
                    atomicAdd(&tile[3][n], (double)acc23[3 * sy + sx].y);
                }
        }
    }
    return nruns;
}

// Tile write-out.  Only the box of the 8^3 tile that holds support nodes goes to HBM, compactly: the 6^3 core plus, per axis and side,
// the margin plane a particle's support reaches (`tile_mode`: the OR of Scatter::shell over the block).  The lattice at rest writes
// 216 of 512 nodes (-58 % tile traffic here and in the grid update); the flowing column, whose blocks nearly all hold a particle that
// has sunk into the margin below, mostly 6 x 7 x 6 or 7 x 7 x 6 instead of all 512.
struct TileBox {
    int lx, ly, lz, wx, wy, wz;
};
SPK_DEV TileBox tile_box(int mode) {
    TileBox t;
    t.lx = 1 - (mode & 1); t.ly = 1 - ((mode >> 2) & 1); t.lz = 1 - ((mode >> 4) & 1);
    t.wx = 6 + (mode & 1) + ((mode >> 1) & 1); t.wy = 6 + ((mode >> 2) & 1) + ((mode >> 3) & 1); t.wz = 6 + ((mode >> 4) & 1) + ((mode >> 5) & 1);
    return t;
}
// index of tile coordinate (tx, ty, tz) in [0, 8)^3 inside a tile stored with `mode`, or -1 where nothing was stored (zero)
SPK_DEV int tile_box_index(int mode, int tx, int ty, int tz) {
    const TileBox t = tile_box(mode);
    const int x = tx - t.lx, y = ty - t.ly, z = tz - t.lz;
    if ((unsigned)x >= (unsigned)t.wx || (unsigned)y >= (unsigned)t.wy || (unsigned)z >= (unsigned)t.wz) return -1;
    return x + t.wx * (y + t.wy * z);
}
constexpr int TILE_MODE_FULL = 63;
constexpr int TILE_MODE_MASS = 64;   // a fluid mass pass wrote the tile: all 8^3 nodes, one float each (2 KB instead of 8)
// OR over the wave as a scalar (DPP inside the 16-lane rows, the four row results through readlane); every lane must call it.  The
// margin faces a block uses are kept this way, wave-uniform in one SGPR, rather than per lane: a VGPR more across the particle loop
// of k_g2p2g is a spill.
SPK_DEV int wave_or_i32(int v) {
    v |= __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
    v |= __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false);
    v |= __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xF, 0xF, false);
    v |= __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xF, 0xF, false);
    return __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
}
SPK_DEV void tile_write(double (*tile)[ACC_N], const GridArrays& G, int pb, int tid, int shell_bits /* wave-uniform */) {
    __shared__ int s_mode[BLK_THREADS / 64];
    if ((tid & 63) == 0) s_mode[tid >> 6] = shell_bits;   // wave-uniform (wave_or_i32 by the caller)
    __syncthreads();   // also: every deposit of the workgroup has landed
    int mode = 0;
#pragma unroll
    for (int i = 0; i < BLK_THREADS / 64; ++i) mode |= s_mode[i];
    const TileBox t = tile_box(mode);
    float4* out = G.tiles + (size_t)pb * TILE_N;
    const int n = t.wx * t.wy * t.wz;
    const float iwx = 1.f / (float)t.wx, iwy = 1.f / (float)t.wy;
    for (int l = tid; l < n; l += BLK_THREADS) {
        const int r = (int)(((float)l + 0.5f) * iwx);          // l / wx, exact for l < 2^20
        const int z = (int)(((float)r + 0.5f) * iwy);
        const int x = l - r * t.wx, y = r - z * t.wy;
        const int s = acc_slot((x + t.lx) + TILE_W * (y + t.ly) + TILE_W * TILE_W * (z + t.lz));
        out[l] = make_float4((float)tile[0][s], (float)tile[1][s], (float)tile[2][s], (float)tile[3][s]);
    }
    if (tid == 0) G.tile_mode[pb] = mode;
}
// (mass, momentum) partial of tile `t` at tile coordinate (tx, ty, tz) in [0, 8)^3

// One thread per launch: account for the heads of the previous scatter (index `prev`), arm the counter, maybe abort.
// `parity` is the launch's substep parity: like a margin violation, the abort takes effect from the next substep.
SPK_DEV void disorder_policy(Ctl* ctl, int prev, int parity) {
    const uint32_t hcount = ctl->heads[prev];
    ctl->heads[prev] = 0u;
    if (hcount == 0u) return;
    ctl->last_heads = hcount;
    const uint32_t base = ctl->heads_base;
    if (base == 0u) { ctl->heads_base = hcount; return; }
    if (hcount > base) {
        const float acc = ctl->excess_acc + (float)(hcount - base);
        ctl->excess_acc = acc;
        if (ctl->reorder_limit > 0.f && acc > ctl->reorder_limit) ctl->reorder_request = 1u;
        if (ctl->excess_limit > 0.f && acc > ctl->excess_limit && !ctl->abort_flag[(parity + 1) & 1]) {
            raise_abort(ctl, (parity + 1) & 1);
            ctl->disorder_aborts += 1u;
        }
    }
}

// A scatter's contribution to the disorder books: the launch total (Ctl::heads, read by the next launch's disorder_policy) and the block's own
// record (pb_heads: .x = heads of the block's last scatter, .y = of its first one since the block was last put in order; k_block_reorder skips
// the blocks whose count has not grown).  One thread per workgroup.
SPK_DEV void record_heads(Ctl* ctl, const GridArrays& G, int pb, int slot, int heads) {
    atomicAdd(&ctl->heads[slot], (uint32_t)heads);
    int2 hd = G.pb_heads[pb];
    hd.x = heads;
    if (hd.y == 0) hd.y = max(heads, 1);
    G.pb_heads[pb] = hd;
}

// One workgroup per particle block: scatter its particles into the LDS tile, write the tile once.
template <int REDUCE_STEPS, bool LDS_MODELS = true>
__global__ __launch_bounds__(BLK_THREADS) void k_p2g(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models,
                                                     Ctl* __restrict__ ctl, StepParams S) {
    if (ctl->abort_flag[S.parity]) return;
    __shared__ double tile[4][ACC_N];
    __shared__ int s_heads;
    const int tid = threadIdx.x;
    __shared__ DevModel s_models[LDS_MODELS ? MAX_LDS_MODELS : 1];
    if (LDS_MODELS) stage_models(s_models, models, S.num_models, tid);
    if (tid == 0) s_heads = 0;
    if (blockIdx.x == 0 && tid == 0) disorder_policy(ctl, (S.parity + 1) & 1, S.parity);
    const int pb = launch_pblock(G);
    for (int n = tid; n < ACC_N; n += BLK_THREADS) { tile[0][n] = 0.0; tile[1][n] = 0.0; tile[2][n] = 0.0; tile[3][n] = 0.0; }
    const int4 rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    const float h = S.h;
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    // software pipeline: the 7 x 16 B of the next batch are in flight while this batch computes
    float4 n0, n1, n2, n3, n4, n5, nk;
    {
        const int i0 = pfirst + min(tid, pcount - 1);
        n0 = P.q[0][i0]; n1 = P.q[1][i0]; n2 = P.q[2][i0]; n3 = P.q[3][i0]; n4 = P.q[4][i0]; n5 = P.q[5][i0]; nk = P.k[i0];
    }
    __syncthreads();

    int shell_used = 0;
    int wave_heads = 0;
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const int i = pfirst + (active ? li : 0);
        const float4 q0 = n0, q1 = n1, q2 = n2, q3 = n3, q4 = n4, q5 = n5, kk = nk;
        if (it + 1 < niter) {
            const int inx = pfirst + min(li + BLK_THREADS, pcount - 1);
            n0 = P.q[0][inx]; n1 = P.q[1][inx]; n2 = P.q[2][inx]; n3 = P.q[3][inx]; n4 = P.q[4][inx]; n5 = P.q[5][inx]; nk = P.k[inx];
        }
        const uint32_t flags = __float_as_uint(kk.w);
        const bool live = active && !(flags & FLAG_INVALID);
        Scatter sc;
        scatter_clear(sc);
        if (live) {
            const float u[3] = {q0.x, q0.y, q0.z}, v[3] = {q3.x, q3.y, q3.z};
            float fx[3]; int lc[3];
            cell_frac(u, fx, lc);
            const M3 F = {{q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}};   // F - I as stored
            const M3 C = {{q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w}};
            const DevModel& mo = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
            const float phase = rare_load((flags & FLAG_PHASE_NOT_ONE) != 0, &P.aux[i].x, 1.f);
            const float jp = mo.is_fluid ? P.d[i].z : 1.f;
            sc = p2g_prepare(mo, flags, fx, lc, v, C, F, kk.x, kk.y, kk.z, phase, jp, h, S.inv_d, S.dt);
        }
        shell_used |= wave_or_i32(sc.shell);
        wave_heads += p2g_scatter<REDUCE_STEPS>(tile, sc, live, tid);
    }
    if ((tid & 63) == 0) atomicAdd(&s_heads, wave_heads);
    tile_write(tile, G, pb, tid, shell_used);
    if (tid == 0 && S.count_heads) record_heads(ctl, G, pb, S.parity, s_heads);
}

// k_p2g with two particles per lane (see p2g_scatter_pair).
template <int REDUCE_STEPS, bool LDS_MODELS = true>
__global__ __launch_bounds__(BLK_THREADS) void k_p2g_pair(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models,
                                                          Ctl* __restrict__ ctl, StepParams S) {
    if (ctl->abort_flag[S.parity]) return;
    __shared__ double tile[4][ACC_N];
    __shared__ int s_heads;
    const int tid = threadIdx.x;
    __shared__ DevModel s_models[LDS_MODELS ? MAX_LDS_MODELS : 1];
    if (LDS_MODELS) stage_models(s_models, models, S.num_models, tid);
    if (tid == 0) s_heads = 0;
    if (blockIdx.x == 0 && tid == 0) disorder_policy(ctl, (S.parity + 1) & 1, S.parity);
    const int pb = launch_pblock(G);
    for (int n = tid; n < ACC_N; n += BLK_THREADS) { tile[0][n] = 0.0; tile[1][n] = 0.0; tile[2][n] = 0.0; tile[3][n] = 0.0; }
    const int4 rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    const float h = S.h;
    __syncthreads();
    int shell_used = 0;
    int wave_heads = 0;
    for (int base = 0; base < pcount; base += 2 * BLK_THREADS) {
        Scatter sc[2];
        bool live[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int li = base + 2 * tid + k;
            const bool active = li < pcount;
            const int i = pfirst + (active ? li : 0);
            const float4 q0 = P.q[0][i], q1 = P.q[1][i], q2 = P.q[2][i], q3 = P.q[3][i], q4 = P.q[4][i], q5 = P.q[5][i], kk = P.k[i];
            const uint32_t flags = __float_as_uint(kk.w);
            live[k] = active && !(flags & FLAG_INVALID);
            scatter_clear(sc[k]);
            if (live[k]) {
                const float u[3] = {q0.x, q0.y, q0.z}, v[3] = {q3.x, q3.y, q3.z};
                float fx[3]; int lc[3];
                cell_frac(u, fx, lc);
                const M3 F = {{q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}};   // F - I as stored
                const M3 C = {{q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w}};
                const DevModel& mo = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
                const float phase = rare_load((flags & FLAG_PHASE_NOT_ONE) != 0, &P.aux[i].x, 1.f);
                const float jp = mo.is_fluid ? P.d[i].z : 1.f;
                sc[k] = p2g_prepare(mo, flags, fx, lc, v, C, F, kk.x, kk.y, kk.z, phase, jp, h, S.inv_d, S.dt);
            }
        }
        shell_used |= wave_or_i32(sc[0].shell | sc[1].shell);
        wave_heads += p2g_scatter_pair<REDUCE_STEPS>(tile, sc[0], sc[1], live[0], live[1], tid);
    }
    if ((tid & 63) == 0) atomicAdd(&s_heads, wave_heads);
    tile_write(tile, G, pb, tid, shell_used);
    if (tid == 0 && S.count_heads) record_heads(ctl, G, pb, S.parity, s_heads);
}

// ----------------------------------------------------------------------------------- grid update
// parry3d Cuboid::project_local_point restated (clamp outside, closest face inside, solid = false)
SPK_DEV void quat_rot(const float q[4], const float v[3], float sgn, float out[3]) {
    const float ux = sgn * q[0], uy = sgn * q[1], uz = sgn * q[2], w = q[3];
    const float cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
    const float dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
    out[0] = v[0] + 2.f * (w * cx + dx);
    out[1] = v[1] + 2.f * (w * cy + dy);
    out[2] = v[2] + 2.f * (w * cz + dz);
}
SPK_DEV bool cuboid_project(const DevCuboid& c, const float ptw[3], float projw[3]) {
    float rel[3] = {ptw[0] - c.t[0], ptw[1] - c.t[1], ptw[2] - c.t[2]};
    float pt[3];
    quat_rot(c.q, rel, -1.f, pt);
    float mins_pt[3], pt_maxs[3], shift[3];
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        mins_pt[i] = -c.he[i] - pt[i];
        pt_maxs[i] = pt[i] - c.he[i];
        shift[i] = fmaxf(mins_pt[i], 0.f) - fmaxf(pt_maxs[i], 0.f);
        if (shift[i] != 0.f) inside = false;
    }
    float proj[3];
    if (!inside) {
#pragma unroll
        for (int i = 0; i < 3; ++i) proj[i] = pt[i] + shift[i];
    } else {
        float best = -3.402823466e+38f;
        bool is_mins = false;
        int best_id = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (mins_pt[i] < pt_maxs[i]) {
                if (pt_maxs[i] > best) { best_id = i; is_mins = false; best = pt_maxs[i]; }
            } else if (mins_pt[i] > best) { best_id = i; is_mins = true; best = mins_pt[i]; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) proj[i] = pt[i] + (i == best_id ? (is_mins ? best : -best) : 0.f);
    }
    quat_rot(c.q, proj, 1.f, projw);
    projw[0] += c.t[0]; projw[1] += c.t[1]; projw[2] += c.t[2];
    return inside;
}

// closest point on triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5 -- the algorithm of parry's
// Triangle::project_local_point)
// Returns the feature the closest point lies on: 0 1 2 = vertex a b c, 3 4 5 = edge ab ac bc, 6 = face.
SPK_DEV int closest_on_triangle(const float p[3], const float a[3], const float b[3], const float c[3], float out[3]) {
    float ab[3], ac[3], ap[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const float d1 = ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2], d2 = ac[0] * ap[0] + ac[1] * ap[1] + ac[2] * ap[2];
    if (d1 <= 0.f && d2 <= 0.f) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return 0; }
    float bp[3] = {p[0] - b[0], p[1] - b[1], p[2] - b[2]};
    const float d3 = ab[0] * bp[0] + ab[1] * bp[1] + ab[2] * bp[2], d4 = ac[0] * bp[0] + ac[1] * bp[1] + ac[2] * bp[2];
    if (d3 >= 0.f && d4 <= d3) { out[0] = b[0]; out[1] = b[1]; out[2] = b[2]; return 1; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float v = d1 / (d1 - d3); out[0] = a[0] + v * ab[0]; out[1] = a[1] + v * ab[1]; out[2] = a[2] + v * ab[2]; return 3; }
    float cp[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    const float d5 = ab[0] * cp[0] + ab[1] * cp[1] + ab[2] * cp[2], d6 = ac[0] * cp[0] + ac[1] * cp[1] + ac[2] * cp[2];
    if (d6 >= 0.f && d5 <= d6) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return 2; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float w = d2 / (d2 - d6); out[0] = a[0] + w * ac[0]; out[1] = a[1] + w * ac[1]; out[2] = a[2] + w * ac[2]; return 4; }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        out[0] = b[0] + w * (c[0] - b[0]); out[1] = b[1] + w * (c[1] - b[1]); out[2] = b[2] + w * (c[2] - b[2]);
        return 5;
    }
    const float denom = 1.f / (va + vb + vc), v = vb * denom, w = vc * denom;
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = a[i] + ab[i] * v + ac[i] * w;
    return 6;
}

// Closest point of a triangle mesh within `reach` of a world point (parry TriMesh::project_local_point walks a BVH over all
// triangles; here the grid cell of the point lists every triangle that can be within reach of it).
SPK_DEV bool trimesh_project(const DevTriMesh& tm, const float ptw[3], float reach, float projw[3], bool& inside) {
    float rel[3] = {ptw[0] - tm.t[0], ptw[1] - tm.t[1], ptw[2] - tm.t[2]};
    float lp[3];
    quat_rot(tm.q, rel, -1.f, lp);
    int cell[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (lp[a] < tm.lo[a] || lp[a] > tm.hi[a]) return false;
        cell[a] = min((int)((lp[a] - tm.lo[a]) * tm.inv_cell), tm.dims[a] - 1);
    }
    const int ci = cell[0] + tm.dims[0] * (cell[1] + tm.dims[1] * cell[2]);
    float best = 3.402823466e+38f, bp[3] = {0.f, 0.f, 0.f};
    int best_tri = -1, best_feat = 6;
    for (int k = tm.cell_start[ci]; k < tm.cell_start[ci + 1]; ++k) {
        const int ti = tm.cell_tris[k];
        const float* T = tm.tris + (size_t)ti * 30;
        const float a[3] = {T[0], T[1], T[2]}, b[3] = {T[3], T[4], T[5]}, c[3] = {T[6], T[7], T[8]};
        float cpt[3];
        const int feat = closest_on_triangle(lp, a, b, c, cpt);
        const float d = (cpt[0] - lp[0]) * (cpt[0] - lp[0]) + (cpt[1] - lp[1]) * (cpt[1] - lp[1]) + (cpt[2] - lp[2]) * (cpt[2] - lp[2]);
        if (d < best) { best = d; bp[0] = cpt[0]; bp[1] = cpt[1]; bp[2] = cpt[2]; best_tri = ti; best_feat = feat; }
    }
    if (best_tri < 0 || best > reach * reach) return false;
    inside = false;
    if (tm.oriented) {
        const float* N = tm.tris + (size_t)best_tri * 30 + 9 + 3 * best_feat;
        inside = ((lp[0] - bp[0]) * N[0] + (lp[1] - bp[1]) * N[1] + (lp[2] - bp[2]) * N[2]) <= 0.f;
        if (tm.flip_interior) inside = !inside;
    }
    quat_rot(tm.q, bp, 1.f, projw);
    projw[0] += tm.t[0]; projw[1] += tm.t[1]; projw[2] += tm.t[2];
    return true;
}

// gpu_collider.rs:99-201 (`polyline_project_point`, restated): the closest of the polygon's segments in 3-D, brute force like the reference; the
// projection keeps the point's own local z (the shape is the polygon extruded along z); inside / outside from the segment's planar normal
// (dir.y, -dir.x, 0) -- a point ON the side the normal points away from is inside a counter-clockwise polygon --, at a vertex from the adjacent
// segment's direction (reentrant vertex = inside) unless the two are collinear within 1e-3.  (The reference dots the WORLD point minus the LOCAL
// projection with the normal, which is only right for an identity pose; the local point is used here.)
SPK_DEV bool polyline_project(const DevPolyline& pl, const float ptw[3], float projw[3], bool& inside) {
    if (pl.n < 2) return false;
    float rel[3] = {ptw[0] - pl.t[0], ptw[1] - pl.t[1], ptw[2] - pl.t[2]};
    float lp[3];
    quat_rot(pl.q, rel, -1.f, lp);
    float best = 3.402823466e+38f, bp[3] = {0.f, 0.f, 0.f}, bt = 0.f;
    int bi = 0;
    for (int i = 0; i < pl.n; ++i) {
        const float* A = pl.verts + 3 * i;
        const float* B = pl.verts + 3 * ((i + 1) % pl.n);
        const float ab[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, ap[3] = {lp[0] - A[0], lp[1] - A[1], lp[2] - A[2]};
        const float ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
        float tt = ab2 > 0.f ? (ap[0] * ab[0] + ap[1] * ab[1] + ap[2] * ab[2]) / ab2 : 0.f;
        tt = fminf(fmaxf(tt, 0.f), 1.f);
        const float c[3] = {A[0] + tt * ab[0], A[1] + tt * ab[1], A[2] + tt * ab[2]};
        const float d = sqrtf((c[0] - lp[0]) * (c[0] - lp[0]) + (c[1] - lp[1]) * (c[1] - lp[1]) + (c[2] - lp[2]) * (c[2] - lp[2]));
        if (d < best) { best = d; bp[0] = c[0]; bp[1] = c[1]; bp[2] = c[2]; bt = tt; bi = i; }
    }
    inside = false;
    {
        const float* A = pl.verts + 3 * bi;
        const float* B = pl.verts + 3 * ((bi + 1) % pl.n);
        const float dx = B[0] - A[0], dy = B[1] - A[1];
        const float nl = sqrtf(dx * dx + dy * dy);
        if (nl > 0.f) {   // Segment::planar_normal(2)
            const float n1[2] = {dy / nl, -dx / nl};
            const float side = (lp[0] - bp[0]) * n1[0] + (lp[1] - bp[1]) * n1[1];
            if (bt > 0.f && bt < 1.f) inside = side <= 0.f;                 // SegmentPointLocation::OnEdge
            else {                                                          // OnVertex(0 | 1): the adjacent segment decides
                float d2[3];
                if (bt <= 0.f) {
                    const float* P = pl.verts + 3 * ((bi + pl.n - 1) % pl.n);
                    d2[0] = -(A[0] - P[0]); d2[1] = -(A[1] - P[1]); d2[2] = -(A[2] - P[2]);
                } else {
                    const float* Nn = pl.verts + 3 * ((bi + 2) % pl.n);
                    d2[0] = Nn[0] - B[0]; d2[1] = Nn[1] - B[1]; d2[2] = Nn[2] - B[2];
                }
                const float dot = n1[0] * d2[0] + n1[1] * d2[1];
                const float thr = 1.0e-3f * sqrtf(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
                inside = fabsf(dot) > thr ? dot >= 0.f : side <= 0.f;
            }
        }
    }
    bp[2] = lp[2];
    if (pl.flip_interior) inside = !inside;
    quat_rot(pl.q, bp, 1.f, projw);
    projw[0] += pl.t[0]; projw[1] += pl.t[1]; projw[2] += pl.t[2];
    return true;
}

// Closest point of a heightfield to a world point, searching only the cells within `reach` (world units) of the point
// horizontally: the boundary rule never needs a projection farther than a couple of cells (grid_update.rs:103-127).
// Returns false when no triangle is in reach.  The CPU reference brute-forces every triangle (parry HeightField::
// project_local_point); the results agree whenever the true distance is below `reach`.
SPK_DEV bool heightfield_project(const DevHeightField& hf, const float ptw[3], float reach, float projw[3], bool& inside) {
    float rel[3] = {ptw[0] - hf.t[0], ptw[1] - hf.t[1], ptw[2] - hf.t[2]};
    float lp[3];
    quat_rot(hf.q, rel, -1.f, lp);
    // vertical early-out: no triangle within `reach` of a point that is more than `reach` above the highest sample or below the lowest
    // one (the search below returns false for those as well, after visiting up to 128 triangles).  In a 25 m column this is 99 % of the
    // grid nodes: the grid update with the sand3 ground went 0.124 -> 0.05 ms at 10M particles.
    if (lp[1] > hf.ymax + reach || lp[1] < hf.ymin - reach) return false;
    const float cw = 1.f / (float)(hf.ncols - 1), ch = 1.f / (float)(hf.nrows - 1);
    const float ux = lp[0] / hf.scale[0] + 0.5f, uz = lp[2] / hf.scale[2] + 0.5f;
    const float rx = reach / fabsf(hf.scale[0]), rz = reach / fabsf(hf.scale[2]);
    int j0 = (int)floorf((ux - rx) / cw), j1 = (int)floorf((ux + rx) / cw), i0 = (int)floorf((uz - rz) / ch), i1 = (int)floorf((uz + rz) / ch);
    j0 = max(j0, 0); i0 = max(i0, 0); j1 = min(j1, hf.ncols - 2); i1 = min(i1, hf.nrows - 2);
    // bounded work even for heightfield cells much finer than the MPM grid: an 8 x 8 window centred on the point
    if (j1 - j0 > 7) { j0 = min(max((int)floorf(ux / cw) - 3, 0), max(hf.ncols - 9, 0)); j1 = min(j0 + 7, hf.ncols - 2); }
    if (i1 - i0 > 7) { i0 = min(max((int)floorf(uz / ch) - 3, 0), max(hf.nrows - 9, 0)); i1 = min(i0 + 7, hf.nrows - 2); }
    if (i1 < i0 || j1 < j0) return false;
    // the same vertical test against the height range of the window only (blocks of 4 x 4 cells): under a sinusoidal ground most nodes
    // of the band |y - ground range| < reach are still too far from the piece of ground below them
    if (hf.coarse) {
        float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
        for (int ci = i0 >> 2; ci <= (i1 >> 2); ++ci)
            for (int cj = j0 >> 2; cj <= (j1 >> 2); ++cj) {
                const float2 r = hf.coarse[ci + hf.crows * cj];
                lo = fminf(lo, r.x); hi = fmaxf(hi, r.y);
            }
        if (lp[1] > hi + reach || lp[1] < lo - reach) return false;
    }
    float best = 3.402823466e+38f, bp[3] = {0.f, 0.f, 0.f};
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j) {
            const float z0 = (-0.5f + ch * (float)i) * hf.scale[2], z1 = (-0.5f + ch * (float)(i + 1)) * hf.scale[2];
            const float x0 = (-0.5f + cw * (float)j) * hf.scale[0], x1 = (-0.5f + cw * (float)(j + 1)) * hf.scale[0];
            const float p00[3] = {x0, hf.heights[i + hf.nrows * j] * hf.scale[1], z0};
            const float p10[3] = {x0, hf.heights[i + 1 + hf.nrows * j] * hf.scale[1], z1};
            const float p01[3] = {x1, hf.heights[i + hf.nrows * (j + 1)] * hf.scale[1], z0};
            const float p11[3] = {x1, hf.heights[i + 1 + hf.nrows * (j + 1)] * hf.scale[1], z1};
            float c[3];
            closest_on_triangle(lp, p00, p10, p01, c);
            float d = (c[0] - lp[0]) * (c[0] - lp[0]) + (c[1] - lp[1]) * (c[1] - lp[1]) + (c[2] - lp[2]) * (c[2] - lp[2]);
            if (d < best) { best = d; bp[0] = c[0]; bp[1] = c[1]; bp[2] = c[2]; }
            closest_on_triangle(lp, p10, p11, p01, c);
            d = (c[0] - lp[0]) * (c[0] - lp[0]) + (c[1] - lp[1]) * (c[1] - lp[1]) + (c[2] - lp[2]) * (c[2] - lp[2]);
            if (d < best) { best = d; bp[0] = c[0]; bp[1] = c[1]; bp[2] = c[2]; }
        }
    if (best > reach * reach) return false;
    inside = false;
    if (hf.below_is_solid) inside = (ux >= 0.f) && (ux <= 1.f) && (uz >= 0.f) && (uz <= 1.f) && (hf.flip_interior ? lp[1] >= bp[1] : lp[1] <= bp[1]);
    quat_rot(hf.q, bp, 1.f, projw);
    projw[0] += hf.t[0]; projw[1] += hf.t[1]; projw[2] += hf.t[2];
    return true;
}

// grid_update.rs:84-128 (the CPU path) for one collider: `proj` = projection of the node position, `inside` = proj.is_inside.  Every collider
// acts in turn under SolverParameters::boundary_handling; a set with per-collider options runs gpu_boundary_rule instead.
SPK_DEV void boundary_rule(float v[3], const float cpos[3], const float proj[3], bool inside, float friction, int boundary_handling, float h, float dt) {
    if (boundary_handling == SPARKL_BOUNDARY_NONE) return;
    if (boundary_handling == SPARKL_BOUNDARY_STICK) {
        if (inside) { v[0] = v[1] = v[2] = 0.f; }
        return;
    }
    float dv[3] = {cpos[0] - proj[0], cpos[1] - proj[1], cpos[2] - proj[2]};
    const float dist = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
    if (!(dist > 1.0e-5f)) return;
    float nrm[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist};
    if (inside) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
    const float nv = v[0] * nrm[0] + v[1] * nrm[1] + v[2] * nrm[2];
    if (nv < 0.f) {
        const float dwm = dist - h;
        if (inside || dwm <= 0.f) {
            const float tv[3] = {v[0] - nv * nrm[0], v[1] - nv * nrm[1], v[2] - nv * nrm[2]};
            const float tvn = sqrtf(tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]);
            v[0] = tv[0]; v[1] = tv[1]; v[2] = tv[2];
            if (tvn > 1.0e-10f) {
                const float f = fmaxf(tvn + nv * friction, 0.f);
                v[0] = tv[0] / tvn * f; v[1] = tv[1] / tvn * f; v[2] = tv[2] / tvn * f;
            }
        } else if (-nv * dt > dwm) {
            const float kq = dwm / dt + nv;
            v[0] -= kq * nrm[0]; v[1] -= kq * nrm[1]; v[2] -= kq * nrm[2];
        }
    }
}

// ---- the reference's GPU grid update (src_kernels/cuda/grid_update.rs:54-190): the rule of a collider set in which some collider carries a
// boundary handling of its own (SparklColliderOptions::grid_boundary_handling, an option only that path has).  ONE collider acts on a node,
// the one whose projection of the node is closest, and the collision normal is the normalised central-difference gradient of the signed
// distance to ALL colliders (grid_update.rs:54-95), not the direction of that projection.  project_point_with_max_dist
// (src_kernels/gpu_collider.rs:31-92): cuboids and polylines always answer, heightfields within 2 h; triangle meshes without limit there --
// here within the reach of the mesh's uniform grid (4 h), beyond which no rule but Stick-inside could act anyway.
struct ColliderScan {
    bool any;          // some collider answered
    float sdf;         // min over the colliders of +-distance (grid_update.rs:54-70)
    float dist;        // the closest projection (grid_update.rs:110-128): its distance, direction node - projection, side, and the collider's own rule
    float dir[3];
    bool inside;
    float friction;
    int handling;
};
template <bool HEIGHTFIELDS>
SPK_DEV void collider_scan(const GridArrays& G, const DevCuboid* __restrict__ cuboids, const DevHeightField* __restrict__ heightfields, const DevTriMesh* __restrict__ trimeshes,
                           const StepParams& S, const float q[3], ColliderScan& out) {
    out.any = false; out.sdf = 0.f; out.dist = 3.402823466e+38f; out.dir[0] = out.dir[1] = out.dir[2] = 0.f; out.inside = false; out.friction = 0.f; out.handling = SPARKL_BOUNDARY_NONE;
    bool have_best = false;
    auto take = [&](const float proj[3], bool inside, float friction, int handling) {
        const float d[3] = {q[0] - proj[0], q[1] - proj[1], q[2] - proj[2]};
        const float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float sd = inside ? -dist : dist;
        if (!out.any || sd < out.sdf) out.sdf = sd;
        out.any = true;
        if (!have_best || dist < out.dist) { have_best = true; out.dist = dist; out.dir[0] = d[0]; out.dir[1] = d[1]; out.dir[2] = d[2]; out.inside = inside; out.friction = friction; out.handling = handling; }
    };
    auto rule_of = [&](int own) { return own >= 0 ? own : S.boundary_handling; };
    for (int ci = 0; ci < S.num_cuboids; ++ci) {
        const DevCuboid col = cuboids[ci];
        if (rule_of(col.handling) == SPARKL_BOUNDARY_NONE) continue;
        float proj[3];
        const bool inside = cuboid_project(col, q, proj);
        take(proj, inside, col.friction, rule_of(col.handling));
    }
    if (HEIGHTFIELDS) {
        for (int hi = 0; hi < S.num_heightfields; ++hi) {
            const DevHeightField& hf = heightfields[hi];
            if (rule_of(hf.handling) == SPARKL_BOUNDARY_NONE) continue;
            float proj[3]; bool inside = false;
            if (heightfield_project(hf, q, 2.0f * S.h, proj, inside)) take(proj, inside, hf.friction, rule_of(hf.handling));
        }
        for (int ti = 0; ti < S.num_trimeshes; ++ti) {
            const DevTriMesh& tm = trimeshes[ti];
            if (rule_of(tm.handling) == SPARKL_BOUNDARY_NONE) continue;
            float proj[3]; bool inside = false;
            if (trimesh_project(tm, q, 4.0f * S.h, proj, inside)) take(proj, inside, tm.friction, rule_of(tm.handling));
        }
        for (int pi = 0; pi < G.num_polylines; ++pi) {
            const DevPolyline& pl = G.polylines[pi];
            if (rule_of(pl.handling) == SPARKL_BOUNDARY_NONE) continue;
            float proj[3]; bool inside = false;
            if (polyline_project(pl, q, proj, inside)) take(proj, inside, pl.friction, rule_of(pl.handling));
        }
    }
}
// grid_update.rs:97-190 (update_single_cell) on the velocity (momentum + m g dt) / m of the node at cpos
// What the reference keeps in a grid node between two grid allocations (GpuGridNode::projection_status / projection_scaled_dir / collision_normal,
// src_kernels/cuda/grid_update.rs:110-146): the closest collider's projection of the node and the signed-distance normal.  Colliders are static between two
// uploads and node blocks between two re-bins, so the record is computed once per binning and replayed (GridArrays::node_proj, two float4 per node:
// (dir.xyz, status bits) and (normal.xyz, friction); node_proj_blk[b] says whether block b's records exist) -- the same floats, hence the same velocities.
struct GpuProj {
    float dir[3];      // node - projection
    float nrm[3];      // zero: no usable gradient
    float friction;
    int handling;      // of the closest collider (never NONE when `any`)
    bool any, inside;
};
template <bool HEIGHTFIELDS>
SPK_DEV GpuProj gpu_projection(const float cpos[3], const GridArrays& G, const DevCuboid* __restrict__ cuboids, const DevHeightField* __restrict__ heightfields,
                               const DevTriMesh* __restrict__ trimeshes, const StepParams& S) {
    GpuProj P;
    ColliderScan best;
    collider_scan<HEIGHTFIELDS>(G, cuboids, heightfields, trimeshes, S, cpos, best);
    P.any = best.any; P.inside = best.inside; P.friction = best.friction; P.handling = best.handling;
    P.dir[0] = best.dir[0]; P.dir[1] = best.dir[1]; P.dir[2] = best.dir[2];
    P.nrm[0] = P.nrm[1] = P.nrm[2] = 0.f;
    if (!best.any || (best.handling != SPARKL_BOUNDARY_FRICTION && best.handling != SPARKL_BOUNDARY_FRICTION_Z_UP)) return P;   // (Stick needs no normal)
    // collision normal: central differences of the signed distance at +-0.1 h (grid_update.rs:72-95); a side with no collider in reach gives 0
    float g[3] = {0.f, 0.f, 0.f};
    const float step = S.h * 0.1f;
    for (int a = 0; a < 3; ++a) {
        float qp[3] = {cpos[0], cpos[1], cpos[2]}, qn[3] = {cpos[0], cpos[1], cpos[2]};
        qp[a] += step; qn[a] -= step;
        ColliderScan sp, sn;
        collider_scan<HEIGHTFIELDS>(G, cuboids, heightfields, trimeshes, S, qp, sp);
        collider_scan<HEIGHTFIELDS>(G, cuboids, heightfields, trimeshes, S, qn, sn);
        if (sp.any && sn.any) g[a] = (sp.sdf - sn.sdf) / S.h / 0.1f / 2.f;
    }
    const float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    if (gn > 1.0e-5f) { P.nrm[0] = g[0] / gn; P.nrm[1] = g[1] / gn; P.nrm[2] = g[2] / gn; }
    return P;
}
// grid_update.rs:148-190 on the velocity (momentum + m g dt) / m of the node
SPK_DEV void gpu_boundary_apply(float v[3], const GpuProj& P, const StepParams& S) {
    if (!P.any) return;
    if (P.handling == SPARKL_BOUNDARY_STICK) {
        if (P.inside) { v[0] = v[1] = v[2] = 0.f; }
        return;
    }
    if (P.handling != SPARKL_BOUNDARY_FRICTION && P.handling != SPARKL_BOUNDARY_FRICTION_Z_UP) return;
    if (P.nrm[0] == 0.f && P.nrm[1] == 0.f && P.nrm[2] == 0.f) return;
    if (P.handling == SPARKL_BOUNDARY_FRICTION_Z_UP && !(P.nrm[2] >= 0.f)) return;
    const float dist = sqrtf(P.dir[0] * P.dir[0] + P.dir[1] * P.dir[1] + P.dir[2] * P.dir[2]);
    const float nv = v[0] * P.nrm[0] + v[1] * P.nrm[1] + v[2] * P.nrm[2];
    if (nv < 0.f) {
        const float dwm = dist - S.h;
        if (P.inside || dwm <= 0.f) {
            const float tv[3] = {v[0] - nv * P.nrm[0], v[1] - nv * P.nrm[1], v[2] - nv * P.nrm[2]};
            const float tvn = sqrtf(tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]);
            v[0] = tv[0]; v[1] = tv[1]; v[2] = tv[2];
            if (tvn > 1.0e-10f) {
                const float f = fmaxf(tvn + nv * P.friction, 0.f);
                v[0] = tv[0] / tvn * f; v[1] = tv[1] / tvn * f; v[2] = tv[2] / tvn * f;
            }
        } else if (-nv * S.dt > dwm) {
            const float kq = dwm / S.dt + nv;
            v[0] -= kq * P.nrm[0]; v[1] -= kq * P.nrm[1]; v[2] -= kq * P.nrm[2];
        }
    }
}

// One wave per node block, 4 blocks per workgroup.  MASS_ONLY: fluid volume recompute (fluids_volume.rs).
// HEIGHTFIELDS keeps the triangle search out of the kernel every scene without a heightfield runs.
// GPU_RULE: some collider carries its own boundary handling: the reference's GPU rule for the whole set (gpu_projection / gpu_boundary_apply), the node's
// projection record cached like the reference caches it in the node.
template <bool MASS_ONLY, bool HEIGHTFIELDS = false, bool GPU_RULE = false>
__global__ __launch_bounds__(BLK_THREADS) void k_grid_update(GridArrays G, const DevCuboid* __restrict__ cuboids, const DevHeightField* __restrict__ heightfields, const DevTriMesh* __restrict__ trimeshes,
                                                             Ctl* ctl, StepParams S, DirectWait W = DirectWait{nullptr, 0, 0u, {0, 0}, 0ull, 0}) {
    // The abort flag is read first but only consulted behind the first table loads: the topology arrays stay valid (stale) after an
    // abort, and the dependent-load chain stays at  blk_tnbr -> tile_mode -> tile  (3 hops; it used to be 5).
    int aborted = ctl->abort_flag[S.parity];
    const int nb_total = G.nb_list ? G.nb_count : G.num_blocks;
    const int bi = xcd_remap(blockIdx.x, (nb_total + 3) / 4) * 4 + (threadIdx.x >> 6);
    if (bi >= nb_total) return;
    // (one node block per wave: b is wave-uniform, and as a scalar it turns the block's origin into a scalar load issued HERE, beside the table
    // read, instead of one more dependent round trip behind the tile reads where the collider loop first needs it)
    const int b = __builtin_amdgcn_readfirstlane(G.nb_list ? G.nb_list[bi] : bi);
    const int lane = threadIdx.x & 63;
    const int cx = lane & 3, cy = (lane >> 2) & 3, cz = lane >> 4;
    const int4 org = G.blk_origin[b];
    const int dx0 = cx <= 2 ? -1 : 0, dy0 = cy <= 2 ? -1 : 0, dz0 = cz <= 2 ? -1 : 0;
    // The 27 neighbour tiles of the block and their storage modes are the same for the whole wave (one node block per wave): lanes 0..26
    // load one table entry and one mode each -- two coalesced loads instead of sixteen 4-byte gathers per lane, a third of the kernel's
    // vector-memory instructions -- and every lane picks its eight through the LDS crossbar (ds_bpermute).
    const int my_t = G.blk_tnbr[b * 27 + min(lane, 26)];
    if (W.box && !W.pre_agreed) {
        // peer-direct exchange: the flag of the substep is the MAX over the mail of ALL members (every wave derives it for itself; the first wave of
        // the launch leaves it in the control block for the particle kernel behind), then the neighbours' records must have landed
        bool any = false;
        const bool first = blockIdx.x == 0 && threadIdx.x == 0;
        if (!direct_wait_mail(W, lane, any)) { if (first) direct_give_up(ctl, S.parity, 1u); return; }
        if (any) { if (first) raise_abort(ctl, S.parity); return; }
        aborted = 0;
    }
    // the flag (a scalar load issued first) has arrived long before the table entry: an aborted launch leaves here, 34 -> ~8 us
    // at 10M particles (a batch cut by a re-bin request leaves up to 127 of them behind)
    if (aborted) return;
    const int my_mode = MASS_ONLY ? 0 : G.tile_mode[max(my_t, 0)];
    int tl[8], mode[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int dx = dx0 + (i & 1), dy = dy0 + ((i >> 1) & 1), dz = dz0 + (i >> 2);
        const int src = ((dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)) << 2;
        tl[i] = __builtin_amdgcn_ds_bpermute(src, my_t);
        mode[i] = MASS_ONLY ? 0 : __builtin_amdgcn_ds_bpermute(src, my_mode);
    }
    // The eight tile reads of a node are issued back to back, without a branch around any of them: a read that has nothing to fetch (no such tile, or a
    // node outside the box the tile stored) goes to element 0 of tile 0 and is replaced by zeros -- x + 0 is x, so the sums are those of the guarded
    // loop bit for bit.  With a branch per read the compiler waited for each one inside its branch: eight dependent round trips to memory per wave
    // behind the three of the tables (2.7 TB/s at 10 M particles; profiles/r05_kernel_experiments.txt, 10).
    float m = 0.f, px = 0.f, py = 0.f, pz = 0.f;
    if (MASS_ONLY) {   // every tile was written by a mass pass (TILE_MODE_MASS)
        float tm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int dx = dx0 + (i & 1), dy = dy0 + ((i >> 1) & 1), dz = dz0 + (i >> 2);
            const int tx = cx - 4 * dx + 1, ty = cy - 4 * dy + 1, tz = cz - 4 * dz + 1;
            tm[i] = ((const float*)(G.tiles + (size_t)max(tl[i], 0) * TILE_N))[tx + TILE_W * ty + TILE_W * TILE_W * tz];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) m += tl[i] >= 0 ? tm[i] : 0.f;
    } else {
        float4 tv[8];
        bool ok[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int dx = dx0 + (i & 1), dy = dy0 + ((i >> 1) & 1), dz = dz0 + (i >> 2);
            const int tx = cx - 4 * dx + 1, ty = cy - 4 * dy + 1, tz = cz - 4 * dz + 1;
            const int off = tile_box_index(mode[i], tx, ty, tz);   // -1: outside what the tile stored
            ok[i] = (tl[i] >= 0) & (off >= 0);
            tv[i] = G.tiles[ok[i] ? (size_t)tl[i] * TILE_N + off : (size_t)0];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            m += ok[i] ? tv[i].x : 0.f; px += ok[i] ? tv[i].y : 0.f; py += ok[i] ? tv[i].z : 0.f; pz += ok[i] ? tv[i].w : 0.f;
        }
    }
    if (G.halo_in) {  // partial sums of the neighbour slab(s): a + b is bit-identical on both sides of a cut
        const int bp = (G.halo_axis == 0 ? org.x : (G.halo_axis == 1 ? org.y : org.z)) >> 2;
        if (bp == G.halo_lo || bp == G.halo_hi || bp + 1 == G.halo_lo || bp + 1 == G.halo_hi) {   // wave-uniform: one block per wave
            float4 hv;
            if (W.box) {   // peer-direct: only the waves of the cut planes wait for the neighbours' records, and read them past the caches
                if (!W.pre_agreed && !direct_wait_halo(W, lane)) { if (lane == 0) direct_give_up(ctl, S.parity, 2u); return; }
                hv = halo_load_wt(&G.halo_in[(size_t)b * 64 + lane]);
            } else hv = G.halo_in[(size_t)b * 64 + lane];
            m += hv.x; px += hv.y; py += hv.z; pz += hv.w;
        }
    }
    if (MASS_ONLY) {
        G.node_vel[(size_t)b * 64 + lane] = make_float4(0.f, 0.f, 0.f, m);
        return;
    }
    if (S.mass_from_sweep) m = G.node_vel[(size_t)b * 64 + lane].w;   // wave-uniform; merged by the MASS_ONLY pass of this substep (halo included)
    // velocity incl. gravity (particle_to_grid.rs:88-89)
    const float im = inv_exact(m);
    float v[3] = {(px + m * S.gx * S.dt) * im, (py + m * S.gy * S.dt) * im, (pz + m * S.gz * S.dt) * im};
    // grid_update.rs:60-74
    if (S.dofs & 1u) { px = 0.f; v[0] = 0.f; }
    if (S.dofs & 2u) { py = 0.f; v[1] = 0.f; }
    if (S.dofs & 4u) { pz = 0.f; v[2] = 0.f; }
    if (GPU_RULE) {
        const float cpos[3] = {(float)(org.x + cx) * S.h, (float)(org.y + cy) * S.h, (float)(org.z + cz) * S.h};
        GpuProj P;
        // one status word per node block (wave-uniform: one node block per wave): 2 = its 64 records are there
        int* blk_rec = G.node_proj ? G.node_proj_blk + b : nullptr;
        const int blk_status = blk_rec ? __builtin_amdgcn_readfirstlane(*blk_rec) : 0;
        float4* rec = G.node_proj ? G.node_proj + (size_t)b * 64 + lane : nullptr;
        const size_t rec_stride = (size_t)G.num_blocks * 64;
        if (blk_status == 2) {
            const float4 r0 = rec[0], r1 = rec[rec_stride];
            const uint32_t bits = __float_as_uint(r0.w);
            P.dir[0] = r0.x; P.dir[1] = r0.y; P.dir[2] = r0.z; P.nrm[0] = r1.x; P.nrm[1] = r1.y; P.nrm[2] = r1.z; P.friction = r1.w;
            P.any = (bits & 1u) != 0u; P.inside = (bits & 2u) != 0u; P.handling = (int)(bits >> 2);
        } else {
            P = gpu_projection<HEIGHTFIELDS>(cpos, G, cuboids, heightfields, trimeshes, S);
            if (rec) {
                rec[0] = make_float4(P.dir[0], P.dir[1], P.dir[2], __uint_as_float((P.any ? 1u : 0u) | (P.inside ? 2u : 0u) | ((uint32_t)P.handling << 2)));
                rec[rec_stride] = make_float4(P.nrm[0], P.nrm[1], P.nrm[2], P.friction);
                if (lane == 0) *blk_rec = 2;
            }
        }
        gpu_boundary_apply(v, P, S);
    } else if ((S.num_cuboids > 0 || HEIGHTFIELDS) && S.boundary_handling != SPARKL_BOUNDARY_NONE) {   // (HEIGHTFIELDS: some collider that is not a cuboid)
        const float cpos[3] = {(float)(org.x + cx) * S.h, (float)(org.y + cy) * S.h, (float)(org.z + cz) * S.h};
        for (int ci = 0; ci < S.num_cuboids; ++ci) {
            const DevCuboid col = cuboids[ci];
            float proj[3];
            const bool inside = cuboid_project(col, cpos, proj);
            boundary_rule(v, cpos, proj, inside, col.friction, S.boundary_handling, S.h, S.dt);
        }
        // Heightfields and triangle meshes search triangles: the projection of a node on a static collider is computed once per binning and
        // kept (GridArrays::node_proj); a cached record replays the very floats the search returned, so the velocities are bit for bit those
        // of the uncached kernel.  Colliders of both kinds in the order of the uncached loop.
        const size_t cache_stride = (size_t)G.num_blocks * 64;
        auto cached = [&](int slot, auto&& search, float friction) {
            float proj[3] = {0.f, 0.f, 0.f};
            bool inside = false, hit;
            // one word per node block first (wave-uniform: one node block per wave): most blocks of a scene are nowhere near the collider
            int* blk_rec = G.node_proj ? G.node_proj_blk + (size_t)slot * G.num_blocks + b : nullptr;
            const int blk_status = blk_rec ? __builtin_amdgcn_readfirstlane(*blk_rec) : 0;
            if (blk_status == 1) return;
            float4* rec = G.node_proj ? G.node_proj + (size_t)slot * cache_stride + (size_t)b * 64 + lane : nullptr;
            if (blk_status == 2) {
                const float4 c = *rec;
                const uint32_t status = __float_as_uint(c.w);
                hit = status >= 2u; inside = status == 3u;
                proj[0] = c.x; proj[1] = c.y; proj[2] = c.z;
            } else {
                hit = search(proj, inside);
                if (rec) {
                    const bool any_hit = __any(hit);
                    if (any_hit) *rec = make_float4(proj[0], proj[1], proj[2], __uint_as_float(hit ? (inside ? 3u : 2u) : 1u));
                    if (lane == 0) *blk_rec = any_hit ? 2 : 1;
                }
            }
            if (hit) boundary_rule(v, cpos, proj, inside, friction, S.boundary_handling, S.h, S.dt);
        };
        if (HEIGHTFIELDS) for (int hi = 0; hi < S.num_heightfields; ++hi) {
            const DevHeightField& hf = heightfields[hi];
            cached(hi, [&](float proj[3], bool& inside) { return heightfield_project(hf, cpos, 4.0f * S.h, proj, inside); }, hf.friction);
        }
        if (HEIGHTFIELDS) for (int ti = 0; ti < S.num_trimeshes; ++ti) {
            const DevTriMesh& tm = trimeshes[ti];
            cached(S.num_heightfields + ti, [&](float proj[3], bool& inside) { return trimesh_project(tm, cpos, 4.0f * S.h, proj, inside); }, tm.friction);
        }
        if (HEIGHTFIELDS) for (int pi = 0; pi < G.num_polylines; ++pi) {
            const DevPolyline& pl = G.polylines[pi];
            cached(S.num_heightfields + S.num_trimeshes + pi, [&](float proj[3], bool& inside) { return polyline_project(pl, cpos, proj, inside); }, pl.friction);
        }
    }
    G.node_vel[(size_t)b * 64 + lane] = make_float4(v[0], v[1], v[2], m);
    if (G.node_mom) G.node_mom[(size_t)b * 64 + lane] = make_float4(px, py, pz, 0.f);
}

// ------------------------------------------------------------------------------------------- G2P
// Stage the 8^3 node values (vx, vy, vz, mass) of the tile into LDS from the 27 neighbour blocks.
// The same tile as two arrays, (vx, vy) and vz: the gather then reads a node with ds_read_b64 + ds_read_b32 (2 + 2 LDS cycles per
// wave) instead of the ds_read_b96 (8 cycles) a float4 tile whose w is unused compiles to, and the tile shrinks from 8 to 6 KB.
#ifndef SPK_PRIO_GATHER
#define SPK_PRIO_GATHER 2
#endif
struct VelTile {
    float2 xy[TILE_N];
    float z[TILE_N];
};
// Both nodes of a thread at once, every load issued before the first wait.  `my_nb` (optional): entry min(lane, 26) of the block's neighbour table,
// loaded by the caller BEFORE its particle loads (vector loads return in order: behind them it would arrive with them) and handed out with
// ds_bpermute -- the staging is then ONE dependent round trip per workgroup; with the table read here it is two, and the loop this replaces made
// four (one per node and hop).  A workgroup spends a quarter of its life in this prologue with nothing to issue, which the other three workgroups
// of the CU cover at three waves per SIMD instead of four: -1.2 % of the 10 M launch for the batching alone (profiles/r04_kernel_experiments.txt, 13).
// A missing neighbour reads node 0 of block 0 and is zeroed by a select: no branch around the second load.
SPK_DEV void load_vel_tile(VelTile& vt, const GridArrays& G, int b, int tid, const int* my_nb = nullptr) {
    static_assert(TILE_N == 2 * BLK_THREADS, "");
    int nbv[2], off[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int n = tid + k * BLK_THREADS;
        const int tx = n & 7, ty = (n >> 3) & 7, tz = n >> 6;
        const int dx = tx == 0 ? -1 : (tx <= 4 ? 0 : 1), dy = ty == 0 ? -1 : (ty <= 4 ? 0 : 1), dz = tz == 0 ? -1 : (tz <= 4 ? 0 : 1);
        const int cx = tx - 1 - 4 * dx, cy = ty - 1 - 4 * dy, cz = tz - 1 - 4 * dz;
        const int slot = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
        nbv[k] = my_nb ? __builtin_amdgcn_ds_bpermute(slot << 2, *my_nb) : G.blk_nbr[b * 27 + slot];
        off[k] = cx + 4 * cy + 16 * cz;
    }
    float4 v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = G.node_vel[(size_t)max(nbv[k], 0) * 64 + (nbv[k] >= 0 ? off[k] : 0)];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int n = tid + k * BLK_THREADS;
        const bool ok = nbv[k] >= 0;
        vt.xy[n] = make_float2(ok ? v[k].x : 0.f, ok ? v[k].y : 0.f);
        vt.z[n] = ok ? v[k].z : 0.f;
    }
}
SPK_DEV void load_vel_tile(float4* vt, const GridArrays& G, int b, int tid) {
    for (int n = tid; n < TILE_N; n += BLK_THREADS) {
        const int tx = n & 7, ty = (n >> 3) & 7, tz = n >> 6;
        const int dx = tx == 0 ? -1 : (tx <= 4 ? 0 : 1), dy = ty == 0 ? -1 : (ty <= 4 ? 0 : 1), dz = tz == 0 ? -1 : (tz <= 4 ? 0 : 1);
        const int cx = tx - 1 - 4 * dx, cy = ty - 1 - 4 * dy, cz = tz - 1 - 4 * dz;
        const int nb = G.blk_nbr[b * 27 + (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nb >= 0) v = G.node_vel[(size_t)nb * 64 + cx + 4 * cy + 16 * cz];
        vt[n] = v;
    }
}

// timestep_estimator.rs:21-44: velocity (APIC) bound and constitutive-model bound of one particle
SPK_DEV float velocity_dt_bound(const float v[3], const M3& C, float h) {
    const float d = (h * h) / 4.f;
    const float apic_v = (d * m3_frob(C)) * 6.f * sqrtf(3.f) / h;
    return fminf(3.402823466e+38f, h / (sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + apic_v));
}
template <class MO>
SPK_DEV float particle_dt_bound(const MO& mo, uint32_t flags, const float v[3], const M3& C, float mass, float vol0, float eh, float f00, float h) {
    const float d = (h * h) / 4.f;
    const float apic_v = (d * m3_frob(C)) * 6.f * sqrtf(3.f) / h;
    const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + apic_v;
    float dt = fminf(3.402823466e+38f, h / vn);
    if (!(flags & FLAG_FAILED)) dt = fminf(dt, model_timestep_bound(mo, mass, vol0, eh, mo.is_fluid ? f00 : 1.f, v[0], v[1], v[2], h));
    return dt;
}

// failure_maximum_stress.rs (glue): the failure model looks at the stress of the updated particle.  Out of line: no 3-D
// example uses a failure model.
SPK_DEV bool failure_check(const DevModel& mo, float phase, float eh, const M3& F, float mass, float vol0, float Jp, const M3& C) {
    M3 tau = m3_zero();
    if (mo.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR) tau = corotated_kirchhoff(mo, phase, eh, F);
    else if (mo.ctag == SPARKL_CONSTITUTIVE_NEO_HOOKEAN) tau = neo_hookean_kirchhoff(mo, phase, eh, F);
    else if (mo.ctag == SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH) tau = monaghan_kirchhoff(mo, mass, vol0, F.m[0], Jp, C);
    return maximum_stress_failed(mo, tau);
}

// Minimum of the dt bound over the particles (float bits; dt >= 0, so the integer order is the float order).  No global
// atomics: one atomicMin per wave on a single word cost 1.6 ms at 10M particles (the L2 serialises ~10 ns per operation on
// one line, plain or atomic), and a read-before-atomic filter still cost 1.2 ms inside k_g2p2g, whose run-head counters
// share the line.  Each workgroup stores its minimum; k_dt_reduce (one workgroup) folds them into ctl->dt_bits.
SPK_DEV void dt_block_min_store(uint32_t* dt_partial, float dt_bound, bool dt_nan, Ctl* ctl, int tid) {
    __shared__ uint32_t s_dt[BLK_THREADS / 64];
    if (dt_nan) dt_bound = 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dt_bound = fminf(dt_bound, __shfl_xor(dt_bound, off));
    if ((tid & 63) == 0) s_dt[tid >> 6] = __float_as_uint(dt_bound);
    if (__any(dt_nan) && (tid & 63) == 0) ctl->dt_nan = 1u;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = s_dt[0];
#pragma unroll
        for (int w = 1; w < BLK_THREADS / 64; ++w) m = min(m, s_dt[w]);
        dt_partial[blockIdx.x] = m;
    }
}
// `expect_bits` != 0: the substeps enqueued behind this launch were planned for that bound; any other value turns them
// into no-ops (abort chain from `next_parity`) and the host re-plans.  A re-bin request already raised wins: dt_mismatch
// is only set when this launch is the one that starts the chain.
__global__ __launch_bounds__(1024) void k_dt_reduce(const uint32_t* __restrict__ dt_partial, int n, Ctl* ctl, uint32_t expect_bits = 0u, int next_parity = 0) {
    __shared__ uint32_t s[16];
    uint32_t m = 0x7f7fffffu;   // f32::MAX
    for (int i = threadIdx.x; i < n; i += 1024) m = min(m, dt_partial[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = min(m, s[w]);
        m = min(m, ctl->dead_dt_bits);
        ctl->dt_bits = m;
        if (expect_bits != 0u && (m != expect_bits || ctl->dt_nan) && !ctl->abort_flag[next_parity]) {
            raise_abort(ctl, next_parity);
            ctl->dt_mismatch = 1u;
        }
    }
}

// Adaptive stepping with the fluid volume sweep fused (sparkl_hip_pipeline_step): the scatter of k_vol_p2g was built for a step length
// the host PREDICTED; this runs right behind the estimator pass on the recomputed volumes and checks the prediction against
// min(remaining, max_substep_dt, bound) with the min_dt rule of mls_solver.rs:57-59.  A different value turns the rest of THIS
// substep (grid update, G2P) into no-ops; the host repeats it the long way.
__global__ void k_dt_check(Ctl* ctl, float remaining, float max_substep_dt, float min_dt, uint32_t expect_dt_bits, int parity) {
    const float bound = __uint_as_float(ctl->dt_bits);
    float d = fminf(fminf(remaining, max_substep_dt), bound);
    if (d < min_dt && remaining > min_dt) d = min_dt;
    if ((__float_as_uint(d) != expect_dt_bits || ctl->dt_nan) && !ctl->abort_flag[parity]) {
        raise_abort(ctl, parity);
        ctl->dt_mismatch = 1u;
    }
}

// grid_to_particle.rs:146-163 (enable_boundary_particle_projection): a particle found inside a collider is put back on its
// surface and its velocity corrected by the displacement / dt.  Colliders in the order of the grid update.
// `u`: block-relative position (cells), converted to world coordinates for the projection and back.
SPK_DEV void project_particle_on_colliders(const GridArrays& G, const StepParams& S, const int orgv[3], float u[3], float vel[3]) {
    const float inv_dt = 1.0f / S.dt;
    float px[3] = {world_from_rel(u[0], orgv[0], S.h), world_from_rel(u[1], orgv[1], S.h), world_from_rel(u[2], orgv[2], S.h)};
    bool moved = false;
    for (int ci = 0; ci < S.num_cuboids; ++ci) {
        float proj[3];
        if (cuboid_project(G.cuboids[ci], px, proj)) {
            moved = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) { vel[a] += (proj[a] - px[a]) * inv_dt; px[a] = proj[a]; }
        }
    }
    for (int hi = 0; hi < S.num_heightfields; ++hi) {
        float proj[3];
        bool inside = false;
        if (heightfield_project(G.heightfields[hi], px, 4.0f * S.h, proj, inside) && inside) {
            moved = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) { vel[a] += (proj[a] - px[a]) * inv_dt; px[a] = proj[a]; }
        }
    }
    for (int ti = 0; ti < S.num_trimeshes; ++ti) {
        float proj[3];
        bool inside = false;
        if (trimesh_project(G.trimeshes[ti], px, 4.0f * S.h, proj, inside) && inside) {
            moved = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) { vel[a] += (proj[a] - px[a]) * inv_dt; px[a] = proj[a]; }
        }
    }
    for (int pi = 0; pi < G.num_polylines; ++pi) {
        float proj[3];
        bool inside = false;
        if (polyline_project(G.polylines[pi], px, proj, inside) && inside) {
            moved = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) { vel[a] += (proj[a] - px[a]) * inv_dt; px[a] = proj[a]; }
        }
    }
    if (moved) {
#pragma unroll
        for (int a = 0; a < 3; ++a) u[a] = rel_from_world(px[a], orgv[a], S.h);
    }
}

// SparklColliderOptions::penalty_stiffness (src_kernels/cuda/particle_updater.rs:200-225, compiled out there): a particle found inside a collider
// with a stiffness feels  f = (proj - x) k;  the reference adds f dt to the momentum the particle scatters next (g2p2g.rs:286-287: m v + f dt).
// The stored velocity is what the next scatter multiplies by the mass, and G2P overwrites it from the grid every substep, so the kick
// v += f dt / m on the stored velocity -- behind the advection, which used the gathered velocity -- IS that momentum term.
SPK_DEV void penalty_kick(const GridArrays& G, const StepParams& S, const int orgv[3], const float u[3], float mass, float vel[3]) {
    const float px[3] = {world_from_rel(u[0], orgv[0], S.h), world_from_rel(u[1], orgv[1], S.h), world_from_rel(u[2], orgv[2], S.h)};
    const float reach = 100.0f * S.h, k = S.dt * inv_exact(mass);
    float f[3] = {0.f, 0.f, 0.f};
    for (int ci = 0; ci < S.num_cuboids; ++ci) {
        float proj[3];
        if (G.cuboids[ci].penalty > 0.f && cuboid_project(G.cuboids[ci], px, proj))
            for (int a = 0; a < 3; ++a) f[a] += (proj[a] - px[a]) * G.cuboids[ci].penalty;
    }
    for (int hi = 0; hi < S.num_heightfields; ++hi) {
        float proj[3];
        bool inside = false;
        if (G.heightfields[hi].penalty > 0.f && heightfield_project(G.heightfields[hi], px, reach, proj, inside) && inside)
            for (int a = 0; a < 3; ++a) f[a] += (proj[a] - px[a]) * G.heightfields[hi].penalty;
    }
    for (int ti = 0; ti < S.num_trimeshes; ++ti) {
        float proj[3];
        bool inside = false;
        if (G.trimeshes[ti].penalty > 0.f && trimesh_project(G.trimeshes[ti], px, reach, proj, inside) && inside)
            for (int a = 0; a < 3; ++a) f[a] += (proj[a] - px[a]) * G.trimeshes[ti].penalty;
    }
    for (int pi = 0; pi < G.num_polylines; ++pi) {
        float proj[3];
        bool inside = false;
        if (G.polylines[pi].penalty > 0.f && polyline_project(G.polylines[pi], px, proj, inside) && inside)
            for (int a = 0; a < 3; ++a) f[a] += (proj[a] - px[a]) * G.polylines[pi].penalty;
    }
    vel[0] += f[0] * k; vel[1] += f[1] * k; vel[2] += f[2] * k;
}

// State of one particle after grid_to_particle.rs:62-164 (gather .. psi_pos), before it is written back.
struct Advected {
    float px[3], vel[3]; // px: block-relative, in cells (cell_frac)
    M3 C, Fe;            // Fe = F - I: what the particle stores
    float q, lvg, Jp, psi;
    float eh, phase;     // elastic hardening (snow changes it) and phase (a failure model zeroes it) as P2G must see them
    uint32_t flags;
    bool store_k;        // the K stream changed beyond the flags (elastic hardening)
    bool store_aux;      // phase and / or nacc_alpha changed (cold stream)
    float alpha;
    bool newly_failed;   // flags word must be stored
    bool in_tile;        // the NEW position still scatters inside the home block's tile (and inside the domain)
    float rnew[3];       // rint(new position): taken once, the fused scatter derives its cell_frac from it (cell_frac_r)
    SymStress T;         // corotated stress of the final F, formed by the Drucker-Prager return map (fused kernel: scattered next substep)
};

// APIC gather of one particle from the staged velocity tile: v = sum w v_n, C = sum (w inv_d) v_n dpt^T
// (grid_to_particle.rs:47-60).  `u` is the block-relative position the weights are evaluated at (the particle BEFORE advection).
// CLAMP (k_regather_c only): `u` was reconstructed and may round into the cell next to the one the tile covers; the weights
// are continuous across the cell face, so the covered cell is used.
template <bool CLAMP = false>
SPK_DEV void apic_gather(const VelTile& vt, const float u[3], const StepParams& S, float vel[3], M3& C) {
    const float h = S.h;
    float fx[3], wgt[3][3];
    int lc[3];
    cell_frac(u, fx, lc);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (CLAMP) { const int c = min(max(lc[a], 0), 5); fx[a] += (float)(lc[a] - c); lc[a] = c; }
        eval_all(fx[a], wgt[a]);
    }
    const int base = lc[0] + TILE_W * lc[1] + TILE_W * TILE_W * lc[2];
    // APIC gather in separable form (grid_to_particle.rs:47-60):
    //   v = sum w v_n ;  C = inv_d * sum w v_n (d0 + s h)^T = inv_d * (v d0^T + h [Vx Vy Vz]),  V_a = sum w s_a v_n
    //   velocity_gradient_det = sum w (v_n . dpt) inv_d = trace(C)
    // The (x, y) components travel as packed pairs -- the velocity tile holds them as float2, so a ds_read_b64 lands in an aligned
    // register pair and every step of the separable sum is one v_pk_fma_f32 for both -- the z component as scalars beside them:
    // 2 instead of 3 VALU operations per 3-vector step (the kernel is VALU-issue bound).
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64p;   // volatile: keeps the reads apart (see below)
    const lds_u64p vq = (lds_u64p)(vt.xy);
    f2v V0p = {0.f, 0.f}, Vxp = {0.f, 0.f}, Vyp = {0.f, 0.f}, Vzp = {0.f, 0.f};
    float V0z = 0.f, Vxz = 0.f, Vyz = 0.f, Vzz = 0.f;
    const float wx0 = wgt[0][0], wx1 = wgt[0][1], wx2 = wgt[0][2], wx2d = 2.f * wgt[0][2];
#pragma unroll 1
    for (int sz = 0; sz < 3; ++sz) {
        const float wz = sz == 0 ? wgt[2][0] : (sz == 1 ? wgt[2][1] : wgt[2][2]);  // no dynamic register indexing
        f2v Rzp = {0.f, 0.f}, Rzxp = {0.f, 0.f}, Rzyp = {0.f, 0.f};
        float Rzz = 0.f, Rzxz = 0.f, Rzyz = 0.f;
#pragma unroll
        for (int sy = 0; sy < 3; ++sy) {
            const int row = base + TILE_W * sy + TILE_W * TILE_W * sz;
            // three ds_read_b64 (2 LDS cycles each, 64 banks); left to itself the compiler fuses neighbours into ds_read2_b64, which the
            // LDS serves as two 4 x 16-lane accesses on 32 banks: 8 cycles for what two plain reads do in 4 (MI355X_MICROARCH, LDS table)
            const unsigned long long u0 = vq[row], u1 = vq[row + 1], u2 = vq[row + 2];
            const f2v c0 = __builtin_bit_cast(f2v, u0), c1 = __builtin_bit_cast(f2v, u1), c2 = __builtin_bit_cast(f2v, u2);
            const float z0 = vt.z[row], z1 = vt.z[row + 1], z2 = vt.z[row + 2];
            const float wy = wgt[1][sy];
            const f2v A1 = c1 * wx1;
            const float a1z = wx1 * z1;
            const f2v R = c0 * wx0 + A1 + c2 * wx2, Rx = A1 + c2 * wx2d;
            const float Rz_ = wx0 * z0 + a1z + wx2 * z2, Rxz = a1z + wx2d * z2;
            Rzp += R * wy; Rzz += wy * Rz_;
            Rzxp += Rx * wy; Rzxz += wy * Rxz;
            if (sy > 0) { const float wys = wy * (float)sy; Rzyp += R * wys; Rzyz += wys * Rz_; }
        }
        const float wzs = wz * (float)sz;
        V0p += Rzp * wz; V0z += wz * Rzz;
        Vxp += Rzxp * wz; Vxz += wz * Rzxz;
        Vyp += Rzyp * wz; Vyz += wz * Rzyz;
        Vzp += Rzp * wzs; Vzz += wzs * Rzz;
    }
    const float V0[3] = {V0p.x, V0p.y, V0z}, Vx[3] = {Vxp.x, Vxp.y, Vxz}, Vy[3] = {Vyp.x, Vyp.y, Vyz}, Vz[3] = {Vzp.x, Vzp.y, Vzz};
    vel[0] = V0[0]; vel[1] = V0[1]; vel[2] = V0[2];
    // with d0 = -fx h (first support node - particle):  inv_d (v d0^T + h [Vx Vy Vz])  =  (h inv_d) ([Vx Vy Vz] - v fx^T)
    const float hid = h * S.inv_d;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        C.m[e] = hid * (Vx[e] - fx[0] * V0[e]);
        C.m[3 + e] = hid * (Vy[e] - fx[1] * V0[e]);
        C.m[6 + e] = hid * (Vz[e] - fx[2] * V0[e]);
    }
}

#ifndef SPK_RARE_MIN_BLOCKS_G2P
#define SPK_RARE_MIN_BLOCKS_G2P 3
#endif
// RARE: the instantiation that also carries snow / Rankine / NACC plasticity and the failure model; launched only when a
// model of the set needs it, so the elastic / Drucker-Prager kernels keep their register budget (inlining that code,
// or calling it out of line, cost the hot kernel 35 %).
// (as an int: 0 = none of it, 1 = all of it, 2 = the plastic models only -- no failure test, no particle projection, no penalty kick: the snow view)
template <int RARE, bool WANT_STRESS = false, class MO = DevModel>
SPK_DEV Advected g2p_update(const GridArrays& G, const VelTile& vt, const MO& mo, uint32_t flags, const float4& q0, const float4& q1, const float4& q2, const float4& dd,
                            const float4& kk, const float4* kinvel_i, const float4* aux_i, float phase, const int4& org, const TileRange tr_border, const StepParams& S,
                            unsigned* num_invalid, ParkSlot park = ParkSlot()) {
    Advected o;
    const float h = S.h, dt = S.dt;
    float px[3] = {q0.x, q0.y, q0.z};   // block-relative, in cells
    const int orgv[3] = {org.x, org.y, org.z};
    float vel[3];
    M3 C;
    SPK_MARK("gather");
    // the gather is 54 LDS reads with short arithmetic between them: at a raised issue priority its reads go out ahead of the other waves' VALU
    // work instead of queueing behind it (-1.1 % of the launch; the same hint around the scatter's visits or the epilogue does nothing, around
    // the prologue it doubles the launch time: profiles/r04_kernel_experiments.txt, experiment 12)
    __builtin_amdgcn_s_setprio(SPK_PRIO_GATHER);
    apic_gather(vt, px, S, vel, C);
    __builtin_amdgcn_s_setprio(0);
    SPK_MARK("advect_F");
    const float div = C.m[0] + C.m[4] + C.m[8];
    // kinematic override, advection (grid_to_particle.rs:82-86)
    {
        const bool kin = (flags & FLAG_KINEMATIC) != 0;
        const float* kp = reinterpret_cast<const float*>(kinvel_i);
        vel[0] = rare_load(kin, kp, vel[0]); vel[1] = rare_load(kin, kp + 1, vel[1]); vel[2] = rare_load(kin, kp + 2, vel[2]);
    }
    px[0] += vel[0] * S.dt_h; px[1] += vel[1] * S.dt_h; px[2] += vel[2] * S.dt_h;   // x += v dt, in cells
    // deformation gradient update (:91-100)
    // the particle stores E = F - I; Drucker-Prager particles carry E through the update and the return map (device_math.h,
    // "strain-form arithmetic"), everything else works on F and hands F - I back
    const bool dp_e = !mo.is_fluid && mo.ptag == SPARKL_PLASTIC_DRUCKER_PRAGER;
    M3 E = {{q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}};
    M3 F = E;
    F.m[0] += 1.f; F.m[4] += 1.f; F.m[8] += 1.f;
    if (!mo.is_fluid) {
        M3 Cdt;
#pragma unroll
        for (int e = 0; e < 9; ++e) Cdt.m[e] = C.m[e] * dt;
        const M3 dF = m3_mul(Cdt, F);
#pragma unroll
        for (int e = 0; e < 9; ++e) { F.m[e] += dF.m[e]; E.m[e] += dF.m[e]; }
    } else {
        const float dj = (div * dt) * F.m[0];
        F.m[0] += dj; E.m[0] += dj;
    }
    float q = dd.x, lvg = dd.y, Jp = dd.z, psi = dd.w;
    float eh = kk.z;
    float sv[3] = {1.f, 1.f, 1.f};
    bool sv_valid = false;
    o.store_k = false; o.store_aux = false; o.alpha = 0.f;
    o.T.valid = false; o.T.jm1 = 0.f;
    if constexpr (RARE != 0) if (mo.ptag > SPARKL_PLASTIC_DRUCKER_PRAGER && rare_plastic_may_act(mo, E, q)) {   // snow / Rankine / NACC (plastic_model.update_particle, :102-104), behind the elastic predictor test
        float alpha = mo.ptag == SPARKL_PLASTIC_NACC ? aux_i->z : 0.f;
        const float eh0 = eh, alpha0 = alpha;
        rare_plastic_update(mo, F, Jp, q, eh, alpha);
        E = F;
        E.m[0] -= 1.f; E.m[4] -= 1.f; E.m[8] -= 1.f;
        o.store_k = eh != eh0;
        o.store_aux = alpha != alpha0;
        o.alpha = alpha;
    }
    if (dp_e) {
        SPK_MARK("dp_update");
        dp_update_e<WANT_STRESS>(mo, phase, eh, E, Jp, q, lvg, sv, sv_valid, o.T, park);   // sv: sigma' - 1; the stress goes to `park` (device_math.h, lds_fp)
        SPK_MARK("post_dp");
        F = E;
        F.m[0] += 1.f; F.m[4] += 1.f; F.m[8] += 1.f;
    }
    if (flags & FLAG_STATIC) { vel[0] = vel[1] = vel[2] = 0.f; C = m3_zero(); }
    // fail guards (:111-121); density_def_grad() == 0  <=>  mass/volume0 == 0 or det(F) overflowed
    // det(F) - 1 without cancellation, once: from the return map when it formed the stress of this very F, else by minors
    float jm1 = o.T.jm1;
    if (!o.T.valid) jm1 = det_minus_one(E);
    const bool zero_density = (kk.x == 0.f) || !(fabsf(jm1) < 1.0e30f);
    bool newly_failed = false;
    SPK_HOT(!(zero_density || (flags & FLAG_FAILED) || (!mo.is_fluid && fabsf(F.m[0]) > 1.0e4f)) && !(flags & FLAG_STATIC));
    if (zero_density || (flags & FLAG_FAILED) || (!mo.is_fluid && fabsf(F.m[0]) > 1.0e4f)) {
        newly_failed = !(flags & FLAG_FAILED);
        flags |= FLAG_FAILED;
        F = m3_identity();
        E = m3_zero();
        C = m3_zero();
        sv_valid = false;
        o.T.valid = false;
        jm1 = 0.f;
    }
    // positive energy bookkeeping (:127-132)
    SPK_MARK("pos_energy");
    psi = fmaxf(psi, model_pos_energy(mo, phase, eh, F, E, sv, sv_valid, jm1));
    SPK_MARK("tile_checks");
    // failure model (:134-144): stress of the updated particle; a failed particle keeps phase = 0 from now on
    if constexpr (RARE == 1) if (mo.has_failure && phase != 0.f && failure_check(mo, phase, eh, F, kk.x, kk.y, Jp, C)) {
        phase = 0.f;
        if (!o.store_aux) o.alpha = aux_i->z;
        o.store_aux = true;
        if (!(flags & FLAG_PHASE_NOT_ONE)) { flags |= FLAG_PHASE_NOT_ONE; newly_failed = true; }   // the flags word must be stored
    }
    if (RARE == 1 && S.project_particles) project_particle_on_colliders(G, S, orgv, px, vel);
    if (RARE == 1 && S.penalty) penalty_kick(G, S, orgv, px, kk.x, vel);
    // does the particle still sit inside its home block's tile, and inside the domain?  One test against the block's box (TileRange: the
    // tile, cut by the domain where the block touches its border); a NaN coordinate fails it like the reference's range test does.
    o.rnew[0] = rintf(px[0]); o.rnew[1] = rintf(px[1]); o.rnew[2] = rintf(px[2]);
    // rint(u) in [0, 5] on every axis, as |r - 2.5| <= 2.5: two VALU per axis, false for NaN, and no integer cell kept alive across the store
    bool in_tile = (fabsf(o.rnew[0] - 2.5f) <= 2.5f) && (fabsf(o.rnew[1] - 2.5f) <= 2.5f) && (fabsf(o.rnew[2] - 2.5f) <= 2.5f);
    bool invalid = false;
    SPK_HOT(in_tile && !tr_border);
    if (!in_tile || tr_border) {   // cold: out of the tile (re-bin) or out of the domain?  0 <= assoc, assoc + 5 < dom  (sp_grid.rs / particle_set.rs:129-136)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // round(x / h) in integers (nothing for the compiler to hoist out of the particle loop into registers that only this cold side reads)
            const int rw = orgv[a] + (int)fminf(fmaxf(o.rnew[a], -1.0e6f), 1.0e6f);
            if (!(o.rnew[a] == o.rnew[a]) || rw < 1 || rw >= S.dom - 4) invalid = true;
        }
    }
    if (invalid) {  // what the next ParticleSet::sort would do (particle_set.rs:129-136)
        newly_failed = true;
        flags |= FLAG_FAILED | FLAG_INVALID;
        in_tile = false;
        atomicAdd(num_invalid, 1u);
        // num_invalid sits in Ctl: the dead particles' dt bound is its neighbour (see Ctl::dead_dt_bits)
        atomicMin(num_invalid + (offsetof(Ctl, dead_dt_bits) - offsetof(Ctl, num_invalid)) / 4, __float_as_uint(velocity_dt_bound(vel, C, h)));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { o.px[a] = px[a]; o.vel[a] = vel[a]; }
    o.C = C; o.Fe = E; o.q = q; o.lvg = lvg; o.Jp = Jp; o.psi = psi; o.flags = flags; o.newly_failed = newly_failed; o.in_tile = in_tile;
    o.eh = eh; o.phase = phase;
    return o;
}
// `defer_c`: q[3], q[4], q[5] (v and C) are not written; k_regather_c recomputes them if anything but the next fused launch needs
// them (see StepParams::defer_c)
SPK_DEV void g2p_store(const ParticleArrays& P, int i, const Advected& o, int defer_c = 0) {
    P.q[0][i] = make_float4(o.px[0], o.px[1], o.px[2], o.Fe.m[0]);
    P.q[1][i] = make_float4(o.Fe.m[1], o.Fe.m[2], o.Fe.m[3], o.Fe.m[4]);
    P.q[2][i] = make_float4(o.Fe.m[5], o.Fe.m[6], o.Fe.m[7], o.Fe.m[8]);
    // a particle that just left the domain is never touched again (k_regather_c included): its last v and C are stored now
    if (!defer_c || (o.flags & FLAG_INVALID)) {
        P.q[3][i] = make_float4(o.vel[0], o.vel[1], o.vel[2], o.C.m[0]);
        P.q[4][i] = make_float4(o.C.m[1], o.C.m[2], o.C.m[3], o.C.m[4]);
        P.q[5][i] = make_float4(o.C.m[5], o.C.m[6], o.C.m[7], o.C.m[8]);
    }
    P.d[i] = make_float4(o.q, o.lvg, o.Jp, o.psi);
    if (o.newly_failed) { reinterpret_cast<uint32_t*>(&P.k[i])[3] = o.flags; if (P.fl) P.fl[i] = o.flags; }
    if (o.store_k) reinterpret_cast<float*>(&P.k[i])[2] = o.eh;
    if (o.store_aux) { float* a = reinterpret_cast<float*>(&P.aux[i]); a[0] = o.phase; a[2] = o.alpha; }
}

// The same store through block-based stream pointers (wave-uniform: SGPR pairs) and ONE 32-bit byte offset per lane shared by every 16-byte
// stream (global_store ... v_off, s[base]): no 64-bit address arithmetic per access.  `off` = 16 x the particle's index in its block.
template <class T>
SPK_DEV T* at_off(T* base, unsigned off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off); }
SPK_DEV void g2p_store_at(const ParticleArrays& Pb, unsigned off, const Advected& o, int defer_c = 0) {
    *at_off(Pb.q[0], off) = make_float4(o.px[0], o.px[1], o.px[2], o.Fe.m[0]);
    *at_off(Pb.q[1], off) = make_float4(o.Fe.m[1], o.Fe.m[2], o.Fe.m[3], o.Fe.m[4]);
    *at_off(Pb.q[2], off) = make_float4(o.Fe.m[5], o.Fe.m[6], o.Fe.m[7], o.Fe.m[8]);
    if (!defer_c || (o.flags & FLAG_INVALID)) {
        *at_off(Pb.q[3], off) = make_float4(o.vel[0], o.vel[1], o.vel[2], o.C.m[0]);
        *at_off(Pb.q[4], off) = make_float4(o.C.m[1], o.C.m[2], o.C.m[3], o.C.m[4]);
        *at_off(Pb.q[5], off) = make_float4(o.C.m[5], o.C.m[6], o.C.m[7], o.C.m[8]);
    }
    *at_off(Pb.d, off) = make_float4(o.q, o.lvg, o.Jp, o.psi);
    if (o.newly_failed) { reinterpret_cast<uint32_t*>(at_off(Pb.k, off))[3] = o.flags; if (Pb.fl) *at_off(Pb.fl, off >> 2) = o.flags; }
    if (o.store_k) reinterpret_cast<float*>(at_off(Pb.k, off))[2] = o.eh;
    if (o.store_aux) { float* a = reinterpret_cast<float*>(at_off(Pb.aux, off)); a[0] = o.phase; a[2] = o.alpha; }
}

template <bool RARE, bool LDS_MODELS = true>
__global__ __launch_bounds__(BLK_THREADS, (RARE ? SPK_RARE_MIN_BLOCKS_G2P : 4)) void k_g2p(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models, Ctl* ctl, StepParams S) {
    const int tid = threadIdx.x;
    __shared__ DevModel s_models[LDS_MODELS ? MAX_LDS_MODELS : 1];
    if (ctl->abort_flag[S.parity]) {
        // sticky abort: keep every later substep a no-op until the host re-bins
        if (blockIdx.x == 0 && tid == 0) ctl->abort_flag[(S.parity + 1) & 1] = 1u;
        return;
    }
    __shared__ VelTile vt;
    const int pb = launch_pblock(G);
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    // first batch of particle data goes in flight before the tile is staged
    int li = tid;
    bool have = li < pcount;
    float4 n0, n1, n2, nd, nk;
    {
        const int i0 = pfirst + min(li, pcount - 1);
        n0 = P.q[0][i0]; n1 = P.q[1][i0]; n2 = P.q[2][i0]; nd = P.d[i0]; nk = P.k[i0];
    }
    load_vel_tile(vt, G, b, tid);
    if (LDS_MODELS) stage_models(s_models, models, S.num_models, tid);
    if (G.primary && blockIdx.x == 0 && tid == 0) note_step(ctl);
    __syncthreads();
    const TileRange tr = tile_range(org, S.dom);
    bool want_sort = false;
    float dt_bound = 3.402823466e+38f;
    bool dt_nan = false;
    while (have) {
        const int i = pfirst + li;
        const float4 q0 = n0, q1 = n1, q2 = n2, dd = nd, kk = nk;
        li += BLK_THREADS;
        have = li < pcount;
        if (have) { const int inx = pfirst + li; n0 = P.q[0][inx]; n1 = P.q[1][inx]; n2 = P.q[2][inx]; nd = P.d[inx]; nk = P.k[inx]; }
        const uint32_t flags = __float_as_uint(kk.w);
        if (flags & FLAG_INVALID) continue;
        const DevModel& mo = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
        const float phase = rare_load((flags & FLAG_PHASE_NOT_ONE) != 0, &P.aux[i].x, 1.f);
        const Advected o = g2p_update<(RARE ? 1 : 0)>(G, vt, mo, flags, q0, q1, q2, dd, kk, P.kinvel + i, P.aux + i, phase, org, tr, S, &ctl->num_invalid);
        g2p_store(P, i, o);
        want_sort |= !o.in_tile && !(o.flags & FLAG_INVALID);   // a particle that left the DOMAIN is dead (skipped from now on): no re-bin for it
        if (S.want_dt) {  // the bound the next substep's estimator would compute from this state
            const float b = particle_dt_bound(mo, o.flags, o.vel, o.C, kk.x, kk.y, o.eh, 1.f + o.Fe.m[0], S.h);
            dt_nan |= b != b;
            dt_bound = fminf(dt_bound, b);
        }
    }
    if (__any(want_sort)) {
        if ((tid & 63) == 0) raise_abort(ctl, (S.parity + 1) & 1);
    }
    if (S.want_dt) {
        dt_block_min_store(G.dt_partial, dt_bound, dt_nan, ctl, tid);
    }
}

// Fused G2P(substep s) -> P2G(substep s+1) for fixed-dt runs: the advected particle is scattered straight from
// registers, so v / C / F are not re-read (192 instead of 304 B per particle-substep; SURVEY 8(d) "fused schedule").
// `S` carries substep s (parity, dt); the scatter uses the same dt (fixed) and writes the tiles of substep s+1.
// The G2P half is always valid.  The P2G half is only valid if no particle left its tile margin; otherwise the
// re-bin flag is raised, every later kernel is a no-op and the host re-bins and restarts with a standalone k_p2g.
#ifndef SPK_G2P2G_MIN_BLOCKS
#define SPK_G2P2G_MIN_BLOCKS 4
#endif

// REDUCE_STEPS >= 0: DPP pre-reduction over groups of 2^REDUCE_STEPS lanes (p2g_scatter); -1: run tasks (p2g_scatter_runs).
// MSET: 0 = any model table; 1 = every model is a corotated solid with Drucker-Prager plasticity (DpSandModel: no model dispatch); 4 = the same view on
// the flags mirror (ParticleArrays::fl: mass, volume0 and elastic hardening are the same for every binned particle and come from StepParams);
// 3 = every model is a neo-Hookean solid without plasticity (NeoHookeanOnlyModel); 5 = every model is a corotated solid with snow plasticity and no failure
// model (SnowOnlyModel; launched with RARE = true for the register budget, runs g2p_update<2>)
#ifndef SPK_RARE_MIN_BLOCKS
#define SPK_RARE_MIN_BLOCKS 3   // three workgroups per CU (168 VGPRs) instead of four with 244 B/lane of scratch: 1 M snow block k_g2p2g 0.121 -> 0.095 ms (two: 0.115), profiles/r05_kernel_experiments.txt
#endif
#ifndef SPK_SNOW_MIN_BLOCKS
#define SPK_SNOW_MIN_BLOCKS 4
#endif
template <int REDUCE_STEPS, bool RARE, bool LDS_MODELS = true, int MSET = 0>
__global__ __launch_bounds__(BLK_THREADS, (MSET == 5 ? SPK_SNOW_MIN_BLOCKS : (RARE ? SPK_RARE_MIN_BLOCKS : SPK_G2P2G_MIN_BLOCKS))) void k_g2p2g(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models, Ctl* ctl, StepParams S) {
    const int tid = threadIdx.x;
    __shared__ VelTile vt;
    __shared__ double tile[4][ACC_N];
    __shared__ int s_heads;
    __shared__ int s_runs[BLK_THREADS / 64][64];   // run list of each wave (run-task scatter)
    __shared__ float s_A[BLK_THREADS / 64][9][64]; // affine part of the scatter descriptors, per wave
    __shared__ DevModel s_models[LDS_MODELS ? MAX_LDS_MODELS : 1];
    // (read first: its scalar round trip then overlaps the block-list lookup's instead of following it)
    const uint32_t aborted = ctl->abort_flag[S.parity];
    const int pb = launch_pblock(G);
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    // Software pipeline: the five float4 of the next batch are in flight while this batch computes.  The first batch is requested
    // before anything else that touches memory (abort flag, model table, velocity tile): every vector load the wave waits for
    // also waits for all older ones, so whatever is issued first pays the HBM latency for the rest.
    // stream pointers of this block (wave-uniform) + one 32-bit byte offset per lane: see g2p_store_at
    ParticleArrays Pb = P;
#pragma unroll
    for (int sidx = 0; sidx < 6; ++sidx) Pb.q[sidx] = P.q[sidx] + pfirst;
    Pb.d = P.d + pfirst; Pb.k = P.k + pfirst; Pb.kinvel = P.kinvel + pfirst; Pb.aux = P.aux + pfirst;
    constexpr bool KU = MSET == 4;   // flags from the 4-byte mirror, the rest of the k record from StepParams
    if (P.fl) Pb.fl = P.fl + pfirst;
    auto load_k = [&](unsigned o16) -> float4 {
        if constexpr (KU) return make_float4(S.ku_mass, S.ku_vol0, S.ku_eh, __uint_as_float(*at_off(Pb.fl, o16 >> 2)));
        else return *at_off(Pb.k, o16);
    };
    // the small loads of the prologue go out AHEAD of the particle loads (vector loads return in order): the neighbour table of the velocity
    // tile (load_vel_tile) and the model table (stage_models_load); what used to be four dependent round trips before the first barrier is one
    const int my_nb = G.blk_nbr[b * 27 + min(tid & 63, 26)];
    ModelWords mw = {0u, 0u};
    if (LDS_MODELS) mw = stage_models_load(models, S.num_models, tid);
    // The plastic scalars (D stream): prefetched an iteration ahead like the rest where the update reads them early (PREFETCH_D: the
    // neo-Hookean view -- no return map in front of their first use, so a load issued at the top of the iteration is waited for in full:
    // +15 % on the 1 M cube), loaded at the top of the iteration that uses them where the return map sits in front (see the loop).
    // (MSET = 4 frees three of the four registers of the k prefetch and could carry the D prefetch without a spill: measured, no gain)
    constexpr bool PREFETCH_D = MSET == 3;
    float4 n0, n1, n2, nk, nd = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const unsigned o0 = (unsigned)min(tid, pcount - 1) << 4;
        n0 = *at_off(Pb.q[0], o0); n1 = *at_off(Pb.q[1], o0); n2 = *at_off(Pb.q[2], o0); nk = load_k(o0);
        if (PREFETCH_D) nd = *at_off(Pb.d, o0);
    }
    if (aborted) {   // sticky abort: keep every later substep a no-op until the host re-bins
        if (blockIdx.x == 0 && tid == 0) ctl->abort_flag[(S.parity + 1) & 1] = 1u;
        return;
    }
    if (tid == 0) s_heads = 0;
    // this launch scatters for substep s + 1: its heads go to [(parity + 1) & 1], the previous scatter's are at [parity]
    if (G.primary && blockIdx.x == 0 && tid == 0) disorder_policy(ctl, S.parity, S.parity);
    for (int n = tid; n < ACC_N; n += BLK_THREADS) { tile[0][n] = 0.0; tile[1][n] = 0.0; tile[2][n] = 0.0; tile[3][n] = 0.0; }
    // the particle loads above are already in flight: the L2 round trips of the model table and the velocity tile overlap
    // with their HBM latency instead of preceding it
    if (LDS_MODELS) stage_models_store(s_models, models, S.num_models, tid, mw);
    load_vel_tile(vt, G, b, tid, &my_nb);
    if (G.primary && blockIdx.x == 0 && tid == 0) note_step(ctl);
    __syncthreads();
    const float h = S.h;
    const TileRange tr = tile_range(org, S.dom);
    bool want_sort = false;
    int shell_used = 0;
    int wave_heads = 0;
    float dt_bound = 3.402823466e+38f;
    bool dt_nan = false;
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const unsigned off = (unsigned)(active ? li : 0) << 4;    // byte offset of this lane's particle in the block's 16-byte streams
        const float4 q0 = n0, q1 = n1, q2 = n2, kk = nk;
        // The plastic scalars (D stream) are not prefetched an iteration ahead like the rest: nothing reads them before the return map, ~350
        // instructions into the iteration, so their load is issued HERE, ahead of the next batch's (vector loads return in order: the wait
        // for it leaves the four younger loads in flight), and the four registers are free while the previous iteration's scatter runs --
        // the point where this kernel's register pressure peaks (with them prefetched the allocator spilled a float4 of the next batch
        // right behind its load, i.e. waited for the whole prefetch at the top of every iteration: +5 % on the launch).
        const float4 dd = PREFETCH_D ? nd : *at_off(Pb.d, off);
        if (it + 1 < niter) {
            const unsigned on = (unsigned)min(li + BLK_THREADS, pcount - 1) << 4;
            n0 = *at_off(Pb.q[0], on); n1 = *at_off(Pb.q[1], on); n2 = *at_off(Pb.q[2], on); nk = load_k(on);
            if (PREFETCH_D) nd = *at_off(Pb.d, on);
        }
        const uint32_t flags = __float_as_uint(kk.w);
        bool live = active && !(flags & FLAG_INVALID);
        // a lane without a live particle is never listed by the run-task scatter: only its margin bits have to be defined
        Scatter sc;
        if (REDUCE_STEPS < 0) {
            float u = 0.f; int ui = 0;
            sc.fx[0] = sc.fx[1] = sc.fx[2] = sc.val_m = sc.qv[0] = sc.qv[1] = sc.qv[2] = __builtin_nondeterministic_value(u);
#pragma unroll
            for (int e = 0; e < 9; ++e) sc.A.m[e] = __builtin_nondeterministic_value(u);
            sc.base = __builtin_nondeterministic_value(ui);
            sc.shell = 0;
        } else scatter_clear(sc);
        const ParkSlot park((lds_fp)(&s_A[tid >> 6][0][tid & 63]));
        // behind the particle update: store, then the scatter descriptor of the advected particle
        auto finish = [&](const auto& mo, const Advected& o) {
            const bool t_valid = o.T.valid;   // the six components themselves sit in the LDS (see g2p_update's `park`)
            SPK_MARK("store");
            g2p_store_at(Pb, off, o, S.defer_c);
            want_sort |= !o.in_tile && !(o.flags & FLAG_INVALID);   // a particle that left the DOMAIN is dead (skipped from now on): no re-bin for it
            live = o.in_tile;
            SPK_HOT(!S.want_dt);
            if (S.want_dt) {  // the bound the next substep's estimator would compute from this state
                const float bnd = particle_dt_bound(mo, o.flags, o.vel, o.C, kk.x, kk.y, o.eh, 1.f + o.Fe.m[0], S.h);
                dt_nan |= bnd != bnd;
                dt_bound = fminf(dt_bound, bnd);
            }
            SPK_MARK("p2g_prepare");
            if (live) {
                SymStress T2;
                T2.valid = t_valid; T2.jm1 = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) T2.t[k] = park.p[64 * k];   // (unconditionally: whatever the columns hold is ignored unless T2.valid)
                float fxn[3]; int lcn[3];
                cell_frac_r(o.px, o.rnew, fxn, lcn);
                sc = p2g_prepare(mo, o.flags, fxn, lcn, o.vel, o.C, o.Fe, kk.x, kk.y, o.eh, o.phase, o.Jp, h, S.inv_d, S.dt_next, &T2);
            }
            SPK_MARK("scatter");
            probe_extra_work(s_A[tid >> 6], tid, o.px, o.vel, sc);   // probe_hooks.h: nothing unless an analysis build asks for it
        };
        auto particle = [&](const auto& mo) {
            const float phase = rare_load((flags & FLAG_PHASE_NOT_ONE) != 0, &at_off(Pb.aux, off)->x, 1.f);
            const Advected o = g2p_update<(MSET == 5 ? 2 : (RARE ? 1 : 0)), true>(G, vt, mo, flags, q0, q1, q2, dd, kk, at_off(Pb.kinvel, off), at_off(Pb.aux, off), phase, org, tr, S, &ctl->num_invalid, park);
            finish(mo, o);
        };
        if (live) {
            const DevModel& mo_t = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
            if constexpr (MSET == 1 || MSET == 4) particle(DpSandModel(mo_t));
            else if constexpr (MSET == 3) particle(NeoHookeanOnlyModel(mo_t));
            else if constexpr (MSET == 5) particle(SnowOnlyModel(mo_t));
            else particle(mo_t);
        }
        shell_used |= wave_or_i32(sc.shell);
        // (a hybrid -- run tasks for regular waves, DPP for ragged ones -- was measured: the two paths in one kernel spill, +15 %)
        if (REDUCE_STEPS < 0) wave_heads += p2g_scatter_runs(tile, sc, live, tid, s_runs[tid >> 6], s_A[tid >> 6]);
        else wave_heads += p2g_scatter<(REDUCE_STEPS < 0 ? 0 : REDUCE_STEPS)>(tile, sc, live, tid);
    }
    SPK_MARK("epilogue");
    if ((tid & 63) == 0) atomicAdd(&s_heads, wave_heads);
    tile_write(tile, G, pb, tid, shell_used);
    if (tid == 0) record_heads(ctl, G, pb, (S.parity + 1) & 1, s_heads);   // (the host clears the books when the measure changes: heads_kind)
    if (__any(want_sort)) {
        if ((tid & 63) == 0) raise_abort(ctl, (S.parity + 1) & 1);
    }
    if (S.want_dt) {
        dt_block_min_store(G.dt_partial, dt_bound, dt_nan, ctl, tid);
    }
}

// Rebuilds v and C (q[3], q[4], q[5]) after a fixed-dt batch was cut short behind a fused launch that deferred them (StepParams::defer_c).
// The node velocities of the last substep that ran are still in place (every later grid update was a no-op) and the stored position is the
// ADVECTED one, x' = x + v(x) dt with v(x) the APIC gather at x (grid_to_particle.rs:47-86): v is the fixed point of  v = gather(x' - v dt),
// a contraction by |grad v| dt per round (1e-3 .. 1e-2 under any stable step), started from the gather at x' and iterated until two rounds
// agree to 2^-22 of the velocity (three or four gathers; at most eight).  What comes back: v and C to ~1e-6 relative (x' carries one
// rounding of the sum), the same order as the f32 gather's own rounding.  Kinematic particles take their prescribed velocity (exactly the
// kernel's override), static ones v = 0, C = 0, failed ones C = 0 (grid_to_particle.rs:82-121).  Only launches of the non-RARE kernels defer
// (no particle projection, no penalty kick on v).
__global__ __launch_bounds__(BLK_THREADS) void k_regather_c(ParticleArrays P, GridArrays G, StepParams S) {
    const int tid = threadIdx.x;
    __shared__ VelTile vt;
    const int pb = blockIdx.x;
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    load_vel_tile(vt, G, b, tid);
    __syncthreads();
    for (int li = tid; li < pcount; li += BLK_THREADS) {
        const int i = pfirst + li;
        const float4 q0 = P.q[0][i];
        const uint32_t flags = __float_as_uint(P.k[i].w);
        if (flags & FLAG_INVALID) continue;
        M3 C = m3_zero();
        float v[3] = {0.f, 0.f, 0.f};
        if (!(flags & FLAG_STATIC)) {
            const float x1[3] = {q0.x, q0.y, q0.z};
            if (flags & FLAG_KINEMATIC) {
                const float4 kv = P.kinvel[i];
                v[0] = kv.x; v[1] = kv.y; v[2] = kv.z;
                const float u[3] = {x1[0] - v[0] * S.dt_h, x1[1] - v[1] * S.dt_h, x1[2] - v[2] * S.dt_h};
                float vg[3];
                apic_gather<true>(vt, u, S, vg, C);
            } else {
                apic_gather<true>(vt, x1, S, v, C);
#pragma unroll 1
                for (int round = 0; round < 8; ++round) {
                    const float u[3] = {x1[0] - v[0] * S.dt_h, x1[1] - v[1] * S.dt_h, x1[2] - v[2] * S.dt_h};
                    float vn[3];
                    apic_gather<true>(vt, u, S, vn, C);
                    const float d = fmaxf(fmaxf(fabsf(vn[0] - v[0]), fabsf(vn[1] - v[1])), fabsf(vn[2] - v[2]));
                    const float s = fmaxf(fmaxf(fabsf(vn[0]), fabsf(vn[1])), fabsf(vn[2]));
                    v[0] = vn[0]; v[1] = vn[1]; v[2] = vn[2];
                    if (!(d > 2.4e-7f * s)) break;   // (also leaves on NaN)
                }
            }
            if (flags & FLAG_FAILED) C = m3_zero();
        }
        P.q[3][i] = make_float4(v[0], v[1], v[2], C.m[0]);
        P.q[4][i] = make_float4(C.m[1], C.m[2], C.m[3], C.m[4]);
        P.q[5][i] = make_float4(C.m[5], C.m[6], C.m[7], C.m[8]);
    }
}

// ------------------------------------------------------------------- in-place block reorder
// Re-sorts the particles of every particle block by their *current* cell, inside the block's own slot range: a stable
// counting sort over the 512 tile cells, one chunk of 512 particles at a time, the ten float4 of both particles of a
// thread held in registers between the read and the write.  Restores the same-cell lane runs the scatter's pre-reduction
// needs (see Ctl) for one pass over the particle data (~0.6 ms at 10M) instead of a full re-bin with its host round trip.
// Launched blindly every few substeps; does nothing unless the policy asked for it.
__global__ __launch_bounds__(BLK_THREADS, 4) void k_block_reorder(ParticleArrays P, GridArrays G, const Ctl* __restrict__ ctl, float h, int parity) {
    if (ctl->abort_flag[parity] || !ctl->reorder_request) return;
    __shared__ int hist[8][TILE_N];   // [slot-order segment = iteration * 4 + wave][cell]: count, then start offset
    __shared__ int wave_sum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pb = launch_pblock(G);
    {   // a block whose run count has not grown by an eighth since it was last in order has nothing to gain (workgroup-uniform)
        const int2 hd = G.pb_heads[pb];
        if (hd.y > 0 && hd.x <= hd.y + (hd.y >> 3)) return;
        if (tid == 0) G.pb_heads[pb] = make_int2(0, 0);   // re-armed: the next scatter records the new baseline
    }
    const int4 rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    for (int c0 = 0; c0 < pcount; c0 += 2 * BLK_THREADS) {
        const int cnt = min(2 * BLK_THREADS, pcount - c0);
        for (int n = tid; n < 8 * TILE_N; n += BLK_THREADS) (&hist[0][0])[n] = 0;
        __syncthreads();
        float4 r0[2];
        int key[2], rank[2], src[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int li = k * BLK_THREADS + tid;
            const bool valid = li < cnt;
            const int i = pfirst + c0 + (valid ? li : 0);
            src[k] = i;
            r0[k] = P.q[0][i];
            const float u[3] = {r0[k].x, r0[k].y, r0[k].z};
            int kk = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float rc = rintf(u[a]);
                const int lc = min(max(rc == rc ? (int)rc : 0, 0), TILE_W - 1);   // tile coordinate of the associated cell
                kk += lc << (3 * a);
            }
            key[k] = kk;
            rank[k] = valid ? atomicAdd(&hist[4 * k + wv][kk], 1) : 0;
        }
        __syncthreads();
        // exclusive scan over the 512 cells (two per thread), then per-segment start offsets
        int t0 = 0, t1 = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { t0 += hist[w][2 * tid]; t1 += hist[w][2 * tid + 1]; }
        const int both = t0 + t1;
        int incl = both;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            incl += (lane >= d) ? up : 0;
        }
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        int start0 = incl - both;
        for (int j = 0; j < wv; ++j) start0 += wave_sum[j];
        int start1 = start0 + t0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int a0 = hist[w][2 * tid], a1 = hist[w][2 * tid + 1];
            hist[w][2 * tid] = start0; hist[w][2 * tid + 1] = start1;
            start0 += a0; start1 += a1;
        }
        __syncthreads();
        const bool ok0 = tid < cnt, ok1 = BLK_THREADS + tid < cnt;
        const int s0 = src[0], s1 = src[1];
        const int d0 = pfirst + c0 + hist[wv][key[0]] + rank[0], d1 = pfirst + c0 + hist[4 + wv][key[1]] + rank[1];
        // The streams move in three groups (five float4 per particle in flight at a time: 40 registers; all ten at once spilled 80 B per
        // lane to scratch, a quarter more traffic than the move itself).  Per group: every read of the chunk completes (workgroup
        // barrier) before the first write -- the move is in place.
        {   // Q0 (already here) .. Q4
            const float4 a1 = P.q[1][s0], a2 = P.q[2][s0], a3 = P.q[3][s0], a4 = P.q[4][s0];
            const float4 b1 = P.q[1][s1], b2 = P.q[2][s1], b3 = P.q[3][s1], b4 = P.q[4][s1];
            __syncthreads();
            if (ok0) { P.q[0][d0] = r0[0]; P.q[1][d0] = a1; P.q[2][d0] = a2; P.q[3][d0] = a3; P.q[4][d0] = a4; }
            if (ok1) { P.q[0][d1] = r0[1]; P.q[1][d1] = b1; P.q[2][d1] = b2; P.q[3][d1] = b3; P.q[4][d1] = b4; }
        }
        {   // Q5, D, K, kinvel, aux
            const float4 a1 = P.q[5][s0], a2 = P.d[s0], a3 = P.k[s0], a4 = P.kinvel[s0], a5 = P.aux[s0];
            const float4 b1 = P.q[5][s1], b2 = P.d[s1], b3 = P.k[s1], b4 = P.kinvel[s1], b5 = P.aux[s1];
            __syncthreads();
            if (ok0) { P.q[5][d0] = a1; P.d[d0] = a2; P.k[d0] = a3; P.kinvel[d0] = a4; P.aux[d0] = a5; }
            if (ok1) { P.q[5][d1] = b1; P.d[d1] = b2; P.k[d1] = b3; P.kinvel[d1] = b4; P.aux[d1] = b5; }
        }
        if (P.fl) {   // kernel-uniform: the flags mirror follows its particles
            const uint32_t a1 = P.fl[s0], b1 = P.fl[s1];
            __syncthreads();
            if (ok0) P.fl[d0] = a1;
            if (ok1) P.fl[d1] = b1;
        }
        if (P.frac) {   // kernel-uniform
            const float4 a1 = P.frac[s0], b1 = P.frac[s1];
            __syncthreads();
            if (ok0) P.frac[d0] = a1;
            if (ok1) P.frac[d1] = b1;
        }
        __syncthreads();
    }
}
__global__ void k_reorder_done(Ctl* ctl, int parity) {
    if (ctl->abort_flag[parity] || !ctl->reorder_request) return;
    ctl->reorder_request = 0u;
    ctl->excess_acc = 0.f;
    ctl->reorders += 1u;
}

// ------------------------------------------------------------- fluid volume recompute (config 4)
// fluids_volume.rs:9-115: mass-only P2G (here), merge (k_grid_update<true>), then this gather.
SPK_DEV void mass_scatter(double* mtile, const float u[3], float mass, bool live, int tid);   // (below: shared with k_g2p_mass and the psi sweep)
__global__ __launch_bounds__(BLK_THREADS) void k_p2g_mass(ParticleArrays P, GridArrays G, const Ctl* __restrict__ ctl, StepParams S) {
    if (ctl->abort_flag[S.parity]) return;
    // same scheme as the full scatter: f64 LDS tile, same-cell lanes pre-summed over 16-lane groups with DPP (64 particles
    // per cell in the fluid configs: without it every one of the 27 atomics of a wave hits one address 64 times)
    __shared__ double tile[TILE_N];
    const int tid = threadIdx.x;
    const int pb = launch_pblock(G);
    for (int n = tid; n < TILE_N; n += BLK_THREADS) tile[n] = 0.0;
    const int4 rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    __syncthreads();
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const int i = pfirst + (active ? li : 0);
        const float4 q0 = P.q[0][i], kk = P.k[i];
        const bool live = active && !(__float_as_uint(kk.w) & FLAG_INVALID);
        if (!__any(live)) continue;
        // (the same weights as the momentum scatter: cell_frac of the stored block-relative position)
        const float u[3] = {q0.x, q0.y, q0.z};
        mass_scatter(tile, u, kk.x, live, tid);
    }
    __syncthreads();
    float* out = (float*)(G.tiles + (size_t)pb * TILE_N);   // mass tile: 8^3 scalars at the head of the block's slot
    for (int n = tid; n < TILE_N; n += BLK_THREADS) out[n] = (float)tile[n];
    if (tid == 0) G.tile_mode[pb] = TILE_MODE_MASS;
}

__global__ __launch_bounds__(BLK_THREADS) void k_g2p_volume(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models,
                                                            const Ctl* __restrict__ ctl, StepParams S) {
    if (ctl->abort_flag[S.parity]) return;
    __shared__ float4 vt[TILE_N];
    const int tid = threadIdx.x;
    const int pb = launch_pblock(G);
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    load_vel_tile(vt, G, b, tid);
    __syncthreads();
    const float h = S.h;
    for (int li = tid; li < pcount; li += BLK_THREADS) {
        const int i = pfirst + li;
        const float4 q0 = P.q[0][i], kk = P.k[i];
        const uint32_t flags = __float_as_uint(kk.w);
        if (flags & FLAG_INVALID) continue;
        if (!models[flags >> FLAG_MODEL_SHIFT].is_fluid) continue;
        const float u[3] = {q0.x, q0.y, q0.z};
        float fx[3], wgt[3][3];
        int lc[3];
        cell_frac(u, fx, lc);
#pragma unroll
        for (int a = 0; a < 3; ++a) eval_all(fx[a], wgt[a]);
        const int base = lc[0] + TILE_W * lc[1] + TILE_W * TILE_W * lc[2];
        float new_mass = 0.f;
#pragma unroll
        for (int sz = 0; sz < 3; ++sz)
#pragma unroll
            for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx)
                    new_mass += (wgt[0][sx] * wgt[1][sy] * wgt[2][sz]) * vt[base + sx + TILE_W * sy + TILE_W * TILE_W * sz].w;
        const float new_density = new_mass / (h * h * h);
        const float new_volume = kk.x / new_density;
        reinterpret_cast<float*>(&P.q[0][i])[3] = new_volume / kk.y - 1.f;   // stored as F00 - 1
    }
}

// ---- the same sweep fused into the substep kernels (fixed dt and adaptive) ------------------------------------------------
// Unfused, a substep with force_fluids_volume_recomputation is four passes over the particles (mass scatter, volume gather,
// P2G, G2P).  But the mass scatter of substep s+1 only needs the positions G2P of substep s has just written, and the volume
// gather only feeds the P2G that follows it: k_g2p_mass = G2P + mass scatter, k_vol_p2g = volume gather + P2G -- two passes.
// FLUID_ONLY: every model of the table is a plain fluid (no plastic / failure model).  Fluids never touch F beyond F00
// (grid_to_particle.rs:96-100), the plastic scalars or psi_pos (pos_energy == 0), so the F1..F8 and D streams are neither
// read nor written: 96 + 112 B per particle-substep instead of ~400.

// Mass-only scatter of one batch (one particle per lane) into a 512-entry f64 tile: same-cell lanes pre-summed over 16-lane rows
// with DPP (64 particles per cell in the fluid configs), run heads deposit 3 x 9 atomics.
// `fx`, `lc`: cell_frac of the particle's block-relative position (g2p_update already holds it: Advected::fxn, lcn); finite in dead lanes too
// (their sums meet live lanes' in the DPP chains with factor 0, and 0 x NaN is NaN).
SPK_DEV void mass_scatter(double* mtile, const float fx[3], const int lc[3], float mass, bool live, int tid) {
    if (!__any(live)) return;
    const int r = tid & 15;
    float wgt[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) eval_all(fx[a], wgt[a]);
    const int base = live ? lc[0] + TILE_W * lc[1] + TILE_W * TILE_W * lc[2] : 0;
    const float m = live ? mass : 0.f;
    const int key = live ? base : -1;
    const int key_prev = dpp_shr1_i(key, -2), key_next = dpp_shl_i<1>(key);
    const bool head = (r == 0) | (key_prev != key);
    const int i1 = ((r + 1 < 16) & (key_next == key)) ? 1 : 0;
    const int i2 = i1 & dpp_shl_i<1>(i1) & (r + 2 < 16 ? 1 : 0);
    const int i4 = i2 & dpp_shl_i<2>(i2) & (r + 4 < 16 ? 1 : 0);
    const int i8 = i4 & dpp_shl_i<4>(i4) & (r + 8 < 16 ? 1 : 0);
    const float f1 = (float)i1, f2 = (float)i2, f4 = (float)i4, f8 = (float)i8;
    // every row of the wave inside one cell (wave-uniform: the fluid configurations): twelve nodes at a time through the row reduce-scatter (three groups of
    // four travel as the three values of row_reduce_scatter4x3), the last three nodes through the suffix sums below
    const bool rows_uniform = __all(live & ((r == 0) | (key_prev == key)));
    if (rows_uniform) {
        const int quad = (tid >> 2) & 3;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float X[12];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = 12 * b + 4 * c + k, sx = idx % 3, sy = (idx / 3) % 3, sz = idx / 9;
                    X[3 * k + c] = wgt[0][sx] * (wgt[1][sy] * wgt[2][sz] * m);
                }
            row_reduce_scatter4x3(X);
            if ((tid & 3) == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    int nx, ny, nz;
                    support_node_of(12 * b + 4 * c + quad, nx, ny, nz);
                    atomicAdd(&mtile[base + nx + TILE_W * ny + TILE_W * TILE_W * nz], (double)X[c]);
                }
            }
        }
    }
#pragma unroll
    for (int sz = 0; sz < 3; ++sz)
#pragma unroll
        for (int sy = 0; sy < 3; ++sy) {
            if (rows_uniform && !(sz == 2 && sy == 2)) continue;   // (wave-uniform: nodes 0 .. 23 are done)
            const float wyz = wgt[1][sy] * wgt[2][sz] * m;
            float c0 = wgt[0][0] * wyz, c1 = wgt[0][1] * wyz, c2 = wgt[0][2] * wyz;   // three interleaved DPP chains: a dependent one sits two instructions behind
            asm volatile("s_nop 1\n" SPK_FMAC_DPP3("1", "%3") SPK_FMAC_DPP3("2", "%4") SPK_FMAC_DPP3("4", "%5") SPK_FMAC_DPP3("8", "%6")
                         : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(f1), "v"(f2), "v"(f4), "v"(f8));
            if (live && head) {
                const int n = base + TILE_W * sy + TILE_W * TILE_W * sz;
                atomicAdd(&mtile[n], (double)c0);
                atomicAdd(&mtile[n + 1], (double)c1);
                atomicAdd(&mtile[n + 2], (double)c2);
            }
        }
}

SPK_DEV void mass_scatter(double* mtile, const float u[3], float mass, bool live, int tid) {
    float fx[3];
    int lc[3];
    cell_frac(u, fx, lc);
    mass_scatter(mtile, fx, lc, mass, live, tid);
}

// G2P of substep s + mass-only scatter of the advected particles (= the first half of substep s+1's volume sweep).
// KU: the k record comes from the flags mirror + StepParams (ParticleArrays::fl): a plain fluid reads 20 instead of 32 bytes per particle here.
template <bool FLUID_ONLY, bool LDS_MODELS, bool KU = false>
__global__ __launch_bounds__(BLK_THREADS, 4) void k_g2p_mass(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models, Ctl* ctl, StepParams S) {
    const int tid = threadIdx.x;
    if (ctl->abort_flag[S.parity]) {
        if (blockIdx.x == 0 && tid == 0) ctl->abort_flag[(S.parity + 1) & 1] = 1u;
        return;
    }
    __shared__ VelTile vt;
    __shared__ double mtile[TILE_N];
    __shared__ DevModel s_models[LDS_MODELS ? MAX_LDS_MODELS : 1];
    const int pb = launch_pblock(G);
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    float4 n0, n1, n2, nd, nk;
    auto load_k = [&](int i) -> float4 {
        if constexpr (KU) return make_float4(S.ku_mass, S.ku_vol0, S.ku_eh, __uint_as_float(P.fl[i]));
        else return P.k[i];
    };
    {
        const int i0 = pfirst + min(tid, pcount - 1);
        n0 = P.q[0][i0]; nk = load_k(i0);
        if (!FLUID_ONLY) { n1 = P.q[1][i0]; n2 = P.q[2][i0]; nd = P.d[i0]; }
    }
    for (int n = tid; n < TILE_N; n += BLK_THREADS) mtile[n] = 0.0;
    load_vel_tile(vt, G, b, tid);
    if (LDS_MODELS) stage_models(s_models, models, S.num_models, tid);
    if (blockIdx.x == 0 && tid == 0) note_step(ctl);
    __syncthreads();
    const TileRange tr = tile_range(org, S.dom);
    bool want_sort = false;
    float dt_bound = 3.402823466e+38f;
    bool dt_nan = false;
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const int i = pfirst + (active ? li : 0);
        const float4 q0 = n0, kk = nk;
        // a plain fluid keeps F1..F8 = identity and never reads its plastic scalars
        const float4 q1 = FLUID_ONLY ? make_float4(0.f, 0.f, 0.f, 0.f) : n1, q2 = FLUID_ONLY ? make_float4(0.f, 0.f, 0.f, 0.f) : n2;   // F - I
        const float4 dd = FLUID_ONLY ? make_float4(1.f, 0.f, 1.f, 0.f) : nd;
        if (it + 1 < niter) {
            const int inx = pfirst + min(li + BLK_THREADS, pcount - 1);
            n0 = P.q[0][inx]; nk = load_k(inx);
            if (!FLUID_ONLY) { n1 = P.q[1][inx]; n2 = P.q[2][inx]; nd = P.d[inx]; }
        }
        const uint32_t flags = __float_as_uint(kk.w);
        bool live = active && !(flags & FLAG_INVALID);
        float nfx[3] = {1.f, 1.f, 1.f};
        int nlc[3] = {0, 0, 0};
        auto particle = [&](const auto& mo) {
            const float phase = rare_load((flags & FLAG_PHASE_NOT_ONE) != 0, &P.aux[i].x, 1.f);
            const Advected o = g2p_update<0>(G, vt, mo, flags, q0, q1, q2, dd, kk, P.kinvel + i, P.aux + i, phase, org, tr, S, &ctl->num_invalid);
            if (FLUID_ONLY) {
                P.q[0][i] = make_float4(o.px[0], o.px[1], o.px[2], o.Fe.m[0]);
                P.q[3][i] = make_float4(o.vel[0], o.vel[1], o.vel[2], o.C.m[0]);
                P.q[4][i] = make_float4(o.C.m[1], o.C.m[2], o.C.m[3], o.C.m[4]);
                P.q[5][i] = make_float4(o.C.m[5], o.C.m[6], o.C.m[7], o.C.m[8]);
                if (o.newly_failed) { reinterpret_cast<uint32_t*>(&P.k[i])[3] = o.flags; if (P.fl) P.fl[i] = o.flags; }
            } else {
                g2p_store(P, i, o);
            }
            want_sort |= !o.in_tile && !(o.flags & FLAG_INVALID);   // a particle that left the DOMAIN is dead (skipped from now on): no re-bin for it
            live = o.in_tile;
            if (live) {   // (a particle outside its tile scatters nothing; its cell index may be anything)
                cell_frac_r(o.px, o.rnew, nfx, nlc);
            }
            if (S.want_dt) {
                const float bnd = particle_dt_bound(mo, o.flags, o.vel, o.C, kk.x, kk.y, o.eh, 1.f + o.Fe.m[0], S.h);
                dt_nan |= bnd != bnd;
                dt_bound = fminf(dt_bound, bnd);
            }
        };
        if (live) {
            const DevModel& mo_t = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
            if constexpr (FLUID_ONLY) particle(PlainFluidModel(mo_t));   // the host selects FLUID_ONLY when every model is a plain fluid
            else particle(mo_t);
        }
        mass_scatter(mtile, nfx, nlc, kk.x, live, tid);
    }
    __syncthreads();
    float* out = (float*)(G.tiles + (size_t)pb * TILE_N);   // mass tile: 8^3 scalars at the head of the block's slot
    for (int n = tid; n < TILE_N; n += BLK_THREADS) out[n] = (float)mtile[n];
    if (tid == 0) G.tile_mode[pb] = TILE_MODE_MASS;
    if (__any(want_sort)) {
        if ((tid & 63) == 0) raise_abort(ctl, (S.parity + 1) & 1);
    }
    if (S.want_dt) dt_block_min_store(G.dt_partial, dt_bound, dt_nan, ctl, tid);
}

// Volume gather (second half of the sweep: F00 = (m / rho_grid) / V0 for fluid particles, fluids_volume.rs:83-110) + P2G.
// Two particles per lane like k_p2g_pair.
#ifndef SPK_VOLP2G_STEPS
#define SPK_VOLP2G_STEPS 4   // 16-lane groups: 64 particles per cell in the fluid configs (0.88 -> 0.73 ms at 20M against 8-lane groups)
#endif
template <bool FLUID_ONLY, bool LDS_MODELS, bool KU = false>
__global__ __launch_bounds__(BLK_THREADS) void k_vol_p2g(ParticleArrays P, GridArrays G, const DevModel* __restrict__ models, Ctl* __restrict__ ctl, StepParams S) {
    if (ctl->abort_flag[S.parity]) return;
    __shared__ double tile[4][ACC_N];
    __shared__ float vm[TILE_N];          // node masses of the 8^3 tile (the merged mass-only grid)
    __shared__ int s_heads;
    __shared__ DevModel s_models[LDS_MODELS ? MAX_LDS_MODELS : 1];
    // Pair transposition (one buffer per wave).  Lane j owns records 2j and 2j + 1 of the wave's 128 -- neighbours in the cell-sorted order, so
    // that their node values can be added in the lane --, and loading them as such makes every dwordx4 touch 16 bytes of each 32 (3.8 TB/s where
    // k_g2p_mass streams the same arrays at 4.9).  So the wave loads its 128 records with two COALESCED dwordx4 per stream (lane t: records t and
    // 64 + t) and hands them to their owners through the LDS: record e goes to slot (e >> 1) + 72 (e & 1) -- even records in slots 0..63, odd ones
    // in 72..135, the 8-slot gap puts the two halves of a 16-lane write pass on different banks --, lane j reads slots j and 72 + j.  LDS
    // operations of one wave execute in order: no barrier between the streams that reuse the buffer.
    __shared__ float4 s_xp[BLK_THREADS / 64][136];
    __shared__ float s_xj[BLK_THREADS / 64][136];
    const int tid = threadIdx.x;
    if (LDS_MODELS) stage_models(s_models, models, S.num_models, tid);
    if (tid == 0) s_heads = 0;
    if (blockIdx.x == 0 && tid == 0) disorder_policy(ctl, (S.parity + 1) & 1, S.parity);
    const int pb = launch_pblock(G);
    for (int n = tid; n < ACC_N; n += BLK_THREADS) { tile[0][n] = 0.0; tile[1][n] = 0.0; tile[2][n] = 0.0; tile[3][n] = 0.0; }
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    for (int n = tid; n < TILE_N; n += BLK_THREADS) {
        const int tx = n & 7, ty = (n >> 3) & 7, tz = n >> 6;
        const int dx = tx == 0 ? -1 : (tx <= 4 ? 0 : 1), dy = ty == 0 ? -1 : (ty <= 4 ? 0 : 1), dz = tz == 0 ? -1 : (tz <= 4 ? 0 : 1);
        const int cx = tx - 1 - 4 * dx, cy = ty - 1 - 4 * dy, cz = tz - 1 - 4 * dz;
        const int nb = G.blk_nbr[b * 27 + (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)];
        vm[n] = nb >= 0 ? G.node_vel[(size_t)nb * 64 + cx + 4 * cy + 16 * cz].w : 0.f;
    }
    const float h = S.h;
    __syncthreads();
    int shell_used = 0;
    int wave_heads = 0;
    const int lane = tid & 63, wv = tid >> 6;
    const int sA = (lane >> 1) + 72 * (lane & 1), sB = sA + 32;   // slots of records lane and 64 + lane
    float4* xp = s_xp[wv];
    auto hand_over = [&](const float4& a, const float4& b, float4 out[2]) {
        xp[sA] = a; xp[sB] = b;
        __builtin_amdgcn_wave_barrier();
        out[0] = xp[lane]; out[1] = xp[72 + lane];
        __builtin_amdgcn_wave_barrier();
    };
    for (int base = 0; base < pcount; base += 2 * BLK_THREADS) {
        Scatter sc[2];
        bool live[2], f00_store[2] = {false, false};
        const int wbase = base + wv * 128;                                   // first record of this wave's 128 (block-relative)
        const int eA = pfirst + min(wbase + lane, pcount - 1), eB = pfirst + min(wbase + 64 + lane, pcount - 1);
        // every load of the batch is issued here; the velocity / affine streams are only waited for behind the volume gather
        // (KU: the flags words of the mirror travel through the 4-byte hand-over buffer, the rest of the k record is in StepParams: ParticleArrays::fl)
        const float4 a0 = P.q[0][eA], b0 = P.q[0][eB];
        float4 ak = make_float4(0.f, 0.f, 0.f, 0.f), bk = ak;
        uint32_t afl = 0u, bfl = 0u;
        if constexpr (KU) { afl = P.fl[eA]; bfl = P.fl[eB]; } else { ak = P.k[eA]; bk = P.k[eB]; }
        // plastic_deformation_gradient_det, the "fluid J" of the stress (eos_monaghan_sph.rs glue).  Nothing ever changes it for a plain fluid, and
        // every particle generator leaves it at 1: when the host saw only ones at upload (StepParams::jp_one) the D stream -- 16 B per particle of
        // which 4 are used: the 12 % of this kernel's HBM traffic beyond its algorithmic bytes, r04 PMC -- is not read at all
        float ja = 1.f, jb = 1.f;
        if (!(FLUID_ONLY && S.jp_one)) { ja = P.d[eA].z; jb = P.d[eB].z; }
        const float4 a3 = P.q[3][eA], b3 = P.q[3][eB], a4 = P.q[4][eA], b4 = P.q[4][eB], a5 = P.q[5][eA], b5 = P.q[5][eB];
        float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = a1, a2 = a1, b2 = a1;
        if (!FLUID_ONLY) { a1 = P.q[1][eA]; b1 = P.q[1][eB]; a2 = P.q[2][eA]; b2 = P.q[2][eB]; }
        // ---- phase 1: position, mass, flags -> volume gather (F00 = (m / rho_grid) / V0 for fluid particles, fluids_volume.rs:83-110)
        float4 q0[2], kk[2];
        float jp_stored[2], cfx[2][3];
        int clc[2][3];
        hand_over(a0, b0, q0);
        if constexpr (KU) {
            float* xj = s_xj[wv];
            xj[sA] = __uint_as_float(afl); xj[sB] = __uint_as_float(bfl);
            __builtin_amdgcn_wave_barrier();
            kk[0] = make_float4(S.ku_mass, S.ku_vol0, S.ku_eh, xj[lane]); kk[1] = make_float4(S.ku_mass, S.ku_vol0, S.ku_eh, xj[72 + lane]);
            __builtin_amdgcn_wave_barrier();
        } else hand_over(ak, bk, kk);
        jp_stored[0] = jp_stored[1] = 1.f;
        if (!(FLUID_ONLY && S.jp_one)) {
            float* xj = s_xj[wv];
            xj[sA] = ja; xj[sB] = jb;
            __builtin_amdgcn_wave_barrier();
            jp_stored[0] = xj[lane]; jp_stored[1] = xj[72 + lane];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t flags = __float_as_uint(kk[k].w);
            live[k] = (base + 2 * tid + k < pcount) && !(flags & FLAG_INVALID);
            scatter_clear(sc[k]);
            cfx[k][0] = cfx[k][1] = cfx[k][2] = 1.f;
            clc[k][0] = clc[k][1] = clc[k][2] = 0;
            auto volume = [&](const auto& mo) {
                const float u[3] = {q0[k].x, q0[k].y, q0[k].z};
                cell_frac(u, cfx[k], clc[k]);   // once for the gather and the scatter
                if (mo.is_fluid) {   // with the weights of k_g2p_volume
                    float wgt[3][3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) eval_all(cfx[k][a], wgt[a]);
                    const int vb = clc[k][0] + TILE_W * clc[k][1] + TILE_W * TILE_W * clc[k][2];
                    float new_mass = 0.f;
#pragma unroll
                    for (int sz = 0; sz < 3; ++sz)
#pragma unroll
                        for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                            for (int sx = 0; sx < 3; ++sx)
                                new_mass += (wgt[0][sx] * wgt[1][sy] * wgt[2][sz]) * vm[vb + sx + TILE_W * sy + TILE_W * TILE_W * sz];
                    // F00 = (m / (M / h^3)) / V0 as one quotient (the reference divides three times: the same number to 1e-7)
                    q0[k].w = (kk[k].x * (h * h * h)) / (new_mass * kk[k].y) - 1.f;   // stored as F00 - 1
                    f00_store[k] = true;
                }
            };
            if (live[k]) {
                const DevModel& mo_t = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
                if constexpr (FLUID_ONLY) volume(PlainFluidModel(mo_t));
                else volume(mo_t);
            }
        }
        // ---- phase 2: velocity, affine (and F1..F8 of solids) -> scatter descriptors
        float4 q3[2], q4[2], q5[2], q1[2], q2[2];
        hand_over(a3, b3, q3); hand_over(a4, b4, q4); hand_over(a5, b5, q5);
        if (!FLUID_ONLY) { hand_over(a1, b1, q1); hand_over(a2, b2, q2); }
        else { q1[0] = q1[1] = q2[0] = q2[1] = make_float4(0.f, 0.f, 0.f, 0.f); }   // a plain fluid keeps F1..F8 = identity
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t flags = __float_as_uint(kk[k].w);
            auto prepare = [&](const auto& mo) {
                const float v[3] = {q3[k].x, q3[k].y, q3[k].z};
                const M3 F = {{q0[k].w, q1[k].x, q1[k].y, q1[k].z, q1[k].w, q2[k].x, q2[k].y, q2[k].z, q2[k].w}};   // F - I as stored
                const M3 C = {{q3[k].w, q4[k].x, q4[k].y, q4[k].z, q4[k].w, q5[k].x, q5[k].y, q5[k].z, q5[k].w}};
                const float phase = rare_load((flags & FLAG_PHASE_NOT_ONE) != 0, &P.aux[pfirst + base + 2 * tid + k].x, 1.f);
                const float jp = mo.is_fluid ? jp_stored[k] : 1.f;
                sc[k] = p2g_prepare(mo, flags, cfx[k], clc[k], v, C, F, kk[k].x, kk[k].y, kk[k].z, phase, jp, h, S.inv_d, S.dt);
            };
            if (live[k]) {
                const DevModel& mo_t = LDS_MODELS ? s_models[flags >> FLAG_MODEL_SHIFT] : models[flags >> FLAG_MODEL_SHIFT];
                if constexpr (FLUID_ONLY) prepare(PlainFluidModel(mo_t));
                else prepare(mo_t);
            }
        }
        if (__any(f00_store[0] | f00_store[1])) {   // wave-uniform.  The way back: every in-range record is stored (unchanged unless it is a fluid's), coalesced
            xp[lane] = q0[0]; xp[72 + lane] = q0[1];
            __builtin_amdgcn_wave_barrier();
            const float4 ra = xp[sA], rb = xp[sB];
            __builtin_amdgcn_wave_barrier();
            if (wbase + lane < pcount) P.q[0][pfirst + wbase + lane] = ra;
            if (wbase + 64 + lane < pcount) P.q[0][pfirst + wbase + 64 + lane] = rb;
        }
        shell_used |= wave_or_i32(sc[0].shell | sc[1].shell);
        wave_heads += p2g_scatter_pair<SPK_VOLP2G_STEPS, false>(tile, sc[0], sc[1], live[0], live[1], tid);   // the mass is on the grid already (StepParams::mass_from_sweep)
    }
    if ((tid & 63) == 0) atomicAdd(&s_heads, wave_heads);
    tile_write(tile, G, pb, tid, shell_used);
    if (tid == 0) record_heads(ctl, G, pb, S.parity, s_heads);   // launch total + the block's own record (k_block_reorder skips blocks whose count has not grown)
}

// ---- DamageModel::ModifiedEigenerosion: the psi_mass / psi_momentum node transfer -----------------------------------------
// particle_to_grid.rs:62-86 scatters, beside mass and momentum, psi_mass = m (0 for failed / cracked particles and those whose
// crack_propagation_factor is 0) and psi_momentum = psi_mass * psi_pos; grid_to_particle.rs:57-77 gathers
// sum w psi_momentum / psi_mass, and a particle whose crack_propagation_factor * h * that exceeds its crack_threshold gets
// phase = 0 before it is advected.  Three small kernels between the grid update and G2P, launched only under that damage model
// with ParticleFracture uploaded, so the hot kernels do not carry two more node planes.
__global__ __launch_bounds__(BLK_THREADS) void k_p2g_psi(ParticleArrays P, GridArrays G, const Ctl* __restrict__ ctl, StepParams S, float2* __restrict__ psi_tiles) {
    if (ctl->abort_flag[S.parity]) return;
    __shared__ double ta[TILE_N], tb[TILE_N];
    const int tid = threadIdx.x;
    const int pb = launch_pblock(G);
    for (int n = tid; n < TILE_N; n += BLK_THREADS) { ta[n] = 0.0; tb[n] = 0.0; }
    const int4 rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    __syncthreads();
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const int i = pfirst + (active ? li : 0);
        const float4 q0 = P.q[0][i], kk = P.k[i], dd = P.d[i], fr = P.frac[i];
        const uint32_t flags = __float_as_uint(kk.w);
        const bool live = active && !(flags & FLAG_INVALID);
        const float phase = (flags & FLAG_PHASE_NOT_ONE) ? P.aux[i].x : 1.f;
        const float psi_mass = (live && phase > 0.f && fr.x != 0.f && !(flags & FLAG_FAILED)) ? kk.x : 0.f;
        const float u[3] = {q0.x, q0.y, q0.z};
        mass_scatter(ta, u, psi_mass, live, tid);
        mass_scatter(tb, u, psi_mass * dd.w, live, tid);
    }
    __syncthreads();
    float2* out = psi_tiles + (size_t)pb * TILE_N;
    for (int n = tid; n < TILE_N; n += BLK_THREADS) out[n] = make_float2((float)ta[n], (float)tb[n]);
}
// one wave per node block: psi_momentum * inv_exact(psi_mass) of its 64 nodes, summed over the <= 8 overlapping tiles
__global__ __launch_bounds__(BLK_THREADS) void k_grid_psi(GridArrays G, const float2* __restrict__ psi_tiles, float* __restrict__ node_psi, const Ctl* __restrict__ ctl, int parity,
                                                           const float4* __restrict__ psi_halo = nullptr) {
    if (ctl->abort_flag[parity]) return;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= G.num_blocks) return;
    const int lane = threadIdx.x & 63;
    const int cx = lane & 3, cy = (lane >> 2) & 3, cz = lane >> 4;
    const int dx0 = cx <= 2 ? -1 : 0, dy0 = cy <= 2 ? -1 : 0, dz0 = cz <= 2 ? -1 : 0;
    float m = 0.f, pm = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = dx0 + (k & 1), dy = dy0 + ((k >> 1) & 1), dz = dz0 + (k >> 2);
        const int t = G.blk_tnbr[b * 27 + (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)];
        if (t < 0) continue;
        const int tx = cx - 4 * dx + 1, ty = cy - 4 * dy + 1, tz = cz - 4 * dz + 1;
        const float2 v = psi_tiles[(size_t)t * TILE_N + tx + TILE_W * ty + TILE_W * TILE_W * tz];
        m += v.x; pm += v.y;
    }
    if (psi_halo) {   // the neighbour slab's partial sums (slab groups): same planes as the mass / momentum halo
        const int4 ob = G.blk_origin[b];
        const int bp = (G.halo_axis == 0 ? ob.x : (G.halo_axis == 1 ? ob.y : ob.z)) >> 2;
        if (bp == G.halo_lo || bp == G.halo_hi || bp + 1 == G.halo_lo || bp + 1 == G.halo_hi) {
            const float4 hv = psi_halo[(size_t)b * 64 + lane];
            m += hv.x; pm += hv.y;
        }
    }
    node_psi[(size_t)b * 64 + lane] = pm * inv_exact(m);
}
__global__ __launch_bounds__(BLK_THREADS) void k_g2p_psi(ParticleArrays P, GridArrays G, const float* __restrict__ node_psi, const Ctl* __restrict__ ctl, StepParams S) {
    if (ctl->abort_flag[S.parity]) return;
    __shared__ float vp[TILE_N];
    const int tid = threadIdx.x;
    const int pb = launch_pblock(G);
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w, pfirst = rng.x, pcount = rng.y;
    for (int n = tid; n < TILE_N; n += BLK_THREADS) {
        const int tx = n & 7, ty = (n >> 3) & 7, tz = n >> 6;
        const int dx = tx == 0 ? -1 : (tx <= 4 ? 0 : 1), dy = ty == 0 ? -1 : (ty <= 4 ? 0 : 1), dz = tz == 0 ? -1 : (tz <= 4 ? 0 : 1);
        const int cx = tx - 1 - 4 * dx, cy = ty - 1 - 4 * dy, cz = tz - 1 - 4 * dz;
        const int nb = G.blk_nbr[b * 27 + (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)];
        vp[n] = nb >= 0 ? node_psi[(size_t)nb * 64 + cx + 4 * cy + 16 * cz] : 0.f;
    }
    __syncthreads();
    const float h = S.h;
    for (int li = tid; li < pcount; li += BLK_THREADS) {
        const int i = pfirst + li;
        const float4 fr = P.frac[i];
        if (fr.x == 0.f) continue;
        const float4 q0 = P.q[0][i];
        const uint32_t flags = reinterpret_cast<const uint32_t*>(&P.k[i])[3];
        if (flags & FLAG_INVALID) continue;
        const float phase = (flags & FLAG_PHASE_NOT_ONE) ? P.aux[i].x : 1.f;
        if (!(phase > 0.f)) continue;
        const float u[3] = {q0.x, q0.y, q0.z};
        float fx[3], wgt[3][3];
        int lc[3];
        cell_frac(u, fx, lc);
#pragma unroll
        for (int a = 0; a < 3; ++a) eval_all(fx[a], wgt[a]);
        const int base = lc[0] + TILE_W * lc[1] + TILE_W * TILE_W * lc[2];
        float g = 0.f;
#pragma unroll
        for (int sz = 0; sz < 3; ++sz)
#pragma unroll
            for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) g += (wgt[0][sx] * wgt[1][sy] * wgt[2][sz]) * vp[base + sx + TILE_W * sy + TILE_W * TILE_W * sz];
        if (fr.x * h * g > fr.y) {   // crack_energy > crack_threshold: phase = 0 from now on
            reinterpret_cast<float*>(&P.aux[i])[0] = 0.f;
            reinterpret_cast<uint32_t*>(&P.k[i])[3] = flags | FLAG_PHASE_NOT_ONE;
        }
    }
}
// ----------------------------------------------------------------------------------- dt estimate
// timestep_estimator.rs:10-51: min over particles of the velocity bound and the model bound.
__global__ __launch_bounds__(BLK_THREADS) void k_estimate_dt(ParticleArrays P, const DevModel* __restrict__ models, Ctl* ctl, int n, float h, uint32_t* dt_partial) {
    const int i = blockIdx.x * BLK_THREADS + threadIdx.x;
    float dt = 3.402823466e+38f;
    bool nan = false;
    if (i < n) {
        const float4 q3 = P.q[3][i], q4 = P.q[4][i], q5 = P.q[5][i], kk = P.k[i];
        const uint32_t flags = __float_as_uint(kk.w);
        const DevModel& mo = models[flags >> FLAG_MODEL_SHIFT];
        const float v[3] = {q3.x, q3.y, q3.z};
        const M3 C = {{q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w}};
        dt = particle_dt_bound(mo, flags, v, C, kk.x, kk.y, kk.z, mo.is_fluid ? 1.f + P.q[0][i].w : 1.f, h);
        nan = dt != dt;
    }
    dt_block_min_store(dt_partial, dt, nan, ctl, threadIdx.x);
}

// ------------------------------------------------------------------------------- re-bin (resort)
SPK_DEV uint32_t spread10(uint32_t x) {
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
SPK_DEV uint32_t compact10(uint32_t x) {
    x &= 0x09249249u;
    x = (x | (x >> 2)) & 0x030C30C3u;
    x = (x | (x >> 4)) & 0x0300F00Fu;
    x = (x | (x >> 8)) & 0x030000FFu;
    x = (x | (x >> 16)) & 0x3ffu;
    return x;
}
// block key = Morton code of the block coordinate biased by +1 (blocks at -1 are legal node blocks)
SPK_DEV uint32_t block_key(int bx, int by, int bz) { return spread10(bx + 1) | (spread10(by + 1) << 1) | (spread10(bz + 1) << 2); }

// Sort keys: (block key << 6) | cell in block, as 32-bit words whenever they fit (3 x block_bits + 6 + 1 <= 32: every domain up to 2^33
// bytes; the default 2^30-byte grid uses 28 bits), else 64-bit: the radix sort then moves 8 instead of 12 bytes per pair and pass.
// Tail classes: every valid key is < 2^(bits - 1), these are the four largest words and sort after them.
template <class KeyT>
struct TailKeys {
    static constexpr KeyT EMIG_LO = (KeyT)(~(KeyT)0 - 3);      // home block below the slab's low cut
    static constexpr KeyT EMIG_HI = (KeyT)(~(KeyT)0 - 2);      // home block beyond the slab's high cut
    static constexpr KeyT INVALID = (KeyT)(~(KeyT)0 - 1);      // outside the SPGrid domain (frozen, failed)
    static constexpr KeyT REMOVED = (KeyT)(~(KeyT)0);          // already handed to a neighbour slab: dropped by this re-bin
};
// what k_compute_keys leaves for k_permute beside the key: the origin the particle's stored position is relative to (block coordinates,
// 10 bits each), or OORG_WORLD for a particle of the tail (world coordinates)
constexpr uint32_t OORG_WORLD = 0xffffffffu;
SPK_DEV uint32_t pack_org(int ox, int oy, int oz) { return (uint32_t)(ox >> 2) | ((uint32_t)(oy >> 2) << 10) | ((uint32_t)(oz >> 2) << 20); }

// `c`: associated cell (world cell coordinates) when `ok`; `ok` = inside the domain
template <class KeyT>
SPK_DEV void emit_key(const ParticleArrays& P, KeyT* keys, uint32_t* idx, uint32_t* oorg, int i, const int c[3], bool ok, uint32_t org_word, float h,
                      int slab_axis, int slab_lo, int slab_hi, Ctl* ctl) {
    uint32_t flags = reinterpret_cast<const uint32_t*>(&P.k[i])[3];
    KeyT key = TailKeys<KeyT>::INVALID;
    if (flags & FLAG_REMOVED) {
        key = TailKeys<KeyT>::REMOVED;
    } else if (ok) {
        const uint32_t bk = block_key(c[0] >> 2, c[1] >> 2, c[2] >> 2);
        key = ((KeyT)bk << 6) | (KeyT)((c[0] & 3) | ((c[1] & 3) << 2) | ((c[2] & 3) << 4));
        if (flags & FLAG_INVALID) { flags &= ~FLAG_INVALID; reinterpret_cast<uint32_t*>(&P.k[i])[3] = flags; }
        const int bs = c[slab_axis] >> 2;
        if (bs < slab_lo) key = TailKeys<KeyT>::EMIG_LO;
        else if (bs >= slab_hi) key = TailKeys<KeyT>::EMIG_HI;
    } else if ((flags & (FLAG_INVALID | FLAG_FAILED)) != (FLAG_INVALID | FLAG_FAILED)) {
        flags |= FLAG_INVALID | FLAG_FAILED;  // particle_set.rs:132-135
        reinterpret_cast<uint32_t*>(&P.k[i])[3] = flags;
    }
    if (key == TailKeys<KeyT>::INVALID) {   // outside the domain: only its velocity bound still takes part in the adaptive step (Ctl::dead_dt_bits)
        const float4 q3 = P.q[3][i], q4 = P.q[4][i], q5 = P.q[5][i];
        const float v[3] = {q3.x, q3.y, q3.z};
        const M3 C = {{q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w}};
        atomicMin(&ctl->dead_dt_bits, __float_as_uint(velocity_dt_bound(v, C, h)));
    }
    keys[i] = key;
    idx[i] = (uint32_t)i;
    oorg[i] = org_word;
}
// slots [first, n): the tail of the arrays, WORLD coordinates (fresh uploads, appended particles, immigrants, emigrants, dead particles)
template <class KeyT>
__global__ void k_compute_keys_world(ParticleArrays P, KeyT* keys, uint32_t* idx, uint32_t* oorg, int first, int n, float h, int dom, int slab_axis, int slab_lo, int slab_hi, Ctl* ctl) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q0 = P.q[0][i];
    const float px[3] = {q0.x, q0.y, q0.z};
    int c[3];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float r = round_cell_world(px[a], h);
        if (!((r >= 1.0f) && (r < (float)(dom - 4)))) ok = false;
        c[a] = ok ? (int)r - 1 : 0;
    }
    emit_key(P, keys, idx, oorg, i, c, ok, OORG_WORLD, h, slab_axis, slab_lo, slab_hi, ctl);
}
// slots [0, n_binned): one workgroup per particle block of the OLD binning, block-relative coordinates
template <class KeyT>
__global__ __launch_bounds__(BLK_THREADS) void k_compute_keys_binned(ParticleArrays P, const int4* __restrict__ pb_info, KeyT* keys, uint32_t* idx, uint32_t* oorg, float h, int dom,
                                                                     int slab_axis, int slab_lo, int slab_hi, Ctl* ctl) {
    const int pb = blockIdx.x;
    const int4 org = pb_info[2 * pb], rng = pb_info[2 * pb + 1];
    const int orgv[3] = {org.x, org.y, org.z};
    const uint32_t ow = pack_org(org.x, org.y, org.z);
    for (int li = threadIdx.x; li < rng.y; li += BLK_THREADS) {
        const int i = rng.x + li;
        const float4 q0 = P.q[0][i];
        const float u[3] = {q0.x, q0.y, q0.z};
        int c[3];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float r = rintf(u[a]);
            const int rw = orgv[a] + (int)fminf(fmaxf(r, -1.0e6f), 1.0e6f);   // round(x / h)
            if (!(u[a] == u[a]) || rw < 1 || rw >= dom - 4) ok = false;
            c[a] = ok ? rw - 1 : 0;
        }
        emit_key(P, keys, idx, oorg, i, c, ok, ow, h, slab_axis, slab_lo, slab_hi, ctl);
    }
}

template <class KeyT>
__global__ void k_shift_keys(const KeyT* __restrict__ keys, uint64_t* bkeys, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bkeys[i] = (uint64_t)(keys[i] >> 6);   // (the four tail classes share one block key, larger than every valid one)
}

// Moves every stream to the sorted order and re-bases the positions: a particle that lands in a block gets its position relative to THAT
// block's origin (decoded from its key) -- from the origin it was relative to (oorg) by adding the integer difference, which leaves the
// coordinate of a particle that stays in its block bit for bit alone, or from world coordinates with one rounding --; a particle that lands
// in the tail goes back to world coordinates.
template <class KeyT>
__global__ void k_permute(ParticleArrays src, ParticleArrays dst, const uint32_t* __restrict__ perm, const KeyT* __restrict__ keys_sorted, const uint32_t* __restrict__ oorg, int n, float h,
                          uint32_t* __restrict__ fl, uint32_t* __restrict__ k_not_uniform) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    const KeyT key = keys_sorted[i];
    const uint32_t ow = oorg[s];
    float4 q0 = src.q[0][s];
    const int ox = (int)(ow & 1023u) * 4, oy = (int)((ow >> 10) & 1023u) * 4, oz = (int)((ow >> 20) & 1023u) * 4;
    if (key < TailKeys<KeyT>::EMIG_LO) {
        const uint32_t bk = (uint32_t)(key >> 6);
        const int nx = ((int)compact10(bk) - 1) * 4, ny = ((int)compact10(bk >> 1) - 1) * 4, nz = ((int)compact10(bk >> 2) - 1) * 4;
        if (ow == OORG_WORLD) { q0.x = rel_from_world(q0.x, nx, h); q0.y = rel_from_world(q0.y, ny, h); q0.z = rel_from_world(q0.z, nz, h); }
        else { q0.x += (float)(ox - nx); q0.y += (float)(oy - ny); q0.z += (float)(oz - nz); }
    } else if (ow != OORG_WORLD) {
        q0.x = world_from_rel(q0.x, ox, h); q0.y = world_from_rel(q0.y, oy, h); q0.z = world_from_rel(q0.z, oz, h);
    }
    dst.q[0][i] = q0;
#pragma unroll
    for (int a = 1; a < 6; ++a) dst.q[a][i] = src.q[a][s];
    dst.d[i] = src.d[s];
    const float4 kr = src.k[s];
    dst.k[i] = kr;
    if (fl) {   // flags mirror (ParticleArrays::fl) + do all binned particles share the record of the first one?
        fl[i] = __float_as_uint(kr.w);
        const float4 k0 = src.k[perm[0]];
        const bool differs = key < TailKeys<KeyT>::EMIG_LO &&
                             (__float_as_uint(kr.x) != __float_as_uint(k0.x) || __float_as_uint(kr.y) != __float_as_uint(k0.y) || __float_as_uint(kr.z) != __float_as_uint(k0.z));
        if (__any(differs) && (threadIdx.x & 63) == __builtin_ctzll(__ballot(differs))) atomicOr(k_not_uniform, 1u);
    }
    dst.kinvel[i] = src.kinvel[s];
    dst.aux[i] = src.aux[s];
    if (src.frac) dst.frac[i] = src.frac[s];
}

// 27 candidate node blocks around every particle block
// Launch order of the particle blocks (GridArrays::pb_list with pb_plain = 1, of a pipeline that is not split into slab subsets): the fullest blocks
// first, so that the workgroups still running when the grid runs out are the short ones (largest-processing-time-first list scheduling).  Consecutive
// workgroups go to the XCDs round robin, so every XCD gets every eighth block of the sorted list; the spatial run per XCD that xcd_remap keeps is given
// up -- measured: the 20 M dam (6 256 blocks of 0 - 4 096 particles on 1 024 workgroup slots) -7.3 %, the 10 M column deep in its collapse -2.4 %, the
// lattice at rest unchanged; the same order inside each XCD's run: -1.4 % / nothing (profiles/r04_kernel_experiments.txt, 14).
// Key = 0xffffff - count: one stable radix sort of (key, block index) gives the list (ties stay in Morton order).
__global__ void k_order_keys(const uint32_t* __restrict__ pb_counts, int npb, uint32_t* keys, uint32_t* vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npb) return;
    keys[i] = 0xffffffu - min(pb_counts[i], 0xffffffu);
    vals[i] = (uint32_t)i;
}
__global__ void k_candidates(const uint64_t* __restrict__ pb_keys, int npb, uint32_t* cand) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npb * 27) return;
    const int j = t / 27, d = t % 27;
    const uint32_t k = (uint32_t)pb_keys[j];
    const int bx = (int)compact10(k) - 1, by = (int)compact10(k >> 1) - 1, bz = (int)compact10(k >> 2) - 1;
    cand[t] = block_key(bx + (d % 3) - 1, by + ((d / 3) % 3) - 1, bz + (d / 9) - 1);
}

constexpr uint64_t HASH_EMPTY = ~0ull;   // what hipMemset(0xFF) leaves in a slot
struct HashTable {
    uint64_t* keys;   // HASH_EMPTY = free slot
    uint32_t* vals;
    uint32_t mask;    // capacity - 1 (power of two, load <= 50 %)
};
// reference hash (src_kernels/cuda/hashmap.rs:7-14)
SPK_DEV uint64_t hash64(uint64_t key) {
    key ^= key >> 16; key *= 0x85ebca6bull; key ^= key >> 13; key *= 0xc2b2ae35ull; key ^= key >> 16;
    return key;
}
SPK_DEV void hash_insert(const HashTable& H, uint64_t key, uint32_t val) {
    uint32_t slot = (uint32_t)hash64(key) & H.mask;
    for (uint32_t probe = 0; probe <= H.mask; ++probe) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&H.keys[slot]), (unsigned long long)HASH_EMPTY, (unsigned long long)key);
        if (prev == HASH_EMPTY || prev == key) { H.vals[slot] = val; return; }
        slot = (slot + 1) & H.mask;
    }
}
SPK_DEV int hash_lookup(const HashTable& H, uint64_t key) {
    uint32_t slot = (uint32_t)hash64(key) & H.mask;
    for (uint32_t probe = 0; probe <= H.mask; ++probe) {
        const uint64_t k = H.keys[slot];
        if (k == key) return (int)H.vals[slot];
        if (k == HASH_EMPTY) return -1;
        slot = (slot + 1) & H.mask;
    }
    return -1;
}

__global__ void k_setup_blocks(const uint32_t* __restrict__ blk_keys, int nb, int4* origin, int* pfirst, int* pcount, HashTable H) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t k = blk_keys[b];
    origin[b] = make_int4(((int)compact10(k) - 1) * 4, ((int)compact10(k >> 1) - 1) * 4, ((int)compact10(k >> 2) - 1) * 4, -1);
    pfirst[b] = 0;
    pcount[b] = 0;
    hash_insert(H, (uint64_t)k, (uint32_t)b);
}
__global__ void k_link_pblocks(const uint64_t* __restrict__ pb_keys, const uint32_t* __restrict__ pb_counts, const uint32_t* __restrict__ pb_first,
                               int npb, HashTable H, int4* origin, int* pfirst, int* pcount, int* pblk, int4* pb_info) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npb) return;
    const int b = hash_lookup(H, pb_keys[j]);
    pblk[j] = b;
    origin[b].w = j;
    pfirst[b] = (int)pb_first[j];
    pcount[b] = (int)pb_counts[j];
    const int4 o = origin[b];
    pb_info[2 * j] = make_int4(o.x, o.y, o.z, b);
    pb_info[2 * j + 1] = make_int4((int)pb_first[j], (int)pb_counts[j], 0, 0);
}
__global__ void k_link_nbrs(const uint32_t* __restrict__ blk_keys, int nb, HashTable H, int* nbr) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * 27) return;
    const int b = t / 27, d = t % 27;
    const uint32_t k = blk_keys[b];
    const int bx = (int)compact10(k) - 1 + (d % 3) - 1, by = (int)compact10(k >> 1) - 1 + ((d / 3) % 3) - 1, bz = (int)compact10(k >> 2) - 1 + (d / 9) - 1;
    int r = -1;
    if (bx >= -1 && by >= -1 && bz >= -1 && bx < 1022 && by < 1022 && bz < 1022) r = hash_lookup(H, (uint64_t)block_key(bx, by, bz));
    nbr[t] = r;
}
__global__ void k_link_tile_nbrs(const int* __restrict__ nbr, const int4* __restrict__ origin, int n, int* tnbr) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int b = nbr[t];
    tnbr[t] = b < 0 ? -1 : origin[b].w;
}

// --------------------------------------------------------------------------------------- readback
// scatter a component back to caller order: out[id - id_base] = value(i); id_base < 0 keeps device order
SPK_DEV size_t out_slot(const ParticleArrays& P, int i, long long id_base) { return id_base < 0 ? (size_t)i : (size_t)((long long)particle_id(P, i) - id_base); }
__global__ void k_read_positions(ParticleArrays P, float* out, int n, long long id_base, BinnedView B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* o = out + 3 * out_slot(P, i, id_base);
    slot_world_pos(B, P.q[0][i], i, o);
}
__global__ void k_read_velocities(ParticleArrays P, float* out, int n, long long id_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q3 = P.q[3][i];
    float* o = out + 3 * out_slot(P, i, id_base);
    o[0] = q3.x; o[1] = q3.y; o[2] = q3.z;
}
__global__ void k_read_velgrad(ParticleArrays P, float* out, int n, long long id_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q3 = P.q[3][i], q4 = P.q[4][i], q5 = P.q[5][i];
    float* o = out + 9 * out_slot(P, i, id_base);
    o[0] = q3.w; o[1] = q4.x; o[2] = q4.y; o[3] = q4.z; o[4] = q4.w; o[5] = q5.x; o[6] = q5.y; o[7] = q5.z; o[8] = q5.w;
}
// `CudaParticleSet::from_particles`: the caller's seven repr(C) component arrays (raw bytes, copied to the device as they are)
// packed into the float4 streams.  status 24 B | position 12 | velocity 12 | volume 52 | phase 8 | data 16 | velocity gradient 36.
__global__ void k_pack_upload(ParticleArrays P, int first, int n, uint32_t id_base, const uint32_t* __restrict__ status, const float* __restrict__ pos,
                              const float* __restrict__ vel, const float* __restrict__ vol, const float* __restrict__ phase, const float* __restrict__ data,
                              const float* __restrict__ vg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = first + j;
    const float* V = vol + 13 * (size_t)j;       // mass, volume0, radius0, F[9], Jp
    const float* X = pos + 3 * (size_t)j;
    const float* U = vel + 3 * (size_t)j;
    const uint32_t* S = status + 6 * (size_t)j;
    const float* D = data + 4 * (size_t)j;       // nacc_alpha, plastic_hardening, elastic_hardening, log_vol_gain
    float C[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (vg) for (int e = 0; e < 9; ++e) C[e] = vg[9 * (size_t)j + e];
    // the deformation gradient is STORED as F - I (diagonal entries minus one, exact for 0.5 <= F_kk <= 2): the strain of a stiff
    // particle keeps its relative precision from one substep to the next (device_math.h, "strain-form arithmetic")
    P.q[0][i] = make_float4(X[0], X[1], X[2], V[3] - 1.f);
    P.q[1][i] = make_float4(V[4], V[5], V[6], V[7] - 1.f);
    P.q[2][i] = make_float4(V[8], V[9], V[10], V[11] - 1.f);
    P.q[3][i] = make_float4(U[0], U[1], U[2], C[0]);
    P.q[4][i] = make_float4(C[1], C[2], C[3], C[4]);
    P.q[5][i] = make_float4(C[5], C[6], C[7], C[8]);
    const float ph = phase[2 * (size_t)j], psi = phase[2 * (size_t)j + 1];
    P.d[i] = make_float4(D[1], D[3], V[12], psi);
    uint32_t flags = S[4] << FLAG_MODEL_SHIFT;
    if (S[0] & 0xFFu) flags |= FLAG_FAILED;
    if (S[0] & 0xFF00u) flags |= FLAG_STATIC;
    if (S[0] & 0xFF0000u) flags |= FLAG_KINEMATIC;
    if (ph != 1.0f) flags |= FLAG_PHASE_NOT_ONE;
    P.k[i] = make_float4(V[0], V[1], D[2], __uint_as_float(flags));
    P.kinvel[i] = make_float4(__uint_as_float(S[1]), __uint_as_float(S[2]), __uint_as_float(S[3]), 0.f);
    P.aux[i] = make_float4(ph, V[2], D[0], __uint_as_float(id_base + (uint32_t)i));
}
// the repr(C) component structs written on the device, one kernel per component (what = 0 volume 13 floats, 1 status 6 words,
// 2 data 4 floats), so that the host side is one memcpy into the caller's array
__global__ void k_read_struct(ParticleArrays P, uint32_t* out, int n, long long id_base, int what) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = out_slot(P, i, id_base);
    if (what == 0) {          // SparklParticleVolume: mass, volume0, radius0, F[9], plastic_deformation_gradient_det
        const float4 q0 = P.q[0][i], q1 = P.q[1][i], q2 = P.q[2][i], dd = P.d[i], kk = P.k[i], ax = P.aux[i];
        float* w = reinterpret_cast<float*>(out) + 13 * o;
        w[0] = kk.x; w[1] = kk.y; w[2] = ax.y;
        w[3] = 1.f + q0.w; w[4] = q1.x; w[5] = q1.y; w[6] = q1.z; w[7] = 1.f + q1.w; w[8] = q2.x; w[9] = q2.y; w[10] = q2.z; w[11] = 1.f + q2.w;   // stored as F - I
        w[12] = dd.z;
    } else if (what == 1) {   // SparklParticleStatus: failed, is_static, kinematic_vel_enabled, pad | kinematic_vel[3] | model_index u64
        const float4 kk = P.k[i], kv = P.kinvel[i];
        const uint32_t flags = __float_as_uint(kk.w);
        uint32_t* w = out + 6 * o;
        w[0] = ((flags & FLAG_FAILED) ? 1u : 0u) | ((flags & FLAG_STATIC) ? 0x100u : 0u) | ((flags & FLAG_KINEMATIC) ? 0x10000u : 0u);
        w[1] = __float_as_uint(kv.x); w[2] = __float_as_uint(kv.y); w[3] = __float_as_uint(kv.z);
        w[4] = flags >> FLAG_MODEL_SHIFT; w[5] = 0u;
    } else {                  // SparklParticleData: nacc_alpha, plastic_hardening, elastic_hardening, log_vol_gain
        const float4 dd = P.d[i], kk = P.k[i], ax = P.aux[i];
        float* w = reinterpret_cast<float*>(out) + 4 * o;
        w[0] = ax.z; w[1] = dd.x; w[2] = kk.z; w[3] = dd.y;
    }
}
// the three per-frame render components in one pass (any output may be null)
__global__ void k_read_render(ParticleArrays P, float* pos, float* vel, float* phase, int n, long long id_base, BinnedView B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = out_slot(P, i, id_base);
    if (pos) slot_world_pos(B, P.q[0][i], i, pos + 3 * o);
    if (vel) { const float4 q3 = P.q[3][i]; vel[3 * o] = q3.x; vel[3 * o + 1] = q3.y; vel[3 * o + 2] = q3.z; }
    if (phase) { phase[2 * o] = P.aux[i].x; phase[2 * o + 1] = P.d[i].w; }
}
// `CudaVec::remove_range` on every component at once (cuda_vec.rs:83-110): callers index particles by insertion order, so
// the ids in [start, end) go away (dropped by the re-bin that follows) and every later id moves down by end - start.
__global__ void k_remove_ids(ParticleArrays P, int n, uint32_t id_base, uint32_t start, uint32_t end) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* aux_w = reinterpret_cast<uint32_t*>(&P.aux[i]) + 3;
    const uint32_t id = *aux_w - id_base;
    if (id >= start && id < end) reinterpret_cast<uint32_t*>(&P.k[i])[3] |= FLAG_REMOVED;
    else if (id >= end) *aux_w = id_base + id - (end - start);
}
__global__ void k_read_ids(ParticleArrays P, uint32_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = particle_id(P, i);
}
// ParticleFracture in caller order <-> the device stream (what = 0: upload, 1: read back, 2: defaults for [first, first + n))
__global__ void k_fracture(ParticleArrays P, float2* host_order, int first, int n, long long id_base, int what) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = first + j;
    if (what == 2) { P.frac[i] = make_float4(0.f, 3.402823466e+38f, 0.f, 0.f); return; }
    const size_t o = out_slot(P, i, id_base);
    if (what == 0) { const float2 v = host_order[o]; P.frac[i] = make_float4(v.x, v.y, 0.f, 0.f); }
    else { const float4 v = P.frac[i]; host_order[o] = make_float2(v.x, v.y); }
}


// ------------------------------------------------------------------------------ slabs (multi-GPU)
// a rank that owns no particle block still has to take part in the step counting / abort chain
__global__ void k_bump_steps(Ctl* ctl, int parity) {
    if (ctl->abort_flag[parity]) ctl->abort_flag[(parity + 1) & 1] = 1u;
    else note_step(ctl);
}
// Migration record: the eleven float4 of one particle (176 B; the last is ParticleFracture or its default).
constexpr int MIG_F4 = 11;
__global__ void k_pack_particles(ParticleArrays P, int first, int count, float4* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int i = first + t;
    float4* o = out + (size_t)t * MIG_F4;
#pragma unroll
    for (int a = 0; a < 6; ++a) o[a] = P.q[a][i];
    o[6] = P.d[i]; o[7] = P.k[i]; o[8] = P.kinvel[i]; o[9] = P.aux[i];
    o[10] = P.frac ? P.frac[i] : make_float4(0.f, 3.402823466e+38f, 0.f, 0.f);
    reinterpret_cast<uint32_t*>(&P.k[i])[3] |= FLAG_REMOVED;
}
__global__ void k_unpack_particles(ParticleArrays P, int first, int count, const float4* in) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int i = first + t;
    const float4* o = in + (size_t)t * MIG_F4;
#pragma unroll
    for (int a = 0; a < 6; ++a) P.q[a][i] = o[a];
    P.d[i] = o[6]; P.k[i] = o[7]; P.kinvel[i] = o[8]; P.aux[i] = o[9];
    if (P.frac) P.frac[i] = o[10];
}
// first index of every tail class in the sorted key array: out[k] = first i with class(i) >= k,
// classes 0..3 = EMIG_LO, EMIG_HI, INVALID, REMOVED (out[] preset to n)
template <class KeyT>
__global__ void k_tail_bounds(const KeyT* __restrict__ keys, int n, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KeyT a = keys[i], b = i ? keys[i - 1] : (KeyT)0;
    const KeyT T[4] = {TailKeys<KeyT>::EMIG_LO, TailKeys<KeyT>::EMIG_HI, TailKeys<KeyT>::INVALID, TailKeys<KeyT>::REMOVED};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (a >= T[k] && (i == 0 || b < T[k])) atomicMin(&out[k], i);
}

// Halo record: the 64 nodes two slabs share in ONE column of blocks across a cut.  A particle sits at most one cell outside its home
// block (the tile margin), so the slab below the cut reaches the first three node planes of the block above it (nodes 4 cut .. 4 cut + 2)
// and the slab above reaches the last plane of the block below (node 4 cut - 1): 16 + 48 nodes, not the 2 x 64 of both blocks.
//   nodes[0..15]  = plane 3 of the LOWER block (axis coordinate 3; the other two coordinates u + 4 v)
//   nodes[16..63] = planes 0, 1, 2 of the UPPER block (16 each)
// `key` = block key of the UPPER block position (whether or not that block is active on the sender).  1040 B, 16-B aligned payload.
struct HaloRecord {
    unsigned long long key;
    unsigned long long pad;
    float4 nodes[64];
};
SPK_DEV int nbr_slot(int axis, int d) { return axis == 0 ? (d + 1) + 3 + 9 : (axis == 1 ? 1 + 3 * (d + 1) + 9 : 1 + 3 + 9 * (d + 1)); }
SPK_DEV int plane_node(int axis, int c, int u, int v) { return axis == 0 ? c + 4 * u + 16 * v : (axis == 1 ? u + 4 * c + 16 * v : u + 4 * v + 16 * c); }
// columns with an active block on either side of the cut.  Entry = (upper block << 1), or (lower block << 1) | 1 when the block above it
// is not active here (a column is listed once)
__global__ void k_halo_list(const int4* __restrict__ blk_origin, const int* __restrict__ blk_nbr, int nb, int axis, int cut, int* list, int* count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int4 o = blk_origin[b];
    const int bs = (axis == 0 ? o.x : (axis == 1 ? o.y : o.z)) >> 2;
    if (bs == cut) list[atomicAdd(count, 1)] = b << 1;
    else if (bs == cut - 1 && blk_nbr[b * 27 + nbr_slot(axis, +1)] < 0) list[atomicAdd(count, 1)] = (b << 1) | 1;
}
// flags for the interior / cut-adjacent split of a slab (see GridArrays::pb_list): a particle block whose G2P reads a node plane that
// receives halo (home plane within two of a cut) and a node block that receives halo itself (planes lo - 1, lo, hi - 1, hi)
__global__ void k_split_flags(const int4* __restrict__ pb_info, int npb, const int4* __restrict__ blk_origin, int nb, int axis, int lo, int hi, unsigned char* pb_flag,
                              unsigned char* nb_flag) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < npb) {
        const int4 o = pb_info[2 * j];
        const long long pl = (axis == 0 ? o.x : (axis == 1 ? o.y : o.z)) >> 2;
        pb_flag[j] = (pl < (long long)lo + 2 || pl >= (long long)hi - 2) ? 1 : 0;
    }
    if (j < nb) {
        const int4 o = blk_origin[j];
        const long long pl = (axis == 0 ? o.x : (axis == 1 ? o.y : o.z)) >> 2;
        nb_flag[j] = (pl == lo || pl == hi || pl + 1 == lo || pl + 1 == hi) ? 1 : 0;
    }
}
// the particle-block flags in the order of a launch list (fullest first): what DevicePartition::Flagged needs beside that list
__global__ void k_gather_flags(const uint32_t* __restrict__ order, const unsigned char* __restrict__ flag, int n, unsigned char* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = flag[order[i]];
}
// this slab's partial (mass, momentum) of node (cx, cy, cz) of block b, gathered from its own tiles.
// Three dependent round trips -- the eight tile-neighbour entries, their storage modes, the eight node records, each set issued back to back without a
// branch (a missing tile reads element 0 and is replaced by zeros: the sums are those of a guarded loop bit for bit) -- where the loop
// blk_nbr -> blk_origin.w -> tile_mode -> tile per neighbour made thirty-two, on the critical path of every substep of a slab group.
SPK_DEV float4 halo_node_sum(const GridArrays& G, int b, int cx, int cy, int cz) {
    const int dx0 = cx <= 2 ? -1 : 0, dy0 = cy <= 2 ? -1 : 0, dz0 = cz <= 2 ? -1 : 0;
    int t[8], mode[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = dx0 + (k & 1), dy = dy0 + ((k >> 1) & 1), dz = dz0 + (k >> 2);
        t[k] = G.blk_tnbr[b * 27 + (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) mode[k] = G.tile_mode[max(t[k], 0)];
    float4 v[8];
    bool ok[8], mass_tile[8], any_mass = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = dx0 + (k & 1), dy = dy0 + ((k >> 1) & 1), dz = dz0 + (k >> 2);
        const int tx = cx - 4 * dx + 1, ty = cy - 4 * dy + 1, tz = cz - 4 * dz + 1;
        mass_tile[k] = (t[k] >= 0) & (mode[k] == TILE_MODE_MASS);
        any_mass |= mass_tile[k];
        const int i = tile_box_index(mode[k], tx, ty, tz);
        ok[k] = (t[k] >= 0) & !mass_tile[k] & (i >= 0);
        v[k] = G.tiles[ok[k] ? (size_t)t[k] * TILE_N + i : (size_t)0];
    }
    if (__any(any_mass)) {   // tiles of a fluid mass pass: one float per node, all 8^3 of them
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dx = dx0 + (k & 1), dy = dy0 + ((k >> 1) & 1), dz = dz0 + (k >> 2);
            const int tx = cx - 4 * dx + 1, ty = cy - 4 * dy + 1, tz = cz - 4 * dz + 1;
            const float f = ((const float*)(G.tiles + (size_t)max(t[k], 0) * TILE_N))[tx + TILE_W * ty + TILE_W * TILE_W * tz];
            if (mass_tile[k]) { v[k] = make_float4(f, 0.f, 0.f, 0.f); ok[k] = true; }
        }
    }
    float m = 0.f, px = 0.f, py = 0.f, pz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { m += ok[k] ? v[k].x : 0.f; px += ok[k] ? v[k].y : 0.f; py += ok[k] ? v[k].z : 0.f; pz += ok[k] ? v[k].w : 0.f; }
    return make_float4(m, px, py, pz);
}
// one wave per listed column
__global__ __launch_bounds__(BLK_THREADS) void k_halo_pack(GridArrays G, const uint32_t* __restrict__ blk_keys, const int* __restrict__ list, int n, int axis,
                                                            HaloRecord* out, const Ctl* __restrict__ ctl, int parity) {
    if (ctl->abort_flag[parity]) return;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int e = list[r];
    const int lane = threadIdx.x & 63;
    int upper = (e & 1) ? -1 : (e >> 1), lower = (e & 1) ? (e >> 1) : G.blk_nbr[(e >> 1) * 27 + nbr_slot(axis, -1)];
    const bool lo_part = lane < 16;
    const int b = lo_part ? lower : upper;
    const int c = lo_part ? 3 : (lane - 16) >> 4, uv = lo_part ? lane : (lane - 16) & 15;
    const int node = plane_node(axis, c, uv & 3, uv >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b >= 0) v = halo_node_sum(G, b, node & 3, (node >> 2) & 3, node >> 4);
    out[r].nodes[lane] = v;
    if (lane == 0) {
        unsigned long long key;
        if (upper >= 0) key = (unsigned long long)blk_keys[upper];
        else {
            const int4 o = G.blk_origin[lower];
            int bc[3] = {o.x >> 2, o.y >> 2, o.z >> 2};
            bc[axis] += 1;
            key = (unsigned long long)block_key(bc[0], bc[1], bc[2]);
        }
        out[r].key = key; out[r].pad = 0ull;
    }
}
// add the neighbour's partial sums into halo_in (zeroed every substep); blocks unknown here are ignored: no local particle touches them
__global__ __launch_bounds__(BLK_THREADS) void k_halo_merge(const HaloRecord* __restrict__ in, int n, int axis, HashTable H, float4* halo_in,
                                                             const Ctl* __restrict__ ctl, int parity) {
    if (ctl->abort_flag[parity]) return;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    const uint32_t key = (uint32_t)in[r].key;
    const bool lo_part = lane < 16;
    int b;
    if (!lo_part) b = hash_lookup(H, (uint64_t)key);
    else {
        int bc[3] = {(int)compact10(key) - 1, (int)compact10(key >> 1) - 1, (int)compact10(key >> 2) - 1};
        bc[axis] -= 1;
        b = bc[axis] >= -1 ? hash_lookup(H, (uint64_t)block_key(bc[0], bc[1], bc[2])) : -1;
    }
    if (b < 0) return;
    const int c = lo_part ? 3 : (lane - 16) >> 4, uv = lo_part ? lane : (lane - 16) & 15;
    const int node = plane_node(axis, c, uv & 3, uv >> 2);
    float4 v = halo_in[(size_t)b * 64 + node];
    const float4 a = in[r].nodes[lane];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    halo_in[(size_t)b * 64 + node] = v;
}
// Both cuts of a slab in one launch each (sparkl_hip_group_substeps): waves [0, n0) serve the low cut, [n0, n0 + n1) the high one.
// The merge STORES: between two re-bins the same (block, node) pairs receive a record every substep, each from exactly one
// neighbour (the last node plane of the block below a cut, the first three of the block above it), and halo_in is zeroed by the
// re-bin -- no per-substep memset of the whole array (28 MB at 10 M particles).
struct HaloSides {
    const int* list[2];
    int n[2];
    HaloRecord* out[2];
    const HaloRecord* in[2];
    int nin[2];
};
// (psi_mass, psi_momentum) partial of node (cx, cy, cz) of block b from this slab's psi tiles (DamageModel::ModifiedEigenerosion)
SPK_DEV float4 halo_node_sum_psi(const GridArrays& G, const float2* __restrict__ psi_tiles, int b, int cx, int cy, int cz) {
    const int dx0 = cx <= 2 ? -1 : 0, dy0 = cy <= 2 ? -1 : 0, dz0 = cz <= 2 ? -1 : 0;
    int t[8];
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = dx0 + (k & 1), dy = dy0 + ((k >> 1) & 1), dz = dz0 + (k >> 2);
        t[k] = G.blk_tnbr[b * 27 + (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // (branch-free like halo_node_sum: two round trips instead of sixteen)
        const int dx = dx0 + (k & 1), dy = dy0 + ((k >> 1) & 1), dz = dz0 + (k >> 2);
        v[k] = psi_tiles[(size_t)max(t[k], 0) * TILE_N + (cx - 4 * dx + 1) + TILE_W * (cy - 4 * dy + 1) + TILE_W * TILE_W * (cz - 4 * dz + 1)];
    }
    float m = 0.f, pm = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { m += t[k] >= 0 ? v[k].x : 0.f; pm += t[k] >= 0 ? v[k].y : 0.f; }
    return make_float4(m, pm, 0.f, 0.f);
}
// `psi_tiles` != NULL: the records carry (psi_mass, psi_momentum, 0, 0) of the psi sweep instead of (mass, momentum)
__global__ __launch_bounds__(BLK_THREADS) void k_halo_pack2(GridArrays G, const uint32_t* __restrict__ blk_keys, HaloSides hs, int axis, const Ctl* __restrict__ ctl, int parity,
                                                             const float2* __restrict__ psi_tiles = nullptr) {
    if (ctl->abort_flag[parity]) return;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int side = r >= hs.n[0] ? 1 : 0;
    if (side) r -= hs.n[0];
    if (r >= hs.n[side]) return;
    const int e = hs.list[side][r];
    HaloRecord* out = hs.out[side];
    const int lane = threadIdx.x & 63;
    int upper = (e & 1) ? -1 : (e >> 1), lower = (e & 1) ? (e >> 1) : G.blk_nbr[(e >> 1) * 27 + nbr_slot(axis, -1)];
    const bool lo_part = lane < 16;
    const int b = lo_part ? lower : upper;
    const int c = lo_part ? 3 : (lane - 16) >> 4, uv = lo_part ? lane : (lane - 16) & 15;
    const int node = plane_node(axis, c, uv & 3, uv >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b >= 0) v = psi_tiles ? halo_node_sum_psi(G, psi_tiles, b, node & 3, (node >> 2) & 3, node >> 4) : halo_node_sum(G, b, node & 3, (node >> 2) & 3, node >> 4);
    out[r].nodes[lane] = v;
    if (lane == 0) {
        unsigned long long key;
        if (upper >= 0) key = (unsigned long long)blk_keys[upper];
        else {
            const int4 o = G.blk_origin[lower];
            int bc[3] = {o.x >> 2, o.y >> 2, o.z >> 2};
            bc[axis] += 1;
            key = (unsigned long long)block_key(bc[0], bc[1], bc[2]);
        }
        out[r].key = key; out[r].pad = 0ull;
    }
}
__global__ __launch_bounds__(BLK_THREADS) void k_halo_merge2(HaloSides hs, int axis, HashTable H, float4* halo_in, const Ctl* __restrict__ ctl, int parity) {
    if (ctl->abort_flag[parity]) return;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int side = r >= hs.nin[0] ? 1 : 0;
    if (side) r -= hs.nin[0];
    if (r >= hs.nin[side]) return;
    const HaloRecord* in = hs.in[side];
    const int lane = threadIdx.x & 63;
    const uint32_t key = (uint32_t)in[r].key;
    const bool lo_part = lane < 16;
    int b;
    if (!lo_part) b = hash_lookup(H, (uint64_t)key);
    else {
        int bc[3] = {(int)compact10(key) - 1, (int)compact10(key >> 1) - 1, (int)compact10(key >> 2) - 1};
        bc[axis] -= 1;
        b = bc[axis] >= -1 ? hash_lookup(H, (uint64_t)block_key(bc[0], bc[1], bc[2])) : -1;
    }
    if (b < 0) return;
    const int c = lo_part ? 3 : (lane - 16) >> 4, uv = lo_part ? lane : (lane - 16) & 15;
    halo_in[(size_t)b * 64 + plane_node(axis, c, uv & 3, uv >> 2)] = in[r].nodes[lane];
}
// record keys of the halo columns of one cut, in list order: what the neighbour resolves to its own node-block slots (k_halo_slots)
__global__ void k_halo_keys(GridArrays G, const uint32_t* __restrict__ blk_keys, const int* __restrict__ list, int n, int axis, uint32_t* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int e = list[r];
    if (!(e & 1)) { out[r] = blk_keys[e >> 1]; return; }
    const int4 o = G.blk_origin[e >> 1];
    int bc[3] = {o.x >> 2, o.y >> 2, o.z >> 2};
    bc[axis] += 1;
    out[r] = block_key(bc[0], bc[1], bc[2]);
}
// (.x, .y) = slot of the block of that key and of the block below it along the axis in THIS member's block list (what k_halo_merge2 looked up every substep)
__global__ void k_halo_slots(const uint32_t* __restrict__ keys, int n, int axis, HashTable H, int2* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t key = keys[r];
    int bc[3] = {(int)compact10(key) - 1, (int)compact10(key >> 1) - 1, (int)compact10(key >> 2) - 1};
    bc[axis] -= 1;
    out[r] = make_int2(hash_lookup(H, (uint64_t)key), bc[axis] >= -1 ? hash_lookup(H, (uint64_t)block_key(bc[0], bc[1], bc[2])) : -1);
}
// one wave per halo column, both cuts in one launch (waves [0, n0) serve the low cut); at least one workgroup even without columns: the mail
__global__ __launch_bounds__(BLK_THREADS) void k_halo_push(GridArrays G, DirectPush D, int axis, Ctl* ctl, int parity) {
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned int aborted = ctl->abort_flag[parity];   // this member's own flag for the substep: final, the particle kernel in front has finished
    if (blockIdx.x == 0 && tid < D.n_mail)
        __hip_atomic_store((global_u64p)D.mail[tid], ((unsigned long long)D.epoch << 32) | (aborted ? 1ull : 0ull), SPK_RLX_SYS);
    if (aborted) return;   // (every member's grid update will read the flag in the mail and skip the substep without waiting for records)
    int r = blockIdx.x * 4 + (tid >> 6);
    const int side = r >= D.n[0] ? 1 : 0;
    if (side) r -= D.n[0];
    if (r < D.n[side]) {
        const int e = D.list[side][r];
        const int2 dst = D.dst[side][r];
        int upper = (e & 1) ? -1 : (e >> 1), lower = (e & 1) ? (e >> 1) : G.blk_nbr[(e >> 1) * 27 + nbr_slot(axis, -1)];
        const bool lo_part = lane < 16;
        const int b = lo_part ? lower : upper;
        const int c = lo_part ? 3 : (lane - 16) >> 4, uv = lo_part ? lane : (lane - 16) & 15;
        const int node = plane_node(axis, c, uv & 3, uv >> 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b >= 0) v = halo_node_sum(G, b, node & 3, (node >> 2) & 3, node >> 4);
        const int slot = lo_part ? dst.y : dst.x;
        if (slot >= 0) halo_store_wt(&D.remote_halo[side][(size_t)slot * 64 + node], v);
    }
    // every record of this launch has landed (write-through stores, drained by every wave) before the neighbours are told so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned int old = __hip_atomic_fetch_add((global_u32p)D.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == gridDim.x) {
            __hip_atomic_store((global_u32p)D.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch is a later kernel
            for (int sd = 0; sd < 2; ++sd)
                if (D.remote_epoch[sd]) __hip_atomic_store((global_u32p)D.remote_epoch[sd], D.epoch, SPK_RLX_SYS);
        }
    }
}
// One wave: waits for the mail of all members, leaves the agreed flag in the control block and -- unless the substep is skipped -- waits for the
// neighbours' records too.  In front of k_bump_steps for a member without particle blocks, and in front of the grid update when members of
// different processes share a device (DirectWait::pre_agreed).
__global__ void k_direct_agree(DirectWait W, Ctl* ctl, int parity) {
    bool aborted = false;
    if (!direct_wait_mail(W, threadIdx.x & 63, aborted)) { if (threadIdx.x == 0) direct_give_up(ctl, parity, 1u); return; }
    if (aborted) { if (threadIdx.x == 0) raise_abort(ctl, parity); return; }
    if (!direct_wait_halo(W, threadIdx.x & 63)) { if (threadIdx.x == 0) direct_give_up(ctl, parity, 2u); }
}

// MAX of the members' abort flags of the upcoming substep, written back to all of them (members of one process on one stream)
struct CtlList { Ctl* c[16]; int n; };
__global__ void k_flags_max(CtlList L, int parity) {
    uint32_t m = 0u;
    for (int i = 0; i < L.n; ++i) m = max(m, L.c[i]->abort_flag[parity]);
    for (int i = 0; i < L.n; ++i) L.c[i]->abort_flag[parity] = m;
}
// extent of the particle blocks along the slab axis (block plane indices): out[0] = min, out[1] = max (preset to INT_MAX / INT_MIN)
__global__ void k_plane_range(const int4* __restrict__ pb_info, int npb, int axis, int* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npb) return;
    const int4 o = pb_info[2 * j];
    const int plane = (axis == 0 ? o.x : (axis == 1 ? o.y : o.z)) >> 2;
    atomicMin(&out[0], plane);
    atomicMax(&out[1], plane);
}
// particles per block plane along the slab axis (index plane + 1; the block keys carry 10 bits per axis with that bias): the histogram
// the slab cuts are re-chosen on
__global__ void k_plane_hist(const int4* __restrict__ pb_info, int npb, int axis, long long* hist, int nplanes) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npb) return;
    const int4 o = pb_info[2 * j], r = pb_info[2 * j + 1];
    const int plane = ((axis == 0 ? o.x : (axis == 1 ? o.y : o.z)) >> 2) + 1;
    if (plane >= 0 && plane < nplanes) atomicAdd((unsigned long long*)&hist[plane], (unsigned long long)r.y);
}
// ---- structure validator (cuda_sparse_grid.rs:492-650 `check_active_blocks`, the reference's debug check of the binning) ----
// counters: [0] particle ids out of range or duplicated, [1] node blocks the hash does not resolve to themselves,
//           [2] particles whose associated cell is outside the tile of the particle block that holds them (or outside the domain
//               without being flagged), [3] missing blocks in the 27-neighbourhood of a particle block, [4] dispatch coverage errors
//               (block ranges not contiguous / not disjoint)
__global__ void k_check_ids(ParticleArrays P, int n, uint32_t id_base, uint32_t* seen, unsigned long long* cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = particle_id(P, i) - id_base;
    if (id >= (uint32_t)n || atomicAdd(&seen[id], 1u) != 0u) atomicAdd(&cnt[0], 1ull);
}
__global__ void k_check_blocks(const uint32_t* __restrict__ blk_keys, int nb, HashTable H, unsigned long long* cnt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    if (hash_lookup(H, (uint64_t)blk_keys[b]) != b) atomicAdd(&cnt[1], 1ull);
    if (b > 0 && !(blk_keys[b - 1] < blk_keys[b])) atomicAdd(&cnt[1], 1ull);      // Morton-sorted, unique
}
__global__ void k_check_pblocks(GridArrays G, int n_valid, unsigned long long* cnt) {
    const int pb = blockIdx.x * blockDim.x + threadIdx.x;
    if (pb >= G.num_pblocks) return;
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int b = org.w;
    if (b < 0 || b >= G.num_blocks) { atomicAdd(&cnt[3], 27ull); return; }
    for (int k = 0; k < 27; ++k) if (G.blk_nbr[b * 27 + k] < 0) atomicAdd(&cnt[3], 1ull);
    const int4 o = G.blk_origin[b];
    if (o.x != org.x || o.y != org.y || o.z != org.z || o.w != pb) atomicAdd(&cnt[4], 1ull);   // block <-> tile cross links
    const int expect = pb == 0 ? 0 : G.pb_info[2 * (pb - 1) + 1].x + G.pb_info[2 * (pb - 1) + 1].y;
    if (rng.x != expect || rng.y <= 0) atomicAdd(&cnt[4], 1ull);
    if (pb == G.num_pblocks - 1 && rng.x + rng.y != n_valid) atomicAdd(&cnt[4], 1ull);
}
__global__ __launch_bounds__(BLK_THREADS) void k_check_particles(ParticleArrays P, GridArrays G, float h, int dom, unsigned long long* cnt) {
    const int pb = blockIdx.x;
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int orgv[3] = {org.x, org.y, org.z};
    for (int li = threadIdx.x; li < rng.y; li += BLK_THREADS) {
        const int i = rng.x + li;
        const float4 q0 = P.q[0][i];
        const uint32_t flags = reinterpret_cast<const uint32_t*>(&P.k[i])[3];
        if (flags & FLAG_INVALID) continue;
        const float u[3] = {q0.x, q0.y, q0.z};   // block-relative, in cells
        bool bad = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float r = rintf(u[a]);
            if (!(r >= 0.0f && r <= 5.0f)) { bad = true; continue; }       // one cell of drift around the block is what the tile covers
            const int rw = orgv[a] + (int)r;                                // round(x / h)
            if (!(rw >= 1 && rw < dom - 4)) bad = true;
        }
        if (bad) atomicAdd(&cnt[2], 1ull);
    }
}

// compact the nodes with mass != 0 for grid_read_nodes
__global__ void k_dump_nodes(GridArrays G, SparklGridNode* out, unsigned long long cap, unsigned long long* count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G.num_blocks * 64) return;
    const float4 v = G.node_vel[t];
    if (v.w == 0.f) return;
    const unsigned long long slot = atomicAdd(count, 1ull);
    if (slot >= cap) return;
    const int b = t >> 6, l = t & 63;
    const int4 org = G.blk_origin[b];
    SparklGridNode nd;
    nd.cell[0] = org.x + (l & 3); nd.cell[1] = org.y + ((l >> 2) & 3); nd.cell[2] = org.z + (l >> 4);
    nd.mass = v.w;
    nd.velocity[0] = v.x; nd.velocity[1] = v.y; nd.velocity[2] = v.z;
    const float4 mm = G.node_mom ? G.node_mom[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    nd.momentum[0] = mm.x; nd.momentum[1] = mm.y; nd.momentum[2] = mm.z;
    out[slot] = nd;
}

}  // namespace spk
