// libsparkl_hip.so -- host side of the C ABI declared in include/sparkl_hip.h.
// Owns all device memory, the stream, the sparse-grid structures and the re-bin logic.
// The product path is HIP only: there is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <array>
#include <chrono>
#include <thread>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "probes.h"
#include "host_transport.h"

using namespace spk;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
static int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess)                                                                               \
            return set_error(e__ == hipErrorOutOfMemory ? SPARKL_ERR_OUT_OF_MEMORY : SPARKL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)                  \
    do {                           \
        int r__ = (expr);          \
        if (r__ != SPARKL_OK) return r__; \
    } while (0)

// growable device buffer
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n, bool keep = false, hipStream_t st = nullptr) {
        if (n <= cap) return SPARKL_OK;
        size_t ncap = std::max(n, cap + cap / 2);
        T* np = nullptr;
        HIPCHK(hipMalloc((void**)&np, ncap * sizeof(T)));
        if (keep && p && cap) {
            HIPCHK(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = ncap;
        return SPARKL_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct ParticleStore {
    DevBuf<float4> q[6], d, k, kinvel, aux, frac;
    bool has_frac = false;   // ParticleFracture uploaded (sparkl_hip_particles_set_fracture): an eleventh stream travels with the particles
    int reserve(size_t n, bool keep = false, hipStream_t st = nullptr) {
        for (auto& b : q) CHK(b.reserve(n, keep, st));
        CHK(d.reserve(n, keep, st)); CHK(k.reserve(n, keep, st)); CHK(kinvel.reserve(n, keep, st)); CHK(aux.reserve(n, keep, st));
        if (has_frac) CHK(frac.reserve(n, keep, st));
        return SPARKL_OK;
    }
    ParticleArrays arrays() const {
        ParticleArrays a;
        for (int i = 0; i < 6; ++i) a.q[i] = q[i].p;
        a.d = d.p; a.k = k.p; a.kinvel = kinvel.p; a.aux = aux.p;
        a.frac = has_frac ? frac.p : nullptr;
        return a;
    }
    void release() { for (auto& b : q) b.release(); d.release(); k.release(); kinvel.release(); aux.release(); frac.release(); }
};

struct EventPair { hipEvent_t a, b; int kind; };

struct SparklPipeline {
    int device = 0;
    float h = 0.f;
    int dom = 256;
    uint32_t domain_log2 = 30;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // particles
    uint64_t n = 0;
    // slots [0, n_binned) belong to the particle blocks of the current binning and hold BLOCK-RELATIVE positions (kernels.h, cell_frac); the
    // tail [n_binned, n) -- fresh uploads, appended particles, immigrants, emigrants, particles outside the domain -- holds world coordinates
    uint64_t n_binned = 0;
    ParticleStore cur, alt;
    uint32_t max_model_index = 0;
    // every particle uploaded so far carries plastic_deformation_gradient_det == 1 (StepParams::jp_one) AND nothing that may change the value has
    // been stepped since: a model table that is not all plain fluid clears it for good (a later switch to plain fluids must read the device
    // values again) -- only a fresh upload sets it back
    bool all_jp_one = true;
    // slab decomposition (multi-GPU): owned home blocks along `slab_axis` are [slab_lo, slab_hi)
    bool slab_on = false;
    int slab_axis = 0, slab_lo = INT32_MIN, slab_hi = INT32_MAX;
    uint64_t id_offset = 0;
    int n_emig[2] = {0, 0}, emig_first[2] = {0, 0};   // emigrant ranges after the last re-bin (low, high)
    DevBuf<int> halo_list[2];
    int halo_count[2] = {0, 0};
    DevBuf<float4> halo_in;
    DevBuf<int> d_tail;
    StepParams pending{};          // split-phase substep state
    bool in_substep = false;
    bool pending_clock_saved = false;
    bool tiles_ready = false;      // the previous substep_finish_fused already scattered this substep's tiles
    // how the last split-phase substep ended (substep_finish_impl): decides whether the standalone P2G of substep_begin feeds the disorder
    // books.  A schedule that finishes fused scatters with k_g2p2g from then on and reaches the standalone P2G only behind a re-bin (an
    // incidental launch: stays out of the books); one that finishes plain (sparkl_hip_group_step: adaptive, one substep per call; slab.py's
    // step()) scatters with it EVERY substep and is the only measure the policy will ever see (it was never counted there, so
    // k_block_reorder and disorder re-bins never fired for adaptive multi-slab runs).  Survives re-bins on purpose.
    bool last_finish_fused = false;
    bool halo_store_mode = false;  // inside sparkl_hip_group_substeps: halo records are stored, not added (no per-substep memset)
    // interior / cut-adjacent split of a slab (built by the re-bin): split_pb = [cut-adjacent particle blocks in Morton order | interior
    // ones], split_nb likewise for node blocks; the slab-group loop runs the cut-adjacent part on `halo_stream` behind the exchange
    DevBuf<int> split_pb, split_nb;
    DevBuf<uint32_t> pb_order, order_keys, order_keys_out, order_vals;   // fullest-first launch order of the particle blocks (k_order_keys); SPARKL_HIP_BLOCK_ORDER=0: off
    bool pb_order_valid = false;
    DevBuf<unsigned char> split_flags;
    int pb_bnd = 0, nb_bnd = 0;
    bool split_ordered = false;   // split_pb was cut out of the fullest-first launch order (pb_order): [cut-adjacent, fullest first | interior, EMPTIEST first: read back to front]
    // Halo / compute overlap of the slab-group loop: -1 = automatic (on between RCCL ranks, where the exchange has a network latency to
    // hide; off for the members of one process on one device, where the cross-stream events cost more than they hide: 2 slabs of 10 M
    // on one MI355X 1.19 ms per substep in line, 1.23 overlapped), 0 / 1 = SPARKL_HIP_OVERLAP forces it
    int overlap_mode = -1;
    // automatic mode between RCCL ranks: the first two undisturbed 16-substep batches are timed, one per schedule, the MAX over the ranks of
    // each time is taken and the faster schedule kept on every rank (0 = nothing measured, 1 = in line measured, 2 = decided)
    int overlap_cal_state = 0, overlap_choice = -1;
    double overlap_cal_us[2] = {0.0, 0.0};   // per substep: [0] in line, [1] overlapped
    // ... and the peer-direct exchange is a THIRD candidate there, not a foregone conclusion: what a store into a neighbour's memory costs is a property of
    // the link between the two devices (one MI355X: nothing; xGMI: every record is a remote write), and the only honest way to know is to time it on the job's
    // own hardware.  Between ranks of a communicator the loop times one undisturbed 16-substep batch on the direct exchange (after one it lets pass: the first
    // batch of a job pays for cold caches and clocks), then the two staged schedules as before, takes the MAX over the ranks of each and keeps the fastest of
    // the three on every rank.  SPARKL_HIP_DIRECT = 1: direct without asking, 0: never, cal: calibrate even inside one process (tests).
    int direct_cal_state = 0;     // 0 = the batch to let pass, 1 = measuring next, 2 = measured
    int direct_choice = -1;       // -1 undecided, 0 = staged (overlap_choice says which schedule), 1 = direct
    double direct_cal_us = 0.0;
    hipStream_t halo_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_halo = nullptr, ev_flag = nullptr, ev_grid = nullptr, ev_xfer = nullptr;
    // Tile carry-over between fixed-dt batches and calls: the LAST substep of a fused batch also runs k_g2p2g (complete state stored), so
    // the tiles of substep `carry_index` for step length `carry_dt` are already scattered and the next batch skips its standalone
    // P2G + the plain G2P it would have ended with (0.35 + 0.34 -> 0.53 ms per batch boundary at 10M).  Anything that touches the
    // particles, the binning or the tiles in between clears it.  SPARKL_HIP_CARRY=0: off.
    bool carry_on = true, carry_valid = false;
    float carry_dt = 0.f;
    uint32_t carry_index = 0;

    // models / colliders
    DevBuf<DevModel> d_models;
    std::vector<DevModel> h_models;
    bool any_fluid = false;
    // slab group stepping (sparkl_hip_group_substeps): RCCL communicator and exchange buffers live in the handle
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    bool comm_aborted = false;   // ncclCommAbort was called on the communicator this handle held (RCCL failure path)
    DevBuf<uint8_t> x_send[2], x_recv[2], x_mig_send[2], x_mig_recv[2];
    uint64_t x_nsend[2] = {0, 0}, x_nrecv[2] = {0, 0};
    DevBuf<long long> x_dev;          // small device staging area for counts / flags
    long long* x_host = nullptr;     // pinned
    uint64_t group_rebins = 0, cut_moves = 0;
    // Peer-direct halo exchange (kernels.h, PeerBox): the arena peers write into, this member's view of theirs, and the epoch counter.
    struct Direct {
        bool ready = false;              // arena mapped everywhere, slots negotiated for the current binning (agreed by every rank)
        bool failed = false;             // a setup step failed on some rank: the job stays on the RCCL / copy exchange
        void* arena = nullptr;           // [PeerBox | halo buffer parity 0 | halo buffer parity 1], fine-grained device memory
        size_t arena_bytes = 0, cap_blocks = 0;
        uint64_t arena_serial = 0;       // bumped by every (re)allocation
        bool shared_device = false;      // members of different processes (or streams) sit on one physical device: no spinning grid update (DirectWait::pre_agreed)
        bool arena_fine = false;         // hipDeviceMallocFinegrained was granted (else plain device memory: enough inside one device only)
        int index = 0, n = 1;            // this member's mailbox slot and the number of members in the job
        uint32_t epoch = 0;              // group substeps enqueued so far (the same on every member of the job)
        uint32_t handshakes = 0;         // handshakes through the mapped arenas so far (group_direct_setup)
        DevBuf<unsigned int> arrive;     // arrival counter of k_halo_push's workgroups (ordinary device memory: an RMW on the fine-grained arena is a trip to memory)
        DevBuf<int2> dst[2];             // per halo column of each cut: the neighbour's node-block slots (k_halo_slots)
        void* peer_arena[MAX_DIRECT] = {nullptr};   // every member's arena in THIS process' address space ([index] = own)
        void* peer_ipc[MAX_DIRECT] = {nullptr};     // the ones opened with hipIpcOpenMemHandle (closed when replaced / at destroy)
        uint64_t peer_serial[MAX_DIRECT] = {0};
        uint64_t peer_pid[MAX_DIRECT] = {0};
        size_t peer_cap[MAX_DIRECT] = {0};          // cap_blocks of the peer's arena (where its second halo buffer starts)
        PeerBox* box() const { return reinterpret_cast<PeerBox*>(arena); }
    } direct;
    DirectWait dw{nullptr, 0, 0u, {0, 0}, 0ull, 0};   // handed to the grid update of the substep in flight (group loop, direct exchange), else box == NULL
    DevBuf<long long> bal_dev;       // plane histogram + cut list of the slab re-balancing (group_rebalance)
    long long* bal_host = nullptr; int bal_host_words = 0;
    bool slab_balance = true;        // sparkl_hip_slab_balance
    int heads_kind = -1;            // which scatter's measure the disorder books hold (note_heads_kind)
    bool all_plain_fluid = false;   // every model is a fluid without plastic / failure model: the FLUID_ONLY kernels apply
    bool all_snow = false;          // every model is a corotated solid with snow plasticity, no failure model: k_g2p2g<.., MSET = 5>
    bool all_neo_elastic = false;   // every model is a neo-Hookean solid without plasticity / failure: k_g2p2g<.., MSET = 3>
    // flags mirror (ParticleArrays::fl): built by every re-bin; ku_ok = every binned particle shares ku[] = (mass, volume0, elastic hardening) and nothing
    // has touched a flags word behind the mirror's back since (the fluid and psi sweeps do: they clear it until the next re-bin).  SPARKL_HIP_KUNIFORM=0: off
    DevBuf<uint32_t> fl;
    bool ku_on = true, ku_ok = false;
    float ku[3] = {0.f, 0.f, 0.f};
    bool all_dp_sand = false;       // every model is a corotated solid with (always active) Drucker-Prager plasticity: k_g2p2g<.., MSET = 1>
    bool mass_tiles_ready = false;  // k_g2p_mass of the previous substep already scattered the next volume sweep's masses
    bool fluid_bound_valid = false; // adaptive step with the volume sweep: the last dt bound seen (predicts the next step length)
    float fluid_bound = 0.f;
    uint64_t fluid_fused_substeps = 0, fluid_fused_mismatches = 0;
    DevBuf<DevCuboid> d_cuboids;
    int num_cuboids = 0;
    // host copies of the collider records (SparklColliderOptions patches them), what the options add up to, and the projection cache
    std::vector<DevCuboid> h_cuboids;
    std::vector<DevHeightField> h_heightfields;
    std::vector<DevTriMesh> h_trimeshes;
    std::vector<DevPolyline> h_polylines;
    DevBuf<DevPolyline> d_polylines;
    std::vector<float*> d_poly_verts;
    int num_polylines = 0;
    bool collider_handling = false, collider_penalty = false, rare_by_models = false;
    // The reference compiles its collider penalty force out (`if false`, src_kernels/cuda/particle_updater.rs:203-221): a penalty_stiffness is stored and
    // changes nothing.  Same here unless the caller opts in (sparkl_hip_colliders_enable_penalty_force, or SPARKL_HIP_PENALTY_FORCE=1 at create).
    bool penalty_force_on = false;
    DevBuf<float4> node_proj;        // GridArrays::node_proj: [heightfields + trimeshes][num_blocks][64]
    DevBuf<int> node_proj_blk;       // GridArrays::node_proj_blk: [heightfields + trimeshes][num_blocks]
    bool proj_cache_on = true;       // SPARKL_HIP_PROJ_CACHE=0: off (A/B, tests)
    bool proj_cache_clear = true;    // colliders changed or a re-bin rebuilt the node blocks: zero the cache before the next grid update
    DevBuf<DevHeightField> d_heightfields;
    std::vector<float*> d_heights;   // one device array per heightfield
    int num_heightfields = 0;

    // control block
    Ctl* d_ctl = nullptr;
    bool ctl_fresh = false;          // h_ctl->steps_done is the device's as of substep `ctl_fresh_index` (set where sparkl_hip_pipeline_substeps returns)
    uint32_t ctl_fresh_index = 0;
    Ctl* h_ctl = nullptr;  // pinned
    volatile uint32_t* h_words = nullptr;   // Ctl::host_words: [0] abort raised, [1] steps_done mirror (pinned, device-mapped)
    uint32_t enq_steps = 0;                 // steps_done the device will show once everything enqueued so far has run (valid while nothing aborts)
    // substeps a fixed-dt batch may be enqueued ahead of the GPU.  -1 (default) = by size: ~2 ms of work, i.e. 4 substeps of a 10 M scene and
    // 40 of a 1 M one (a small scene runs 0.07 ms substeps: four of them are gone before the host has enqueued the next); SPARKL_HIP_AHEAD=N
    // forces N, 0 = no limit (batches enqueued whole, as before round 5)
    int ahead_limit = -1;
    uint32_t substep_index = 0;  // parity source
    bool sorted = false;

    // sparse grid
    int num_blocks = 0, num_pblocks = 0;
    DevBuf<int4> blk_origin;
    DevBuf<int> blk_pfirst, blk_pcount, blk_nbr, blk_tnbr, pblk, tile_mode;
    DevBuf<int2> pb_heads;
    DevBuf<int4> pb_info;
    DevBuf<uint32_t> blk_keys;
    DevBuf<float4> tiles, node_vel, node_mom;
    bool capture_mom = false;
    DevBuf<uint64_t> hash_keys;
    DevBuf<uint32_t> hash_vals;
    uint32_t hash_mask = 0;

    // re-bin scratch
    DevBuf<uint64_t> keys_a, keys_b, pb_keys, bkeys;
    DevBuf<uint32_t> idx_a, idx_b, pb_counts, pb_first, cand_a, cand_b, oorg;
    DevBuf<uint8_t> cub_temp;
    DevBuf<int> d_scalars;  // [0] num_runs, [1] num_unique
    DevBuf<DevTriMesh> d_trimeshes;
    std::vector<void*> d_mesh_bufs;         // triangle / grid arrays of the uploaded meshes
    int num_trimeshes = 0;
    DevBuf<uint32_t> dt_partial;            // per-workgroup dt-bound minima (adaptive stepping)
    DevBuf<float2> psi_tiles;               // DamageModel::ModifiedEigenerosion: per-tile (psi_mass, psi_momentum) partials
    DevBuf<float> node_psi;                 // ... and psi_momentum / psi_mass per node
    DevBuf<float> staging;
    float reorder_limit = -1.f, rebin_limit = 0.f;   // sparkl_hip_reorder_policy_set: < 0 = default, 0 = off
    DevBuf<float> rb_staging;               // asynchronous render readback: 8 floats per particle, own copy stream
    hipStream_t rb_stream = nullptr;
    hipEvent_t rb_gathered = nullptr, rb_done = nullptr;
    bool rb_pending = false;
    DevBuf<SparklGridNode> node_staging;
    DevBuf<unsigned long long> d_count;

    // clock: per-kernel hipEvent spans on every clock_stride-th substep (events cost ~2 % when on every launch)
    uint32_t spec_prev_bound_bits = 0;   // adaptive step: speculative fusion bookkeeping (sparkl_hip_pipeline_step)
    uint64_t spec_hits = 0, spec_misses = 0;
    bool lookahead = true;               // adaptive step: enqueue the rest of the frame while the bound repeats (SPARKL_HIP_LOOKAHEAD=0: off)
    bool lookahead_force = false;        // test hook (SPARKL_HIP_LOOKAHEAD=force): plan ahead even when the bound has not repeated, so the mismatch path runs
    uint64_t lookahead_substeps = 0, lookahead_mismatches = 0;
    bool defer_c = true;           // fused fixed-dt batches skip the store of C.m[1..8] (StepParams::defer_c); SPARKL_HIP_DEFER_C=0: off
    bool c_stale = false;          // a batch was cut short behind such a launch: k_regather_c runs before the re-bin
    StepParams c_stale_S{};
    uint64_t c_regathers = 0;
    bool rare_models = false;      // some model needs the kernels that carry snow / Rankine / NACC / failure code
    // ... and whether the FUSED launches really are the generic RARE instantiation: not for a table the snow view covers (k_g2p2g<.., MSET = 5>: no particle
    // projection, no penalty kick, no failure test -- v and C are the plain gather, so the launch may defer them and its tiles may be carried over like the
    // Drucker-Prager and elastic ones)
    bool rare_fused() const { return rare_models && !(all_snow && !collider_penalty && h_models.size() <= (size_t)MAX_LDS_MODELS); }
    bool fuse_fixed_dt = true;     // SPARKL_HIP_NO_FUSE=1 in the environment turns the fused schedule off (A/B)
    bool clock_on = false;
    uint32_t clock_stride = 1;
    uint64_t clock_sampled = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { size_t a, b; int kind; uint32_t sub; };   // sub = substep_index when the span was opened (per-substep log of step())
    std::vector<Span> spans;
    // per-substep record of the last sparkl_hip_pipeline_step(.., timings != NULL) (CudaTimestepTimings::substeps, cuda_mpm_pipeline.rs:28-85)
    bool sublog_on = false;
    uint32_t sublog_base = 0;
    std::vector<SparklSubstepTimings> sublog;
    SparklKernelClock clock{};
    float span_ref_ms[16] = {0.f};       // duration of a working launch per span kind (see clock_collect)
    uint64_t total_resorts = 0;
};

static void comm_destroy(SparklPipeline* p);
static void direct_release(SparklPipeline* p);
enum { SPAN_P2G = 0, SPAN_GRID = 1, SPAN_G2P = 2, SPAN_SORT = 3, SPAN_DT = 4, SPAN_HALO = 5, SPAN_FUSED = 6, SPAN_REORDER = 7, SPAN_P2G_MASS = 8, SPAN_GRID_MASS = 9, SPAN_G2P_VOLUME = 10 };

static int clock_begin(SparklPipeline* p, int kind, size_t* tok, hipStream_t st = nullptr) {
    if (!p->clock_on) return SPARKL_OK;
    if (!st) st = p->stream;
    if (p->ev_used + 2 > p->ev_pool.size()) {
        size_t old = p->ev_pool.size();
        p->ev_pool.resize(old + 256);
        for (size_t i = old; i < p->ev_pool.size(); ++i) HIPCHK(hipEventCreate(&p->ev_pool[i]));
    }
    *tok = p->spans.size();
    p->spans.push_back({p->ev_used, p->ev_used + 1, kind, p->substep_index});
    HIPCHK(hipEventRecord(p->ev_pool[p->ev_used], st));
    p->ev_used += 2;
    return SPARKL_OK;
}
static int clock_end(SparklPipeline* p, size_t tok, hipStream_t st = nullptr) {
    if (!p->clock_on) return SPARKL_OK;
    HIPCHK(hipEventRecord(p->ev_pool[p->spans[tok].b], st ? st : p->stream));
    return SPARKL_OK;
}
// requires the stream to be idle
// A launch that found the abort flag raised (the substeps of a batch behind a re-bin request) returns at once: ~8 us against the
// 50-700 us of a working launch.  Averaged in, such spans would make the kernel look faster than it is (the deep collapse re-bins
// every ~100 substeps and leaves up to 127 of them per cut batch), so a span shorter than a quarter of the longest recent span of the
// same kernel is counted as a no-op launch (SparklKernelClock::noop_*), not as a launch of the kernel.
static int clock_collect(SparklPipeline* p) {
    std::vector<float> ms(p->spans.size(), 0.f);
    float* longest = p->span_ref_ms;   // decays slowly across collections: a chunk that lies wholly behind a re-bin request holds no-ops only
    for (int k = 0; k < 16; ++k) longest[k] *= 0.9f;
    for (size_t i = 0; i < p->spans.size(); ++i) {
        const auto& s = p->spans[i];
        HIPCHK(hipEventElapsedTime(&ms[i], p->ev_pool[s.a], p->ev_pool[s.b]));
        if (s.kind >= 0 && s.kind < 16) longest[s.kind] = std::max(longest[s.kind], ms[i]);
    }
    for (size_t i = 0; i < p->spans.size(); ++i) {
        const auto& s = p->spans[i];
        const float t = ms[i];
        const bool kernel_span = s.kind == SPAN_P2G || s.kind == SPAN_GRID || s.kind == SPAN_G2P || s.kind == SPAN_FUSED || s.kind == SPAN_P2G_MASS || s.kind == SPAN_GRID_MASS ||
                                 s.kind == SPAN_G2P_VOLUME;
        if (kernel_span && t < 0.25f * longest[s.kind]) { p->clock.noop_ms += t; p->clock.noop_launches++; continue; }
        if (p->sublog_on && s.sub - p->sublog_base < 65536u) {
            // A planned substep that did not run leaves no-op spans (dropped above) and its index is reused by the substep that does run, so
            // summing by index gives every executed substep exactly its own launches; a re-bin or an estimator pass carries the index of the
            // substep it precedes ("at the start of the substep", like the reference's grid_resize_and_sort / estimate_timestep).
            const uint32_t k = s.sub - p->sublog_base;
            if (k >= p->sublog.size()) p->sublog.resize(k + 1, SparklSubstepTimings{});
            SparklSubstepTimings& L = p->sublog[k];
            switch (s.kind) {
                case SPAN_P2G: case SPAN_P2G_MASS: L.p2g_ms += t; break;
                case SPAN_GRID: case SPAN_GRID_MASS: L.grid_update_ms += t; break;
                case SPAN_G2P: case SPAN_FUSED: case SPAN_G2P_VOLUME: L.g2p_ms += t; break;
                case SPAN_SORT: case SPAN_REORDER: L.grid_resize_and_sort_ms += t; break;
                case SPAN_DT: L.estimate_timestep_ms += t; break;
                default: L.halo_ms += t; break;
            }
            L.total_ms += t;
        }
        switch (s.kind) {
            case SPAN_P2G: p->clock.p2g_ms += t; p->clock.p2g_launches++; break;
            case SPAN_GRID: p->clock.grid_update_ms += t; p->clock.grid_update_launches++; break;
            case SPAN_G2P: p->clock.g2p_ms += t; p->clock.g2p_launches++; break;
            case SPAN_FUSED: p->clock.g2p2g_ms += t; p->clock.g2p2g_launches++; break;
            case SPAN_SORT: p->clock.sort_ms += t; break;
            case SPAN_REORDER: p->clock.reorder_ms += t; p->clock.reorder_launches++; break;
            case SPAN_DT: p->clock.dt_ms += t; break;
            case SPAN_P2G_MASS: p->clock.p2g_mass_ms += t; p->clock.p2g_mass_launches++; break;
            case SPAN_GRID_MASS: p->clock.grid_update_mass_ms += t; p->clock.grid_update_mass_launches++; break;
            case SPAN_G2P_VOLUME: p->clock.g2p_volume_ms += t; p->clock.g2p_volume_launches++; break;
            default: p->clock.halo_ms += t; break;
        }
    }
    p->spans.clear();
    p->ev_used = 0;
    return SPARKL_OK;
}

static inline int grid_for(size_t n, int threads) { return (int)((n + threads - 1) / threads); }

// Default reorder budget (see Ctl): the excess run heads per particle, summed over the substeps since the last reorder, that make
// k_block_reorder act.  0.6 while every reorder moved every block (~0.6 ms at 10M, profiles/r01_reorder_policy.txt); since the
// kernel skips the blocks whose runs have not fragmented (GridArrays::pb_heads) a pass costs what the flowing part of the scene
// costs and 0.2 is the optimum of a threshold scan (profiles/r03_kernel_experiments.txt section 5) (late / deep phase of the 10M collapse: 0.683 / 0.815 ms per
// substep at 0.6, 0.673 / 0.797 at 0.2, 0.676 / 0.799 at 0.1).  Since the run-task scatter groups an unsorted wave's lanes by cell its
// heads are the DISTINCT CELLS of a wave, which grow far slower, and the scan is flat from 0.05 to 2 (profiles/r03_kernel_experiments.txt
// section 9).  Scenes with a fluid run the DPP scatters (k_vol_p2g, the mass scatter), whose heads are lane-group heads and whose
// reorder moves every block: 0.6 there (20M dam in flow: 1.27 ms per substep at 0.2, 1.245 at 0.6 and at 2, 1.6 without the reorder:
// profiles/r03b_soak.txt).
static constexpr float DEFAULT_REORDER_HEADS_PER_PARTICLE = 0.2f;
static constexpr float DEFAULT_REORDER_HEADS_PER_PARTICLE_FLUID = 0.6f;
#ifndef SPK_REORDER_PERIOD
#define SPK_REORDER_PERIOD 16
#endif
static constexpr uint32_t REORDER_PERIOD = SPK_REORDER_PERIOD;   // substeps between (mostly no-op) k_block_reorder launches

static GridArrays grid_arrays(const SparklPipeline* p) {
    GridArrays g;
    g.blk_origin = p->blk_origin.p; g.blk_pfirst = p->blk_pfirst.p; g.blk_pcount = p->blk_pcount.p; g.blk_nbr = p->blk_nbr.p; g.blk_tnbr = p->blk_tnbr.p; g.dt_partial = p->dt_partial.p;
    g.cuboids = p->d_cuboids.p; g.heightfields = p->d_heightfields.p; g.trimeshes = p->d_trimeshes.p;
    g.polylines = p->d_polylines.p; g.num_polylines = p->num_polylines;
    g.node_proj = nullptr; g.node_proj_blk = nullptr;   // only launch_grid_update hands the cache out (it also sizes and clears it, in stream order)
    g.pblk = p->pblk.p; g.pb_info = p->pb_info.p; g.tile_mode = p->tile_mode.p; g.pb_heads = p->pb_heads.p; g.tiles = p->tiles.p; g.node_vel = p->node_vel.p; g.node_mom = p->capture_mom ? p->node_mom.p : nullptr;
    g.halo_in = (p->slab_on && p->num_blocks) ? p->halo_in.p : nullptr;
    g.halo_axis = p->slab_axis; g.halo_lo = p->slab_lo; g.halo_hi = p->slab_hi;
    g.pb_list = nullptr; g.pb_count = 0; g.nb_list = nullptr; g.nb_count = 0; g.primary = 1; g.pb_plain = 0;
    if (p->pb_order_valid && p->num_pblocks > 0) { g.pb_list = reinterpret_cast<const int*>(p->pb_order.p); g.pb_count = p->num_pblocks; g.pb_plain = 1; }
    g.num_blocks = p->num_blocks; g.num_pblocks = p->num_pblocks;
    return g;
}

static int cub_reserve(SparklPipeline* p, size_t bytes) { return p->cub_temp.reserve(std::max<size_t>(bytes, 1)); }

// --------------------------------------------------------------------------------- re-bin (resort)
// Sort particles by (block Morton key, cell in block), move the data, rebuild the node-block list,
// its hash and neighbour tables.  Host-orchestrated; runs only when a particle left its tile margin.
static float effective_reorder_limit(const SparklPipeline* p) {
    return p->reorder_limit < 0.f ? (p->any_fluid ? DEFAULT_REORDER_HEADS_PER_PARTICLE_FLUID : DEFAULT_REORDER_HEADS_PER_PARTICLE) * (float)p->n : p->reorder_limit;
}
static int upload_policy(SparklPipeline* p) {
    const float lim[2] = {effective_reorder_limit(p), p->rebin_limit < 0.f ? 0.f : p->rebin_limit};
    uint32_t bits[2];
    memcpy(bits, lim, 8);
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->reorder_limit, (int)bits[0], 1, p->stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->excess_limit, (int)bits[1], 1, p->stream));
    return SPARKL_OK;
}
// every REORDER_PERIOD substeps: the in-place block reorder, which returns at once unless the policy asked for it
static ParticleArrays with_mirror(const SparklPipeline* p, const ParticleArrays& P0);
static int maybe_reorder(SparklPipeline* p, int parity) {
    if (p->num_pblocks == 0 || (p->substep_index % REORDER_PERIOD) != 0 || !(effective_reorder_limit(p) > 0.f)) return SPARKL_OK;
    size_t tok = 0;
    CHK(clock_begin(p, SPAN_REORDER, &tok));
    hipLaunchKernelGGL(k_block_reorder, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, with_mirror(p, p->cur.arrays()), grid_arrays(p), p->d_ctl, p->h, parity);
    hipLaunchKernelGGL(k_reorder_done, dim3(1), dim3(1), 0, p->stream, p->d_ctl, parity);
    CHK(clock_end(p, tok));
    return SPARKL_OK;
}

static bool may_enqueue_next(SparklPipeline* p);
static int resort(SparklPipeline* p) {
    hipStream_t st = p->stream;
    p->carry_valid = false;   // the tiles belong to the old binning
    if (p->h_words) {   // the host-visible abort word is cleared once nothing that could still raise it is in flight (the callers re-bin behind a read of the control block: the stream is idle)
        if (hipStreamQuery(st) != hipSuccess) HIPCHK(hipStreamSynchronize(st));
        p->h_words[0] = 0u;
    }
    const int n = (int)p->n;
    size_t tok = 0;
    CHK(clock_begin(p, SPAN_SORT, &tok));
    if (p->c_stale) {
        // on the OLD binning: the node velocities and block ranges of the last substep that ran
        if (p->num_pblocks > 0) hipLaunchKernelGGL(k_regather_c, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, st, p->cur.arrays(), grid_arrays(p), p->c_stale_S);
        p->c_stale = false;
        p->c_regathers++;
    }
    const int old_npb = p->num_pblocks, old_binned = (int)std::min<uint64_t>(p->n_binned, p->n);
    p->num_blocks = p->num_pblocks = 0;
    p->pb_order_valid = false;
    p->ku_ok = false;
    if (n > 0) {
        CHK(p->keys_a.reserve(n)); CHK(p->keys_b.reserve(n)); CHK(p->idx_a.reserve(n)); CHK(p->idx_b.reserve(n)); CHK(p->oorg.reserve(n));
        CHK(p->bkeys.reserve(n)); CHK(p->pb_keys.reserve(n)); CHK(p->pb_counts.reserve(n)); CHK(p->pb_first.reserve(n));
        CHK(p->d_scalars.reserve(8));
        p->alt.has_frac = p->cur.has_frac;
        CHK(p->alt.reserve(p->cur.q[0].cap));
        CHK(p->d_tail.reserve(8));
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->dead_dt_bits, 0x7f7fffff, 1, st));   // rebuilt from the particles that are outside the domain now
        // key bits actually in use: Morton code of the block coordinate biased by +1 (3 x block_bits) + 6 cell bits; the tail
        // classes (emigrants, out of domain, removed) are all-ones words, so ONE more bit keeps them behind every valid key and
        // their two low bits order them among themselves.  dom = 256 (the default 2^30-byte grid): 28 bits = 4 radix passes
        // instead of the 5 that the worst case (37 bits) needs -- and 32-bit keys: 8 instead of 12 bytes per pair and pass.
        int block_bits = 1;
        while ((1 << block_bits) < p->dom / 4 + 2) ++block_bits;
        const int end_bit = std::min(37, 3 * block_bits + 6 + 1);
        auto sort_and_move = [&](auto key_tag) -> int {
            using KeyT = decltype(key_tag);
            KeyT* ka = reinterpret_cast<KeyT*>(p->keys_a.p);
            KeyT* kb = reinterpret_cast<KeyT*>(p->keys_b.p);
            const int slo = p->slab_on ? p->slab_lo : INT32_MIN, shi = p->slab_on ? p->slab_hi : INT32_MAX;
            // the particles of the old binning (block-relative positions), one workgroup per old particle block, then the tail (world coordinates)
            if (old_binned > 0 && old_npb > 0)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compute_keys_binned<KeyT>), dim3(old_npb), dim3(BLK_THREADS), 0, st, p->cur.arrays(), (const int4*)p->pb_info.p, ka, p->idx_a.p,
                                   p->oorg.p, p->h, p->dom, p->slab_axis, slo, shi, p->d_ctl);
            if (n > old_binned)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compute_keys_world<KeyT>), dim3(grid_for(n - old_binned, 256)), dim3(256), 0, st, p->cur.arrays(), ka, p->idx_a.p, p->oorg.p,
                                   old_binned, n, p->h, p->dom, p->slab_axis, slo, shi, p->d_ctl);
            size_t tb = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ka, kb, p->idx_a.p, p->idx_b.p, n, 0, end_bit, st));
            CHK(cub_reserve(p, tb));
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(p->cub_temp.p, tb, ka, kb, p->idx_a.p, p->idx_b.p, n, 0, end_bit, st));
            // (d_tail[4]: "some binned particle's (mass, volume0, hardening) differs from the first one's", see ParticleArrays::fl)
            if (p->ku_on) { CHK(p->fl.reserve(p->cur.q[0].cap)); HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(p->d_tail.p + 4), 0, 1, st)); }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_permute<KeyT>), dim3(grid_for(n, 256)), dim3(256), 0, st, p->cur.arrays(), p->alt.arrays(), p->idx_b.p, (const KeyT*)kb,
                               (const uint32_t*)p->oorg.p, n, p->h, p->ku_on ? p->fl.p : nullptr, reinterpret_cast<uint32_t*>(p->d_tail.p + 4));
            // block keys of the sorted particles (run-length encoded below) and the tail classes
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shift_keys<KeyT>), dim3(grid_for(n, 256)), dim3(256), 0, st, (const KeyT*)kb, p->bkeys.p, n);
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)p->d_tail.p, n, 4, st));   // no host staging buffer to outlive
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tail_bounds<KeyT>), dim3(grid_for(n, 256)), dim3(256), 0, st, (const KeyT*)kb, n, p->d_tail.p);
            return SPARKL_OK;
        };
        if (end_bit <= 32) CHK(sort_and_move(uint32_t{})); else CHK(sort_and_move(uint64_t{}));
        std::swap(p->cur, p->alt);
        size_t tb2 = 0;
        HIPCHK(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb2, p->bkeys.p, p->pb_keys.p, p->pb_counts.p, p->d_scalars.p, n, st));
        CHK(cub_reserve(p, tb2));
        HIPCHK(hipcub::DeviceRunLengthEncode::Encode(p->cub_temp.p, tb2, p->bkeys.p, p->pb_keys.p, p->pb_counts.p, p->d_scalars.p, n, st));
        // tail classes (removed | emigrants low | emigrants high | out of domain) sit after the valid keys
        int h_runs = 0, tail[4] = {n, n, n, n}, k_differs = 1;
        float4 k_first = make_float4(0.f, 0.f, 0.f, 0.f);
        HIPCHK(hipMemcpyAsync(&h_runs, p->d_scalars.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(tail, p->d_tail.p, sizeof(tail), hipMemcpyDeviceToHost, st));
        if (p->ku_on) {
            HIPCHK(hipMemcpyAsync(&k_differs, p->d_tail.p + 4, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&k_first, p->cur.k.p, sizeof(float4), hipMemcpyDeviceToHost, st));   // slot 0 of the new order
        }
        HIPCHK(hipStreamSynchronize(st));
        // tail[k] = first index with class >= k; a missing class keeps the bound of the next one
        for (int k = 2; k >= 0; --k) tail[k] = std::min(tail[k], tail[k + 1]);
        // every tail class shares one block key (all ones >> 6): if any such record exists, the last run is theirs, not a block
        if (h_runs > 0 && tail[0] < n) --h_runs;
        p->emig_first[0] = tail[0]; p->n_emig[0] = tail[1] - tail[0];
        p->emig_first[1] = tail[1]; p->n_emig[1] = tail[2] - tail[1];
        p->n = (uint64_t)tail[3];   // particles already handed to a neighbour slab are dropped here
        p->n_binned = (uint64_t)tail[0];   // k_permute left block-relative positions in front of the tail classes, world coordinates behind
        if (p->ku_on && tail[0] > 0 && k_differs == 0) { p->ku_ok = true; p->ku[0] = k_first.x; p->ku[1] = k_first.y; p->ku[2] = k_first.z; }
        const int npb = h_runs;
        if (npb > 0) {
            size_t tb3 = 0;
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb3, p->pb_counts.p, p->pb_first.p, npb, st));
            CHK(cub_reserve(p, tb3));
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(p->cub_temp.p, tb3, p->pb_counts.p, p->pb_first.p, npb, st));
            // candidate node blocks: the 27-neighbourhood of every particle block
            const int nc = npb * 27;
            CHK(p->cand_a.reserve(nc)); CHK(p->cand_b.reserve(nc)); CHK(p->blk_keys.reserve(nc));
            hipLaunchKernelGGL(k_candidates, dim3(grid_for(nc, 256)), dim3(256), 0, st, p->pb_keys.p, npb, p->cand_a.p);
            size_t tb4 = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb4, p->cand_a.p, p->cand_b.p, nc, 0, 30, st));
            CHK(cub_reserve(p, tb4));
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(p->cub_temp.p, tb4, p->cand_a.p, p->cand_b.p, nc, 0, 30, st));
            size_t tb5 = 0;
            HIPCHK(hipcub::DeviceSelect::Unique(nullptr, tb5, p->cand_b.p, p->blk_keys.p, p->d_scalars.p + 1, nc, st));
            CHK(cub_reserve(p, tb5));
            HIPCHK(hipcub::DeviceSelect::Unique(p->cub_temp.p, tb5, p->cand_b.p, p->blk_keys.p, p->d_scalars.p + 1, nc, st));
            int nb = 0;
            HIPCHK(hipMemcpyAsync(&nb, p->d_scalars.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            // node-block arrays + hash (<= 50 % load, as cuda_sparse_grid.rs:219-222)
            uint32_t hcap = 1024;
            while (hcap < (uint32_t)nb * 2u) hcap <<= 1;
            CHK(p->hash_keys.reserve(hcap)); CHK(p->hash_vals.reserve(hcap));
            p->hash_mask = hcap - 1;
            HIPCHK(hipMemsetAsync(p->hash_keys.p, 0xFF, (size_t)hcap * sizeof(uint64_t), st));
            CHK(p->blk_origin.reserve(nb)); CHK(p->blk_pfirst.reserve(nb)); CHK(p->blk_pcount.reserve(nb));
            CHK(p->blk_nbr.reserve((size_t)nb * 27)); CHK(p->blk_tnbr.reserve((size_t)nb * 27)); CHK(p->pblk.reserve(npb)); CHK(p->pb_info.reserve((size_t)npb * 2)); CHK(p->tile_mode.reserve(npb));
            CHK(p->pb_heads.reserve(npb));
            HIPCHK(hipMemsetAsync(p->pb_heads.p, 0, (size_t)npb * sizeof(int2), st));   // new binning: every block is in order, baselines not taken yet
            p->heads_kind = -1;
            CHK(p->tiles.reserve((size_t)npb * TILE_N)); CHK(p->node_vel.reserve((size_t)nb * 64));
            if (p->capture_mom) CHK(p->node_mom.reserve((size_t)nb * 64));
            HashTable H{p->hash_keys.p, p->hash_vals.p, p->hash_mask};
            hipLaunchKernelGGL(k_setup_blocks, dim3(grid_for(nb, 256)), dim3(256), 0, st, p->blk_keys.p, nb, p->blk_origin.p, p->blk_pfirst.p, p->blk_pcount.p, H);
            hipLaunchKernelGGL(k_link_pblocks, dim3(grid_for(npb, 256)), dim3(256), 0, st, p->pb_keys.p, p->pb_counts.p, p->pb_first.p, npb, H,
                               p->blk_origin.p, p->blk_pfirst.p, p->blk_pcount.p, p->pblk.p, p->pb_info.p);
            hipLaunchKernelGGL(k_link_nbrs, dim3(grid_for((size_t)nb * 27, 256)), dim3(256), 0, st, p->blk_keys.p, nb, H, p->blk_nbr.p);
            hipLaunchKernelGGL(k_link_tile_nbrs, dim3(grid_for((size_t)nb * 27, 256)), dim3(256), 0, st, p->blk_nbr.p, p->blk_origin.p, nb * 27, p->blk_tnbr.p);
            HIPCHK(hipMemsetAsync(p->node_vel.p, 0, (size_t)nb * 64 * sizeof(float4), st));
            p->num_blocks = nb;
            p->num_pblocks = npb;
            // launch order: fullest blocks first (kernels.h, k_order_keys)
            p->pb_order_valid = false;
            {
                const char* oe = getenv("SPARKL_HIP_BLOCK_ORDER");   // "0": off, "1": on whatever the size (tests), unset: by size
                const bool order_on = !(oe && !strcmp(oe, "0")), order_forced = oe && !strcmp(oe, "1");
                // (only for scenes that do not fit the 256 MB last-level cache anyway -- ~180 B per particle --: the 1 M cube, which does, loses 3 % without its
                // Morton launch order)
                if (order_on && (order_forced || p->n >= 2000000ull)) {
                    CHK(p->pb_order.reserve(npb)); CHK(p->order_keys.reserve(npb)); CHK(p->order_keys_out.reserve(npb)); CHK(p->order_vals.reserve(npb));
                    hipLaunchKernelGGL(k_order_keys, dim3(grid_for(npb, 256)), dim3(256), 0, st, p->pb_counts.p, npb, p->order_keys.p, p->order_vals.p);
                    size_t tbo = 0;
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tbo, p->order_keys.p, p->order_keys_out.p, p->order_vals.p, p->pb_order.p, npb, 0, 24, st));
                    CHK(cub_reserve(p, tbo));
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(p->cub_temp.p, tbo, p->order_keys.p, p->order_keys_out.p, p->order_vals.p, p->pb_order.p, npb, 0, 24, st));
                    p->pb_order_valid = true;
                }
            }
            // the G2P kernels store one dt-bound minimum per workgroup (= particle block): keep the buffer large enough for
            // whatever block count this re-bin produced (a re-bin inside step() used to leave it at the old size)
            CHK(p->dt_partial.reserve(std::max<size_t>(grid_for(n, BLK_THREADS), (size_t)npb)));
            p->halo_count[0] = p->halo_count[1] = 0;
            if (p->slab_on) {
                CHK(p->halo_in.reserve((size_t)nb * 64));
                HIPCHK(hipMemsetAsync(p->halo_in.p, 0, (size_t)nb * 64 * sizeof(float4), st));
                for (int side = 0; side < 2; ++side) {
                    const int cut = side == 0 ? p->slab_lo : p->slab_hi;
                    if (cut == INT32_MIN || cut == INT32_MAX) continue;
                    CHK(p->halo_list[side].reserve(nb));
                    HIPCHK(hipMemsetAsync(p->d_scalars.p + 2 + side, 0, sizeof(int), st));
                    hipLaunchKernelGGL(k_halo_list, dim3(grid_for(nb, 256)), dim3(256), 0, st, p->blk_origin.p, p->blk_nbr.p, nb, p->slab_axis, cut, p->halo_list[side].p, p->d_scalars.p + 2 + side);
                    HIPCHK(hipMemcpyAsync(&p->halo_count[side], p->d_scalars.p + 2 + side, sizeof(int), hipMemcpyDeviceToHost, st));
                }
                // interior / cut-adjacent split: the cut-adjacent part keeps the Morton order, the interior part comes out in REVERSE Morton order
                // (DevicePartition::Flagged writes the rejected items back to front) -- still one contiguous run of space per XCD chunk, which is
                // all the XCD mapping needs (speed only)
                {
                    const int m = std::max(nb, npb);
                    CHK(p->split_flags.reserve((size_t)3 * m));
                    CHK(p->split_pb.reserve(npb)); CHK(p->split_nb.reserve(nb));
                    hipLaunchKernelGGL(k_split_flags, dim3(grid_for(m, 256)), dim3(256), 0, st, p->pb_info.p, npb, p->blk_origin.p, nb, p->slab_axis, p->slab_lo, p->slab_hi,
                                       p->split_flags.p, p->split_flags.p + m);
                    hipcub::CountingInputIterator<int> iota(0);
                    size_t tb = 0;
                    HIPCHK(hipcub::DevicePartition::Flagged(nullptr, tb, iota, p->split_flags.p, p->split_pb.p, p->d_scalars.p + 6, m, st));
                    CHK(cub_reserve(p, tb));
                    // Particle blocks: the subsets keep the FULLEST-FIRST launch order of the whole list where there is one (k_order_keys: a grid of
                    // unequal workgroups ends with whatever comes last; -1.6 ... -6.9 % on one GPU, round 4 experiment 14 -- round 4 left the slab
                    // subsets in Morton order).  The rejected (interior) items come out back to front: that subset is read backwards (pb_plain = 2).
                    p->split_ordered = p->pb_order_valid;
                    if (p->split_ordered) {
                        CHK(p->split_flags.reserve((size_t)3 * m));
                        unsigned char* ford = p->split_flags.p + 2 * m;
                        hipLaunchKernelGGL(k_gather_flags, dim3(grid_for(npb, 256)), dim3(256), 0, st, (const uint32_t*)p->pb_order.p, (const unsigned char*)p->split_flags.p, npb, ford);
                        size_t tbo = 0;
                        HIPCHK(hipcub::DevicePartition::Flagged(nullptr, tbo, reinterpret_cast<const int*>(p->pb_order.p), ford, p->split_pb.p, p->d_scalars.p + 6, npb, st));
                        CHK(cub_reserve(p, std::max(tb, tbo)));
                        HIPCHK(hipcub::DevicePartition::Flagged(p->cub_temp.p, tbo, reinterpret_cast<const int*>(p->pb_order.p), ford, p->split_pb.p, p->d_scalars.p + 6, npb, st));
                    } else
                    HIPCHK(hipcub::DevicePartition::Flagged(p->cub_temp.p, tb, iota, p->split_flags.p, p->split_pb.p, p->d_scalars.p + 6, npb, st));
                    HIPCHK(hipcub::DevicePartition::Flagged(p->cub_temp.p, tb, iota, p->split_flags.p + m, p->split_nb.p, p->d_scalars.p + 7, nb, st));
                    HIPCHK(hipMemcpyAsync(&p->pb_bnd, p->d_scalars.p + 6, sizeof(int), hipMemcpyDeviceToHost, st));
                    HIPCHK(hipMemcpyAsync(&p->nb_bnd, p->d_scalars.p + 7, sizeof(int), hipMemcpyDeviceToHost, st));
                }
                HIPCHK(hipStreamSynchronize(st));
            }
        }
    }
    // clear the abort chain; the substep counter restarts so parity 0 is clean
    HIPCHK(hipMemsetAsync(p->d_ctl, 0, 2 * sizeof(uint32_t), st));
    HIPCHK(hipMemsetAsync(&p->d_ctl->heads[0], 0, 4 * sizeof(uint32_t), st));   // heads[2], heads_base, excess_acc
    HIPCHK(hipMemsetAsync(&p->d_ctl->reorder_request, 0, sizeof(uint32_t), st));
    CHK(upload_policy(p));
    HIPCHK(hipGetLastError());
    CHK(clock_end(p, tok));
    if (n == 0) p->n_binned = 0;
    p->proj_cache_clear = true;   // the node blocks are new
    p->sorted = true;
    p->tiles_ready = false;
    p->mass_tiles_ready = false;
    p->clock.resorts++;
    p->total_resorts++;
    return SPARKL_OK;
}

// ----------------------------------------------------------------------------------- one substep
static StepParams make_params(const SparklPipeline* p, const SparklSolverParams* sp, const float g[3], float dt) {
    StepParams S;
    S.h = p->h; S.inv_d = 4.0f / (p->h * p->h); S.inv_h = 1.0f / p->h; S.dt = dt; S.dt_h = dt / p->h;
    S.gx = g[0]; S.gy = g[1]; S.gz = g[2];
    S.boundary_handling = sp->boundary_handling; S.dofs = sp->simulation_dofs;
    S.parity = (int)(p->substep_index & 1u); S.dom = p->dom; S.num_cuboids = p->num_cuboids; S.num_heightfields = p->num_heightfields; S.num_trimeshes = p->num_trimeshes; S.want_dt = 0; S.defer_c = 0;
    S.project_particles = sp->enable_boundary_particle_projection ? 1 : 0;
    S.collider_handling = p->collider_handling ? 1 : 0;
    S.penalty = p->collider_penalty ? 1 : 0;
    S.ku_mass = p->ku[0]; S.ku_vol0 = p->ku[1]; S.ku_eh = p->ku[2];
    S.jp_one = (p->all_jp_one && p->all_plain_fluid && !p->slab_on) ? 1 : 0;   // (slabs receive immigrants whose value the host has not seen)
    S.dt_next = dt;
    S.count_heads = 1;
    S.mass_from_sweep = 0;
    S.num_models = (int)p->h_models.size();
    return S;
}

#ifndef SPK_P2G_REDUCE_STEPS
#define SPK_P2G_REDUCE_STEPS 4   // 16-lane groups: same speed as 3 on the aligned initial lattice, 5-8 % faster once cells hold uneven counts
#endif

// the projection cache of the grid update: sized for the current node blocks, zeroed (= "not computed") when they or the colliders changed
static void prepare_proj_cache(SparklPipeline* p, hipStream_t st) {
    // (GPU rule: two records per node -- closest projection, signed-distance normal -- and one status word per block, whatever the collider kinds)
    const size_t slots = p->collider_handling ? 2 : (size_t)(p->num_heightfields + p->num_trimeshes + p->num_polylines);
    if (!p->proj_cache_on || slots == 0 || p->num_blocks == 0) return;
    const size_t need = slots * (size_t)p->num_blocks * 64, need_blk = slots * (size_t)p->num_blocks;
    if (need > p->node_proj.cap || need_blk > p->node_proj_blk.cap) {
        if (p->node_proj.reserve(need) != SPARKL_OK || p->node_proj_blk.reserve(need_blk) != SPARKL_OK) { p->node_proj.release(); p->node_proj_blk.release(); return; }
        p->proj_cache_clear = true;
    }
    // only the per-block words are cleared: a node record is never read before its block's word says "computed"
    if (p->proj_cache_clear) { (void)hipMemsetAsync(p->node_proj_blk.p, 0, need_blk * sizeof(int), st); p->proj_cache_clear = false; }
}
// the instantiation of k_grid_update a collider set needs: triangle-searching shapes or not, and the reference's GPU rule (closest collider,
// signed-distance normal: kernels.h gpu_boundary_rule) as soon as one collider carries a boundary handling of its own
static float4* direct_halo(const SparklPipeline* p, const void* arena, size_t cap_blocks, int parity) {
    (void)p;
    return reinterpret_cast<float4*>((char*)arena + sizeof(PeerBox)) + (size_t)parity * cap_blocks * 64;
}
static void launch_grid_kernel(SparklPipeline* p, const GridArrays& G_in, const StepParams& S, dim3 grid, hipStream_t st) {
    const dim3 block(BLK_THREADS);
    GridArrays G = G_in;
    if (p->dw.box) G.halo_in = p->num_blocks ? direct_halo(p, p->direct.arena, p->direct.cap_blocks, (int)(p->dw.epoch & 1u)) : nullptr;   // what the neighbours' k_halo_push wrote for this epoch
    const bool shapes = p->num_heightfields > 0 || p->num_trimeshes > 0 || p->num_polylines > 0;
#define SPK_GU(HF, GR) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_update<false, HF, GR>), grid, block, 0, st, G, p->d_cuboids.p, p->d_heightfields.p, p->d_trimeshes.p, p->d_ctl, S, p->dw)
    if (p->collider_handling) { if (shapes) SPK_GU(true, true); else SPK_GU(false, true); }
    else { if (shapes) SPK_GU(true, false); else SPK_GU(false, false); }
#undef SPK_GU
}
static void launch_grid_update(SparklPipeline* p, const GridArrays& G_in, const StepParams& S, hipStream_t st) {
    GridArrays G = G_in;
    if (p->num_heightfields + p->num_trimeshes + p->num_polylines > 0 || p->collider_handling) {
        prepare_proj_cache(p, st);
        G.node_proj = (p->proj_cache_on && p->node_proj.p && p->node_proj_blk.p && !p->proj_cache_clear) ? p->node_proj.p : nullptr;
        G.node_proj_blk = p->node_proj_blk.p;
    }
    launch_grid_kernel(p, G, S, dim3((p->num_blocks + 3) / 4), st);
}

// Particle kernels stage the model table in LDS when it fits (see stage_models); longer tables are read from global memory.
#define SPK_LAUNCH_MODELS(KERNEL, ...)                                                                                              \
    do {                                                                                                                            \
        if (S.num_models <= MAX_LDS_MODELS) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<__VA_ARGS__, true>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);   \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<__VA_ARGS__, false>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);                                \
    } while (0)

// The disorder books (Ctl::heads_base / excess_acc, pb_heads) hold ONE measure at a time: 0 = lane-group heads of the standalone DPP scatters,
// 1 = distinct cells per wave of the fused run-task scatter.  A scatter that feeds them with the other measure starts them afresh
// (stream-ordered memsets); a re-bin clears them anyway (heads_kind = -1).
static void note_heads_kind(SparklPipeline* p, int kind, hipStream_t st) {
    if (p->heads_kind == kind) return;
    if (p->heads_kind >= 0 && p->num_pblocks > 0) {
        (void)hipMemsetAsync(p->pb_heads.p, 0, (size_t)p->num_pblocks * sizeof(int2), st);
        (void)hipMemsetAsync(&p->d_ctl->heads_base, 0, 2 * sizeof(uint32_t), st);   // heads_base, excess_acc
    }
    p->heads_kind = kind;
}
// `counts`: the launch feeds the disorder policy (StepParams::count_heads)
static void launch_p2g(SparklPipeline* p, const ParticleArrays& P, const GridArrays& G, const StepParams& S_in, hipStream_t st, bool counts) {
    StepParams S = S_in;
    S.count_heads = counts ? 1 : 0;
    if (counts) note_heads_kind(p, 0, st);
    const dim3 grid(p->num_pblocks), block(BLK_THREADS);
    // standalone P2G (adaptive-dt substeps, first substep of a fused batch): two particles per lane, 0.38 -> 0.357 ms at 10M.
    // SPARKL_HIP_P2G_PAIR=0 selects the one-particle-per-lane kernel (A/B).
    static const int pair_steps = getenv("SPARKL_HIP_P2G_PAIR") ? atoi(getenv("SPARKL_HIP_P2G_PAIR")) : 3;
    if (pair_steps == 4) SPK_LAUNCH_MODELS(k_p2g_pair, 4);
    else if (pair_steps == 0) SPK_LAUNCH_MODELS(k_p2g, SPK_P2G_REDUCE_STEPS);
    else SPK_LAUNCH_MODELS(k_p2g_pair, 3);
}
// the flags mirror travels with every launch that may change a binned particle's flags while it is valid (ParticleArrays::fl)
static ParticleArrays with_mirror(const SparklPipeline* p, const ParticleArrays& P0) {
    ParticleArrays P = P0;
    // Handed out whatever the models are, so that every G2P store keeps it in step (which KERNEL reads it is the launchers' choice: the Drucker-Prager
    // view of k_g2p2g, the plain-fluid sweep); has_frac: the psi sweep's crack test sets FLAG_PHASE_NOT_ONE behind the mirror's back.
    P.fl = (p->ku_ok && !p->cur.has_frac) ? p->fl.p : nullptr;
    return P;
}
static void launch_g2p(SparklPipeline* p, const ParticleArrays& P0, const GridArrays& G, const StepParams& S, hipStream_t st) {
    const ParticleArrays P = with_mirror(p, P0);
    const dim3 grid(p->num_pblocks), block(BLK_THREADS);
    if (p->rare_models || S.project_particles) SPK_LAUNCH_MODELS(k_g2p, true);
    else SPK_LAUNCH_MODELS(k_g2p, false);
}
static void launch_g2p2g(SparklPipeline* p, const ParticleArrays& P0, const GridArrays& G, const StepParams& S, hipStream_t st) {
    const ParticleArrays P = with_mirror(p, P0);
    const dim3 grid(G.pb_list ? G.pb_count : p->num_pblocks), block(BLK_THREADS);   // the whole list or the launch's subset (slab groups)
    note_heads_kind(p, (getenv("SPARKL_HIP_SCATTER") && !strcmp(getenv("SPARKL_HIP_SCATTER"), "dpp")) ? 0 : 1, st);
    // SPARKL_HIP_SCATTER=dpp selects the DPP pre-reduction scatter (A/B); the default is the run-task scatter
    static const bool dpp = getenv("SPARKL_HIP_SCATTER") && !strcmp(getenv("SPARKL_HIP_SCATTER"), "dpp");
    if (p->all_snow && !dpp && !S.project_particles && !S.penalty && S.num_models <= MAX_LDS_MODELS)   // (rare_models is set for it: in front of the generic RARE launch)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_g2p2g<-1, true, true, 5>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);
    else if (p->rare_models || S.project_particles) SPK_LAUNCH_MODELS(k_g2p2g, -1, true);
    else if (dpp) SPK_LAUNCH_MODELS(k_g2p2g, SPK_P2G_REDUCE_STEPS, false);
    else if (p->all_dp_sand && S.num_models <= MAX_LDS_MODELS && P.fl)   // the same view on the 4-byte flags mirror (every binned particle shares mass, volume0, hardening)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_g2p2g<-1, false, true, 4>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);
    else if (p->all_dp_sand && S.num_models <= MAX_LDS_MODELS)   // SPARKL_HIP_MSET=0 keeps the generic kernel (A/B)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_g2p2g<-1, false, true, 1>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);
    else if (p->all_neo_elastic && S.num_models <= MAX_LDS_MODELS)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_g2p2g<-1, false, true, 3>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);
    else SPK_LAUNCH_MODELS(k_g2p2g, -1, false);
}

static int enqueue_fluid_volumes(SparklPipeline* p, const StepParams& S) {
    hipStream_t st = p->stream;
    GridArrays G = grid_arrays(p);
    G.node_mom = nullptr;
    // sampled like the substep the sweep belongs to
    const bool was_on = p->clock_on;
    if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
    size_t t0 = 0, t1 = 0, t2 = 0;
    CHK(clock_begin(p, SPAN_P2G_MASS, &t0));
    hipLaunchKernelGGL(k_p2g_mass, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, st, p->cur.arrays(), G, p->d_ctl, S);
    CHK(clock_end(p, t0));
    CHK(clock_begin(p, SPAN_GRID_MASS, &t1));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_update<true>), dim3((p->num_blocks + 3) / 4), dim3(BLK_THREADS), 0, st, G, p->d_cuboids.p, p->d_heightfields.p, p->d_trimeshes.p, p->d_ctl, S);
    CHK(clock_end(p, t1));
    CHK(clock_begin(p, SPAN_G2P_VOLUME, &t2));
    hipLaunchKernelGGL(k_g2p_volume, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, st, p->cur.arrays(), G, p->d_models.p, p->d_ctl, S);
    CHK(clock_end(p, t2));
    p->clock_on = was_on;
    return SPARKL_OK;
}

static bool damage_supported(const SparklSolverParams* params) { return params->damage_model == SPARKL_DAMAGE_NONE || params->damage_model == SPARKL_DAMAGE_MODIFIED_EIGENEROSION; }
// the psi sweep only has something to do when some particle carries a crack_propagation_factor
static bool psi_sweep_on(const SparklPipeline* p, const SparklSolverParams* params) { return params->damage_model == SPARKL_DAMAGE_MODIFIED_EIGENEROSION && p->cur.has_frac; }
// DamageModel::ModifiedEigenerosion: psi scatter, node merge, crack test -- between the grid update and G2P (see k_p2g_psi)
static int enqueue_psi_sweep(SparklPipeline* p, const StepParams& S) {
    CHK(p->psi_tiles.reserve((size_t)p->num_pblocks * TILE_N));
    CHK(p->node_psi.reserve((size_t)p->num_blocks * 64));
    const GridArrays G = grid_arrays(p);
    hipLaunchKernelGGL(k_p2g_psi, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), G, p->d_ctl, S, p->psi_tiles.p);
    hipLaunchKernelGGL(k_grid_psi, dim3((p->num_blocks + 3) / 4), dim3(BLK_THREADS), 0, p->stream, G, p->psi_tiles.p, p->node_psi.p, p->d_ctl, S.parity);
    hipLaunchKernelGGL(k_g2p_psi, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), G, p->node_psi.p, p->d_ctl, S);
    return SPARKL_OK;
}

// enqueue P2G -> grid update -> G2P for the current substep index (no host sync)
static int enqueue_substep(SparklPipeline* p, const SparklSolverParams* sp, const float g[3], float dt, bool want_dt = false) {
    if (p->num_pblocks == 0) { p->substep_index++; return SPARKL_OK; }
    hipStream_t st = p->stream;
    StepParams S = make_params(p, sp, g, dt);
    S.want_dt = want_dt ? 1 : 0;
    const GridArrays G = grid_arrays(p);
    const ParticleArrays P = p->cur.arrays();
    if (sp->force_fluids_volume_recomputation && p->any_fluid) CHK(enqueue_fluid_volumes(p, S));
    size_t t0 = 0, t1 = 0, t2 = 0;
    const bool was_on = p->clock_on;
    if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;  // sample this substep or not
    if (p->clock_on) p->clock_sampled++;
    CHK(maybe_reorder(p, S.parity));
    CHK(clock_begin(p, SPAN_P2G, &t0));
    launch_p2g(p, P, G, S, st, true);       // the unfused schedule scatters with it every substep
    CHK(clock_end(p, t0));
    CHK(clock_begin(p, SPAN_GRID, &t1));
    launch_grid_update(p, G, S, st);
    CHK(clock_end(p, t1));
    if (psi_sweep_on(p, sp)) CHK(enqueue_psi_sweep(p, S));
    CHK(clock_begin(p, SPAN_G2P, &t2));
    launch_g2p(p, P, G, S, st);
    CHK(clock_end(p, t2));
    p->clock_on = was_on;
    p->substep_index++;
    p->enq_steps++;
    return SPARKL_OK;
}

// One substep of the adaptive loop.  `reuse_tiles`: the previous substep's fused kernel already scattered for exactly this dt.
// `dt_guess` > 0: run G2P fused with the scatter of the next substep built for that step length (speculation, see
// sparkl_hip_pipeline_step); otherwise the plain G2P.  Both reduce the next dt bound.
static int enqueue_adaptive_substep(SparklPipeline* p, const SparklSolverParams* sp, const float g[3], float dt, bool reuse_tiles, float dt_guess, uint32_t expect_bound_bits = 0u,
                                    bool defer_c = false) {
    if (p->num_pblocks == 0) { p->substep_index++; return SPARKL_OK; }
    hipStream_t st = p->stream;
    StepParams S = make_params(p, sp, g, dt);
    S.want_dt = 1;
    const GridArrays G = grid_arrays(p);
    const ParticleArrays P = p->cur.arrays();
    size_t t0 = 0, t1 = 0, t2 = 0;
    const bool was_on = p->clock_on;
    if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
    if (p->clock_on) p->clock_sampled++;
    CHK(maybe_reorder(p, S.parity));
    if (!reuse_tiles) {
        CHK(clock_begin(p, SPAN_P2G, &t0));
        launch_p2g(p, P, G, S, st, !(dt_guess > 0.f));   // counts in the per-substep (bound-limited) regime, not as the odd scatter between fused ones
        CHK(clock_end(p, t0));
    }
    CHK(clock_begin(p, SPAN_GRID, &t1));
    launch_grid_update(p, G, S, st);
    CHK(clock_end(p, t1));
    if (psi_sweep_on(p, sp)) CHK(enqueue_psi_sweep(p, S));
    if (dt_guess > 0.f) {
        S.dt_next = dt_guess;
        S.defer_c = (defer_c && p->defer_c && !p->rare_fused() && !S.project_particles) ? 1 : 0;   // look-ahead plans only (the caller rebuilds C if the plan is cut)
        if (S.defer_c) p->c_stale_S = S;
        CHK(clock_begin(p, SPAN_FUSED, &t2));
        launch_g2p2g(p, P, G, S, st);
        CHK(clock_end(p, t2));
    } else {
        CHK(clock_begin(p, SPAN_G2P, &t2));
        launch_g2p(p, P, G, S, st);
        CHK(clock_end(p, t2));
    }
    hipLaunchKernelGGL(k_dt_reduce, dim3(1), dim3(1024), 0, st, p->dt_partial.p, p->num_pblocks, p->d_ctl, expect_bound_bits, (S.parity + 1) & 1);
    p->clock_on = was_on;
    p->substep_index++;
    p->enq_steps++;
    return SPARKL_OK;
}

// Fixed-dt batch with the fused schedule:  P2G_s ; { GU ; G2P+P2G } x (n-1) ; GU ; G2P.  Same arithmetic as the
// unfused order (the scatter of substep s+1 starts from the registers G2P of substep s just produced).
static int enqueue_fused_batch(SparklPipeline* p, const SparklSolverParams* sp, const float g[3], float dt, uint32_t n) {
    if (p->num_pblocks == 0) { p->substep_index += n; return SPARKL_OK; }
    hipStream_t st = p->stream;
    const GridArrays G = grid_arrays(p);
    const ParticleArrays P = p->cur.arrays();
    const bool was_on = p->clock_on;
    size_t tk = 0;
    const bool carry_in = p->carry_on && p->carry_valid && p->carry_dt == dt && p->carry_index == p->substep_index;
    const bool carry_out = p->carry_on && !p->rare_fused() && !sp->enable_boundary_particle_projection;
    p->carry_valid = false;
    if (!carry_in) {
        const StepParams S = make_params(p, sp, g, dt);
        if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
        CHK(clock_begin(p, SPAN_P2G, &tk));
        launch_p2g(p, P, G, S, st, false);
        CHK(clock_end(p, tk));
        p->clock_on = was_on;
    }
    for (uint32_t j = 0; j < n; ++j) {
        if (j > 0 && !may_enqueue_next(p)) return SPARKL_OK;   // a re-bin request is up: the caller reads the control block and re-bins (carry_valid stays false)
        const StepParams S = make_params(p, sp, g, dt);
        if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
        if (p->clock_on) p->clock_sampled++;
        if (j > 0) CHK(maybe_reorder(p, S.parity));   // between two fused kernels: the tiles of this substep are already scattered
        CHK(clock_begin(p, SPAN_GRID, &tk));
        launch_grid_update(p, G, S, st);
        CHK(clock_end(p, tk));
        if (j + 1 < n) {
            StepParams Sf = S;
            Sf.defer_c = (p->defer_c && !p->rare_fused() && !S.project_particles) ? 1 : 0;
            if (Sf.defer_c) p->c_stale_S = S;   // what k_regather_c needs should the batch stop here
            CHK(clock_begin(p, SPAN_FUSED, &tk));
            launch_g2p2g(p, P, G, Sf, st);
            CHK(clock_end(p, tk));
        } else if (carry_out) {
            // the batch ends with the fused kernel too: complete particle state (C included) + the tiles of the next substep
            CHK(clock_begin(p, SPAN_FUSED, &tk));
            launch_g2p2g(p, P, G, S, st);
            CHK(clock_end(p, tk));
        } else {
            CHK(clock_begin(p, SPAN_G2P, &tk));
            launch_g2p(p, P, G, S, st);
            CHK(clock_end(p, tk));
        }
        p->clock_on = was_on;
        p->substep_index++;
        p->enq_steps++;
    }
    if (carry_out) { p->carry_valid = true; p->carry_dt = dt; p->carry_index = p->substep_index; }
    return SPARKL_OK;
}

// Fixed-dt batch with force_fluids_volume_recomputation:  M ; { GUm ; V+P ; GU ; G+M } x n  (see k_g2p_mass / k_vol_p2g).
// Same arithmetic as the unfused sweep (mass scatter, mass merge, volume gather, P2G, grid update, G2P), two passes over the
// particles per substep instead of four.
// `check` (adaptive step, n == 1): {remaining, min_dt}; the estimator runs on the recomputed volumes right behind k_vol_p2g and
// k_dt_check validates `dt` against it (see k_dt_check)
static int enqueue_fluid_batch(SparklPipeline* p, const SparklSolverParams* sp, const float g[3], float dt, uint32_t n, const float* check = nullptr) {
    if (p->num_pblocks == 0) { p->substep_index += n; return SPARKL_OK; }
    hipStream_t st = p->stream;
    GridArrays G = grid_arrays(p);
    const ParticleArrays P = with_mirror(p, p->cur.arrays());
    const bool was_on = p->clock_on;
    const dim3 grid(p->num_pblocks), block(BLK_THREADS), ggrid((p->num_blocks + 3) / 4);
    size_t tk = 0;
    const bool fluid_only = p->all_plain_fluid && !sp->enable_boundary_particle_projection;
    const bool ku = fluid_only && P.fl != nullptr && p->h_models.size() <= (size_t)MAX_LDS_MODELS;   // the k record from the flags mirror + StepParams (ParticleArrays::fl)
    note_heads_kind(p, 0, st);   // k_vol_p2g counts lane-group heads like the standalone P2G: one measure in the books at a time
    if (!p->mass_tiles_ready) {
        const StepParams S = make_params(p, sp, g, dt);
        if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
        CHK(clock_begin(p, SPAN_P2G_MASS, &tk));
        hipLaunchKernelGGL(k_p2g_mass, grid, block, 0, st, P, G, p->d_ctl, S);
        CHK(clock_end(p, tk));
        p->clock_on = was_on;
    }
    for (uint32_t j = 0; j < n; ++j) {
        // the same bounded enqueue-ahead as the fused sand batch (a re-bin request is up: the caller reads the control block and re-bins)
        if (j > 0 && !may_enqueue_next(p)) { p->mass_tiles_ready = false; return SPARKL_OK; }
        const StepParams S = make_params(p, sp, g, dt);
        if (was_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
        if (p->clock_on) p->clock_sampled++;
        GridArrays Gm = G;
        Gm.node_mom = nullptr;
        CHK(clock_begin(p, SPAN_GRID_MASS, &tk));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_update<true>), ggrid, block, 0, st, Gm, p->d_cuboids.p, p->d_heightfields.p, p->d_trimeshes.p, p->d_ctl, S);
        CHK(clock_end(p, tk));
        CHK(clock_begin(p, SPAN_P2G, &tk));
        if (ku) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vol_p2g<true, true, true>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);
        else if (fluid_only) SPK_LAUNCH_MODELS(k_vol_p2g, true); else SPK_LAUNCH_MODELS(k_vol_p2g, false);
        CHK(clock_end(p, tk));
        if (check) {
            uint32_t dt_bits;
            memcpy(&dt_bits, &dt, 4);
            CHK(clock_begin(p, SPAN_DT, &tk));
            hipLaunchKernelGGL(k_estimate_dt, dim3(grid_for(p->n, BLK_THREADS)), block, 0, st, P, p->d_models.p, p->d_ctl, (int)p->n, p->h, p->dt_partial.p);
            hipLaunchKernelGGL(k_dt_reduce, dim3(1), dim3(1024), 0, st, p->dt_partial.p, (int)grid_for(p->n, BLK_THREADS), p->d_ctl, 0u, 0);
            hipLaunchKernelGGL(k_dt_check, dim3(1), dim3(1), 0, st, p->d_ctl, check[0], sp->max_substep_dt, check[1], dt_bits, S.parity);
            CHK(clock_end(p, tk));
        }
        CHK(maybe_reorder(p, S.parity));
        CHK(clock_begin(p, SPAN_GRID, &tk));
        {
            StepParams Sg = S;
            Sg.mass_from_sweep = 1;   // k_vol_p2g scattered the momentum only: the node masses are the ones GUm left in node_vel.w
            launch_grid_update(p, G, Sg, st);
        }
        CHK(clock_end(p, tk));
        CHK(clock_begin(p, SPAN_G2P, &tk));
        if (ku) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_g2p_mass<true, true, true>), grid, block, 0, st, P, G, p->d_models.p, p->d_ctl, S);
        else if (fluid_only) SPK_LAUNCH_MODELS(k_g2p_mass, true); else SPK_LAUNCH_MODELS(k_g2p_mass, false);
        CHK(clock_end(p, tk));
        p->clock_on = was_on;
        p->substep_index++;
        p->enq_steps++;
    }
    p->mass_tiles_ready = true;
    return SPARKL_OK;
}

static int comm_wait(SparklPipeline* p, hipStream_t st);   // hipStreamSynchronize that a dead RCCL peer cannot block for ever (below)
static int read_ctl(SparklPipeline* p) {
    HIPCHK(hipMemcpyAsync(p->h_ctl, p->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, p->stream));
    CHK(comm_wait(p, p->stream));
    HIPCHK(hipGetLastError());
    p->enq_steps = p->h_ctl->steps_done;   // the stream is idle: nothing is enqueued ahead
    return SPARKL_OK;
}
// Before enqueueing one more substep of a fixed-dt batch: wait until the GPU is at most `ahead_limit` substeps behind (polling the
// steps_done mirror in pinned memory: no event, no synchronisation, the queue never drains -- four substeps are ~2 ms of work), and report
// whether the abort chain has been raised (false = stop enqueueing: everything behind it would be a no-op launch).
static bool may_enqueue_next(SparklPipeline* p) {
    if (!p->h_words || p->ahead_limit == 0) return true;
    const int limit = p->ahead_limit > 0 ? p->ahead_limit : (int)std::max<uint64_t>(4, std::min<uint64_t>(64, 40000000ull / std::max<uint64_t>(p->n, 1)));
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (p->h_words[0]) return false;
        if ((int32_t)(p->enq_steps - p->h_words[1]) < limit) return true;
        if ((spin & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) return true;   // never a hang: fall back to enqueueing
    }
}

static int validate_models(const SparklPipeline* p) {
    if (p->n && p->max_model_index >= p->h_models.size())
        return set_error(SPARKL_ERR_INVALID_ARGUMENT, "a particle uses model %u but only %zu models are set", p->max_model_index, p->h_models.size());
    return SPARKL_OK;
}

// ------------------------------------------------------------------------------------- C ABI: misc
extern "C" {

const char* sparkl_hip_last_error_string(void) { return g_last_error.c_str(); }
const char* sparkl_hip_version(void) { return "sparkl_hip 0.1.0 gfx950"; }

int sparkl_hip_pipeline_create(const SparklPipelineDesc* desc, SparklPipeline** out) {
    if (!desc || !out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!(desc->cell_width > 0.f)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "cell_width must be > 0");
    uint32_t bits = desc->domain_log2_bytes ? desc->domain_log2_bytes : 30;
    if (bits != 30 && bits != 33 && bits != 36 && bits != 39) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "domain_log2_bytes must be 30, 33, 36 or 39");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_error(SPARKL_ERR_NO_DEVICE, "no HIP device available");
    if (desc->device_id < 0 || desc->device_id >= ndev) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "device_id %d out of range (%d devices)", desc->device_id, ndev);
    if (hipSetDevice(desc->device_id) != hipSuccess) return set_error(SPARKL_ERR_NO_DEVICE, "hipSetDevice(%d) failed", desc->device_id);
    SparklPipeline* p = new SparklPipeline;
    p->fuse_fixed_dt = getenv("SPARKL_HIP_NO_FUSE") == nullptr;
    p->defer_c = !(getenv("SPARKL_HIP_DEFER_C") && !strcmp(getenv("SPARKL_HIP_DEFER_C"), "0"));
    p->ku_on = !(getenv("SPARKL_HIP_KUNIFORM") && !strcmp(getenv("SPARKL_HIP_KUNIFORM"), "0"));
    if (const char* e = getenv("SPARKL_HIP_CARRY")) p->carry_on = atoi(e) != 0;
    if (const char* e = getenv("SPARKL_HIP_PENALTY_FORCE")) p->penalty_force_on = atoi(e) != 0;
    if (const char* e = getenv("SPARKL_HIP_PROJ_CACHE")) p->proj_cache_on = atoi(e) != 0;
    if (const char* e = getenv("SPARKL_HIP_OVERLAP")) p->overlap_mode = !strcmp(e, "cal") ? -2 : (atoi(e) != 0 ? 1 : 0);   // "cal": calibrate even on one device (tests)
    p->lookahead = !(getenv("SPARKL_HIP_LOOKAHEAD") && !strcmp(getenv("SPARKL_HIP_LOOKAHEAD"), "0"));
    p->lookahead_force = getenv("SPARKL_HIP_LOOKAHEAD") && !strcmp(getenv("SPARKL_HIP_LOOKAHEAD"), "force");
    p->device = desc->device_id;
    p->h = desc->cell_width;
    p->domain_log2 = bits;
    p->dom = 4 << ((bits - 12) / 3);
    if (desc->stream) { p->stream = (hipStream_t)desc->stream; p->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { delete p; return set_error(SPARKL_ERR_HIP, "hipStreamCreate failed"); }
        p->own_stream = true;
    }
    if (hipMalloc((void**)&p->d_ctl, sizeof(Ctl)) != hipSuccess || hipHostMalloc((void**)&p->h_ctl, sizeof(Ctl)) != hipSuccess) {
        delete p;
        return set_error(SPARKL_ERR_OUT_OF_MEMORY, "control block allocation failed");
    }
    (void)hipMemset(p->d_ctl, 0, sizeof(Ctl));
    (void)hipMemsetD32((hipDeviceptr_t)&p->d_ctl->dead_dt_bits, 0x7f7fffff, 1);
    memset(p->h_ctl, 0, sizeof(Ctl));
    if (const char* e = getenv("SPARKL_HIP_AHEAD")) p->ahead_limit = atoi(e);
    if (p->ahead_limit != 0) {   // the host-visible words of the control block (Ctl::host_words); without them the batches are enqueued whole
        uint32_t* hw = nullptr;
        if (hipHostMalloc((void**)&hw, 64, hipHostMallocMapped) == hipSuccess) {
            hw[0] = hw[1] = 0u;
            void* dev = nullptr;
            if (hipHostGetDevicePointer(&dev, hw, 0) == hipSuccess && hipMemcpy(&p->d_ctl->host_words, &dev, sizeof(void*), hipMemcpyHostToDevice) == hipSuccess) p->h_words = hw;
            else (void)hipHostFree(hw);
        }
    }
    *out = p;
    return SPARKL_OK;
}

void sparkl_hip_pipeline_destroy(SparklPipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->stream);
    p->cur.release(); p->alt.release();
    p->d_models.release(); p->d_cuboids.release(); p->d_heightfields.release();
    for (float* d : p->d_heights) (void)hipFree(d);
    for (void* d : p->d_mesh_bufs) (void)hipFree(d);
    for (float* d : p->d_poly_verts) (void)hipFree(d);
    p->d_trimeshes.release(); p->d_polylines.release();
    p->blk_origin.release(); p->blk_pfirst.release(); p->blk_pcount.release(); p->blk_nbr.release(); p->blk_tnbr.release(); p->pblk.release(); p->pb_info.release(); p->tile_mode.release(); p->pb_heads.release(); p->blk_keys.release();
    p->tiles.release(); p->node_vel.release(); p->node_mom.release(); p->hash_keys.release(); p->hash_vals.release();
    p->keys_a.release(); p->keys_b.release(); p->pb_keys.release(); p->bkeys.release(); p->idx_a.release(); p->idx_b.release();
    p->pb_counts.release(); p->pb_first.release(); p->cand_a.release(); p->cand_b.release(); p->cub_temp.release(); p->d_scalars.release();
    p->staging.release(); p->node_staging.release(); p->d_count.release(); p->dt_partial.release(); p->psi_tiles.release(); p->node_psi.release();
    p->oorg.release(); p->node_proj.release(); p->node_proj_blk.release(); p->pb_order.release(); p->fl.release(); p->order_keys.release(); p->order_keys_out.release(); p->order_vals.release();
    p->split_pb.release(); p->split_nb.release(); p->split_flags.release(); p->bal_dev.release();
    p->halo_list[0].release(); p->halo_list[1].release(); p->halo_in.release(); p->d_tail.release();
    for (int k = 0; k < 2; ++k) { p->x_send[k].release(); p->x_recv[k].release(); p->x_mig_send[k].release(); p->x_mig_recv[k].release(); }
    p->x_dev.release();
    direct_release(p);
    if (p->x_host) (void)hipHostFree(p->x_host);
    if (p->bal_host) (void)hipHostFree(p->bal_host);
    comm_destroy(p);
    if (p->rb_stream) { (void)hipStreamSynchronize(p->rb_stream); (void)hipStreamDestroy(p->rb_stream); (void)hipEventDestroy(p->rb_gathered); (void)hipEventDestroy(p->rb_done); }
    p->rb_staging.release();
    for (auto e : p->ev_pool) (void)hipEventDestroy(e);
    if (p->halo_stream) { (void)hipStreamSynchronize(p->halo_stream); (void)hipStreamDestroy(p->halo_stream); }
    for (hipEvent_t e : {p->ev_main, p->ev_halo, p->ev_flag, p->ev_grid, p->ev_xfer}) if (e) (void)hipEventDestroy(e);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    if (p->h_ctl) (void)hipHostFree(p->h_ctl);
    if (p->h_words) (void)hipHostFree((void*)p->h_words);
    if (p->own_stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int sparkl_hip_reorder_policy_set(SparklPipeline* p, float reorder_limit, float rebin_limit) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (reorder_limit != reorder_limit || rebin_limit != rebin_limit) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "NaN limit");
    HIPCHK(hipSetDevice(p->device));
    p->reorder_limit = reorder_limit;
    p->rebin_limit = rebin_limit;
    return upload_policy(p);
}
int sparkl_hip_reorder_policy_stats(SparklPipeline* p, SparklReorderStats* out) {
    if (!p || !out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    CHK(read_ctl(p));
    out->last_heads = p->h_ctl->last_heads; out->base_heads = p->h_ctl->heads_base;
    out->reorders = p->h_ctl->reorders; out->policy_rebins = p->h_ctl->disorder_aborts;
    return SPARKL_OK;
}

int sparkl_hip_pipeline_sync(SparklPipeline* p) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    return SPARKL_OK;
}

// ------------------------------------------------------------------------------ models / colliders
int sparkl_hip_models_set(SparklPipeline* p, uint32_t n, const SparklParticleModel* models) {
    if (p) p->carry_valid = false;   // tiles scattered with the old model table
    if (!p || (n && !models)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    std::vector<DevModel> dm(n);
    bool any_fluid = false;
    for (uint32_t i = 0; i < n; ++i) {
        const SparklParticleModel& m = models[i];
        DevModel d;
        memset(&d, 0, sizeof(d));
        d.ctag = (int)m.constitutive_model.tag;
        d.ptag = -1;
        switch (m.constitutive_model.tag) {
            case SPARKL_CONSTITUTIVE_COROTATED_LINEAR:
                d.split_stress_on_failure = m.constitutive_model.u.corotated.split_stress_on_failure;
                d.cfl_coeff = m.constitutive_model.u.corotated.cfl_coeff;
                d.lambda = m.constitutive_model.u.corotated.lambda; d.mu = m.constitutive_model.u.corotated.mu;
                break;
            case SPARKL_CONSTITUTIVE_NEO_HOOKEAN:
                d.cfl_coeff = m.constitutive_model.u.neo_hookean.cfl_coeff;
                d.lambda = m.constitutive_model.u.neo_hookean.lambda; d.mu = m.constitutive_model.u.neo_hookean.mu;
                break;
            case SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH:
                d.is_fluid = 1; any_fluid = true;
                d.pressure0 = m.constitutive_model.u.monaghan.pressure0; d.gamma = (int)m.constitutive_model.u.monaghan.gamma;
                d.viscosity = m.constitutive_model.u.monaghan.viscosity; d.max_neg_pressure = m.constitutive_model.u.monaghan.max_neg_pressure;
                break;
            default:
                return set_error(SPARKL_ERR_UNSUPPORTED, "model %u: CoreConstitutiveModel::Custom cannot cross the C ABI", i);
        }
        if (m.has_plastic_model) {
            d.ptag = (int)m.plastic_model.tag;
            switch (m.plastic_model.tag) {
                case SPARKL_PLASTIC_DRUCKER_PRAGER: {
                    const auto& dp = m.plastic_model.u.drucker_prager;
                    d.h0 = dp.h0; d.h1 = dp.h1; d.h2 = dp.h2; d.h3 = dp.h3; d.p_lambda = dp.lambda; d.p_mu = dp.mu;
                    d.volume_correction = dp.volume_correction; d.only_active_when_failed = dp.only_active_when_failed;
                    d.p_k = (3.0f * dp.lambda + 2.0f * dp.mu) / (2.0f * dp.mu);
                    break;
                }
                case SPARKL_PLASTIC_SNOW: {
                    const auto& sn = m.plastic_model.u.snow;
                    d.x0 = sn.min_epsilon; d.x1 = sn.max_epsilon; d.x2 = sn.hardening_coeff;
                    break;
                }
                case SPARKL_PLASTIC_RANKINE: {
                    const auto& r = m.plastic_model.u.rankine;
                    d.x0 = r.mu; d.x1 = r.lambda; d.x2 = r.tensile_strength; d.x3 = r.softening_rate;
                    break;
                }
                case SPARKL_PLASTIC_NACC: {
                    const auto& na = m.plastic_model.u.nacc;
                    d.x0 = na.mu; d.x1 = na.kappa; d.x2 = na.hardening_factor; d.x3 = na.cohesion; d.x4 = na.friction; d.xi = na.hardening_enabled;
                    break;
                }
                default:
                    return set_error(SPARKL_ERR_UNSUPPORTED, "model %u: CorePlasticModel::Custom cannot cross the C ABI", i);
            }
        }
        if (m.has_failure_model) {
            if (m.failure_model.tag != SPARKL_FAILURE_MAXIMUM_STRESS)
                return set_error(SPARKL_ERR_UNSUPPORTED, "model %u: CoreFailureModel::Custom cannot cross the C ABI", i);
            d.has_failure = 1;
            d.max_principal_stress = m.failure_model.u.maximum_stress.max_principal_stress;
            d.max_shear_stress = m.failure_model.u.maximum_stress.max_shear_stress;
        }
        dm[i] = d;
    }
    CHK(p->d_models.reserve(std::max<uint32_t>(n, 1)));
    if (n) HIPCHK(hipMemcpyAsync(p->d_models.p, dm.data(), n * sizeof(DevModel), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->h_models = dm;
    p->any_fluid = any_fluid;
    p->rare_by_models = false;
    p->all_plain_fluid = n > 0;
    p->all_dp_sand = n > 0 && !(getenv("SPARKL_HIP_MSET") && !strcmp(getenv("SPARKL_HIP_MSET"), "0"));
    p->all_neo_elastic = p->all_dp_sand;
    p->all_snow = p->all_dp_sand;
    for (const DevModel& d : dm) {
        p->all_snow &= !d.is_fluid && d.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR && d.ptag == SPARKL_PLASTIC_SNOW && !d.only_active_when_failed && !d.has_failure;
        p->rare_by_models |= d.ptag > SPARKL_PLASTIC_DRUCKER_PRAGER || d.has_failure;
        p->all_plain_fluid &= d.is_fluid && d.ptag < 0 && !d.has_failure;
        p->all_neo_elastic &= !d.is_fluid && d.ctag == SPARKL_CONSTITUTIVE_NEO_HOOKEAN && d.ptag < 0 && !d.has_failure;
        p->all_dp_sand &= !d.is_fluid && d.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR && d.ptag == SPARKL_PLASTIC_DRUCKER_PRAGER && !d.only_active_when_failed && !d.has_failure;
    }
    p->rare_models = p->rare_by_models || p->collider_penalty;
    if (p->d_ctl) CHK(upload_policy(p));   // the default reorder budget depends on whether a fluid is present
    return SPARKL_OK;
}

// what the per-collider options add up to (make_params), which kernels they need, and a stale projection cache
static void collider_options_changed(SparklPipeline* p) {
    p->collider_handling = false; p->collider_penalty = false;
    for (const auto& c : p->h_cuboids) { p->collider_handling |= c.handling >= 0; p->collider_penalty |= c.penalty > 0.f; }
    for (const auto& c : p->h_heightfields) { p->collider_handling |= c.handling >= 0; p->collider_penalty |= c.penalty > 0.f; }
    for (const auto& c : p->h_trimeshes) { p->collider_handling |= c.handling >= 0; p->collider_penalty |= c.penalty > 0.f; }
    for (const auto& c : p->h_polylines) { p->collider_handling |= c.handling >= 0; p->collider_penalty |= c.penalty > 0.f; }
    p->collider_penalty = p->collider_penalty && p->penalty_force_on;   // stored, not applied, unless opted in (see penalty_force_on)
    p->rare_models = p->rare_by_models || p->collider_penalty;   // the penalty force lives in the RARE instantiations of the G2P kernels
    p->proj_cache_clear = true;
    p->carry_valid = false;
}
static int check_options(uint32_t n, const SparklColliderOptions* o, size_t have, const char* kind) {
    if (n != have) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "%u option records for %zu %s colliders", n, have, kind);
    if (n && !o) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t i = 0; i < n; ++i) {
        if (o[i].grid_boundary_handling < SPARKL_COLLIDER_HANDLING_FROM_PARAMS || o[i].grid_boundary_handling > SPARKL_BOUNDARY_NONE)
            return set_error(SPARKL_ERR_INVALID_ARGUMENT, "%s collider %u: grid_boundary_handling %d", kind, i, (int)o[i].grid_boundary_handling);
        if (!(o[i].penalty_stiffness >= 0.f)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "%s collider %u: penalty_stiffness must be >= 0", kind, i);
    }
    return SPARKL_OK;
}
int sparkl_hip_colliders_enable_penalty_force(SparklPipeline* p, int enable) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->penalty_force_on = enable != 0;
    collider_options_changed(p);
    return SPARKL_OK;
}
int sparkl_hip_colliders_set_cuboid_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* o) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    CHK(check_options(n, o, p->h_cuboids.size(), "cuboid"));
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (uint32_t i = 0; i < n; ++i) { p->h_cuboids[i].handling = o[i].grid_boundary_handling; p->h_cuboids[i].penalty = o[i].penalty_stiffness; }
    if (n) HIPCHK(hipMemcpy(p->d_cuboids.p, p->h_cuboids.data(), n * sizeof(DevCuboid), hipMemcpyHostToDevice));
    collider_options_changed(p);
    return SPARKL_OK;
}
int sparkl_hip_colliders_set_heightfield_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* o) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    CHK(check_options(n, o, p->h_heightfields.size(), "heightfield"));
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (uint32_t i = 0; i < n; ++i) {
        p->h_heightfields[i].handling = o[i].grid_boundary_handling; p->h_heightfields[i].penalty = o[i].penalty_stiffness; p->h_heightfields[i].flip_interior = o[i].flip_interior ? 1 : 0;
    }
    if (n) HIPCHK(hipMemcpy(p->d_heightfields.p, p->h_heightfields.data(), n * sizeof(DevHeightField), hipMemcpyHostToDevice));
    collider_options_changed(p);
    return SPARKL_OK;
}
int sparkl_hip_colliders_set_trimesh_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* o) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    CHK(check_options(n, o, p->h_trimeshes.size(), "trimesh"));
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (uint32_t i = 0; i < n; ++i) {
        p->h_trimeshes[i].handling = o[i].grid_boundary_handling; p->h_trimeshes[i].penalty = o[i].penalty_stiffness; p->h_trimeshes[i].flip_interior = o[i].flip_interior ? 1 : 0;
    }
    if (n) HIPCHK(hipMemcpy(p->d_trimeshes.p, p->h_trimeshes.data(), n * sizeof(DevTriMesh), hipMemcpyHostToDevice));
    collider_options_changed(p);
    return SPARKL_OK;
}
int sparkl_hip_colliders_set_polylines(SparklPipeline* p, uint32_t n, const SparklPolylineCollider* c) {
    if (!p || (n && !c)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (c[i].num_vertices < 3 || !c[i].vertices) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "polyline %u: a closed polygon needs >= 3 vertices", i);
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (float* d : p->d_poly_verts) (void)hipFree(d);
    p->d_poly_verts.clear();
    std::vector<DevPolyline> dp(n);
    for (uint32_t i = 0; i < n; ++i) {
        float* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, (size_t)c[i].num_vertices * 3 * sizeof(float)));
        p->d_poly_verts.push_back(d);
        HIPCHK(hipMemcpy(d, c[i].vertices, (size_t)c[i].num_vertices * 3 * sizeof(float), hipMemcpyHostToDevice));
        dp[i].verts = d; dp[i].n = (int)c[i].num_vertices;
        for (int a = 0; a < 4; ++a) dp[i].q[a] = c[i].rotation[a];
        for (int a = 0; a < 3; ++a) dp[i].t[a] = c[i].translation[a];
        dp[i].friction = c[i].friction; dp[i].handling = SPARKL_COLLIDER_HANDLING_FROM_PARAMS; dp[i].penalty = 0.f; dp[i].flip_interior = 0;
    }
    CHK(p->d_polylines.reserve(std::max<uint32_t>(n, 1)));
    if (n) HIPCHK(hipMemcpy(p->d_polylines.p, dp.data(), n * sizeof(DevPolyline), hipMemcpyHostToDevice));
    p->num_polylines = (int)n;
    p->h_polylines = dp;
    collider_options_changed(p);
    return SPARKL_OK;
}
int sparkl_hip_colliders_set_polyline_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* o) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    CHK(check_options(n, o, p->h_polylines.size(), "polyline"));
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (uint32_t i = 0; i < n; ++i) {
        p->h_polylines[i].handling = o[i].grid_boundary_handling; p->h_polylines[i].penalty = o[i].penalty_stiffness; p->h_polylines[i].flip_interior = o[i].flip_interior ? 1 : 0;
    }
    if (n) HIPCHK(hipMemcpy(p->d_polylines.p, p->h_polylines.data(), n * sizeof(DevPolyline), hipMemcpyHostToDevice));
    collider_options_changed(p);
    return SPARKL_OK;
}
int sparkl_hip_colliders_set_cuboids(SparklPipeline* p, uint32_t n, const SparklCuboidCollider* c) {
    if (!p || (n && !c)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    std::vector<DevCuboid> dc(n);
    for (uint32_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) { dc[i].he[a] = c[i].half_extents[a]; dc[i].t[a] = c[i].translation[a]; }
        for (int a = 0; a < 4; ++a) dc[i].q[a] = c[i].rotation[a];
        dc[i].friction = c[i].friction;
    }
    for (auto& d : dc) { d.handling = SPARKL_COLLIDER_HANDLING_FROM_PARAMS; d.penalty = 0.f; }
    CHK(p->d_cuboids.reserve(std::max<uint32_t>(n, 1)));
    if (n) HIPCHK(hipMemcpyAsync(p->d_cuboids.p, dc.data(), n * sizeof(DevCuboid), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->num_cuboids = (int)n;
    p->h_cuboids = dc;
    collider_options_changed(p);
    return SPARKL_OK;
}

int sparkl_hip_colliders_set_heightfields(SparklPipeline* p, uint32_t n, const SparklHeightFieldCollider* c) {
    if (!p || (n && !c)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (uint32_t i = 0; i < n; ++i)
        if (c[i].nrows < 2 || c[i].ncols < 2 || !c[i].heights || c[i].scale[0] == 0.f || c[i].scale[2] == 0.f)
            return set_error(SPARKL_ERR_INVALID_ARGUMENT, "heightfield %u: needs >= 2x2 heights and a non-zero x/z scale", i);
    for (float* d : p->d_heights) (void)hipFree(d);
    p->d_heights.clear();
    std::vector<DevHeightField> dh(n);
    for (uint32_t i = 0; i < n; ++i) {
        const size_t cnt = (size_t)c[i].nrows * c[i].ncols;
        float* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, cnt * sizeof(float)));
        p->d_heights.push_back(d);
        HIPCHK(hipMemcpy(d, c[i].heights, cnt * sizeof(float), hipMemcpyHostToDevice));
        dh[i].heights = d; dh[i].nrows = (int)c[i].nrows; dh[i].ncols = (int)c[i].ncols;
        for (int a = 0; a < 3; ++a) { dh[i].scale[a] = c[i].scale[a]; dh[i].t[a] = c[i].translation[a]; }
        for (int a = 0; a < 4; ++a) dh[i].q[a] = c[i].rotation[a];
        dh[i].friction = c[i].friction; dh[i].below_is_solid = c[i].below_is_solid;
        dh[i].handling = SPARKL_COLLIDER_HANDLING_FROM_PARAMS; dh[i].penalty = 0.f; dh[i].flip_interior = 0;
        {
            float lo = c[i].heights[0] * c[i].scale[1], hi = lo;
            for (size_t k = 1; k < cnt; ++k) { const float y = c[i].heights[k] * c[i].scale[1]; lo = std::min(lo, y); hi = std::max(hi, y); }
            dh[i].ymin = lo; dh[i].ymax = hi;
            // (min, max) over blocks of 4 x 4 cells = 5 x 5 samples; cell (r, c) spans samples r..r+1, c..c+1
            const int nr = (int)c[i].nrows, nc = (int)c[i].ncols;
            const int cr = (nr - 1 + 3) / 4, cc = (nc - 1 + 3) / 4;
            std::vector<float2> coarse((size_t)cr * cc);
            for (int bj = 0; bj < cc; ++bj)
                for (int bi = 0; bi < cr; ++bi) {
                    float l = std::numeric_limits<float>::max(), h2 = -l;
                    for (int cj = 4 * bj; cj <= std::min(4 * bj + 4, nc - 1); ++cj)
                        for (int ri = 4 * bi; ri <= std::min(4 * bi + 4, nr - 1); ++ri) {
                            const float y = c[i].heights[ri + (size_t)nr * cj] * c[i].scale[1];
                            l = std::min(l, y); h2 = std::max(h2, y);
                        }
                    coarse[bi + (size_t)cr * bj] = make_float2(l, h2);
                }
            float2* dc = nullptr;
            HIPCHK(hipMalloc((void**)&dc, coarse.size() * sizeof(float2)));
            p->d_heights.push_back(reinterpret_cast<float*>(dc));     // freed with the height arrays
            HIPCHK(hipMemcpy(dc, coarse.data(), coarse.size() * sizeof(float2), hipMemcpyHostToDevice));
            dh[i].coarse = dc; dh[i].crows = cr; dh[i].ccols = cc;
        }
    }
    CHK(p->d_heightfields.reserve(std::max<uint32_t>(n, 1)));
    if (n) HIPCHK(hipMemcpy(p->d_heightfields.p, dh.data(), n * sizeof(DevHeightField), hipMemcpyHostToDevice));
    p->num_heightfields = (int)n;
    p->h_heightfields = dh;
    collider_options_changed(p);
    return SPARKL_OK;
}

// Triangle meshes: pseudo-normals (Baerentzen & Aanaes, as parry's TriMesh::compute_pseudo_normals: angle-weighted on
// vertices, summed face normals on edges) and a uniform grid over the bounding box, both built here on the host.
int sparkl_hip_colliders_set_trimeshes(SparklPipeline* p, uint32_t n, const SparklTriMeshCollider* c) {
    if (!p || (n && !c)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (uint32_t i = 0; i < n; ++i) {
        if (!c[i].num_triangles || !c[i].num_vertices || !c[i].vertices || !c[i].indices)
            return set_error(SPARKL_ERR_INVALID_ARGUMENT, "trimesh %u: empty mesh", i);
        for (uint64_t k = 0; k < 3ull * c[i].num_triangles; ++k)
            if (c[i].indices[k] >= c[i].num_vertices) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "trimesh %u: index %u out of range", i, c[i].indices[k]);
    }
    for (void* d : p->d_mesh_bufs) (void)hipFree(d);
    p->d_mesh_bufs.clear();
    const float reach = 4.0f * p->h;
    std::vector<DevTriMesh> dm(n);
    auto to_device = [&](const void* src, size_t bytes, const void** out) -> int {
        void* d = nullptr;
        HIPCHK(hipMalloc(&d, std::max<size_t>(bytes, 4)));
        p->d_mesh_bufs.push_back(d);
        if (bytes) HIPCHK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *out = d;
        return SPARKL_OK;
    };
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t nt = c[i].num_triangles, nv = c[i].num_vertices;
        const float* V = c[i].vertices;
        std::vector<float> tris((size_t)nt * 30, 0.f);
        std::vector<double> vn((size_t)nv * 3, 0.0);
        std::map<std::pair<uint32_t, uint32_t>, std::array<double, 3>> en;
        std::vector<std::array<double, 3>> fn(nt, std::array<double, 3>{{0, 0, 0}});
        float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        auto sub = [](const float* a, const float* b, double* o) { for (int k = 0; k < 3; ++k) o[k] = (double)a[k] - (double)b[k]; };
        auto angle = [](const double* u, const double* v) {
            const double nu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), nvv = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (nu == 0.0 || nvv == 0.0) return 0.0;
            return std::acos(std::max(-1.0, std::min(1.0, (u[0] * v[0] + u[1] * v[1] + u[2] * v[2]) / (nu * nvv))));
        };
        for (uint32_t t = 0; t < nt; ++t) {
            const uint32_t* id = c[i].indices + 3 * t;
            const float *a = V + 3 * id[0], *b = V + 3 * id[1], *cc = V + 3 * id[2];
            for (int k = 0; k < 3; ++k) {
                tris[30 * t + k] = a[k]; tris[30 * t + 3 + k] = b[k]; tris[30 * t + 6 + k] = cc[k];
                lo[k] = std::min(lo[k], std::min(a[k], std::min(b[k], cc[k]))); hi[k] = std::max(hi[k], std::max(a[k], std::max(b[k], cc[k])));
            }
            if (!c[i].oriented) continue;
            double ab[3], ac[3], bc[3], ba[3], ca[3], cb[3];
            sub(b, a, ab); sub(cc, a, ac); sub(cc, b, bc); sub(a, b, ba); sub(a, cc, ca); sub(b, cc, cb);
            double nrm[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
            const double l = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
            if (l == 0.0) continue;   // degenerate triangle: no normal (Triangle::normal() == None)
            for (int k = 0; k < 3; ++k) nrm[k] /= l;
            fn[t] = {{nrm[0], nrm[1], nrm[2]}};
            const double ang[3] = {angle(ab, ac), angle(ba, bc), angle(ca, cb)};
            for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) vn[3 * id[v] + k] += nrm[k] * ang[v];
            const uint32_t e[3][2] = {{id[0], id[1]}, {id[0], id[2]}, {id[1], id[2]}};
            for (int q = 0; q < 3; ++q) {
                auto key = std::make_pair(std::min(e[q][0], e[q][1]), std::max(e[q][0], e[q][1]));
                auto& acc = en[key];
                for (int k = 0; k < 3; ++k) acc[k] += nrm[k];
            }
        }
        if (c[i].oriented)
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t* id = c[i].indices + 3 * t;
                for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) tris[30 * t + 9 + 3 * v + k] = (float)vn[3 * id[v] + k];
                const uint32_t e[3][2] = {{id[0], id[1]}, {id[0], id[2]}, {id[1], id[2]}};
                for (int q = 0; q < 3; ++q) {
                    const auto& acc = en[std::make_pair(std::min(e[q][0], e[q][1]), std::max(e[q][0], e[q][1]))];
                    for (int k = 0; k < 3; ++k) tris[30 * t + 18 + 3 * q + k] = (float)acc[k];
                }
                for (int k = 0; k < 3; ++k) tris[30 * t + 27 + k] = (float)fn[t][k];
            }
        // uniform grid over the box grown by the reach
        DevTriMesh& d = dm[i];
        float ext = 0.f;
        for (int k = 0; k < 3; ++k) { d.lo[k] = lo[k] - reach; d.hi[k] = hi[k] + reach; ext = std::max(ext, d.hi[k] - d.lo[k]); }
        const float cell = std::max(2.0f * p->h, ext / 64.0f);
        d.inv_cell = 1.0f / cell;
        size_t ncell = 1;
        for (int k = 0; k < 3; ++k) { d.dims[k] = std::max(1, (int)std::ceil((d.hi[k] - d.lo[k]) / cell)); ncell *= (size_t)d.dims[k]; }
        std::vector<int> count(ncell + 1, 0);
        auto cell_range = [&](uint32_t t, int r0[3], int r1[3]) {
            for (int k = 0; k < 3; ++k) {
                const float mn = std::min(tris[30 * t + k], std::min(tris[30 * t + 3 + k], tris[30 * t + 6 + k])) - reach;
                const float mx = std::max(tris[30 * t + k], std::max(tris[30 * t + 3 + k], tris[30 * t + 6 + k])) + reach;
                r0[k] = std::max(0, std::min(d.dims[k] - 1, (int)std::floor((mn - d.lo[k]) * d.inv_cell)));
                r1[k] = std::max(0, std::min(d.dims[k] - 1, (int)std::floor((mx - d.lo[k]) * d.inv_cell)));
            }
        };
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<int> fill;
            std::vector<int> list;
            if (pass == 1) {
                for (size_t k = 0; k < ncell; ++k) count[k + 1] += count[k];   // count[k + 1] held the size of cell k
                fill.assign(count.begin(), count.end() - 1);
                list.resize((size_t)count[ncell]);
            }
            for (uint32_t t = 0; t < nt; ++t) {
                int r0[3], r1[3];
                cell_range(t, r0, r1);
                for (int z = r0[2]; z <= r1[2]; ++z) for (int y = r0[1]; y <= r1[1]; ++y) for (int x = r0[0]; x <= r1[0]; ++x) {
                    const size_t ci = (size_t)x + (size_t)d.dims[0] * ((size_t)y + (size_t)d.dims[1] * (size_t)z);
                    if (pass == 0) count[ci + 1]++; else list[(size_t)fill[ci]++] = (int)t;
                }
            }
            if (pass == 1) {
                CHK(to_device(tris.data(), tris.size() * sizeof(float), (const void**)&d.tris));
                CHK(to_device(count.data(), count.size() * sizeof(int), (const void**)&d.cell_start));
                CHK(to_device(list.data(), list.size() * sizeof(int), (const void**)&d.cell_tris));
            }
        }
        for (int a = 0; a < 3; ++a) d.t[a] = c[i].translation[a];
        for (int a = 0; a < 4; ++a) d.q[a] = c[i].rotation[a];
        d.friction = c[i].friction; d.oriented = c[i].oriented; d.flip_interior = c[i].flip_interior;
        d.handling = SPARKL_COLLIDER_HANDLING_FROM_PARAMS; d.penalty = 0.f;
    }
    CHK(p->d_trimeshes.reserve(std::max<uint32_t>(n, 1)));
    if (n) HIPCHK(hipMemcpy(p->d_trimeshes.p, dm.data(), n * sizeof(DevTriMesh), hipMemcpyHostToDevice));
    p->num_trimeshes = (int)n;
    p->h_trimeshes = dm;
    collider_options_changed(p);
    return SPARKL_OK;
}

// --------------------------------------------------------------------------------------- particles
static int upload_range(SparklPipeline* p, uint64_t first, uint64_t n, const SparklParticleStatus* status, const SparklParticlePosition* pos,
                        const SparklParticleVelocity* vel, const SparklParticleVolume* volume, const SparklParticlePhase* phase,
                        const SparklParticleData* data, const SparklVelocityGradient* vg) {
    // the caller's arrays go to the device as they are (160 B per particle) and are packed into the float4 streams there
    static_assert(sizeof(SparklParticleStatus) == 24 && sizeof(SparklParticlePosition) == 12 && sizeof(SparklParticleVelocity) == 12 &&
                  sizeof(SparklParticleVolume) == 52 && sizeof(SparklParticlePhase) == 8 && sizeof(SparklParticleData) == 16 && sizeof(SparklVelocityGradient) == 36, "");
    for (uint64_t i = 0; i < n; ++i) p->max_model_index = std::max(p->max_model_index, (uint32_t)status[i].model_index);
    for (uint64_t i = 0; i < n && p->all_jp_one; ++i) p->all_jp_one = volume[i].plastic_deformation_gradient_det == 1.0f;
    hipStream_t st = p->stream;
    // chunks bound the staging area (40 floats per particle) for very large sets
    const uint64_t CHUNK = 1ull << 24;
    CHK(p->staging.reserve((size_t)std::min(n, CHUNK) * 40));
    for (uint64_t c0 = 0; c0 < n; c0 += CHUNK) {
        const uint64_t m = std::min(CHUNK, n - c0);
        float* base = p->staging.p;
        uint32_t* d_status = (uint32_t*)base;            // 6 words
        float* d_pos = base + 6 * m;                     // 3
        float* d_vel = d_pos + 3 * m;                    // 3
        float* d_vol = d_vel + 3 * m;                    // 13
        float* d_phase = d_vol + 13 * m;                 // 2
        float* d_data = d_phase + 2 * m;                 // 4
        float* d_vg = d_data + 4 * m;                    // 9
        HIPCHK(hipMemcpyAsync(d_status, status + c0, m * 24, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_pos, pos + c0, m * 12, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_vel, vel + c0, m * 12, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_vol, volume + c0, m * 52, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_phase, phase + c0, m * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_data, data + c0, m * 16, hipMemcpyHostToDevice, st));
        if (vg) HIPCHK(hipMemcpyAsync(d_vg, vg + c0, m * 36, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_pack_upload, dim3(grid_for(m, 256)), dim3(256), 0, st, p->cur.arrays(), (int)(first + c0), (int)m, (uint32_t)p->id_offset, d_status, d_pos, d_vel,
                           d_vol, d_phase, d_data, vg ? d_vg : (const float*)nullptr);
        HIPCHK(hipStreamSynchronize(st));   // the staging area is reused by the next chunk (and by the read paths)
    }
    if (p->cur.has_frac)   // appended particles get ParticleFracture::default() (particle.rs:187-194)
        hipLaunchKernelGGL(k_fracture, dim3(grid_for(n, 256)), dim3(256), 0, st, p->cur.arrays(), (float2*)nullptr, (int)first, (int)n, -1ll, 2);
    return SPARKL_OK;
}

int sparkl_hip_particles_upload(SparklPipeline* p, uint64_t n, const SparklParticleStatus* status, const SparklParticlePosition* pos,
                                const SparklParticleVelocity* vel, const SparklParticleVolume* volume, const SparklParticlePhase* phase,
                                const SparklParticleData* data, const SparklVelocityGradient* vg) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (n && (!status || !pos || !vel || !volume || !phase || !data)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null particle array");
    if (n >= (1ull << 31) || p->id_offset + n > 0xFFFFFFFFull) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "at most 2^31 - 1 particles per device, 2^32 ids per job");
    for (uint64_t i = 0; i < n; ++i)
        if (status[i].model_index >= (1ull << 24)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "model_index too large");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->cur.has_frac = false;
    CHK(p->cur.reserve(std::max<uint64_t>(n, 1)));
    p->n = n;
    p->n_binned = 0;   // world coordinates until the first re-bin
    p->max_model_index = 0;
    p->all_jp_one = true;
    if (n) CHK(upload_range(p, 0, n, status, pos, vel, volume, phase, data, vg));
    p->sorted = false;
    return SPARKL_OK;
}

uint64_t sparkl_hip_particles_len(const SparklPipeline* p) { return p ? p->n : 0; }

// id_base >= 0: scatter to out[id - id_base] (caller order); id_base < 0: device (sorted) order
static BinnedView binned_view(const SparklPipeline* p) { return BinnedView{p->pb_info.p, p->num_pblocks, p->num_pblocks > 0 ? (int)p->n_binned : 0, p->h}; }
static int launch_read(SparklPipeline* p, int what, size_t floats_per, long long id_base) {
    const int n = (int)p->n;
    CHK(p->staging.reserve(std::max<size_t>((size_t)n * floats_per, 1)));
    const ParticleArrays P = p->cur.arrays();
    hipStream_t st = p->stream;
    const dim3 g(grid_for(n, 256)), b(256);
    switch (what) {
        case 0: hipLaunchKernelGGL(k_read_positions, g, b, 0, st, P, p->staging.p, n, id_base, binned_view(p)); break;
        case 1: hipLaunchKernelGGL(k_read_velocities, g, b, 0, st, P, p->staging.p, n, id_base); break;
        default: hipLaunchKernelGGL(k_read_velgrad, g, b, 0, st, P, p->staging.p, n, id_base); break;
    }
    return SPARKL_OK;
}
static long long read_base(const SparklPipeline* p) { return p->slab_on ? -1ll : (long long)p->id_offset; }
#define READ_PROLOGUE                                                                                      \
    if (!p || (n && !out)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");                 \
    if (n != p->n) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "n (%llu) != particle count (%llu)", (unsigned long long)n, (unsigned long long)p->n); \
    HIPCHK(hipSetDevice(p->device));                                                                      \
    if (n == 0) return SPARKL_OK;

int sparkl_hip_particles_read_positions(SparklPipeline* p, SparklParticlePosition* out, uint64_t n) {
    READ_PROLOGUE
    CHK(launch_read(p, 0, 3, read_base(p)));
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * 12, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}
int sparkl_hip_particles_read_velocities(SparklPipeline* p, SparklParticleVelocity* out, uint64_t n) {
    READ_PROLOGUE
    CHK(launch_read(p, 1, 3, read_base(p)));
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * 12, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}
static int read_struct(SparklPipeline* p, void* out, uint64_t n, int what, size_t words) {
    CHK(p->staging.reserve(n * words));
    hipLaunchKernelGGL(k_read_struct, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (uint32_t*)p->staging.p, (int)n, read_base(p), what);
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * words * 4, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}
int sparkl_hip_particles_read_volumes(SparklPipeline* p, SparklParticleVolume* out, uint64_t n) {
    READ_PROLOGUE
    static_assert(sizeof(SparklParticleVolume) == 52, "");
    return read_struct(p, out, n, 0, 13);
}
int sparkl_hip_particles_read_statuses(SparklPipeline* p, SparklParticleStatus* out, uint64_t n) {
    READ_PROLOGUE
    static_assert(sizeof(SparklParticleStatus) == 24, "");
    return read_struct(p, out, n, 1, 6);
}
int sparkl_hip_particles_read_phases(SparklPipeline* p, SparklParticlePhase* out, uint64_t n) {
    READ_PROLOGUE
    CHK(p->staging.reserve(n * 2));
    hipLaunchKernelGGL(k_read_render, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (float*)nullptr, (float*)nullptr, p->staging.p, (int)n, read_base(p), binned_view(p));
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * sizeof(SparklParticlePhase), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}
int sparkl_hip_particles_read_data(SparklPipeline* p, SparklParticleData* out, uint64_t n) {
    READ_PROLOGUE
    static_assert(sizeof(SparklParticleData) == 16, "");
    return read_struct(p, out, n, 2, 4);
}
int sparkl_hip_particles_read_velocity_gradients(SparklPipeline* p, SparklVelocityGradient* out, uint64_t n) {
    READ_PROLOGUE
    CHK(launch_read(p, 3, 9, read_base(p)));
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * 36, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}
int sparkl_hip_particles_read_sorted_ids(SparklPipeline* p, uint32_t* out, uint64_t n) {
    READ_PROLOGUE
    CHK(p->staging.reserve(n));
    hipLaunchKernelGGL(k_read_ids, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (uint32_t*)p->staging.p, (int)n);
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}

// ---- ParticleFracture (src_core/dynamics/particle.rs:180-195): optional component, caller order
int sparkl_hip_particles_set_fracture(SparklPipeline* p, const SparklParticleFracture* fracture, uint64_t n) {
    if (!p || (n && !fracture)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (n != p->n) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "n (%llu) != particle count (%llu)", (unsigned long long)n, (unsigned long long)p->n);
    // a slab numbers its particles id_offset .. id_offset + n in upload order; once a collective re-bin has migrated particles the
    // local set is no longer that range (migration records carry the component along: k_pack_particles)
    if (p->slab_on && p->group_rebins > 0) return set_error(SPARKL_ERR_UNSUPPORTED, "ParticleFracture of a slab pipeline must be set before the first collective re-bin");
    HIPCHK(hipSetDevice(p->device));
    if (n == 0) return SPARKL_OK;
    static_assert(sizeof(SparklParticleFracture) == 8, "");
    HIPCHK(hipStreamSynchronize(p->stream));
    p->cur.has_frac = true;
    CHK(p->cur.frac.reserve(p->cur.q[0].cap));
    CHK(p->staging.reserve(n * 2));
    HIPCHK(hipMemcpyAsync(p->staging.p, fracture, n * sizeof(SparklParticleFracture), hipMemcpyHostToDevice, p->stream));
    hipLaunchKernelGGL(k_fracture, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (float2*)p->staging.p, 0, (int)n, (long long)p->id_offset, 0);
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}
/* 1 when the caller has uploaded a ParticleFracture component (sparkl_hip_particles_set_fracture; slab groups under ModifiedEigenerosion
 * create a default one themselves), 0 when every particle still holds the default */
int sparkl_hip_particles_has_fracture(const SparklPipeline* p) { return p && p->cur.has_frac ? 1 : 0; }
int sparkl_hip_particles_read_fracture(SparklPipeline* p, SparklParticleFracture* out, uint64_t n) {
    READ_PROLOGUE
    if (!p->cur.has_frac) {   // never uploaded: every particle holds the default
        for (uint64_t i = 0; i < n; ++i) { out[i].crack_propagation_factor = 0.f; out[i].crack_threshold = std::numeric_limits<float>::max(); }
        return SPARKL_OK;
    }
    CHK(p->staging.reserve(n * 2));
    hipLaunchKernelGGL(k_fracture, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (float2*)p->staging.p, 0, (int)n, read_base(p), 1);
    HIPCHK(hipMemcpyAsync(out, p->staging.p, n * sizeof(SparklParticleFracture), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}

// ---- asynchronous render readback (testbed_plugin.rs:533-588)
int sparkl_hip_host_alloc(void** out, uint64_t bytes) {
    if (!out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    return SPARKL_OK;
}
int sparkl_hip_host_free(void* ptr) {
    if (ptr) HIPCHK(hipHostFree(ptr));
    return SPARKL_OK;
}
int sparkl_hip_readback_begin(SparklPipeline* p, SparklParticlePosition* pos, SparklParticleVelocity* vel, SparklParticlePhase* phase, uint64_t n) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (n != p->n) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "n (%llu) != particle count (%llu)", (unsigned long long)n, (unsigned long long)p->n);
    HIPCHK(hipSetDevice(p->device));
    if (n == 0 || (!pos && !vel && !phase)) return SPARKL_OK;
    if (!p->rb_stream) {
        HIPCHK(hipStreamCreateWithFlags(&p->rb_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&p->rb_gathered, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p->rb_done, hipEventDisableTiming));
    }
    // the gather may not overwrite the staging area while an earlier copy still reads it
    if (p->rb_pending) HIPCHK(hipStreamWaitEvent(p->stream, p->rb_done, 0));
    if ((size_t)n * 8 > p->rb_staging.cap && p->rb_pending) HIPCHK(hipEventSynchronize(p->rb_done));   // growing frees the old area
    CHK(p->rb_staging.reserve((size_t)n * 8));
    float* d_pos = p->rb_staging.p, *d_vel = d_pos + 3 * n, *d_phase = d_vel + 3 * n;
    hipLaunchKernelGGL(k_read_render, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), pos ? d_pos : nullptr, vel ? d_vel : nullptr,
                       phase ? d_phase : nullptr, (int)n, read_base(p), binned_view(p));
    HIPCHK(hipEventRecord(p->rb_gathered, p->stream));
    HIPCHK(hipStreamWaitEvent(p->rb_stream, p->rb_gathered, 0));
    if (pos) HIPCHK(hipMemcpyAsync(pos, d_pos, n * sizeof(SparklParticlePosition), hipMemcpyDeviceToHost, p->rb_stream));
    if (vel) HIPCHK(hipMemcpyAsync(vel, d_vel, n * sizeof(SparklParticleVelocity), hipMemcpyDeviceToHost, p->rb_stream));
    if (phase) HIPCHK(hipMemcpyAsync(phase, d_phase, n * sizeof(SparklParticlePhase), hipMemcpyDeviceToHost, p->rb_stream));
    HIPCHK(hipEventRecord(p->rb_done, p->rb_stream));
    p->rb_pending = true;
    return SPARKL_OK;
}
int sparkl_hip_readback_wait(SparklPipeline* p) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (!p->rb_pending) return SPARKL_OK;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventSynchronize(p->rb_done));
    p->rb_pending = false;
    return SPARKL_OK;
}

// ---- append / remove_range / truncate: device side, the resident particles never travel to the host
int sparkl_hip_particles_append(SparklPipeline* p, uint64_t n, const SparklParticleStatus* status, const SparklParticlePosition* pos,
                                const SparklParticleVelocity* vel, const SparklParticleVolume* volume, const SparklParticlePhase* phase,
                                const SparklParticleData* data, const SparklVelocityGradient* vg) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (n == 0) return SPARKL_OK;
    if (!status || !pos || !vel || !volume || !phase || !data) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null particle array");
    if (p->slab_on) return set_error(SPARKL_ERR_UNSUPPORTED, "append on a slab pipeline: ids are global there, add particles before sparkl_hip_slab_set");
    if (p->n + n >= (1ull << 31) || p->id_offset + p->n + n > 0xFFFFFFFFull) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "at most 2^31 - 1 particles per device");
    for (uint64_t i = 0; i < n; ++i)
        if (status[i].model_index >= (1ull << 24)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "model_index too large");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    // `CudaVec::append` (cuda_vec.rs:56-81): grow (x1.5, old contents kept), write the new records behind the old ones; the
    // resident particles never leave the device.  The next step re-bins.
    CHK(p->cur.reserve(p->n + n, true, p->stream));
    CHK(upload_range(p, p->n, n, status, pos, vel, volume, phase, data, vg));
    p->n += n;
    p->sorted = false;
    return SPARKL_OK;
}
int sparkl_hip_particles_remove_range(SparklPipeline* p, uint64_t start, uint64_t end) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (start > end || end > p->n) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad range [%llu, %llu) of %llu", (unsigned long long)start, (unsigned long long)end, (unsigned long long)p->n);
    if (start == end) return SPARKL_OK;
    if (p->slab_on) return set_error(SPARKL_ERR_UNSUPPORTED, "remove_range on a slab pipeline");
    HIPCHK(hipSetDevice(p->device));
    hipLaunchKernelGGL(k_remove_ids, dim3(grid_for(p->n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (int)p->n, (uint32_t)p->id_offset, (uint32_t)start, (uint32_t)end);
    p->sorted = false;
    return resort(p);   // compacts: the removed records sort behind every live one and p->n drops
}
int sparkl_hip_particles_truncate(SparklPipeline* p, uint64_t new_len) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (new_len >= p->n) return SPARKL_OK;
    return sparkl_hip_particles_remove_range(p, new_len, p->n);
}

// ------------------------------------------------------------------------------------------ stepping
// DamageModel::{CdMpm, Eigenerosion} (mls_solver.rs:68-83) are not built: refuse instead of silently stepping with
// DamageModel::None.  ModifiedEigenerosion is built for the single-pipeline entry points (enqueue_psi_sweep); the split-phase
// and slab-group entry points refuse it (the psi planes are not exchanged across cuts).
static int unsupported_damage_model(const SparklSolverParams* params) {
    return set_error(SPARKL_ERR_UNSUPPORTED, "damage_model %d is not implemented (SPARKL_DAMAGE_NONE and SPARKL_DAMAGE_MODIFIED_EIGENEROSION are)", (int)params->damage_model);
}

static int ensure_ready(SparklPipeline* p) {
    HIPCHK(hipSetDevice(p->device));
    CHK(validate_models(p));
    if (!p->all_plain_fluid) p->all_jp_one = false;   // about to step under models that may move det Fp (see all_jp_one)
    if (!p->sorted) CHK(resort(p));
    return SPARKL_OK;
}

int sparkl_hip_pipeline_substeps(SparklPipeline* p, const SparklSolverParams* params, const float gravity[3], float dt, uint32_t nsubsteps) {
    if (!p || !params || !gravity) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (p->slab_on && (p->slab_lo != INT32_MIN || p->slab_hi != INT32_MAX))
        return set_error(SPARKL_ERR_INVALID_ARGUMENT, "a slab with finite cuts must be stepped through substep_begin/halo_*/substep_finish");
    if (!damage_supported(params)) return unsupported_damage_model(params);
    if (p->n == 0 || nsubsteps == 0) return SPARKL_OK;
    CHK(ensure_ready(p));
    // the step counter this call starts from: the host's copy if nothing has stepped since the last call returned (every path that runs a
    // substep advances substep_index), else one read-back -- a stream sync of its own before the first launch (~0.15 ms of the driver's
    // 20-substep window)
    if (!(p->ctl_fresh && p->ctl_fresh_index == p->substep_index)) CHK(read_ctl(p));
    p->ctl_fresh = false;
    const uint32_t base_done = p->h_ctl->steps_done;
    uint32_t done = 0;
    // fixed dt: G2P of substep s and P2G of substep s+1 run as one kernel (see k_g2p2g); a batch pays one standalone
    // P2G + G2P, so fused batches are longer.  After a re-bin request the rest of a batch is no-op launches (~3 us each).
    const bool fluid_sweep = params->force_fluids_volume_recomputation && p->any_fluid;
    const bool fuse = p->fuse_fixed_dt && !fluid_sweep && !psi_sweep_on(p, params);
    const bool fuse_fluid = p->fuse_fixed_dt && fluid_sweep && !p->rare_models && !psi_sweep_on(p, params);
    const uint32_t CHUNK = fuse ? 128 : 32;
    p->mass_tiles_ready = false;   // only trusted from one batch of this call to the next
    int guard = 0;
    while (done < nsubsteps) {
        const uint32_t batch = std::min(CHUNK, nsubsteps - done);
        if (fuse) CHK(enqueue_fused_batch(p, params, gravity, dt, batch));
        else if (fuse_fluid) CHK(enqueue_fluid_batch(p, params, gravity, dt, batch));
        else for (uint32_t i = 0; i < batch; ++i) CHK(enqueue_substep(p, params, gravity, dt));
        HIPCHK(hipGetLastError());   // a launch of this batch that failed (bad configuration, out of resources) surfaces here, not as a stalled re-bin loop
        CHK(read_ctl(p));
        if (p->clock_on) CHK(clock_collect(p));
        if (getenv("SPARKL_HIP_DEBUG"))
            fprintf(stderr, "[sparkl_hip] batch %u: steps_done %u (base %u) abort %u %u invalid %u n %llu pblocks %d blocks %d parity-next %u\n", batch, p->h_ctl->steps_done,
                    base_done, p->h_ctl->abort_flag[0], p->h_ctl->abort_flag[1], p->h_ctl->num_invalid, (unsigned long long)p->n, p->num_pblocks, p->num_blocks, p->substep_index & 1u);
        const uint32_t now_done = p->num_pblocks ? (p->h_ctl->steps_done - base_done) : done + batch;
        const bool aborted = now_done < done + batch;
        // cut short behind a fused launch that deferred C.m[1..8]: the last substep of a batch that runs to its end stores
        // everything (plain k_g2p, or k_g2p2g without the deferral when the tiles are carried over)
        if (fuse && aborted && now_done > done && p->defer_c && !p->rare_fused() && !params->enable_boundary_particle_projection) p->c_stale = true;
        done = now_done;
        if (aborted || p->h_ctl->abort_flag[0] || p->h_ctl->abort_flag[1]) {
            CHK(resort(p));  // a particle left its tile margin: re-bin, then carry on from `done`
            if (++guard > (int)nsubsteps * 2 + 16) return set_error(SPARKL_ERR_INTERNAL, "re-bin loop did not make progress");
        }
    }
    p->clock.substeps += nsubsteps;
    p->ctl_fresh = true; p->ctl_fresh_index = p->substep_index;   // the loop's last action was read_ctl (a re-bin behind it leaves steps_done alone)
    return SPARKL_OK;
}

int sparkl_hip_pipeline_step(SparklPipeline* p, const SparklSolverParams* params, const float gravity[3], SparklTimings* timings, float* simulated_time) {
    if (!p || !params || !gravity) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (timings) memset(timings, 0, sizeof(*timings));
    if (p->slab_on && (p->slab_lo != INT32_MIN || p->slab_hi != INT32_MAX))
        return set_error(SPARKL_ERR_UNSUPPORTED, "adaptive step on a slab with finite cuts: the dt bound is per rank; drive fixed-length substeps through substep_begin/finish");
    if (!damage_supported(params)) return unsupported_damage_model(params);
    // mls_solver.rs:25-28
    if (p->n == 0 || params->dt == 0.0f) { if (simulated_time) *simulated_time = params->dt; return SPARKL_OK; }
    CHK(ensure_ready(p));
    const bool was_clock = p->clock_on;
    SparklKernelClock saved = p->clock;
    const uint32_t saved_stride = p->clock_stride;
    p->sublog_on = false;
    if (timings) {
        p->clock_on = true; memset(&p->clock, 0, sizeof(p->clock));
        p->clock_stride = 1; p->sublog_on = true; p->sublog_base = p->substep_index; p->sublog.clear();
    }
    auto note_dt = [&](uint32_t index, float d) {   // the step length of an EXECUTED substep (per-substep log)
        if (!timings) return;
        const uint32_t k = index - p->sublog_base;
        if (k >= p->sublog.size()) p->sublog.resize(k + 1, SparklSubstepTimings{});
        p->sublog[k].dt = d;
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timings) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, p->stream)); }
    const float min_dt = params->dt / (float)params->max_num_substeps;  // mls_solver.rs:33
    float remaining = params->dt;
    uint32_t nsub = 0;
    float result = params->dt;
    int rc = SPARKL_OK;
    bool have_bound = false;   // after the first substep, G2P leaves the next bound in ctl->dt_bits (no extra pass over the particles)
    const float fmax = std::numeric_limits<float>::max();
    uint32_t init_bits;
    memcpy(&init_bits, &fmax, 4);
    // Speculative fusion.  The P2G of substep s+1 depends on dt_{s+1}, which is only known after G2P of substep s -- but in
    // the usual regime (stiffness-bound: alpha h / c_sound, the same f32 number every substep) it can be predicted exactly:
    // dt' = min(remaining, max_substep_dt, bound_s).  While two consecutive bounds agree bit for bit, G2P runs fused with the
    // scatter for dt'; if the next substep then asks for another step length the tiles are simply scattered again.
    // The substep sequence is the reference's in every case.
    CHK(p->dt_partial.reserve(std::max<size_t>(grid_for(p->n, BLK_THREADS), (size_t)std::max(p->num_pblocks, 1))));
    const bool may_fuse = p->fuse_fixed_dt && !(params->force_fluids_volume_recomputation && p->any_fluid) && !params->stop_after_one_substep && !psi_sweep_on(p, params);
    bool spec_tiles = false;
    uint32_t spec_dt_bits = 0;
    // mls_solver.rs:46-48: with force_fluids_volume_recomputation the fluid volumes are recomputed after the sort and BEFORE
    // the dt estimate of every substep, so the bound G2P left behind (computed from the volumes before the recompute) is not
    // the reference's: the estimator pass runs every substep in that mode.
    const bool fluid_volumes = params->force_fluids_volume_recomputation && p->any_fluid;
    // With force_fluids_volume_recomputation the step length of a substep is only known after ITS volume sweep, which sits in
    // front of ITS scatter -- the unfused order costs seven kernels (2.5 ms at 20M particles).  The fused sweep of the fixed-dt
    // batches (k_g2p_mass / k_vol_p2g) is used here too: the scatter is built for the PREDICTED length min(remaining,
    // max_substep_dt, last bound), the estimator runs right behind the volume gather and k_dt_check compares; a wrong prediction
    // no-ops the rest of the substep, which is then repeated in the reference's order.  Still one host read per substep.
    const bool fluid_fused = fluid_volumes && p->fuse_fixed_dt && !p->rare_models && !psi_sweep_on(p, params) && !(getenv("SPARKL_HIP_FLUID_STEP_FUSE") && !strcmp(getenv("SPARKL_HIP_FLUID_STEP_FUSE"), "0"));
    if (fluid_fused) CHK(read_ctl(p));   // steps_done as of now
    uint32_t pending = 0;                // substeps enqueued since the last read of the control block (they always run: the read before them found no re-bin request)
    while (remaining > 0.0f) {
        // ... and only when the prediction does not hinge on the bound itself: a step limited by `remaining` or max_substep_dt survives a
        // bound that moves a little, a bound-limited one is refuted whenever the bound changes in its last bit (every substep of an
        // accelerating flow), and a refuted attempt costs 60 % of a substep
        float d = p->fluid_bound_valid ? std::fmin(std::fmin(remaining, params->max_substep_dt), p->fluid_bound) : 0.f;
        if (fluid_fused && p->fluid_bound_valid && d < p->fluid_bound && p->num_pblocks > 0 && !(p->h_ctl->abort_flag[0] || p->h_ctl->abort_flag[1])) {
            if (d < min_dt && remaining > min_dt) d = min_dt;
            const float chk[2] = {remaining, min_dt};
            const uint32_t idx0 = p->substep_index, done0 = p->h_ctl->steps_done + pending;
            pending = 0;
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->dt_bits, (int)init_bits, 1, p->stream));
            CHK(enqueue_fluid_batch(p, params, gravity, d, 1, chk));
            HIPCHK(hipGetLastError());
            CHK(read_ctl(p));
            if (p->h_ctl->steps_done == done0 + 1) {          // the substep ran: the prediction held
                memcpy(&p->fluid_bound, &p->h_ctl->dt_bits, 4);
                p->fluid_fused_substeps++;
                note_dt(idx0, d);
                ++nsub;
                remaining -= d;
                have_bound = false;
                if (p->h_ctl->abort_flag[0] || p->h_ctl->abort_flag[1]) { CHK(resort(p)); CHK(read_ctl(p)); }   // it also asked for a re-bin
                if (params->stop_after_one_substep) { result = d; break; }
                continue;
            }
            p->substep_index = idx0;
            p->mass_tiles_ready = false;
            if (!p->h_ctl->dt_mismatch) {                     // a re-bin request stopped it
                CHK(resort(p));
                CHK(read_ctl(p));
                continue;
            }
            // wrong prediction (or a NaN bound): clear the chain, take this substep in the reference's order
            p->fluid_fused_mismatches++;
            if (p->h_words) p->h_words[0] = 0u;   // (the stream is idle behind read_ctl: nothing in flight can raise it again)
            HIPCHK(hipMemsetAsync(p->d_ctl, 0, 2 * sizeof(uint32_t), p->stream));
            HIPCHK(hipMemsetAsync(&p->d_ctl->dt_mismatch, 0, sizeof(uint32_t), p->stream));
            HIPCHK(hipMemsetAsync(&p->d_ctl->dt_nan, 0, sizeof(uint32_t), p->stream));
        }
        // one host sync per substep, as in the reference (cuda_mpm_pipeline.rs:393-398)
        if (fluid_volumes) have_bound = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (fluid_volumes && p->num_pblocks > 0) {
                const float g0[3] = {gravity[0], gravity[1], gravity[2]};
                CHK(enqueue_fluid_volumes(p, make_params(p, params, g0, 0.f)));
            }
            if (!have_bound) {
                size_t tok = 0;
                HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->dt_bits, (int)init_bits, 1, p->stream));
                CHK(clock_begin(p, SPAN_DT, &tok));
                hipLaunchKernelGGL(k_estimate_dt, dim3(grid_for(p->n, BLK_THREADS)), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), p->d_models.p, p->d_ctl, (int)p->n, p->h,
                                   p->dt_partial.p);
                hipLaunchKernelGGL(k_dt_reduce, dim3(1), dim3(1024), 0, p->stream, p->dt_partial.p, (int)grid_for(p->n, BLK_THREADS), p->d_ctl);
                CHK(clock_end(p, tok));
            }
            CHK(read_ctl(p));
            pending = 0;
            if (!(p->h_ctl->abort_flag[0] || p->h_ctl->abort_flag[1])) break;
            CHK(resort(p));
            spec_tiles = false;
            // the volumes (and with them the bound) must be recomputed on the new binning; without fluids the bound read
            // above is a function of the particle state alone and stays valid
            if (!fluid_volumes) break;
        }
        if (p->h_ctl->dt_nan) { rc = set_error(SPARKL_ERR_NAN_TIMESTEP, "NaN timestep bound (the reference panics: timestep_estimator.rs:46)"); break; }
        float bound;
        memcpy(&bound, &p->h_ctl->dt_bits, 4);
        if (fluid_volumes) { p->fluid_bound = bound; p->fluid_bound_valid = true; p->mass_tiles_ready = false; }
        auto pick = [&](float rem) {
            float d = std::fmin(std::fmin(rem, params->max_substep_dt), bound);   // timestep_estimator.rs:25,35,44
            if (d < min_dt && rem > min_dt) d = min_dt;                              // mls_solver.rs:57-59
            return d;
        };
        const float dt = pick(remaining);
        uint32_t dt_bits, bound_bits;
        memcpy(&dt_bits, &dt, 4);
        memcpy(&bound_bits, &bound, 4);
        const bool reuse = spec_tiles && spec_dt_bits == dt_bits;
        if (spec_tiles) { if (reuse) p->spec_hits++; else p->spec_misses++; }
        const float rem_after = remaining - dt;
        float dt_guess = -1.f;
        if (may_fuse && rem_after > 0.0f && (bound_bits == p->spec_prev_bound_bits || p->lookahead_force)) dt_guess = pick(rem_after);
        p->spec_prev_bound_bits = bound_bits;
        // Look-ahead.  While the bound repeats, the rest of the frame is known: dt_k = pick(remaining_k) with the same bound.
        // All of it is enqueued at once; every k_dt_reduce checks the bound it just produced against the planned one and, on
        // a difference, turns the substeps behind it into no-ops.  One host sync per frame instead of one per substep in the
        // stiffness-bound regime; the substep sequence stays the reference's (a mismatch re-plans from the exact state the
        // per-substep loop would have had).
        if (dt_guess > 0.f && p->lookahead && p->num_pblocks > 0) {
            std::vector<float> plan;
            float r = remaining;
            while (r > 0.0f && plan.size() < 64) { const float d = pick(r); plan.push_back(d); r -= d; }
            if (plan.size() >= 2) {
                const uint32_t base_done = p->h_ctl->steps_done;
                const uint32_t base_index = p->substep_index;
                for (size_t k = 0; k < plan.size(); ++k) {
                    const bool last = k + 1 == plan.size();
                    const float guess_k = !last ? plan[k + 1] : (r > 0.0f ? pick(r) : -1.f);
                    // every launch but the last may leave C.m[1..8] to the launch that ends the plan (StepParams::defer_c)
                    CHK(enqueue_adaptive_substep(p, params, gravity, plan[k], k == 0 ? reuse : true, guess_k, last ? 0u : bound_bits, !last));
                }
                HIPCHK(hipGetLastError());
                CHK(read_ctl(p));
                const uint32_t executed = std::min<uint32_t>(p->h_ctl->steps_done - base_done, (uint32_t)plan.size());
                for (uint32_t k = 0; k < executed; ++k) { remaining -= plan[k]; ++nsub; note_dt(base_index + k, plan[k]); }
                p->lookahead_substeps += executed;
                have_bound = true;
                if (executed == plan.size()) {
                    spec_tiles = r > 0.0f;
                    const float g_last = r > 0.0f ? pick(r) : -1.f;
                    memcpy(&spec_dt_bits, &g_last, 4);
                } else {
                    // stopped early.  The tiles hold the scatter substep executed-1 built for plan[executed].
                    if (executed > 0 && p->defer_c && !p->rare_fused() && !params->enable_boundary_particle_projection) {
                        // ... and the last launch that ran had deferred C.m[1..8]: rebuilt from the node velocities it used
                        // (every grid update behind it was a no-op), before the next P2G or the re-bin reads them
                        StepParams Sr = p->c_stale_S;
                        Sr.dt = plan[executed - 1];
                        hipLaunchKernelGGL(k_regather_c, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), grid_arrays(p), Sr);
                        p->c_regathers++;
                    }
                    spec_tiles = executed > 0;
                    memcpy(&spec_dt_bits, &plan[executed], 4);
                    p->substep_index = base_index + executed;   // parity continues as if the no-op substeps had not been enqueued
                    if (p->h_ctl->dt_mismatch) {
                        p->lookahead_mismatches++;
                        // the abort word in pinned host memory goes down with the device flags (only resort() used to clear it: the next
                        // fixed-dt batch then stopped enqueueing behind its first substep and re-binned for nothing)
                        if (p->h_words) p->h_words[0] = 0u;
                        HIPCHK(hipMemsetAsync(p->d_ctl, 0, 2 * sizeof(uint32_t), p->stream));
                        HIPCHK(hipMemsetAsync(&p->d_ctl->dt_mismatch, 0, sizeof(uint32_t), p->stream));
                        p->h_ctl->abort_flag[0] = p->h_ctl->abort_flag[1] = 0;
                        p->h_ctl->dt_mismatch = 0;
                    }
                    // else: a real re-bin request; the loop head re-bins
                }
                continue;
            }
        }
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->dt_bits, (int)init_bits, 1, p->stream));
        note_dt(p->substep_index, dt);
        CHK(enqueue_adaptive_substep(p, params, gravity, dt, reuse, dt_guess));
        HIPCHK(hipGetLastError());
        pending = 1;
        spec_tiles = dt_guess > 0.f && p->num_pblocks > 0;
        memcpy(&spec_dt_bits, &dt_guess, 4);
        have_bound = p->num_pblocks > 0;
        ++nsub;
        remaining -= dt;
        if (params->stop_after_one_substep) { result = dt; break; }
    }
    CHK(read_ctl(p));
    if (p->h_ctl->abort_flag[0] || p->h_ctl->abort_flag[1]) p->sorted = false;  // re-bin lazily at the next call
    if (getenv("SPARKL_HIP_DEBUG"))
        fprintf(stderr, "[sparkl_hip] step: %u substeps; fused-scatter hits %llu misses %llu; look-ahead substeps %llu mismatches %llu\n", nsub, (unsigned long long)p->spec_hits,
                (unsigned long long)p->spec_misses, (unsigned long long)p->lookahead_substeps, (unsigned long long)p->lookahead_mismatches);
    if (timings) {
        HIPCHK(hipEventRecord(e1, p->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(clock_collect(p));
        timings->dt = result; timings->num_substeps = nsub; timings->num_resorts = (uint32_t)p->clock.resorts;
        timings->grid_resize_and_sort_ms = (float)p->clock.sort_ms; timings->estimate_timestep_ms = (float)p->clock.dt_ms;
        timings->p2g_ms = (float)p->clock.p2g_ms; timings->grid_update_ms = (float)p->clock.grid_update_ms; timings->g2p_ms = (float)(p->clock.g2p_ms + p->clock.g2p2g_ms);   // fused launches count as G2P
        timings->halo_ms = (float)p->clock.halo_ms;
        timings->total_ms = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        p->clock = saved; p->clock_on = was_clock;
        p->clock_stride = saved_stride; p->sublog_on = false;
        p->sublog.resize(nsub, SparklSubstepTimings{});   // (spans of a re-bin requested by the LAST substep would open one more entry: it belongs to the next call)
    } else if (p->clock_on) {
        CHK(clock_collect(p));
        p->clock.substeps += nsub;
    }
    if (simulated_time) *simulated_time = result;
    return rc;
}

// cuda_sparse_grid.rs:492-650 (`check_active_blocks`, the reference's debug validator of the binning), on the current state
int sparkl_hip_pipeline_step_substep_timings(SparklPipeline* p, SparklSubstepTimings* out, uint32_t cap, uint32_t* n_out) {
    if (!p || !n_out || (cap && !out)) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    *n_out = (uint32_t)p->sublog.size();
    for (uint32_t k = 0; k < cap && k < p->sublog.size(); ++k) out[k] = p->sublog[k];
    return SPARKL_OK;
}
int sparkl_hip_grid_check(SparklPipeline* p, SparklGridCheck* out) {
    if (!p || !out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    memset(out, 0, sizeof(*out));
    HIPCHK(hipSetDevice(p->device));
    CHK(validate_models(p));
    if (!p->sorted) CHK(resort(p));
    hipStream_t st = p->stream;
    const int n = (int)p->n;
    DevBuf<unsigned long long> cnt;
    DevBuf<uint32_t> seen;
    CHK(cnt.reserve(8));
    HIPCHK(hipMemsetAsync(cnt.p, 0, 8 * sizeof(unsigned long long), st));
    if (n > 0) {
        CHK(seen.reserve((size_t)n));
        HIPCHK(hipMemsetAsync(seen.p, 0, (size_t)n * sizeof(uint32_t), st));
        if (!p->slab_on) hipLaunchKernelGGL(k_check_ids, dim3(grid_for(n, 256)), dim3(256), 0, st, p->cur.arrays(), n, (uint32_t)p->id_offset, seen.p, cnt.p);
    }
    if (p->num_blocks > 0) {
        HashTable H{p->hash_keys.p, p->hash_vals.p, p->hash_mask};
        hipLaunchKernelGGL(k_check_blocks, dim3(grid_for(p->num_blocks, 256)), dim3(256), 0, st, p->blk_keys.p, p->num_blocks, H, cnt.p);
    }
    if (p->num_pblocks > 0) {
        // particles flagged out of the domain sit behind the last block range
        int n_valid = 0;
        {
            std::vector<int4> last(2);
            HIPCHK(hipMemcpyAsync(last.data(), p->pb_info.p + 2 * (size_t)(p->num_pblocks - 1), 2 * sizeof(int4), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            n_valid = last[1].x + last[1].y;
            if (n_valid > n) n_valid = -1;    // reported as a coverage error by the kernel
        }
        const GridArrays G = grid_arrays(p);
        hipLaunchKernelGGL(k_check_pblocks, dim3(grid_for(p->num_pblocks, 256)), dim3(256), 0, st, G, n_valid, cnt.p);
        hipLaunchKernelGGL(k_check_particles, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, st, p->cur.arrays(), G, p->h, p->dom, cnt.p);
    }
    unsigned long long h[8] = {0};
    HIPCHK(hipMemcpyAsync(h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    out->bad_particle_ids = h[0]; out->blocks_not_in_hash = h[1]; out->particles_outside_their_block = h[2];
    out->missing_neighbour_blocks = h[3]; out->coverage_errors = h[4];
    out->node_blocks = (uint64_t)p->num_blocks; out->particle_blocks = (uint64_t)p->num_pblocks;
    cnt.release(); seen.release();
    return SPARKL_OK;
}

int sparkl_hip_pipeline_step_stats(SparklPipeline* p, uint64_t out[7]) {
    if (!p || !out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    out[0] = p->spec_hits; out[1] = p->spec_misses; out[2] = p->lookahead_substeps; out[3] = p->lookahead_mismatches;
    out[4] = p->c_regathers;
    out[5] = p->fluid_fused_substeps; out[6] = p->fluid_fused_mismatches;
    return SPARKL_OK;
}

int sparkl_hip_pipeline_clock_enable(SparklPipeline* p, int enable) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (p->clock_on) CHK(clock_collect(p));
    p->clock_on = enable != 0;
    p->clock_stride = enable > 1 ? (uint32_t)enable : 1u;  // enable = N > 1: sample every N-th substep
    return SPARKL_OK;
}
int sparkl_hip_pipeline_clock_read(SparklPipeline* p, SparklKernelClock* out, int reset) {
    if (!p || !out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (p->clock_on) CHK(clock_collect(p));
    *out = p->clock;
    out->sampled_substeps = p->clock_sampled;
    if (reset) { memset(&p->clock, 0, sizeof(p->clock)); p->clock_sampled = 0; }
    return SPARKL_OK;
}

// ------------------------------------------------------------------------------ slabs (multi-GPU)
int sparkl_hip_slab_set(SparklPipeline* p, const SparklSlab* slab) {
    if (!p || !slab) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (slab->axis < 0 || slab->axis > 2 || slab->block_lo >= slab->block_hi) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad slab");
    if (p->n) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "set the slab before uploading particles");
    p->slab_on = true;
    p->slab_axis = slab->axis; p->slab_lo = slab->block_lo; p->slab_hi = slab->block_hi; p->id_offset = slab->id_offset;
    return SPARKL_OK;
}
uint64_t sparkl_hip_halo_record_bytes(void) { return sizeof(HaloRecord); }
uint64_t sparkl_hip_migration_record_bytes(void) { return MIG_F4 * sizeof(float4); }

int sparkl_hip_rebin(SparklPipeline* p) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    HIPCHK(hipSetDevice(p->device));
    CHK(validate_models(p));
    return resort(p);
}
int sparkl_hip_dt_bound_export(SparklPipeline* p, void* dev_u32) {
    if (!p || !dev_u32) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    CHK(validate_models(p));
    const float fmax = std::numeric_limits<float>::max();
    uint32_t bits;
    memcpy(&bits, &fmax, 4);
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->dt_bits, (int)bits, 1, p->stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&p->d_ctl->dt_nan, 0, 1, p->stream));
    if (p->n > 0) {
        const int nblk = (int)grid_for(p->n, BLK_THREADS);
        CHK(p->dt_partial.reserve(std::max<size_t>(nblk, (size_t)std::max(p->num_pblocks, 1))));
        hipLaunchKernelGGL(k_estimate_dt, dim3(nblk), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), p->d_models.p, p->d_ctl, (int)p->n, p->h, p->dt_partial.p);
        hipLaunchKernelGGL(k_dt_reduce, dim3(1), dim3(1024), 0, p->stream, p->dt_partial.p, nblk, p->d_ctl);
    }
    HIPCHK(hipMemcpyAsync(dev_u32, &p->d_ctl->dt_bits, 4, hipMemcpyDeviceToDevice, p->stream));
    return SPARKL_OK;
}
int sparkl_hip_flags_export(SparklPipeline* p, void* dev_u32) {
    if (!p || !dev_u32) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipMemcpyAsync(dev_u32, &p->d_ctl->abort_flag[p->substep_index & 1u], 4, hipMemcpyDeviceToDevice, p->stream));
    return SPARKL_OK;
}
int sparkl_hip_flags_import(SparklPipeline* p, const void* dev_u32) {
    if (!p || !dev_u32) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipMemcpyAsync(&p->d_ctl->abort_flag[p->substep_index & 1u], dev_u32, 4, hipMemcpyDeviceToDevice, p->stream));
    return SPARKL_OK;
}
int sparkl_hip_substep_status(SparklPipeline* p, uint32_t* steps_done, uint32_t* need_rebin) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    HIPCHK(hipSetDevice(p->device));
    CHK(read_ctl(p));
    if (p->clock_on) CHK(clock_collect(p));
    if (steps_done) *steps_done = p->h_ctl->steps_done;
    if (need_rebin) *need_rebin = (p->h_ctl->abort_flag[0] | p->h_ctl->abort_flag[1]) ? 1u : 0u;
    return SPARKL_OK;
}

int sparkl_hip_substep_begin(SparklPipeline* p, const SparklSolverParams* params, const float gravity[3], float dt) {
    if (!p || !params || !gravity) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (p->in_substep) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "substep_begin called twice");
    // split-phase callers move the halo themselves and have no psi round; the group loop (halo_store_mode) runs one
    if (params->damage_model != SPARKL_DAMAGE_NONE && !(p->halo_store_mode && damage_supported(params))) return unsupported_damage_model(params);
    if (params->force_fluids_volume_recomputation && p->any_fluid && p->slab_on && (p->slab_lo != INT32_MIN || p->slab_hi != INT32_MAX) && !p->halo_store_mode)
        return set_error(SPARKL_ERR_UNSUPPORTED, "fluid volume recomputation across slab cuts needs the mass planes exchanged: use sparkl_hip_group_substeps");
    HIPCHK(hipSetDevice(p->device));
    if (!p->sorted) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "particles are not binned: call sparkl_hip_rebin (collectively) first");
    p->pending = make_params(p, params, gravity, dt);
    p->in_substep = true;
    p->pending_clock_saved = p->clock_on;   // sample the spans of this substep or not (restored by substep_finish)
    if (p->clock_on && (p->substep_index % p->clock_stride) != 0) p->clock_on = false;
    if (p->clock_on) p->clock_sampled++;
    if (p->num_pblocks == 0) return SPARKL_OK;
    hipStream_t st = p->stream;
    const GridArrays G = grid_arrays(p);
    // split-phase callers ADD their records (sparkl_hip_halo_merge, any number of calls per substep): start from zero.  The group
    // loop stores its records instead (k_halo_merge2) and relies on the zero fill of the re-bin.
    if (G.halo_in && !p->halo_store_mode) HIPCHK(hipMemsetAsync(p->halo_in.p, 0, (size_t)p->num_blocks * 64 * sizeof(float4), st));
    if (p->tiles_ready) { p->tiles_ready = false; return SPARKL_OK; }   // scattered by the fused kernel of the previous substep
    // (the slab-group loop runs the volume sweep itself, with the mass planes of the cuts exchanged: halo_store_mode)
    if (params->force_fluids_volume_recomputation && p->any_fluid && !p->halo_store_mode) CHK(enqueue_fluid_volumes(p, p->pending));
    size_t t0 = 0;
    CHK(clock_begin(p, SPAN_P2G, &t0));
    launch_p2g(p, p->cur.arrays(), G, p->pending, st, !p->fuse_fixed_dt || !p->last_finish_fused);   // counts unless the schedule finishes fused (see last_finish_fused)
    CHK(clock_end(p, t0));
    return SPARKL_OK;
}
int sparkl_hip_halo_count(SparklPipeline* p, int side, uint64_t* n) {
    if (!p || !n || side < 0 || side > 1) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    *n = (uint64_t)p->halo_count[side];
    return SPARKL_OK;
}
int sparkl_hip_halo_pack(SparklPipeline* p, int side, void* dev_out) {
    if (!p || side < 0 || side > 1) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    if (!p->in_substep) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "halo_pack outside substep_begin/finish");
    const int n = p->halo_count[side];
    if (n == 0) return SPARKL_OK;
    if (!dev_out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null buffer");
    size_t tok = 0;
    CHK(clock_begin(p, SPAN_HALO, &tok));
    hipLaunchKernelGGL(k_halo_pack, dim3((n + 3) / 4), dim3(BLK_THREADS), 0, p->stream, grid_arrays(p), p->blk_keys.p, p->halo_list[side].p, n, p->slab_axis,
                       (HaloRecord*)dev_out, p->d_ctl, p->pending.parity);
    CHK(clock_end(p, tok));
    return SPARKL_OK;
}
int sparkl_hip_halo_merge(SparklPipeline* p, int side, const void* dev_in, uint64_t n) {
    if (!p || side < 0 || side > 1) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    if (!p->in_substep) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "halo_merge outside substep_begin/finish");
    if (n == 0 || p->num_blocks == 0) return SPARKL_OK;
    if (!dev_in) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null buffer");
    HashTable H{p->hash_keys.p, p->hash_vals.p, p->hash_mask};
    size_t tok = 0;
    CHK(clock_begin(p, SPAN_HALO, &tok));
    hipLaunchKernelGGL(k_halo_merge, dim3((unsigned)((n + 3) / 4)), dim3(BLK_THREADS), 0, p->stream, (const HaloRecord*)dev_in, (int)n, p->slab_axis, H, p->halo_in.p,
                       p->d_ctl, p->pending.parity);
    CHK(clock_end(p, tok));
    return SPARKL_OK;
}
// `part`: 1 = the grid update only (the substep stays open: the slab-group loop runs the psi sweep with its own halo round behind
// it), 2 = the particle update only, 3 = both
static int substep_finish_impl(SparklPipeline* p, bool fuse_next, bool defer_c = false, int part = 3) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (!p->in_substep) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "substep_finish without substep_begin");
    if (part & 2) { p->in_substep = false; p->tiles_ready = false; }
    if (p->num_pblocks == 0) {
        if (!(part & 2)) return SPARKL_OK;
        if (p->dw.box) hipLaunchKernelGGL(k_direct_agree, dim3(1), dim3(64), 0, p->stream, p->dw, p->d_ctl, p->pending.parity);   // the agreed flag first
        hipLaunchKernelGGL(k_bump_steps, dim3(1), dim3(1), 0, p->stream, p->d_ctl, p->pending.parity);
        p->clock_on = p->pending_clock_saved;
        p->substep_index++;
        return SPARKL_OK;
    }
    hipStream_t st = p->stream;
    const GridArrays G = grid_arrays(p);
    size_t t1 = 0, t2 = 0;
    if (part & 1) {
        // (direct exchange: the grid update is where the substep's flag is agreed; the reorder, which reads it, goes behind)
        if (!p->dw.box) CHK(maybe_reorder(p, p->pending.parity));
        CHK(clock_begin(p, SPAN_GRID, &t1));
        launch_grid_update(p, G, p->pending, st);
        CHK(clock_end(p, t1));
        if (p->dw.box) CHK(maybe_reorder(p, p->pending.parity));
    }
    if (!(part & 2)) return SPARKL_OK;
    p->last_finish_fused = fuse_next && p->fuse_fixed_dt;
    if (fuse_next && p->fuse_fixed_dt) {
        // G2P of this substep + P2G of the next one (same dt) in one kernel; the next substep_begin is then free
        // defer_c: only from sparkl_hip_group_substeps, which owns the batch and rebuilds C if it is cut (the public
        // substep_finish_fused leaves a complete particle state behind)
        StepParams Sf = p->pending;
        Sf.defer_c = (defer_c && p->defer_c && !p->rare_fused() && !Sf.project_particles) ? 1 : 0;
        if (Sf.defer_c) p->c_stale_S = p->pending;
        CHK(clock_begin(p, SPAN_FUSED, &t2));
        launch_g2p2g(p, p->cur.arrays(), G, Sf, st);
        CHK(clock_end(p, t2));
        p->tiles_ready = true;
    } else {
        CHK(clock_begin(p, SPAN_G2P, &t2));
        launch_g2p(p, p->cur.arrays(), G, p->pending, st);
        CHK(clock_end(p, t2));
    }
    p->clock_on = p->pending_clock_saved;
    p->substep_index++;
    p->clock.substeps++;
    return SPARKL_OK;
}
int sparkl_hip_substep_finish(SparklPipeline* p) { return substep_finish_impl(p, false); }
int sparkl_hip_substep_finish_fused(SparklPipeline* p) { return substep_finish_impl(p, true); }

int sparkl_hip_slab_emigrant_count(SparklPipeline* p, int side, uint64_t* n) {
    if (!p || !n || side < 0 || side > 1) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    *n = (uint64_t)p->n_emig[side];
    return SPARKL_OK;
}
int sparkl_hip_slab_emigrants_pack(SparklPipeline* p, int side, void* dev_out) {
    if (!p || side < 0 || side > 1) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    const int n = p->n_emig[side];
    if (n == 0) return SPARKL_OK;
    if (!dev_out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null buffer");
    HIPCHK(hipSetDevice(p->device));
    hipLaunchKernelGGL(k_pack_particles, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), p->emig_first[side], n, (float4*)dev_out);
    p->n_emig[side] = 0;
    p->sorted = false;   // the FLAG_REMOVED particles are dropped by the next re-bin
    return SPARKL_OK;
}
int sparkl_hip_slab_immigrants_add(SparklPipeline* p, const void* dev_in, uint64_t n) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (n == 0) return SPARKL_OK;
    if (!dev_in) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null buffer");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    CHK(p->cur.reserve(p->n + n, true, p->stream));
    hipLaunchKernelGGL(k_unpack_particles, dim3(grid_for(n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (int)p->n, (int)n, (const float4*)dev_in);
    p->n += n;
    p->sorted = false;
    return SPARKL_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ slab groups: the exchange inside the library
// cuda_mpm_pipeline.rs:262-271 steps `&mut [SingleGpuMpmContext]` from one host thread and moves the halo itself
// (cuda_mpm_pipeline.rs:407-533).  sparkl_hip_group_substeps is that entry point for the slab decomposition: the members are the
// slabs this process owns (ascending along the slab axis); their outer neighbours are other ranks of an RCCL communicator
// (sparkl_hip_comm_init: one process per GPU, or several members per process with one communicator each).  Without a
// communicator the members must share one device and one stream and exchange by device copies (tests, single GPU).
// Per substep: abort-flag MAX all-reduce (4 bytes), P2G, halo pack, grouped ncclSend / ncclRecv with both neighbours, halo
// merge, grid update, G2P (+ next P2G).  Nothing synchronises with the host inside a batch of substeps.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                         // optional (older librccl: absent -> hangs stay hangs)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
};
static Rccl g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return SPARKL_OK;
    // SPARKL_HIP_TRANSPORT=host: the same function table on shared memory + host staging (host_transport.h) -- the TEST transport that lets
    // several OS processes sharing ONE GPU run the group loop with world > 1 (RCCL refuses two ranks on one device)
    if (const char* tr = getenv("SPARKL_HIP_TRANSPORT"); tr && !strcmp(tr, "host")) {
        g_rccl.GetUniqueId = hostcomm::GetUniqueId; g_rccl.CommInitRank = hostcomm::CommInitRank; g_rccl.CommDestroy = hostcomm::CommDestroy;
        g_rccl.GroupStart = hostcomm::GroupStart; g_rccl.GroupEnd = hostcomm::GroupEnd; g_rccl.Send = hostcomm::Send; g_rccl.Recv = hostcomm::Recv;
        g_rccl.AllReduce = hostcomm::AllReduce; g_rccl.GetErrorString = hostcomm::GetErrorString; g_rccl.CommAbort = hostcomm::CommAbort;
        g_rccl.CommGetAsyncError = hostcomm::CommGetAsyncError;
        g_rccl.lib = (void*)&g_rccl;
        return SPARKL_OK;
    }
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_error(SPARKL_ERR_UNSUPPORTED, "cannot load librccl.so.1: %s", dlerror());
#define SPK_SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return set_error(SPARKL_ERR_UNSUPPORTED, "librccl: missing %s", name); } while (0)
    SPK_SYM(GetUniqueId, "ncclGetUniqueId"); SPK_SYM(CommInitRank, "ncclCommInitRank"); SPK_SYM(CommDestroy, "ncclCommDestroy");
    SPK_SYM(GroupStart, "ncclGroupStart"); SPK_SYM(GroupEnd, "ncclGroupEnd"); SPK_SYM(Send, "ncclSend"); SPK_SYM(Recv, "ncclRecv");
    SPK_SYM(AllReduce, "ncclAllReduce"); SPK_SYM(GetErrorString, "ncclGetErrorString");
#undef SPK_SYM
    *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
    *(void**)(&g_rccl.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
    g_rccl.lib = h;
    return SPARKL_OK;
}
#define NCCLCHK(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r__ = (expr);                                                                                 \
        if (r__ != ncclSuccess) return set_error(SPARKL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)
static void comm_destroy(SparklPipeline* p) {
    if (p->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(p->comm);
    p->comm = nullptr;
}
// a new communicator is a new job: other processes may sit behind the ranks now.  Mappings go, the mailbox starts empty (mail of the old job must
// not answer an epoch of the new one), a failed setup may be tried again.
static void direct_reset(SparklPipeline* p) {
    SparklPipeline::Direct& D = p->direct;
    for (int k = 0; k < MAX_DIRECT; ++k) { if (D.peer_ipc[k]) (void)hipIpcCloseMemHandle(D.peer_ipc[k]); D.peer_ipc[k] = nullptr; D.peer_arena[k] = nullptr; D.peer_serial[k] = 0; }
    if (D.arena) { (void)hipStreamSynchronize(p->stream); (void)hipMemset(D.arena, 0, sizeof(PeerBox)); }
    D.ready = false; D.failed = false; D.epoch = 0; D.handshakes = 0;
}
// RCCL failure path.  A rank that leaves the group loop on an error would leave its peers blocked in their next collective for ever, and a
// rank whose peer has died would sit in hipStreamSynchronize behind an RCCL kernel that never completes.  So (1) every host wait of a
// pipeline that holds a communicator polls the stream AND ncclCommGetAsyncError, with a deadline (SPARKL_HIP_COMM_TIMEOUT_S, default 300 s),
// and (2) any error inside a group call aborts the communicators of all members (ncclCommAbort: the peers' pending operations fail with
// an error instead of hanging) before it is returned.  After an abort the handle has no communicator: the caller tears the job down.
static double comm_timeout_s() {
    static const double t = getenv("SPARKL_HIP_COMM_TIMEOUT_S") ? atof(getenv("SPARKL_HIP_COMM_TIMEOUT_S")) : 300.0;
    return t;
}
static int comm_wait(SparklPipeline* p, hipStream_t st) {
    if (!p->comm || !g_rccl.CommGetAsyncError) { HIPCHK(hipStreamSynchronize(st)); return SPARKL_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return SPARKL_OK;
        if (q != hipErrorNotReady) return set_error(SPARKL_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
        if ((spin & 255u) == 255u) {
            ncclResult_t ar = ncclSuccess;
            const ncclResult_t r = g_rccl.CommGetAsyncError(p->comm, &ar);
            if (r != ncclSuccess || ar != ncclSuccess)
                return set_error(SPARKL_ERR_HIP, "RCCL communicator of rank %d reports an asynchronous error: %s", p->comm_rank, g_rccl.GetErrorString(r != ncclSuccess ? r : ar));
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (comm_timeout_s() > 0.0 && el > comm_timeout_s())
                return set_error(SPARKL_ERR_HIP, "rank %d waited %.0f s for its stream behind an RCCL operation (a peer is gone?): giving up (SPARKL_HIP_COMM_TIMEOUT_S)", p->comm_rank, el);
            if (el > 0.001) std::this_thread::sleep_for(std::chrono::microseconds(50));   // short waits spin, long ones yield the core
        }
    }
}
static void group_abort(SparklPipeline** members, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        SparklPipeline* p = members[i];
        if (!p || !p->comm) continue;
        (void)hipSetDevice(p->device);
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(p->comm);
        p->comm = nullptr;
        p->comm_aborted = true;
    }
}


// every member of a slab group carries the ParticleFracture stream under DamageModel::ModifiedEigenerosion (immigrants bring theirs)
static int ensure_fracture(SparklPipeline* p) {
    if (p->cur.has_frac) return SPARKL_OK;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->cur.has_frac = true;
    CHK(p->cur.frac.reserve(std::max<size_t>(p->cur.q[0].cap, 1)));
    if (p->n) hipLaunchKernelGGL(k_fracture, dim3(grid_for(p->n, 256)), dim3(256), 0, p->stream, p->cur.arrays(), (float2*)nullptr, 0, (int)p->n, 0ll, 2);
    return SPARKL_OK;
}

namespace {
struct Group {
    SparklPipeline** m;
    int n;
    bool rccl;    // every member has a communicator
    bool balance; // move the cuts towards equal particle counts at collective re-bins
    // neighbour of member i on `side` (0 = low, 1 = high): -2 none, -1 remote rank (RCCL), >= 0 local member index
    int nbr(int i, int side) const {
        SparklPipeline* p = m[i];
        if ((side == 0 ? p->slab_lo == INT32_MIN : p->slab_hi == INT32_MAX)) return -2;
        if (rccl) return (side == 0 ? p->comm_rank > 0 : p->comm_rank < p->comm_world - 1) ? -1 : -2;
        const int j = side == 0 ? i - 1 : i + 1;
        return (j >= 0 && j < n) ? j : -2;
    }
    int peer_rank(int i, int side) const { return m[i]->comm_rank + (side == 0 ? -1 : 1); }
    // stream the halo chain of member i runs on: its own when the exchange is serial, the halo stream when it overlaps the interior
    // work (members of one process share ONE halo stream, like they share one main stream: ordering by enqueue order)
    bool overlap = false;
    hipStream_t xs(int i) const { return overlap ? (rccl ? m[i]->halo_stream : m[0]->halo_stream) : m[i]->stream; }
};

static int x_staging(SparklPipeline* p) {
    CHK(p->x_dev.reserve(64));
    if (!p->x_host) HIPCHK(hipHostMalloc((void**)&p->x_host, 64 * sizeof(long long)));
    return SPARKL_OK;
}

// MAX over all members and ranks of one host value per member (host-synchronous: only used around re-bins).  `veto` (optional)
// travels in the same all-reduce: out_veto = MAX of every member's veto[i] on every rank.
static int group_max(const Group& G, const long long* per_member, long long* out, const long long* veto = nullptr, long long* out_veto = nullptr) {
    long long mx = 0, vx = 0;
    for (int i = 0; i < G.n; ++i) { mx = std::max(mx, per_member[i]); if (veto) vx = std::max(vx, veto[i]); }
    if (G.rccl) {
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            p->x_host[0] = mx; p->x_host[1] = vx;
            HIPCHK(hipMemcpyAsync(p->x_dev.p, p->x_host, 2 * sizeof(long long), hipMemcpyHostToDevice, p->stream));
        }
        NCCLCHK(g_rccl.GroupStart());
        for (int i = 0; i < G.n; ++i) NCCLCHK(g_rccl.AllReduce(G.m[i]->x_dev.p, G.m[i]->x_dev.p, 2, ncclInt64, ncclMax, G.m[i]->comm, G.m[i]->stream));
        NCCLCHK(g_rccl.GroupEnd());
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            HIPCHK(hipMemcpyAsync(p->x_host, p->x_dev.p, 2 * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
            CHK(comm_wait(p, p->stream));
            mx = std::max(mx, p->x_host[0]); vx = std::max(vx, p->x_host[1]);
        }
    }
    *out = mx;
    if (out_veto) *out_veto = vx;
    return SPARKL_OK;
}
// Whether the coming collective re-bin may also move the cuts (group_rebalance: two more count exchanges).  Every rank must take
// the same branch or the Send / Recv pairs of those exchanges stay unmatched: the decision is rank-local state (a re-bin has
// happened before, sparkl_hip_slab_balance), so it travels as a veto in the all-reduce that decides the re-bin itself.
static long long balance_veto(const Group& G) {
    bool ok = G.balance;
    for (int i = 0; i < G.n; ++i) ok = ok && G.m[i]->group_rebins > 0 && G.m[i]->slab_balance;
    return ok ? 0 : 1;
}

// counts[i][side] announced to the neighbour on that side -> what the neighbours announced (host-synchronous)
static int group_swap_counts(const Group& G, const uint64_t (*out_counts)[2], uint64_t (*in_counts)[2]) {
    for (int i = 0; i < G.n; ++i) in_counts[i][0] = in_counts[i][1] = 0;
    if (!G.rccl) {
        for (int i = 0; i < G.n; ++i)
            for (int side = 0; side < 2; ++side) { const int j = G.nbr(i, side); if (j >= 0) in_counts[i][side] = out_counts[j][1 - side]; }
        return SPARKL_OK;
    }
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        p->x_host[0] = (long long)out_counts[i][0]; p->x_host[1] = (long long)out_counts[i][1]; p->x_host[2] = 0; p->x_host[3] = 0;
        HIPCHK(hipMemcpyAsync(p->x_dev.p, p->x_host, 4 * sizeof(long long), hipMemcpyHostToDevice, p->stream));
    }
    NCCLCHK(g_rccl.GroupStart());
    for (int i = 0; i < G.n; ++i)
        for (int side = 0; side < 2; ++side)
            if (G.nbr(i, side) == -1) {
                NCCLCHK(g_rccl.Send(G.m[i]->x_dev.p + side, 1, ncclInt64, G.peer_rank(i, side), G.m[i]->comm, G.m[i]->stream));
                NCCLCHK(g_rccl.Recv(G.m[i]->x_dev.p + 2 + side, 1, ncclInt64, G.peer_rank(i, side), G.m[i]->comm, G.m[i]->stream));
            }
    NCCLCHK(g_rccl.GroupEnd());
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        HIPCHK(hipMemcpyAsync(p->x_host, p->x_dev.p, 4 * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
        CHK(comm_wait(p, p->stream));
        for (int side = 0; side < 2; ++side) if (G.nbr(i, side) == -1) in_counts[i][side] = (uint64_t)p->x_host[2 + side];
    }
    return SPARKL_OK;
}

// send[side] (nsend bytes) of every member to its neighbour, into that neighbour's recv[1 - side]; stream ordered
typedef DevBuf<uint8_t> (SparklPipeline::*BufPair)[2];
static int group_exchange(const Group& G, BufPair send, BufPair recv, const uint64_t (*nsend)[2], const uint64_t (*nrecv)[2]) {
    if (G.rccl) {
        bool any = false;
        for (int i = 0; i < G.n && !any; ++i) for (int side = 0; side < 2; ++side) any |= G.nbr(i, side) == -1 && (nsend[i][side] || nrecv[i][side]);
        if (!any) return SPARKL_OK;
        NCCLCHK(g_rccl.GroupStart());
        for (int i = 0; i < G.n; ++i)
            for (int side = 0; side < 2; ++side) {
                if (G.nbr(i, side) != -1) continue;
                SparklPipeline* p = G.m[i];
                if (nsend[i][side]) NCCLCHK(g_rccl.Send((p->*send)[side].p, nsend[i][side], ncclUint8, G.peer_rank(i, side), p->comm, G.xs(i)));
                if (nrecv[i][side]) NCCLCHK(g_rccl.Recv((p->*recv)[side].p, nrecv[i][side], ncclUint8, G.peer_rank(i, side), p->comm, G.xs(i)));
            }
        NCCLCHK(g_rccl.GroupEnd());
        return SPARKL_OK;
    }
    for (int i = 0; i < G.n; ++i)
        for (int side = 0; side < 2; ++side) {
            const int j = G.nbr(i, side);
            if (j < 0 || !nsend[i][side]) continue;
            HIPCHK(hipMemcpyAsync((G.m[j]->*recv)[1 - side].p, (G.m[i]->*send)[side].p, nsend[i][side], hipMemcpyDeviceToDevice, G.xs(i)));
        }
    return SPARKL_OK;
}

// Load re-balancing at a collective re-bin ("cut ... chosen on the block-sorted histogram so each GPU owns ~ N / world").
// Global: every member bins its particle blocks by block plane along the slab axis (k_plane_hist, <= 1024 planes), the histograms are summed
// over all members and ranks together with the current cuts (one all-reduce of 1024 + world 64-bit words), and every rank derives the SAME
// new cut list from it: cut k goes to the plane boundary whose prefix sum is nearest k N / world, at most `max_shift` planes per re-bin and
// never past a neighbouring cut's old position -- so the particles of the planes that changed owner travel to an ADJACENT rank, as ordinary
// emigrants of the re-bin that follows (a long way takes several re-bins).  Slabs keep at least two block planes.
static constexpr int BAL_PLANES = 1024;
// The cut list every rank derives from the summed histogram (pure host arithmetic: also behind sparkl_hip_debug_balance_cuts for CPU tests).
// hist[c] = particles in the block plane of index c (= block plane + 1); cuts[1 .. world - 1] = current cuts in block planes (cut k = first plane
// of rank k); out[0 .. world]: the new list with the open ends at the occupied range.  false = leave the cuts alone.
static bool balance_cuts(const long long* hist, int nplanes, const long long* cuts, int world, int max_shift, std::vector<long long>& nxt) {
    std::vector<long long> prefix(nplanes + 1, 0);          // prefix[c] = particles in planes (index) < c
    for (int c = 0; c < nplanes; ++c) prefix[c + 1] = prefix[c] + hist[c];
    const long long total = prefix[nplanes];
    if (total <= 0 || world < 2) return false;
    int first = 0, last = nplanes - 1;
    while (first < nplanes && hist[first] == 0) ++first;
    while (last > first && hist[last] == 0) --last;
    std::vector<long long> cur(world + 1);
    cur[0] = (long long)first - 1;                   // in block planes (index - 1); the open ends stand at the occupied range
    cur[world] = (long long)last;                    // = (last + 1) - 1: first plane beyond the occupied range
    for (int k = 1; k < world; ++k) cur[k] = std::min(std::max(cuts[k], cur[0]), cur[world]);   // a cut outside the occupied range owns nothing there: same as at its edge
    for (int k = 1; k <= world; ++k) if (cur[k] < cur[k - 1]) return false;   // cuts out of order: not this function's to repair
    nxt = cur;
    for (int k = 1; k < world; ++k) {   // low to high: cut k settles against the NEW cut below it and the OLD cuts around it
        const double want = (double)total * k / world;
        // boundary c (as a plane index: the cut sits between planes c - 1 and c) whose prefix is nearest `want`, within reach of this re-bin
        long long best = std::min<long long>(std::max<long long>(cur[k] + 1, 0), nplanes);
        double bestd = std::fabs((double)prefix[best] - want);
        for (long long c = std::max<long long>(cur[k] + 1 - max_shift, 0); c <= std::min<long long>(cur[k] + 1 + max_shift, nplanes); ++c) {
            const double d = std::fabs((double)prefix[c] - want);
            if (d < bestd - 0.5) { bestd = d; best = c; }
        }
        long long cut = best - 1;                                                   // back to block planes
        // emigrants travel one rank far: never past the old neighbouring cuts.  Two block planes per slab where the old cuts leave room
        // (a sliver the caller started with is widened as fast as its neighbours' old cuts allow), room for the slabs above.
        const long long lo = std::max(cur[k - 1], nxt[k - 1] + 2);
        const long long hi = std::min(cur[k + 1], cur[world] - 2ll * (world - k));
        cut = std::max(cut, lo);
        cut = std::min(cut, std::max(hi, std::min(cur[k + 1], nxt[k - 1] + 1)));     // lo > hi: as far as the old cut above allows, at least one plane wide
        cut = std::max(cut, nxt[k - 1] + 1);
        nxt[k] = cut;
    }
    for (int k = 1; k < world; ++k)
        if (nxt[k] <= nxt[k - 1] || nxt[k] > cur[world] || nxt[k] < cur[k - 1] || nxt[k] > cur[k + 1]) return false;   // no room: keep the cuts
    return true;
}
extern "C" int sparkl_hip_debug_balance_cuts(const long long* hist, int nplanes, const long long* cuts, int world, int max_shift, long long* out) {
    std::vector<long long> nxt;
    if (!hist || !cuts || !out || nplanes < 1 || world < 1) return -1;
    if (!balance_cuts(hist, nplanes, cuts, world, max_shift, nxt)) return 0;
    for (int k = 0; k <= world; ++k) out[k] = nxt[k];
    return 1;
}
static int group_rebalance(const Group& G) {
    const int world = G.rccl ? G.m[0]->comm_world : G.n;
    if (world < 2) return SPARKL_OK;
    const int words = BAL_PLANES + world;                 // histogram, then cut k (k = 1 .. world - 1) at [BAL_PLANES + k]
    auto rank_of = [&](int i) { return G.rccl ? G.m[i]->comm_rank : i; };
    static const int max_shift = getenv("SPARKL_HIP_BALANCE_SHIFT") ? std::max(1, atoi(getenv("SPARKL_HIP_BALANCE_SHIFT"))) : 4;
    std::vector<long long> sum(words, 0);
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        CHK(p->bal_dev.reserve(words));
        if (!p->bal_host || p->bal_host_words < words) {
            if (p->bal_host) HIPCHK(hipHostFree(p->bal_host));
            HIPCHK(hipHostMalloc((void**)&p->bal_host, words * sizeof(long long)));
            p->bal_host_words = words;
        }
        memset(p->bal_host, 0, words * sizeof(long long));
        const int r = rank_of(i);
        if (r + 1 < world) p->bal_host[BAL_PLANES + r + 1] = (long long)p->slab_hi;   // the cut above this slab, announced by its lower rank
        HIPCHK(hipMemcpyAsync(p->bal_dev.p, p->bal_host, words * sizeof(long long), hipMemcpyHostToDevice, p->stream));
        if (p->num_pblocks > 0)
            hipLaunchKernelGGL(k_plane_hist, dim3(grid_for(p->num_pblocks, 256)), dim3(256), 0, p->stream, p->pb_info.p, p->num_pblocks, p->slab_axis, p->bal_dev.p, BAL_PLANES);
    }
    if (G.rccl) {
        NCCLCHK(g_rccl.GroupStart());
        for (int i = 0; i < G.n; ++i) NCCLCHK(g_rccl.AllReduce(G.m[i]->bal_dev.p, G.m[i]->bal_dev.p, words, ncclInt64, ncclSum, G.m[i]->comm, G.m[i]->stream));
        NCCLCHK(g_rccl.GroupEnd());
    }
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        HIPCHK(hipMemcpyAsync(p->bal_host, p->bal_dev.p, words * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
        CHK(comm_wait(p, p->stream));
        if (G.rccl) { if (i == 0) for (int w = 0; w < words; ++w) sum[w] = p->bal_host[w]; }   // every rank holds the global sum already
        else for (int w = 0; w < words; ++w) sum[w] += p->bal_host[w];
    }
    std::vector<long long> cuts(world + 1, 0), nxt;
    for (int k = 1; k < world; ++k) cuts[k] = sum[BAL_PLANES + k];
    if (!balance_cuts(sum.data(), BAL_PLANES, cuts.data(), world, max_shift, nxt)) return SPARKL_OK;
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        const int r = rank_of(i);
        const int lo = r == 0 ? INT32_MIN : (int)nxt[r], hi = r == world - 1 ? INT32_MAX : (int)nxt[r + 1];
        if (lo != p->slab_lo || hi != p->slab_hi) {
            p->cut_moves += (lo != p->slab_lo) + (hi != p->slab_hi);
            p->slab_lo = lo; p->slab_hi = hi;
            p->sorted = false;
        }
    }
    return SPARKL_OK;
}

static int group_direct_setup(Group& G);
// `balance`: agreed by every rank (balance_veto through group_max); needs the block list of an earlier re-bin
static int group_rebin(const Group& G, bool balance) {
    const uint64_t mig_bytes = MIG_F4 * sizeof(float4), halo_bytes = sizeof(HaloRecord);
    if (balance) CHK(group_rebalance(G));
    std::vector<std::array<uint64_t, 2>> n_out(G.n), n_in(G.n), b_out(G.n), b_in(G.n);
    std::vector<long long> total(G.n);
    for (int round = 0; round < 64; ++round) {
        for (int i = 0; i < G.n; ++i) { HIPCHK(hipSetDevice(G.m[i]->device)); CHK(validate_models(G.m[i])); CHK(resort(G.m[i])); }
        for (int i = 0; i < G.n; ++i) {
            for (int side = 0; side < 2; ++side) n_out[i][side] = G.nbr(i, side) != -2 ? (uint64_t)G.m[i]->n_emig[side] : 0;
            total[i] = (long long)(n_out[i][0] + n_out[i][1]);
        }
        long long any = 0;
        CHK(group_max(G, total.data(), &any));
        if (any == 0) break;
        CHK(group_swap_counts(G, (const uint64_t(*)[2])n_out.data(), (uint64_t(*)[2])n_in.data()));
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            for (int side = 0; side < 2; ++side) {
                b_out[i][side] = n_out[i][side] * mig_bytes; b_in[i][side] = n_in[i][side] * mig_bytes;
                CHK(p->x_mig_send[side].reserve(std::max<uint64_t>(b_out[i][side], 16)));
                CHK(p->x_mig_recv[side].reserve(std::max<uint64_t>(b_in[i][side], 16)));
                if (n_out[i][side]) CHK(sparkl_hip_slab_emigrants_pack(p, side, p->x_mig_send[side].p));
            }
        }
        {   // on the MAIN streams, whatever the substep schedule: the emigrants were packed there and the immigrants are unpacked there
            // (with the overlapped schedule G.xs() is the halo stream, which nothing here has ordered behind the pack kernels)
            Group Gm = G;
            Gm.overlap = false;
            CHK(group_exchange(Gm, &SparklPipeline::x_mig_send, &SparklPipeline::x_mig_recv, (const uint64_t(*)[2])b_out.data(), (const uint64_t(*)[2])b_in.data()));
        }
        for (int i = 0; i < G.n; ++i)
            for (int side = 0; side < 2; ++side)
                if (n_in[i][side]) CHK(sparkl_hip_slab_immigrants_add(G.m[i], G.m[i]->x_mig_recv[side].p, n_in[i][side]));
        if (round == 63) return set_error(SPARKL_ERR_INTERNAL, "slab migration did not settle");
    }
    // halo buffers for the new block lists
    for (int i = 0; i < G.n; ++i) for (int side = 0; side < 2; ++side) n_out[i][side] = G.nbr(i, side) != -2 ? (uint64_t)G.m[i]->halo_count[side] : 0;
    CHK(group_swap_counts(G, (const uint64_t(*)[2])n_out.data(), (uint64_t(*)[2])n_in.data()));
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        for (int side = 0; side < 2; ++side) {
            p->x_nsend[side] = n_out[i][side]; p->x_nrecv[side] = n_in[i][side];
            CHK(p->x_send[side].reserve(std::max<uint64_t>(n_out[i][side] * halo_bytes, 16)));
            CHK(p->x_recv[side].reserve(std::max<uint64_t>(n_in[i][side] * halo_bytes, 16)));
        }
        p->group_rebins++;
        p->direct.ready = false;   // the binning is new: slots and buffers are rebuilt below
    }
    {
        Group Gd = G;
        CHK(group_direct_setup(Gd));
    }
    return SPARKL_OK;
}

// ---------------------------------------------------------------- peer-direct halo exchange: setup (see kernels.h, PeerBox)
// What every member publishes about its arena: one 128-byte record per member of the job, gathered over the communicator.
struct ArenaRecord {
    uint64_t pid, token;          // process identity (pid + a per-process random word: members of one process exchange raw pointers)
    uint64_t ptr, bytes, cap_blocks, serial;
    int32_t device, valid;
    unsigned char handle[64];     // hipIpcMemHandle_t
    char bus_id[24];              // PCI bus id of the device ("0000:05:00.0"): which members share a physical device
    unsigned char pad[16];
};
static_assert(sizeof(ArenaRecord) == 160 && sizeof(hipIpcMemHandle_t) == 64, "");
static uint64_t process_token() {
    static const uint64_t t = [] {
        uint64_t r = 0;
        if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&r, sizeof(r), 1, f) != 1) r = 0; fclose(f); }
        return r ^ ((uint64_t)getpid() << 32);
    }();
    return t;
}
static bool direct_wanted() { return !(getenv("SPARKL_HIP_DIRECT") && !strcmp(getenv("SPARKL_HIP_DIRECT"), "0")); }   // (read at every collective re-bin)
// 1 = SPARKL_HIP_DIRECT=1 (use it whenever it is set up), 2 = "cal" (time it against the staged schedules whatever the group), 0 = unset: time it between the
// ranks of a communicator, use it inside one process
static int direct_policy() { const char* e = getenv("SPARKL_HIP_DIRECT"); return !e ? 0 : (!strcmp(e, "cal") ? 2 : (atoi(e) != 0 ? 1 : 0)); }
// the arena of one member, sized for its current node blocks (grows by half: a reallocation means new mappings on every peer)
static int direct_arena(SparklPipeline* p) {
    const size_t need_blocks = std::max<size_t>((size_t)p->num_blocks, 64);
    SparklPipeline::Direct& D = p->direct;
    if (D.arena && need_blocks <= D.cap_blocks) return SPARKL_OK;
    const size_t cap = need_blocks + need_blocks / 2;
    const size_t bytes = sizeof(PeerBox) + 2 * cap * 64 * sizeof(float4);
    void* a = nullptr;
    // fine-grained: peers store into it while kernels of this device poll and read it (system-scope release / acquire needs coherent memory)
    hipError_t e = hipExtMallocWithFlags(&a, bytes, hipDeviceMallocFinegrained);
    D.arena_fine = e == hipSuccess;
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&a, bytes); }   // (one device, one process: plain device memory does; across devices the handshake decides)
    if (e != hipSuccess) return set_error(SPARKL_ERR_OUT_OF_MEMORY, "direct exchange arena: %s", hipGetErrorString(e));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemset(a, 0, sizeof(PeerBox)));   // a fresh box: no mail of past epochs (the job is quiescent around a re-bin: nothing in flight is lost)
    if (D.arena) (void)hipFree(D.arena);
    D.arena = a; D.arena_bytes = bytes; D.cap_blocks = cap; D.arena_serial++;
    return SPARKL_OK;
}
// Behind a collective re-bin: arenas sized and zeroed, every member's arena mapped into every other member's address space, the halo columns
// of each cut resolved to the neighbour's node-block slots.  A step that fails on ANY rank (no peer access, hipIpc refused, more than MAX_DIRECT
// members) is agreed through one more group_max and the whole job stays on the RCCL / copy exchange -- never a mix.
static int group_direct_setup(Group& G) {
    const int world = G.rccl ? G.m[0]->comm_world : G.n;
    bool ok = direct_wanted() && world <= MAX_DIRECT && world >= 1;
    // (a schedule forced or calibrated through SPARKL_HIP_OVERLAP is a request for the RCCL / copy exchange, whose two-stream variant it selects)
    // (... and a job that has timed it and kept a staged schedule does not set it up again)
    for (int i = 0; i < G.n; ++i) ok = ok && !G.m[i]->direct.failed && G.m[i]->direct_choice != 0 && (G.m[i]->overlap_mode == -1 || G.m[i]->overlap_mode == 0 || direct_policy() == 2);
    auto index_of = [&](int i) { return G.rccl ? G.m[i]->comm_rank : i; };
    std::vector<ArenaRecord> table((size_t)world);
    memset(table.data(), 0, table.size() * sizeof(ArenaRecord));
    if (ok)
        for (int i = 0; i < G.n && ok; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            if (direct_arena(p) != SPARKL_OK) { ok = false; break; }
            p->direct.index = index_of(i); p->direct.n = world;
            ArenaRecord& r = table[(size_t)index_of(i)];
            r.pid = (uint64_t)getpid(); r.token = process_token(); r.ptr = (uint64_t)(uintptr_t)p->direct.arena; r.bytes = p->direct.arena_bytes;
            r.cap_blocks = p->direct.cap_blocks; r.serial = p->direct.arena_serial; r.device = p->device; r.valid = 1;
            if (hipDeviceGetPCIBusId(r.bus_id, (int)sizeof(r.bus_id) - 1, p->device) != hipSuccess) { (void)hipGetLastError(); snprintf(r.bus_id, sizeof(r.bus_id), "dev%d", p->device); }
            if (G.rccl && world > G.n) {   // some member lives in another process: it needs a handle
                hipIpcMemHandle_t h;
                if (hipIpcGetMemHandle(&h, p->direct.arena) != hipSuccess) { (void)hipGetLastError(); r.valid = 0; ok = false; }
                else memcpy(r.handle, &h, 64);
            }
        }
    if (G.rccl && world > 1) {
        // all-gather as a byte-wise SUM over zero-filled tables (every rank fills its own records): one collective RCCL and the test transport both have
        const size_t bytes = table.size() * sizeof(ArenaRecord);
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            CHK(p->x_mig_send[0].reserve(std::max<size_t>(bytes, 16)));
            std::vector<ArenaRecord> mine((size_t)world);
            memset(mine.data(), 0, bytes);
            if (ok) mine[(size_t)index_of(i)] = table[(size_t)index_of(i)];
            HIPCHK(hipMemcpyAsync(p->x_mig_send[0].p, mine.data(), bytes, hipMemcpyHostToDevice, p->stream));
            HIPCHK(hipStreamSynchronize(p->stream));
        }
        NCCLCHK(g_rccl.GroupStart());
        for (int i = 0; i < G.n; ++i) NCCLCHK(g_rccl.AllReduce(G.m[i]->x_mig_send[0].p, G.m[i]->x_mig_send[0].p, bytes, ncclUint8, ncclSum, G.m[i]->comm, G.m[i]->stream));
        NCCLCHK(g_rccl.GroupEnd());
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            HIPCHK(hipMemcpyAsync(table.data(), p->x_mig_send[0].p, bytes, hipMemcpyDeviceToHost, p->stream));
            CHK(comm_wait(p, p->stream));
        }
        for (int q = 0; q < world; ++q) ok = ok && table[(size_t)q].valid == 1;
    }
    // map (a mapping that is still the one of the last setup is kept: `remapped` says whether anything changed -- the same answer on every rank, they
    // all read the same table -- and only then is the handshake below repeated)
    bool remapped = false;
    if (ok)
        for (int i = 0; i < G.n && ok; ++i) {
            SparklPipeline* p = G.m[i];
            SparklPipeline::Direct& D = p->direct;
            HIPCHK(hipSetDevice(p->device));
            for (int q = 0; q < world && ok; ++q) {
                const ArenaRecord& r = table[(size_t)q];
                if (D.peer_arena[q] && D.peer_serial[q] == r.serial && D.peer_pid[q] == r.token) continue;   // still the mapping of the last setup
                remapped = true;
                if (D.peer_ipc[q]) { (void)hipIpcCloseMemHandle(D.peer_ipc[q]); D.peer_ipc[q] = nullptr; }
                D.peer_arena[q] = nullptr;
                if (r.pid == (uint64_t)getpid() && r.token == process_token()) {
                    if (r.device != p->device) {   // another device of this process (sparkl_hip_comm_init_all): peer access, both ways (the peer's loop turn does the other)
                        int can = 0;
                        if (hipDeviceCanAccessPeer(&can, p->device, r.device) != hipSuccess || !can) { (void)hipGetLastError(); ok = false; break; }
                        const hipError_t e = hipDeviceEnablePeerAccess(r.device, 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); ok = false; break; }
                        (void)hipGetLastError();
                    }
                    D.peer_arena[q] = (void*)(uintptr_t)r.ptr;
                } else {
                    hipIpcMemHandle_t h;
                    memcpy(&h, r.handle, 64);
                    void* mapped = nullptr;
                    if (hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
                    D.peer_ipc[q] = mapped; D.peer_arena[q] = mapped;
                }
                D.peer_serial[q] = r.serial; D.peer_pid[q] = r.token; D.peer_cap[q] = (size_t)r.cap_blocks;
            }
            for (int q = 0; q < world; ++q) D.peer_cap[q] = (size_t)table[(size_t)q].cap_blocks;
            // two members on one physical device that do not share this member's process (their kernels sit in another queue): see DirectWait::pre_agreed
            D.shared_device = false;
            for (int a = 0; a < world; ++a)
                for (int b = a + 1; b < world; ++b)
                    if (!strncmp(table[(size_t)a].bus_id, table[(size_t)b].bus_id, sizeof(table[0].bus_id)) && table[(size_t)a].token != table[(size_t)b].token) D.shared_device = true;
        }
    {   // every rank takes the same branch
        std::vector<long long> v(G.n, ok ? 0 : 1);
        long long any_bad = 0;
        CHK(group_max(G, v.data(), &any_bad));
        if (any_bad) {
            for (int i = 0; i < G.n; ++i) { G.m[i]->direct.ready = false; G.m[i]->direct.failed = G.m[i]->direct_choice != 0; }   // (not a failure when the job timed it and kept a staged schedule)
            if (getenv("SPARKL_HIP_DEBUG")) fprintf(stderr, "[sparkl_hip] direct halo exchange not available (local setup %s): staying on the RCCL / copy exchange\n", ok ? "ok" : "failed");
            return SPARKL_OK;
        }
    }
    // slots: my column keys -> neighbour -> its node-block slots -> back (two exchanges per re-bin, on the main streams, like the migration)
    std::vector<std::array<uint64_t, 2>> b_out(G.n), b_in(G.n);
    Group Gm = G;
    Gm.overlap = false;
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        for (int side = 0; side < 2; ++side) {
            const uint64_t ns = G.nbr(i, side) != -2 ? p->x_nsend[side] : 0, nr = G.nbr(i, side) != -2 ? p->x_nrecv[side] : 0;
            b_out[i][side] = ns * sizeof(uint32_t); b_in[i][side] = nr * sizeof(uint32_t);
            CHK(p->x_mig_send[side].reserve(std::max<uint64_t>(std::max(ns, nr) * sizeof(int2), 16)));
            CHK(p->x_mig_recv[side].reserve(std::max<uint64_t>(std::max(ns, nr) * sizeof(int2), 16)));
            if (ns) hipLaunchKernelGGL(k_halo_keys, dim3(grid_for(ns, 256)), dim3(256), 0, p->stream, grid_arrays(p), p->blk_keys.p, p->halo_list[side].p, (int)ns, p->slab_axis, (uint32_t*)p->x_mig_send[side].p);
        }
    }
    CHK(group_exchange(Gm, &SparklPipeline::x_mig_send, &SparklPipeline::x_mig_recv, (const uint64_t(*)[2])b_out.data(), (const uint64_t(*)[2])b_in.data()));
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        HashTable H{p->hash_keys.p, p->hash_vals.p, p->hash_mask};
        for (int side = 0; side < 2; ++side) {
            const uint64_t nr = b_in[i][side] / sizeof(uint32_t);
            if (nr && p->num_blocks) hipLaunchKernelGGL(k_halo_slots, dim3(grid_for(nr, 256)), dim3(256), 0, p->stream, (const uint32_t*)p->x_mig_recv[side].p, (int)nr, p->slab_axis, H, (int2*)p->x_mig_send[side].p);
            else if (nr) HIPCHK(hipMemsetAsync(p->x_mig_send[side].p, 0xFF, nr * sizeof(int2), p->stream));   // no block at all here: every slot -1
            const uint64_t ns = b_out[i][side] / sizeof(uint32_t);
            b_out[i][side] = nr * sizeof(int2); b_in[i][side] = ns * sizeof(int2);
        }
    }
    CHK(group_exchange(Gm, &SparklPipeline::x_mig_send, &SparklPipeline::x_mig_recv, (const uint64_t(*)[2])b_out.data(), (const uint64_t(*)[2])b_in.data()));
    for (int i = 0; i < G.n; ++i) {
        SparklPipeline* p = G.m[i];
        HIPCHK(hipSetDevice(p->device));
        for (int side = 0; side < 2; ++side) {
            const uint64_t ns = b_in[i][side] / sizeof(int2);
            CHK(p->direct.dst[side].reserve(std::max<uint64_t>(ns, 1)));
            if (ns) HIPCHK(hipMemcpyAsync(p->direct.dst[side].p, p->x_mig_recv[side].p, ns * sizeof(int2), hipMemcpyDeviceToDevice, p->stream));
        }
        if (!p->direct.arrive.p) { CHK(p->direct.arrive.reserve(16)); HIPCHK(hipMemsetAsync(p->direct.arrive.p, 0, 16 * sizeof(unsigned int), p->stream)); }
        // both halo buffers start from zero: a node of a cut plane no record covers reads 0, like the halo_in of the copy exchange
        HIPCHK(hipMemsetAsync((char*)p->direct.arena + sizeof(PeerBox), 0, 2 * p->direct.cap_blocks * 64 * sizeof(float4), p->stream));
        CHK(comm_wait(p, p->stream));
    }
    // Nobody may push into a buffer its owner is still clearing: one more agreement (host-synchronous, like everything around a re-bin).  It also
    // carries `remapped`: a function of the gathered table and of what each member had mapped before -- members created at different times could
    // disagree, so the decision to repeat the handshake is the MAX, like every other branch of this function.
    long long hs_nonce = 0;   // (the epoch of this handshake: one more than the most any member has done, so that mail of an earlier one cannot answer it)
    {
        // ... and the epoch counter: members that have stepped in other groups before (or not at all) continue from the largest count among them, so
        // that everybody mails and expects the same number
        std::vector<long long> v(G.n, 0), ep(G.n, 0);
        for (int i = 0; i < G.n; ++i) { v[i] = remapped ? (long long)G.m[i]->direct.handshakes + 1 : 0; ep[i] = (long long)G.m[i]->direct.epoch; }
        long long max_epoch = 0;
        CHK(group_max(G, v.data(), &hs_nonce, ep.data(), &max_epoch));
        remapped = hs_nonce != 0;
        for (int i = 0; i < G.n; ++i) G.m[i]->direct.epoch = (uint32_t)max_epoch;
    }
    // Handshake through the mapped arenas, whenever a mapping is new: every member mails a reserved epoch to all members and waits (bounded) for
    // everybody's.  A mapping that opens but does not carry stores from one device into another device's polling kernel (a platform without working
    // peer / IPC coherence for fine-grained memory) is found HERE, where the job can still agree on the staged exchange, not in the first real
    // substep, where a timeout is fatal.
    if (remapped) {
        const uint32_t E0 = 0xFFFF0000u + (uint32_t)(hs_nonce & 0x7FFF);   // never a real epoch (they count up from 1)
        for (int i = 0; i < G.n; ++i) G.m[i]->direct.handshakes = (uint32_t)hs_nonce;
        static const double hs_s = getenv("SPARKL_HIP_DIRECT_HANDSHAKE_S") ? atof(getenv("SPARKL_HIP_DIRECT_HANDSHAKE_S")) : 5.0;
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            SparklPipeline::Direct& Dp = p->direct;
            HIPCHK(hipSetDevice(p->device));
            // (test hook, value 2: member 0 never mails, so every member's wait below really runs into its bound and the undo path is the one taken)
            if (getenv("SPARKL_HIP_DIRECT_TEST_FAIL") && atoi(getenv("SPARKL_HIP_DIRECT_TEST_FAIL")) == 2 && Dp.index == 0) continue;
            DirectPush D;
            memset(&D, 0, sizeof(D));
            D.n_mail = Dp.n; D.epoch = E0; D.arrive = Dp.arrive.p;
            for (int k = 0; k < Dp.n; ++k) D.mail[k] = &reinterpret_cast<PeerBox*>(Dp.peer_arena[k])->mail[E0 & 1u][Dp.index];
            hipLaunchKernelGGL(k_halo_push, dim3(1), dim3(BLK_THREADS), 0, p->stream, grid_arrays(p), D, p->slab_axis, p->d_ctl, (int)(p->substep_index & 1u));
        }
        std::vector<long long> bad(G.n, 0);
        for (int i = 0; i < G.n; ++i) {
            SparklPipeline* p = G.m[i];
            HIPCHK(hipSetDevice(p->device));
            DirectWait W{p->direct.box(), p->direct.n, E0, {0, 0}, (unsigned long long)(hs_s * 1.0e8), 0};
            hipLaunchKernelGGL(k_direct_agree, dim3(1), dim3(64), 0, p->stream, W, p->d_ctl, (int)(p->substep_index & 1u));
            CHK(read_ctl(p));
            if (p->h_ctl->direct_error) {   // undo what the give-up raised: this is a probe, not a substep
                bad[i] = 1;
                HIPCHK(hipMemsetAsync(p->d_ctl, 0, 2 * sizeof(uint32_t), p->stream));
                HIPCHK(hipMemsetAsync(&p->d_ctl->direct_error, 0, sizeof(uint32_t), p->stream));
                if (p->h_words) p->h_words[0] = 0u;
                CHK(comm_wait(p, p->stream));
            }
        }
        // (test hook: member 0 of the job reports a failed handshake -- the rank-asymmetric case the agreement below exists for)
        if (getenv("SPARKL_HIP_DIRECT_TEST_FAIL") && atoi(getenv("SPARKL_HIP_DIRECT_TEST_FAIL")) == 1) for (int i = 0; i < G.n; ++i) if (G.m[i]->direct.index == 0) bad[i] = 1;
        long long any_bad = 0;
        CHK(group_max(G, bad.data(), &any_bad));
        if (any_bad) {
            for (int i = 0; i < G.n; ++i) { G.m[i]->direct.ready = false; G.m[i]->direct.failed = true; }
            if (getenv("SPARKL_HIP_DEBUG")) fprintf(stderr, "[sparkl_hip] direct halo exchange: the handshake through the mapped arenas timed out: staying on the RCCL / copy exchange\n");
            return SPARKL_OK;
        }
    }
    for (int i = 0; i < G.n; ++i) G.m[i]->direct.ready = true;
    return SPARKL_OK;
}
}  // namespace
static void direct_release(SparklPipeline* p) {
    SparklPipeline::Direct& D = p->direct;
    for (int k = 0; k < MAX_DIRECT; ++k) { if (D.peer_ipc[k]) (void)hipIpcCloseMemHandle(D.peer_ipc[k]); D.peer_ipc[k] = nullptr; D.peer_arena[k] = nullptr; D.peer_serial[k] = 0; }
    if (D.arena) (void)hipFree(D.arena);
    D.arena = nullptr; D.arena_bytes = 0; D.cap_blocks = 0;
    D.dst[0].release(); D.dst[1].release(); D.arrive.release();
    D.ready = false;
}

// halo stream and fork / join events of a member (created on first use)
static int overlap_resources(SparklPipeline* p) {
    HIPCHK(hipSetDevice(p->device));
    if (!p->halo_stream) HIPCHK(hipStreamCreateWithFlags(&p->halo_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&p->ev_main, &p->ev_halo, &p->ev_flag, &p->ev_grid, &p->ev_xfer}) if (!*e) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return SPARKL_OK;
}
// ... and back, once a job has settled on a schedule that does not use the second stream: an idle stream is still a hardware queue (eight processes of one
// device with two queues each ran 12.8 ms per substep where the same eight with one queue each ran 0.69: the queues were time-sliced)
static void overlap_release(SparklPipeline* p) {
    if (!p->halo_stream) return;
    (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->halo_stream);
    (void)hipStreamDestroy(p->halo_stream);
    p->halo_stream = nullptr;
}
// The node / particle kernels of one substep over a subset of the blocks (GridArrays::pb_list / nb_list), on stream `st`.
// which = 0: grid update of the interior node blocks, 1: of the cut-adjacent ones; 2 / 3: the particle kernel likewise.
static int launch_subset(SparklPipeline* p, int which, bool fuse_next, bool primary, hipStream_t st, bool defer = true) {
    GridArrays G = grid_arrays(p);
    const bool bnd = which & 1;
    StepParams S = p->pending;
    if (which < 2) {
        G.nb_list = bnd ? p->split_nb.p : p->split_nb.p + p->nb_bnd;
        G.nb_count = bnd ? p->nb_bnd : p->num_blocks - p->nb_bnd;
        if (G.nb_count == 0) return SPARKL_OK;
        G.node_proj = nullptr;   // (two streams: the projection cache is only handed out by the single-stream launch_grid_update)
        launch_grid_kernel(p, G, S, dim3((G.nb_count + 3) / 4), st);
        return SPARKL_OK;
    }
    G.pb_plain = p->split_ordered ? (bnd ? 1 : 2) : 0;
    G.pb_list = bnd ? p->split_pb.p : p->split_pb.p + p->pb_bnd;
    G.pb_count = bnd ? p->pb_bnd : p->num_pblocks - p->pb_bnd;
    G.primary = primary ? 1 : 0;
    if (G.pb_count == 0) return SPARKL_OK;
    const ParticleArrays P = with_mirror(p, p->cur.arrays());   // (k_g2p below stores flags too)
    const dim3 grid(G.pb_count), block(BLK_THREADS);
    if (fuse_next && p->fuse_fixed_dt) {
        S.defer_c = (defer && p->defer_c && !p->rare_fused() && !S.project_particles) ? 1 : 0;
        if (S.defer_c) p->c_stale_S = p->pending;
        launch_g2p2g(p, P, G, S, st);
    } else {
        if (p->rare_models || S.project_particles) SPK_LAUNCH_MODELS(k_g2p, true);
        else SPK_LAUNCH_MODELS(k_g2p, false);
    }
    return SPARKL_OK;
}

extern "C" {

int sparkl_hip_comm_unique_id(void* out_128_bytes) {
    if (!out_128_bytes) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    CHK(rccl_load());
    static_assert(sizeof(ncclUniqueId) == 128, "");
    NCCLCHK(g_rccl.GetUniqueId((ncclUniqueId*)out_128_bytes));
    return SPARKL_OK;
}
int sparkl_hip_comm_init(SparklPipeline* p, const void* unique_id_128_bytes, int rank, int world) {
    if (!p || !unique_id_128_bytes || world < 1 || rank < 0 || rank >= world) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    CHK(rccl_load());
    HIPCHK(hipSetDevice(p->device));
    comm_destroy(p);
    direct_reset(p);
    p->comm_aborted = false;
    ncclUniqueId id;
    memcpy(&id, unique_id_128_bytes, sizeof(id));
    NCCLCHK(g_rccl.CommInitRank(&p->comm, world, id, rank));
    p->comm_rank = rank; p->comm_world = world;
    return SPARKL_OK;
}
/* One host thread, several devices (cuda_mpm_pipeline.rs:262-271 steps `&mut [SingleGpuMpmContext]` from one call): member i joins the job as
 * rank first_rank + i.  The ncclCommInitRank calls sit inside ncclGroupStart / End -- issued one after the other from one thread they would
 * deadlock, each waiting for the ranks behind it. */
int sparkl_hip_comm_init_all(SparklPipeline** members, uint32_t n_members, const void* unique_id_128_bytes, int first_rank, int world) {
    if (!members || n_members == 0 || !unique_id_128_bytes || world < 1 || first_rank < 0 || first_rank + (int)n_members > world)
        return set_error(SPARKL_ERR_INVALID_ARGUMENT, "bad argument");
    for (uint32_t i = 0; i < n_members; ++i) {
        if (!members[i]) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null member");
        for (uint32_t j = 0; j < i; ++j)
            if (members[j]->device == members[i]->device) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "members %u and %u share device %d: RCCL takes one rank per device and process", j, i, members[i]->device);
    }
    CHK(rccl_load());
    ncclUniqueId id;
    memcpy(&id, unique_id_128_bytes, sizeof(id));
    for (uint32_t i = 0; i < n_members; ++i) { HIPCHK(hipSetDevice(members[i]->device)); comm_destroy(members[i]); direct_reset(members[i]); members[i]->comm_aborted = false; }
    NCCLCHK(g_rccl.GroupStart());
    for (uint32_t i = 0; i < n_members; ++i) {
        HIPCHK(hipSetDevice(members[i]->device));
        const ncclResult_t r = g_rccl.CommInitRank(&members[i]->comm, world, id, first_rank + (int)i);
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return set_error(SPARKL_ERR_HIP, "ncclCommInitRank of member %u failed: %s", i, g_rccl.GetErrorString(r)); }
    }
    NCCLCHK(g_rccl.GroupEnd());
    for (uint32_t i = 0; i < n_members; ++i) { members[i]->comm_rank = first_rank + (int)i; members[i]->comm_world = world; }
    return SPARKL_OK;
}
/* SUM over the ranks of the communicator of one 64-bit value (host-synchronous; the bench uses it to count its ranks) */
int sparkl_hip_comm_allreduce_sum(SparklPipeline* p, long long* value) {
    if (!p || !value) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (!p->comm) return SPARKL_OK;
    HIPCHK(hipSetDevice(p->device));
    CHK(x_staging(p));
    p->x_host[0] = *value;
    HIPCHK(hipMemcpyAsync(p->x_dev.p, p->x_host, sizeof(long long), hipMemcpyHostToDevice, p->stream));
    NCCLCHK(g_rccl.AllReduce(p->x_dev.p, p->x_dev.p, 1, ncclInt64, ncclSum, p->comm, p->stream));
    HIPCHK(hipMemcpyAsync(p->x_host, p->x_dev.p, sizeof(long long), hipMemcpyDeviceToHost, p->stream));
    CHK(comm_wait(p, p->stream));
    *value = p->x_host[0];
    return SPARKL_OK;
}

// `pick_dt` (adaptive steps): called once per substep BEHIND the fluid volume sweep of that substep and before its scatter; returns the length
// the substep is run with (mls_solver.rs:46-59: volumes, then the estimate, then the substep)
typedef std::function<int(float*)> PickDt;
static int group_substeps_impl(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params, const float gravity[3], float dt, uint32_t nsubsteps, const PickDt* pick_dt = nullptr);
static bool any_comm(SparklPipeline** members, uint32_t n) { for (uint32_t i = 0; members && i < n; ++i) if (members[i] && members[i]->comm) return true; return false; }
static int aborted_check(SparklPipeline** members, uint32_t n) {
    for (uint32_t i = 0; members && i < n; ++i)
        if (members[i] && members[i]->comm_aborted) return set_error(SPARKL_ERR_HIP, "member %u: its RCCL communicator was aborted after an earlier error; create a new one (sparkl_hip_comm_init)", i);
    return SPARKL_OK;
}
// A group call that fails BEHIND its first collective leaves the peers waiting in theirs: the members' communicators are aborted so that they
// fail (their own comm_wait: asynchronous error, or the SPARKL_HIP_COMM_TIMEOUT_S deadline -- a local ncclCommAbort does not reliably wake a
// peer on intra-node transports, the deadline is what releases it).  A call that fails in validation, before anything was issued, has not
// desynchronised anybody: the communicator stays (a caller's mistake used to destroy it).
static thread_local bool g_group_engaged = false;
int sparkl_hip_group_substeps(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params, const float gravity[3], float dt, uint32_t nsubsteps) {
    CHK(aborted_check(members, n_members));
    const bool with_comm = any_comm(members, n_members);
    g_group_engaged = false;
    const int rc = group_substeps_impl(members, n_members, params, gravity, dt, nsubsteps);
    if (rc != SPARKL_OK && with_comm && g_group_engaged) group_abort(members, n_members);
    return rc;
}
static int group_substeps_impl(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params, const float gravity[3], float dt, uint32_t nsubsteps, const PickDt* pick_dt) {
    if (!members || n_members == 0 || !params || !gravity) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    Group G{members, (int)n_members, true, true};
    for (uint32_t i = 0; i < n_members; ++i) if (members[i]) G.balance = G.balance && members[i]->slab_balance;
    for (int i = 0; i < G.n; ++i) {
        if (!members[i] || !members[i]->slab_on) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "member %d is not a slab pipeline (sparkl_hip_slab_set)", i);
        G.rccl = G.rccl && members[i]->comm != nullptr;
    }
    if (!G.rccl)
        for (int i = 0; i < G.n; ++i) {
            if (members[i]->comm) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "either every member has a communicator or none");
            if (members[i]->device != members[0]->device || members[i]->stream != members[0]->stream)
                return set_error(SPARKL_ERR_INVALID_ARGUMENT, "members without a communicator must share one device and one stream");
        }
    if (!damage_supported(params)) return unsupported_damage_model(params);
    // Collective by construction: decided by the solver parameters alone, never by what a rank happens to hold (a slab without fluid
    // or without the ParticleFracture component still takes part in the extra halo rounds, with zeros).
    const bool fluid_sweep = params->force_fluids_volume_recomputation != 0;
    const bool psi_sweep = params->damage_model == SPARKL_DAMAGE_MODIFIED_EIGENEROSION;
    if (psi_sweep) for (int i = 0; i < G.n; ++i) CHK(ensure_fracture(members[i]));
    for (int i = 0; i < G.n; ++i) { HIPCHK(hipSetDevice(members[i]->device)); CHK(x_staging(members[i])); }
    g_group_engaged = true;   // validation is over: from here on an error leaves peers waiting (see sparkl_hip_group_substeps)
    bool need_rebin = false;
    for (int i = 0; i < G.n; ++i) need_rebin |= !members[i]->sorted || members[i]->group_rebins == 0;
    {   // a member that still has to (re-)bin makes the whole job do it: the re-bin is collective
        std::vector<long long> v(G.n, need_rebin ? 1 : 0), veto(G.n, balance_veto(G));
        long long any = 0, no_balance = 0;
        CHK(group_max(G, v.data(), &any, veto.data(), &no_balance));
        bool first = true;
        for (int i = 0; i < G.n; ++i) first = first && members[i]->group_rebins == 0;
        if (any) CHK(group_rebin(G, no_balance == 0));
        // The cuts the caller chose meet the particle histogram once, right behind the first collective re-bin (which builds the block list
        // the histogram is taken from): a scene at rest may not ask for another re-bin for a long time.  Agreed like every re-bin.
        // (`any` is the reduction's result and `first` counts collective re-bins: both are the same on every rank, so the second agreement --
        // a host-synchronous all-reduce -- only runs behind the first collective re-bin, not on every call: ADVICE r04)
        long long any2 = 0, no_balance2 = 0;
        if (any && first) {
            std::vector<long long> v2(G.n, 1), veto2(G.n, balance_veto(G));
            CHK(group_max(G, v2.data(), &any2, veto2.data(), &no_balance2));
        }
        if (any2 && no_balance2 == 0) {
            CHK(group_rebalance(G));
            bool moved = false;
            for (int i = 0; i < G.n; ++i) moved = moved || !members[i]->sorted;
            std::vector<long long> v3(G.n, moved ? 1 : 0);
            long long any3 = 0;
            CHK(group_max(G, v3.data(), &any3));
            if (any3) CHK(group_rebin(G, false));
        }
    }
    struct StoreMode {   // halo records are stored, not added, while this loop owns the substeps (see k_halo_merge2)
        const Group& g;
        explicit StoreMode(const Group& gg) : g(gg) { for (int i = 0; i < g.n; ++i) g.m[i]->halo_store_mode = true; }
        ~StoreMode() { for (int i = 0; i < g.n; ++i) g.m[i]->halo_store_mode = false; }
    } store_mode(G);
    std::vector<uint32_t> base(G.n);
    for (int i = 0; i < G.n; ++i) { HIPCHK(hipSetDevice(members[i]->device)); CHK(read_ctl(members[i])); base[i] = members[i]->h_ctl->steps_done; }
    // Halo / compute overlap: forced by SPARKL_HIP_OVERLAP, else decided by measurement between RCCL ranks (never on one device: see
    // overlap_mode).  The two-stream schedule uses ONE communicator from two streams (the flag all-reduce and the halo Send / Recv on the
    // halo stream; the host-synchronous reductions around re-bins on the main stream): correct because every rank issues the same
    // operations in the same order and the two streams are joined at the end of every substep.
    const bool overlap_legal = !fluid_sweep && !psi_sweep;
    bool overlap_auto = overlap_legal;
    bool overlap_forced = overlap_legal;
    for (int i = 0; i < G.n; ++i) {
        overlap_auto = overlap_auto && (members[i]->overlap_mode == -2 || (members[i]->overlap_mode < 0 && G.rccl && members[i]->comm_world > 1));
        overlap_forced = overlap_forced && members[i]->overlap_mode == 1;
    }
    auto schedule_of_batch = [&]() {   // identical on every rank: the calibration state only advances on collectively agreed facts
        if (!overlap_auto) return overlap_forced;
        const SparklPipeline* m0 = members[0];
        return m0->overlap_cal_state == 2 ? m0->overlap_choice == 1 : m0->overlap_cal_state == 1;
    };
    uint32_t done = 0;
    int guard = 0;
    std::vector<std::array<uint64_t, 2>> bs(G.n), br(G.n);
    while (done < nsubsteps) {
        const uint32_t batch = std::min<uint32_t>(16, nsubsteps - done);
        // peer-direct exchange (kernels.h, PeerBox): set up by the last collective re-bin on every rank or on none
        bool direct = !fluid_sweep && !psi_sweep;
        for (int i = 0; i < G.n; ++i) direct = direct && members[i]->direct.ready;
        // three-way calibration (see direct_cal_state): identical on every rank -- the states only advance on collectively agreed facts
        const bool exchange_auto = direct && overlap_auto && (direct_policy() == 2 || (direct_policy() == 0 && G.rccl && members[0]->comm_world > 1));
        if (exchange_auto) {
            const SparklPipeline* m0 = members[0];
            direct = m0->direct_choice == 1 || (m0->direct_choice < 0 && m0->direct_cal_state < 2);
        }
        const bool direct_cal_batch = exchange_auto && direct && members[0]->direct_choice < 0;
        G.overlap = !direct && schedule_of_batch();
        if (G.overlap) for (int i = 0; i < G.n; ++i) CHK(overlap_resources(members[i]));
        const auto batch_t0 = std::chrono::steady_clock::now();
        for (uint32_t j = 0; j < batch; ++j) {
            // 1. the re-bin request of the upcoming substep, MAX over every slab: all ranks skip the same substeps.  In place on the
            //    control blocks: one 4-byte all-reduce (RCCL) or one single-thread kernel (members of one process)
            if (G.overlap)   // fork: the halo stream sees everything the main stream has done (the tiles of this substep among it)
                for (int i = 0; i < G.n; ++i) {
                    HIPCHK(hipSetDevice(members[i]->device));
                    HIPCHK(hipEventRecord(members[i]->ev_main, members[i]->stream));
                    HIPCHK(hipStreamWaitEvent(G.xs(i), members[i]->ev_main, 0));
                }
            if (direct) {
                // nothing here: every member's k_halo_push mails its flag to all members and their grid updates take the MAX
            } else if (G.rccl) {
                NCCLCHK(g_rccl.GroupStart());
                for (int i = 0; i < G.n; ++i) {
                    uint32_t* flag = &members[i]->d_ctl->abort_flag[members[i]->substep_index & 1u];
                    NCCLCHK(g_rccl.AllReduce(flag, flag, 1, ncclUint32, ncclMax, members[i]->comm, G.xs(i)));
                }
                NCCLCHK(g_rccl.GroupEnd());
            } else if (G.n > 1) {
                // up to 16 control blocks per launch; longer groups chain chunks through members[0] and run the chain twice
                const int parity = (int)(members[0]->substep_index & 1u);
                for (int pass = 0; pass < (G.n > 16 ? 2 : 1); ++pass)
                    for (int i0 = 1; i0 < G.n; i0 += 15) {
                        CtlList L;
                        L.n = 0;
                        L.c[L.n++] = members[0]->d_ctl;
                        for (int i = i0; i < G.n && L.n < 16; ++i) L.c[L.n++] = members[i]->d_ctl;
                        hipLaunchKernelGGL(k_flags_max, dim3(1), dim3(1), 0, G.xs(0), L, parity);
                    }
            }
            if (G.overlap)   // every kernel of the substep reads the agreed flag: the main stream waits for it (and for nothing else)
                for (int i = 0; i < G.n; ++i) {
                    HIPCHK(hipSetDevice(members[i]->device));
                    HIPCHK(hipEventRecord(members[i]->ev_flag, G.xs(i)));
                    HIPCHK(hipStreamWaitEvent(members[i]->stream, members[i]->ev_flag, 0));
                }
            // 2. scatter (nothing when the previous fused kernel already did), pack the shared planes of both cuts, swap, store
            // One halo round: every member packs its partial sums of the shared node planes (from the mass / momentum tiles, or from
            // the psi tiles), the records travel to both neighbours and are stored into halo_in, which the next node kernel adds.
            auto halo_round = [&](bool psi) -> int {
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    for (int side = 0; side < 2; ++side) { bs[i][side] = p->x_nsend[side] * sizeof(HaloRecord); br[i][side] = p->x_nrecv[side] * sizeof(HaloRecord); }
                    const int n0 = (int)p->x_nsend[0], n1 = (int)p->x_nsend[1];
                    if (n0 + n1 == 0 || p->num_pblocks == 0) continue;
                    HIPCHK(hipSetDevice(p->device));
                    HaloSides hs{};
                    for (int side = 0; side < 2; ++side) { hs.list[side] = p->halo_list[side].p; hs.n[side] = (int)p->x_nsend[side]; hs.out[side] = (HaloRecord*)p->x_send[side].p; }
                    size_t tok = 0;
                    CHK(clock_begin(p, SPAN_HALO, &tok, G.xs(i)));
                    hipLaunchKernelGGL(k_halo_pack2, dim3((n0 + n1 + 3) / 4), dim3(BLK_THREADS), 0, G.xs(i), grid_arrays(p), p->blk_keys.p, hs, p->slab_axis, p->d_ctl, p->pending.parity,
                                       psi ? (const float2*)p->psi_tiles.p : (const float2*)nullptr);
                    CHK(clock_end(p, tok, G.xs(i)));
                }
                {
                    size_t tok = 0;
                    CHK(clock_begin(members[0], SPAN_HALO, &tok, G.xs(0)));
                    CHK(group_exchange(G, &SparklPipeline::x_send, &SparklPipeline::x_recv, (const uint64_t(*)[2])bs.data(), (const uint64_t(*)[2])br.data()));
                    CHK(clock_end(members[0], tok, G.xs(0)));
                }
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    const int n0 = (int)p->x_nrecv[0], n1 = (int)p->x_nrecv[1];
                    if (n0 + n1 == 0 || p->num_blocks == 0) continue;
                    HIPCHK(hipSetDevice(p->device));
                    HaloSides hs{};
                    for (int side = 0; side < 2; ++side) { hs.in[side] = (const HaloRecord*)p->x_recv[side].p; hs.nin[side] = (int)p->x_nrecv[side]; }
                    HashTable H{p->hash_keys.p, p->hash_vals.p, p->hash_mask};
                    size_t tok = 0;
                    CHK(clock_begin(p, SPAN_HALO, &tok, G.xs(i)));
                    hipLaunchKernelGGL(k_halo_merge2, dim3((n0 + n1 + 3) / 4), dim3(BLK_THREADS), 0, G.xs(i), hs, p->slab_axis, H, p->halo_in.p, p->d_ctl, p->pending.parity);
                    CHK(clock_end(p, tok, G.xs(i)));
                }
                return SPARKL_OK;
            };
            if (fluid_sweep) {
                // fluids_volume.rs:9-115 across the cuts: mass-only scatter, the mass planes of the cuts swapped like the momentum planes
                // (the records carry the mass alone), merge, volume gather -- then the substep proper.  Unfused order.
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    HIPCHK(hipSetDevice(p->device));
                    p->pending = make_params(p, params, gravity, dt);
                    if (p->num_pblocks) hipLaunchKernelGGL(k_p2g_mass, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), grid_arrays(p), p->d_ctl, p->pending);
                }
                CHK(halo_round(false));
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    if (!p->num_pblocks) continue;
                    HIPCHK(hipSetDevice(p->device));
                    GridArrays Gm = grid_arrays(p);
                    Gm.node_mom = nullptr;
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_update<true>), dim3((p->num_blocks + 3) / 4), dim3(BLK_THREADS), 0, p->stream, Gm, p->d_cuboids.p, p->d_heightfields.p,
                                       p->d_trimeshes.p, p->d_ctl, p->pending);
                    hipLaunchKernelGGL(k_g2p_volume, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), Gm, p->d_models.p, p->d_ctl, p->pending);
                }
            }
            if (pick_dt) CHK((*pick_dt)(&dt));   // adaptive: the length of THIS substep, estimated on the volumes the sweep has just rewritten
            for (int i = 0; i < G.n; ++i) CHK(sparkl_hip_substep_begin(members[i], params, gravity, dt));
            // the fused kernel also closes a batch that another one follows inside this call: the next batch then starts from its tiles
            // instead of a standalone P2G (0.35 + 0.34 -> 0.5 ms per 16 substeps at 10M).  That launch stores C in full (`defer`): the host
            // looks at the control block between batches and may re-bin there
            const bool last_of_batch = j + 1 == batch;
            const bool fuse_next = (!last_of_batch || done + batch < nsubsteps) && !fluid_sweep && !psi_sweep;
            const bool defer = !last_of_batch;
            if (G.overlap) {
                // Halo / compute overlap (the reference runs its halo work on a second stream too: cuda_mpm_pipeline.rs:410-430).
                //   main stream:  [P2G, reorder]  grid update of the interior node blocks ; particle kernel of the interior particle blocks
                //   halo stream:  pack ; send / recv ; store ; grid update of the node planes that received ; particle kernel of the
                //                 particle blocks within two block planes of a cut (the ones whose gather reads those planes)
                // The cut-adjacent particle kernel also reads interior node planes (it waits for the interior grid update) and rewrites
                // tiles the interior grid update reads (same wait); nothing on the main stream reads what the halo stream writes before
                // the join at the end of the substep.
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    HIPCHK(hipSetDevice(p->device));
                    if (p->num_pblocks) CHK(maybe_reorder(p, p->pending.parity));
                    HIPCHK(hipEventRecord(p->ev_xfer, p->stream));
                    HIPCHK(hipStreamWaitEvent(G.xs(i), p->ev_xfer, 0));
                }
                CHK(halo_round(false));
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    HIPCHK(hipSetDevice(p->device));
                    if (p->num_pblocks == 0) { hipLaunchKernelGGL(k_bump_steps, dim3(1), dim3(1), 0, p->stream, p->d_ctl, p->pending.parity); continue; }
                    const bool has_interior = p->num_pblocks > p->pb_bnd;
                    size_t t1 = 0, t2 = 0;
                    CHK(clock_begin(p, SPAN_GRID, &t1));
                    CHK(launch_subset(p, 0, fuse_next, false, p->stream));
                    CHK(clock_end(p, t1));
                    HIPCHK(hipEventRecord(p->ev_grid, p->stream));
                    CHK(clock_begin(p, fuse_next && p->fuse_fixed_dt ? SPAN_FUSED : SPAN_G2P, &t2));
                    CHK(launch_subset(p, 2, fuse_next, has_interior, p->stream, defer));
                    CHK(clock_end(p, t2));
                    CHK(launch_subset(p, 1, fuse_next, false, G.xs(i)));
                    HIPCHK(hipStreamWaitEvent(G.xs(i), p->ev_grid, 0));
                    CHK(launch_subset(p, 3, fuse_next, !has_interior, G.xs(i), defer));
                }
                for (int i = 0; i < G.n; ++i) {   // join
                    SparklPipeline* p = members[i];
                    HIPCHK(hipSetDevice(p->device));
                    HIPCHK(hipEventRecord(p->ev_halo, G.xs(i)));
                    HIPCHK(hipStreamWaitEvent(p->stream, p->ev_halo, 0));
                    p->in_substep = false;
                    p->tiles_ready = p->num_pblocks > 0 && fuse_next && p->fuse_fixed_dt;
                    p->clock_on = p->pending_clock_saved;
                    p->substep_index++;
                    if (p->num_pblocks) p->clock.substeps++;
                }
                HIPCHK(hipGetLastError());
                continue;
            }
            if (direct) {
                // ONE kernel per member in front of the grid update: flag mail to everybody, records straight into the neighbours' halo buffers
                static const double tmo_s = std::min(20.0, comm_timeout_s() > 0.0 ? comm_timeout_s() : 20.0);   // (a kernel may not spin for minutes: the driver's compute watchdog)
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    SparklPipeline::Direct& Dp = p->direct;
                    HIPCHK(hipSetDevice(p->device));
                    const uint32_t epoch = ++Dp.epoch;
                    const int par = (int)(epoch & 1u);
                    DirectPush D;
                    memset(&D, 0, sizeof(D));
                    D.n_mail = Dp.n; D.epoch = epoch; D.arrive = Dp.arrive.p;
                    for (int k = 0; k < Dp.n; ++k) D.mail[k] = &reinterpret_cast<PeerBox*>(Dp.peer_arena[k])->mail[par][Dp.index];
                    p->dw.box = Dp.box(); p->dw.n_mail = Dp.n; p->dw.epoch = epoch; p->dw.timeout_ticks = (unsigned long long)(tmo_s * 1.0e8);
                    p->dw.pre_agreed = Dp.shared_device ? 1 : 0;
                    for (int side = 0; side < 2; ++side) {
                        const bool has = G.nbr(i, side) != -2;
                        p->dw.wait_side[side] = has ? 1 : 0;
                        if (!has) continue;
                        const int q = Dp.index + (side == 0 ? -1 : 1);
                        D.list[side] = p->halo_list[side].p; D.dst[side] = Dp.dst[side].p; D.n[side] = p->num_pblocks ? (int)p->x_nsend[side] : 0;
                        D.remote_halo[side] = direct_halo(p, Dp.peer_arena[q], Dp.peer_cap[q], par);
                        D.remote_epoch[side] = &reinterpret_cast<PeerBox*>(Dp.peer_arena[q])->halo_epoch[par][1 - side];
                    }
                    size_t tok = 0;
                    CHK(clock_begin(p, SPAN_HALO, &tok));
                    hipLaunchKernelGGL(k_halo_push, dim3(std::max(1, (D.n[0] + D.n[1] + 3) / 4)), dim3(BLK_THREADS), 0, p->stream, grid_arrays(p), D, p->slab_axis, p->d_ctl, p->pending.parity);
                    // (ranks that share a device: one wave waits for everybody, the grid update behind it does not spin)
                    if (p->dw.pre_agreed && p->num_pblocks) hipLaunchKernelGGL(k_direct_agree, dim3(1), dim3(64), 0, p->stream, p->dw, p->d_ctl, p->pending.parity);
                    CHK(clock_end(p, tok));
                }
                for (int i = 0; i < G.n; ++i) {
                    const int rc = substep_finish_impl(members[i], fuse_next, defer);
                    members[i]->dw.box = nullptr;
                    CHK(rc);
                }
                HIPCHK(hipGetLastError());
                continue;
            }
            CHK(halo_round(false));
            if (psi_sweep) {
                // DamageModel::ModifiedEigenerosion across the cuts: the psi_mass / psi_momentum planes travel like the momentum planes,
                // between the grid update and the crack test (grid_to_particle.rs:57-77)
                for (int i = 0; i < G.n; ++i) CHK(substep_finish_impl(members[i], false, false, 1));
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    if (!p->num_pblocks) continue;
                    HIPCHK(hipSetDevice(p->device));
                    CHK(p->psi_tiles.reserve((size_t)p->num_pblocks * TILE_N));
                    CHK(p->node_psi.reserve((size_t)p->num_blocks * 64));
                    if (p->cur.has_frac) hipLaunchKernelGGL(k_p2g_psi, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), grid_arrays(p), p->d_ctl, p->pending, p->psi_tiles.p);
                    else HIPCHK(hipMemsetAsync(p->psi_tiles.p, 0, (size_t)p->num_pblocks * TILE_N * sizeof(float2), p->stream));   // no component here: zeros to the neighbours
                }
                CHK(halo_round(true));
                for (int i = 0; i < G.n; ++i) {
                    SparklPipeline* p = members[i];
                    if (!p->num_pblocks) continue;
                    HIPCHK(hipSetDevice(p->device));
                    const GridArrays Gp = grid_arrays(p);
                    hipLaunchKernelGGL(k_grid_psi, dim3((p->num_blocks + 3) / 4), dim3(BLK_THREADS), 0, p->stream, Gp, p->psi_tiles.p, p->node_psi.p, p->d_ctl, p->pending.parity,
                                       (const float4*)p->halo_in.p);
                    if (p->cur.has_frac) hipLaunchKernelGGL(k_g2p_psi, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), Gp, p->node_psi.p, p->d_ctl, p->pending);
                }
                for (int i = 0; i < G.n; ++i) CHK(substep_finish_impl(members[i], false, false, 2));
            } else {
                for (int i = 0; i < G.n; ++i) CHK(substep_finish_impl(members[i], fuse_next, defer));
            }
            HIPCHK(hipGetLastError());
        }
        // one host read per batch; the flag is global, so every member ran the same substeps
        long long need = 0;
        uint32_t ran = 0;
        std::vector<long long> needs(G.n, 0);
        for (int i = 0; i < G.n; ++i) {
            HIPCHK(hipSetDevice(members[i]->device));
            CHK(read_ctl(members[i]));
            if (members[i]->clock_on) CHK(clock_collect(members[i]));
            if (members[i]->h_ctl->direct_error)
                return set_error(SPARKL_ERR_HIP, "member %d: a kernel gave up waiting for a peer's %s (direct halo exchange; the peer is gone?)", i, members[i]->h_ctl->direct_error == 1 ? "re-bin flag" : "halo records");
            const uint32_t r = members[i]->h_ctl->steps_done - base[i];
            // cut short behind a fused launch that deferred C.m[1..8] (the last launch of a whole batch stores C, fused or not)
            if (r > done && r < done + batch && !fluid_sweep && !psi_sweep && members[i]->fuse_fixed_dt && members[i]->defer_c && !members[i]->rare_fused() &&
                !params->enable_boundary_particle_projection)
                members[i]->c_stale = true;
            ran = i == 0 ? r : std::min(ran, r);
            needs[i] = (r < done + batch || members[i]->h_ctl->abort_flag[0] || members[i]->h_ctl->abort_flag[1]) ? 1 : 0;
        }
        std::vector<long long> veto(G.n, balance_veto(G));
        long long no_balance = 0;
        CHK(group_max(G, needs.data(), &need, veto.data(), &no_balance));
        if (direct_cal_batch && batch == 16 && ran == done + batch && !need) {
            if (members[0]->direct_cal_state == 0) { for (int i = 0; i < G.n; ++i) members[i]->direct_cal_state = 1; }   // the batch that warms the job up
            else {
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - batch_t0).count() / batch;
                std::vector<long long> a(G.n, (long long)(us * 1000.0));
                long long amax = 0;
                CHK(group_max(G, a.data(), &amax));
                for (int i = 0; i < G.n; ++i) { members[i]->direct_cal_us = amax * 1e-3; members[i]->direct_cal_state = 2; }
            }
        }
        if (exchange_auto && members[0]->direct_choice < 0 && members[0]->direct_cal_state == 2 && members[0]->overlap_cal_state == 2) {
            // all three timed (every rank holds the same three maxima): the fastest stays
            const double staged = members[0]->overlap_cal_us[1] > 0.0 ? std::min(members[0]->overlap_cal_us[0], members[0]->overlap_cal_us[1]) : members[0]->overlap_cal_us[0];
            const int choice = members[0]->direct_cal_us <= staged ? 1 : 0;
            for (int i = 0; i < G.n; ++i) members[i]->direct_choice = choice;
            if (choice == 1 || members[0]->overlap_choice != 1) for (int i = 0; i < G.n; ++i) overlap_release(members[i]);
            if (getenv("SPARKL_HIP_DEBUG"))
                fprintf(stderr, "[sparkl_hip] halo exchange timed on this job: direct %.1f us per substep, staged in line %.1f, staged overlapped %.1f -> %s\n", members[0]->direct_cal_us,
                        members[0]->overlap_cal_us[0], members[0]->overlap_cal_us[1], choice ? "direct" : (members[0]->overlap_choice == 1 ? "staged, overlapped" : "staged, in line"));
        }
        if (!direct && overlap_auto && members[0]->overlap_cal_state < 2 && batch == 16 && ran == done + batch && !need) {
            // an undisturbed batch: its wall clock (enqueue to control-block read) is this schedule's sample
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - batch_t0).count() / batch;
            const int which = members[0]->overlap_cal_state;   // 0: this batch ran in line, 1: overlapped
            for (int i = 0; i < G.n; ++i) { members[i]->overlap_cal_us[which] = us; members[i]->overlap_cal_state = which + 1; }
            // Ranks that share a physical device (a one-GPU box standing in for a node) never time the two-stream schedule: a second hardware queue per
            // process is what it costs there, whatever it hides -- eight processes with two queues each ran 20 ms per substep in that batch and 12.8 ms in every
            // batch after it (the queue outlives its stream), against 0.69 with one queue each.  The in-line time stands for the staged exchange.
            if (which == 0 && members[0]->direct.shared_device) {
                std::vector<long long> a(G.n, (long long)(us * 1000.0));
                long long amax = 0;
                CHK(group_max(G, a.data(), &amax));
                for (int i = 0; i < G.n; ++i) { members[i]->overlap_cal_us[0] = amax * 1e-3; members[i]->overlap_cal_us[1] = 0.0; members[i]->overlap_cal_state = 2; members[i]->overlap_choice = 0; }
            }
            if (which == 1) {
                std::vector<long long> a(G.n), b(G.n);
                for (int i = 0; i < G.n; ++i) { a[i] = (long long)(members[i]->overlap_cal_us[0] * 1000.0); b[i] = (long long)(members[i]->overlap_cal_us[1] * 1000.0); }
                long long amax = 0, bmax = 0;
                CHK(group_max(G, a.data(), &amax, b.data(), &bmax));   // the slowest rank sets the pace of either schedule
                for (int i = 0; i < G.n; ++i) {
                    members[i]->overlap_cal_us[0] = amax * 1e-3; members[i]->overlap_cal_us[1] = bmax * 1e-3;
                    members[i]->overlap_choice = bmax < amax ? 1 : 0;
                }
                if (!exchange_auto && members[0]->overlap_choice == 0) for (int i = 0; i < G.n; ++i) overlap_release(members[i]);
            }
        }
        done = ran;
        if (need) {
            CHK(group_rebin(G, no_balance == 0));
            if (++guard > (int)nsubsteps * 2 + 16) return set_error(SPARKL_ERR_INTERNAL, "re-bin loop did not make progress");
        }
    }
    return SPARKL_OK;
}
// `MpmPipeline::step` across the slabs: params->dt of simulated time in adaptive substeps whose length follows mls_solver.rs:33-59
// with the dt bound (timestep_estimator.rs:10-51) taken over ALL members and ranks: per substep every member exports its local
// bound, the MIN travels through RCCL (f32 bits of a non-negative float order like the float), one host read, then one lockstep
// substep of that length through the same loop as sparkl_hip_group_substeps.
static int group_step_impl(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params, const float gravity[3], float* simulated_time, uint32_t* num_substeps);
int sparkl_hip_group_step(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params, const float gravity[3], float* simulated_time,
                          uint32_t* num_substeps) {
    CHK(aborted_check(members, n_members));
    const bool with_comm = any_comm(members, n_members);
    g_group_engaged = false;
    const int rc = group_step_impl(members, n_members, params, gravity, simulated_time, num_substeps);
    if (rc != SPARKL_OK && with_comm && g_group_engaged) group_abort(members, n_members);
    return rc;
}
static int group_step_impl(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params, const float gravity[3], float* simulated_time,
                           uint32_t* num_substeps) {
    if (!members || n_members == 0 || !params || !gravity) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    if (num_substeps) *num_substeps = 0;
    if (simulated_time) *simulated_time = params->dt;
    if (params->dt == 0.0f) return SPARKL_OK;
    bool rccl = true;
    for (uint32_t i = 0; i < n_members; ++i) {
        if (!members[i]) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null member");
        rccl = rccl && members[i]->comm != nullptr;
    }
    const float min_dt = params->dt / (float)params->max_num_substeps;
    float remaining = params->dt;
    uint32_t nsub = 0;
    // the dt bound over ALL members and ranks (timestep_estimator.rs:10-51), then the substep length (mls_solver.rs:33-59)
    auto pick = [&](float* out) -> int {
        uint32_t bits = 0x7f7fffffu;
        for (uint32_t i = 0; i < n_members; ++i) {
            SparklPipeline* p = members[i];
            HIPCHK(hipSetDevice(p->device));
            CHK(x_staging(p));
            CHK(sparkl_hip_dt_bound_export(p, p->x_dev.p + 16));
        }
        if (rccl) {
            g_group_engaged = true;
            NCCLCHK(g_rccl.GroupStart());
            for (uint32_t i = 0; i < n_members; ++i) NCCLCHK(g_rccl.AllReduce(members[i]->x_dev.p + 16, members[i]->x_dev.p + 16, 1, ncclUint32, ncclMin, members[i]->comm, members[i]->stream));
            NCCLCHK(g_rccl.GroupEnd());
        }
        for (uint32_t i = 0; i < n_members; ++i) {
            SparklPipeline* p = members[i];
            HIPCHK(hipSetDevice(p->device));
            HIPCHK(hipMemcpyAsync(p->x_host + 16, p->x_dev.p + 16, 4, hipMemcpyDeviceToHost, p->stream));
            CHK(comm_wait(p, p->stream));
            uint32_t b;
            memcpy(&b, p->x_host + 16, 4);
            bits = std::min(bits, b);
        }
        float bound;
        memcpy(&bound, &bits, 4);
        // A NaN bound never wins an integer MIN of float bits (0x7fc00000 > 0x7f7fffff): the estimator exports it as 0 (and sets
        // Ctl::dt_nan), so the word that travels is 0 on EVERY rank -- all of them return here, in step, like sparkl_hip_pipeline_step
        // does (the reference panics: timestep_estimator.rs:46).  A true bound of 0 (infinite speed) is no more steppable.
        if (!(bound > 0.0f)) return set_error(SPARKL_ERR_NAN_TIMESTEP, "NaN timestep bound (the reference panics: timestep_estimator.rs:46)");
        float dt = std::fmin(std::fmin(remaining, params->max_substep_dt), bound);   // timestep_estimator.rs:25,35,44
        if (dt < min_dt && remaining > min_dt) dt = min_dt;                          // mls_solver.rs:57-59
        if (!(dt > 0.0f)) return set_error(SPARKL_ERR_NAN_TIMESTEP, "substep length %g: the step cannot make progress", (double)dt);
        *out = dt;
        return SPARKL_OK;
    };
    const PickDt pick_fn = pick;
    while (remaining > 0.0f) {
        float dt = 0.0f;
        if (params->force_fluids_volume_recomputation) {
            // mls_solver.rs:46-48 recomputes the fluid volumes BEFORE the dt estimate of every substep: the group loop runs the sweep
            // (mass planes across the cuts) and asks for the length behind it, in front of the scatter of the same substep.  A substep
            // that a re-bin request turns into a no-op is repeated by the loop, estimate included; the last answer is the one that ran.
            dt = std::fmin(remaining, params->max_substep_dt);
            const PickDt behind_sweep = [&](float* out) -> int { CHK(pick(out)); dt = *out; return SPARKL_OK; };
            CHK(group_substeps_impl(members, n_members, params, gravity, dt, 1, &behind_sweep));
        } else {
            CHK(pick_fn(&dt));
            CHK(group_substeps_impl(members, n_members, params, gravity, dt, 1));
        }
        ++nsub;
        remaining -= dt;
        if (params->stop_after_one_substep) { if (simulated_time) *simulated_time = dt; break; }
    }
    if (num_substeps) *num_substeps = nsub;
    return SPARKL_OK;
}
uint64_t sparkl_hip_group_rebins(const SparklPipeline* p) { return p ? p->group_rebins : 0; }
/* schedule of the slab-group loop between RCCL ranks: *chosen = 1 overlapped, 0 in line, -1 not decided yet (or forced by SPARKL_HIP_OVERLAP:
 * then the forced value); us_per_substep[2] = the MAX over the ranks of the calibration batches, in line and overlapped (0 before) */
int sparkl_hip_overlap_info(const SparklPipeline* p, int* chosen, double* us_per_substep_2) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    if (chosen) *chosen = p->overlap_mode >= 0 ? p->overlap_mode : (p->overlap_cal_state == 2 ? p->overlap_choice : -1);
    if (us_per_substep_2) { us_per_substep_2[0] = p->overlap_cal_us[0]; us_per_substep_2[1] = p->overlap_cal_us[1]; }
    return SPARKL_OK;
}
int sparkl_hip_slab_balance(SparklPipeline* p, int enable) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    p->slab_balance = enable != 0;
    return SPARKL_OK;
}
int sparkl_hip_slab_get(const SparklPipeline* p, SparklSlab* out, uint64_t* cut_moves) {
    if (!p || !out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    out->axis = p->slab_axis; out->block_lo = p->slab_lo; out->block_hi = p->slab_hi; out->_pad0 = 0; out->id_offset = p->id_offset;
    if (cut_moves) *cut_moves = p->cut_moves;
    return SPARKL_OK;
}
void* sparkl_hip_pipeline_stream(const SparklPipeline* p) { return p ? (void*)p->stream : nullptr; }

}  // extern "C"

extern "C" {

// --------------------------------------------------------------------------------- grid inspection
uint64_t sparkl_hip_grid_num_active_blocks(SparklPipeline* p) { return p ? (uint64_t)p->num_blocks : 0; }
uint64_t sparkl_hip_grid_num_particle_blocks(SparklPipeline* p) { return p ? (uint64_t)p->num_pblocks : 0; }
int sparkl_hip_grid_debug_capture(SparklPipeline* p, int enable) {
    if (!p) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null pipeline");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->capture_mom = enable != 0;
    if (p->capture_mom && p->num_blocks) {
        CHK(p->node_mom.reserve((size_t)p->num_blocks * 64));
        HIPCHK(hipMemset(p->node_mom.p, 0, (size_t)p->num_blocks * 64 * sizeof(float4)));
    }
    return SPARKL_OK;
}
int sparkl_hip_grid_read_nodes(SparklPipeline* p, SparklGridNode* out, uint64_t cap, uint64_t* n_out) {
    if (!p || !n_out) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(p->device));
    *n_out = 0;
    if (p->num_blocks == 0) return SPARKL_OK;
    const size_t total = (size_t)p->num_blocks * 64;
    CHK(p->node_staging.reserve(std::max<size_t>(std::min<size_t>(cap, total), 1)));
    CHK(p->d_count.reserve(1));
    HIPCHK(hipMemsetAsync(p->d_count.p, 0, sizeof(unsigned long long), p->stream));
    const unsigned long long dcap = out ? std::min<size_t>(cap, total) : 0;
    hipLaunchKernelGGL(k_dump_nodes, dim3(grid_for(total, 256)), dim3(256), 0, p->stream, grid_arrays(p), p->node_staging.p, dcap, p->d_count.p);
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, p->d_count.p, sizeof(cnt), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    *n_out = cnt;
    if (out && cnt) HIPCHK(hipMemcpy(out, p->node_staging.p, std::min<unsigned long long>(cnt, dcap) * sizeof(SparklGridNode), hipMemcpyDeviceToHost));
    return SPARKL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- diagnostics (probes.h)
// (tests) the row reduce-scatter of the 64-per-cell scatters on known inputs: out[192] = the three values every lane of one wave ends with (probes.h)
extern "C" int sparkl_hip_debug_row_reduce(SparklPipeline* p, float* out192) {
    if (!p || !out192) return -1;
    HIPCHK(hipSetDevice(p->device));
    DevBuf<float> d;
    CHK(d.reserve(192));
    hipLaunchKernelGGL(k_test_row_reduce, dim3(1), dim3(64), 0, p->stream, d.p);
    HIPCHK(hipMemcpyAsync(out192, d.p, 192 * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    d.release();
    return SPARKL_OK;
}
extern "C" int sparkl_hip_debug_run_stats(SparklPipeline* p, unsigned long long* out, uint64_t cap) {
    if (!p || !out || cap < (uint64_t)RS_HISTS * RS_BINS) return set_error(SPARKL_ERR_INVALID_ARGUMENT, "run_stats: need %d counters", RS_HISTS * RS_BINS);
    HIPCHK(hipSetDevice(p->device));
    memset(out, 0, sizeof(unsigned long long) * RS_HISTS * RS_BINS);
    if (p->num_pblocks == 0) return SPARKL_OK;
    CHK(p->d_count.reserve((size_t)RS_HISTS * RS_BINS));
    HIPCHK(hipMemsetAsync(p->d_count.p, 0, sizeof(unsigned long long) * RS_HISTS * RS_BINS, p->stream));
    hipLaunchKernelGGL(k_run_stats, dim3(p->num_pblocks), dim3(BLK_THREADS), 0, p->stream, p->cur.arrays(), grid_arrays(p), p->h, p->d_count.p);
    HIPCHK(hipMemcpyAsync(out, p->d_count.p, sizeof(unsigned long long) * RS_HISTS * RS_BINS, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SPARKL_OK;
}

// (tests, bench) out[0] = 1 while the group loop of this pipeline exchanges its halo peer-direct (kernels.h, PeerBox), out[1] = substeps it has pushed so far,
// out[2] = bit 0: a setup attempt failed and the job fell back to the RCCL / copy exchange, bit 1: the arena is fine-grained memory, bits 8..: handshakes done;
// out[3] = members of the job
extern "C" int sparkl_hip_debug_direct_choice(const SparklPipeline* p, int* choice, double us[3]) {
    if (!p || !choice || !us) return -1;
    *choice = p->direct_choice;
    us[0] = p->direct_cal_us; us[1] = p->overlap_cal_us[0]; us[2] = p->overlap_cal_us[1];
    return 0;
}
extern "C" int sparkl_hip_debug_direct(const SparklPipeline* p, unsigned long long out[4]) {
    if (!p || !out) return -1;
    out[0] = p->direct.ready ? 1 : 0; out[1] = p->direct.epoch; out[2] = (p->direct.failed ? 1 : 0) | (p->direct.arena_fine ? 2 : 0) | ((unsigned long long)p->direct.handshakes << 8);
    out[3] = (unsigned long long)p->direct.n;
    return 0;
}

// (tests) 1 while the fused kernel of this pipeline would run on the flags mirror (ParticleArrays::fl), else 0
extern "C" int sparkl_hip_debug_flags_mirror(const SparklPipeline* p) {
    if (!p) return -1;
    return (p->ku_ok && (p->all_dp_sand || p->all_plain_fluid) && !p->cur.has_frac && !p->rare_models) ? 1 : 0;
}
