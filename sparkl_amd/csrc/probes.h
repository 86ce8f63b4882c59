// Diagnostic kernels (not on the product path; nothing in sparkl_hip_pipeline_* launches them).
//
// sparkl_hip_debug_run_stats: what the run-task scatter of k_g2p2g sees in the CURRENT particle order -- for every wave batch
// (64 consecutive slots of a particle block, the partition k_g2p2g uses) the number of same-cell runs, the longest run, the
// number of distinct cells, the number of <= 4-particle chunks, and the rounds / visit iterations the scatter would take.
#pragma once
#include "kernels.h"

namespace spk {

constexpr int RS_BINS = 65;
enum { RS_NRUNS = 0, RS_MAXLEN = 1, RS_DISTINCT = 2, RS_CHUNKS4 = 3, RS_ROUNDS = 4, RS_ITERS = 5, RS_PCOUNT64 = 6, RS_LIVE = 7, RS_ROWS = 8, RS_LAYERS = 9, RS_STRAIN = 10, RS_HISTS = 11 };

__global__ __launch_bounds__(BLK_THREADS) void k_run_stats(ParticleArrays P, GridArrays G, float h, unsigned long long* out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int pb = blockIdx.x;
    const int4 org = G.pb_info[2 * pb], rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    const float inv_h = 1.0f / h;
    if (tid == 0) atomicAdd(&out[RS_PCOUNT64 * RS_BINS + min(pcount >> 6, 64)], 1ull);
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const int i = pfirst + (active ? li : 0);
        const float4 q0 = P.q[0][i];
        const uint32_t flags = __float_as_uint(P.k[i].w);
        bool live = active && !(flags & FLAG_INVALID);
        const int lx = (int)round_cell(q0.x, h, inv_h) - 1 - org.x + 1, ly = (int)round_cell(q0.y, h, inv_h) - 1 - org.y + 1, lz = (int)round_cell(q0.z, h, inv_h) - 1 - org.z + 1;
        live = live && lx >= 0 && lx <= 5 && ly >= 0 && ly <= 5 && lz >= 0 && lz <= 5;
        const int key = live ? lx + 8 * ly + 64 * lz : -1;
        if (!__any(live)) continue;
        const int prev = wave_shr1_i(key, -2);
        const bool head = (lane == 0) | (prev != key);
        const unsigned long long hb = __ballot(head);
        const unsigned long long above = (lane == 63) ? 0ull : (hb >> (lane + 1));
        const int len = above ? (int)__ffsll((long long)above) : (64 - lane);
        const bool lhead = head && live;
        const int nruns_all = (int)__popcll(hb);
        const int nruns = (int)__popcll(__ballot(lhead));
        const int maxlen = wave_max_i32(lhead ? len : 0);
        bool dup = false, rowdup = false, laydup = false;
        for (int j = 0; j < 63; ++j) {
            const int kj = __builtin_amdgcn_readlane(key, j);
            const int hj = __builtin_amdgcn_readlane((int)lhead, j);
            if (lane > j && hj) { dup |= key == kj; rowdup |= (key >> 3) == (kj >> 3); laydup |= (key >> 6) == (kj >> 6); }
        }
        const int distinct = (int)__popcll(__ballot(lhead && !dup));
        const int rows = (int)__popcll(__ballot(lhead && !rowdup));
        const int layers = (int)__popcll(__ballot(lhead && !laydup));
        int c4 = lhead ? (len + 3) / 4 : 0;
        for (int o = 32; o > 0; o >>= 1) c4 += __shfl_xor(c4, o);
        // what p2g_scatter_runs does today: S halves when 6 * nruns <= 64, rounds of 64 tasks, each as long as its longest task
        const int S = (6 * nruns_all <= 64) ? 2 : 1;
        const int ntasks = nruns_all * 3 * S;
        int rounds = 0, iters = 0;
        for (int t0 = 0; t0 < ntasks; t0 += 64) {
            const int t = t0 + lane;
            const int r = t / (3 * S), half = (S == 2) ? ((t - r * 3 * S) & 1) : 0;
            int cnt = 0;
            {
                // run r: its head lane is the r-th set bit of hb
                const int rr = min(r, nruns_all - 1);
                unsigned long long m = hb;
                for (int q = 0; q < rr; ++q) m &= m - 1;
                const int start = (int)__ffsll((long long)m) - 1;
                const int rl = __shfl(len, start), rlive = __shfl((int)live, start);
                if (t < ntasks) cnt = rlive ? (S == 2 ? (half ? rl - ((rl + 1) >> 1) : ((rl + 1) >> 1)) : rl) : 0;
            }
            rounds++;
            iters += max(wave_max_i32(cnt), 1);
        }
        const int nlive = (int)__popcll(__ballot(live));
        // wave maximum of |F^T F - I|_F (what selects the series path of the return map): bin k = first threshold of
        // {1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 3e-1, inf} above it
        int sbin = 0;
        {
            const float4 q1 = P.q[1][i], q2 = P.q[2][i];
            const M3 E = {{q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}};
            float S6[6];
            strain_metric(E, S6);
            float f = live ? sqrtf(sym_frob2(S6)) : 0.f;
            for (int o = 32; o > 0; o >>= 1) f = fmaxf(f, __shfl_xor(f, o));
            sbin = f < 1e-3f ? 0 : f < 3e-3f ? 1 : f < 1e-2f ? 2 : f < 3e-2f ? 3 : f < 1e-1f ? 4 : f < 3e-1f ? 5 : 6;
        }
        if (lane == 0) {
            atomicAdd(&out[RS_NRUNS * RS_BINS + nruns], 1ull);
            atomicAdd(&out[RS_MAXLEN * RS_BINS + maxlen], 1ull);
            atomicAdd(&out[RS_DISTINCT * RS_BINS + distinct], 1ull);
            atomicAdd(&out[RS_CHUNKS4 * RS_BINS + min(c4, 64)], 1ull);
            atomicAdd(&out[RS_ROUNDS * RS_BINS + min(rounds, 64)], 1ull);
            atomicAdd(&out[RS_ITERS * RS_BINS + min(iters, 64)], 1ull);
            atomicAdd(&out[RS_LIVE * RS_BINS + nlive], 1ull);
            atomicAdd(&out[RS_ROWS * RS_BINS + rows], 1ull);
            atomicAdd(&out[RS_LAYERS * RS_BINS + layers], 1ull);
            atomicAdd(&out[RS_STRAIN * RS_BINS + sbin], 1ull);
        }
    }
}

}  // namespace spk
