// Diagnostic kernels (not on the product path; nothing in sparkl_hip_pipeline_* launches them).
//
// sparkl_hip_debug_run_stats: what the run-task scatter of k_g2p2g sees in the CURRENT particle order -- for every wave batch
// (64 consecutive slots of a particle block, the partition k_g2p2g uses) the number of same-cell lane runs, the longest run, the
// number of distinct cells, the largest cell, the number of <= 4-particle chunks, and the rounds / visit iterations the scatter takes.
#pragma once
#include "kernels.h"

namespace spk {

constexpr int RS_BINS = 65;
enum { RS_NRUNS = 0, RS_MAXLEN = 1, RS_DISTINCT = 2, RS_CHUNKS4 = 3, RS_ROUNDS = 4, RS_ITERS = 5, RS_PCOUNT64 = 6, RS_LIVE = 7, RS_ROWS = 8, RS_LAYERS = 9, RS_STRAIN = 10, RS_MAXCELL = 11, RS_ITERS_MULTI = 12, RS_HISTS = 13 };

__global__ __launch_bounds__(BLK_THREADS) void k_run_stats(ParticleArrays P, GridArrays G, float h, unsigned long long* out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int pb = blockIdx.x;
    const int4 rng = G.pb_info[2 * pb + 1];
    const int pfirst = rng.x, pcount = rng.y;
    const int niter = (pcount + BLK_THREADS - 1) / BLK_THREADS;
    if (tid == 0) atomicAdd(&out[RS_PCOUNT64 * RS_BINS + min(pcount >> 6, 64)], 1ull);
    for (int it = 0; it < niter; ++it) {
        const int li = it * BLK_THREADS + tid;
        const bool active = li < pcount;
        const int i = pfirst + (active ? li : 0);
        const float4 q0 = P.q[0][i];
        const uint32_t flags = __float_as_uint(P.k[i].w);
        bool live = active && !(flags & FLAG_INVALID);
        const int lx = (int)rintf(q0.x), ly = (int)rintf(q0.y), lz = (int)rintf(q0.z);   // block-relative positions (kernels.h, cell_frac)
        live = live && lx >= 0 && lx <= 5 && ly >= 0 && ly <= 5 && lz >= 0 && lz <= 5;
        const int key = live ? lx + 8 * ly + 64 * lz : -1;
        if (!__any(live)) continue;
        const int prev = wave_shr1_i(key, -2);
        const bool head = (lane == 0) | (prev != key);
        const unsigned long long hb = __ballot(head);
        const unsigned long long above = (lane == 63) ? 0ull : (hb >> (lane + 1));
        const int len = above ? (int)__ffsll((long long)above) : (64 - lane);
        const bool lhead = head && live;
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
        // what p2g_scatter_runs does: lanes grouped by cell, a cell longer than L split into two halves, L the smallest of 4, 5, 6, 8, 12
        // whose chunks fit the rounds of 64 tasks the cells need anyway; a round iterates as long as its longest task.  `iters_multi`:
        // the same with cells cut into ceil(len / L) equal chunks instead of two halves.
        const unsigned long long lt = (1ull << lane) - 1ull;
        unsigned long long mine = 0ull;
        {
            unsigned long long todo = __ballot(live);
            while (todo) {
                const int K = __builtin_amdgcn_readlane(key, (int)__ffsll((long long)todo) - 1);
                const unsigned long long m = __ballot(key == K);
                if (key == K) mine = m;
                todo &= ~m;
            }
        }
        const bool chead = live && (mine & lt) == 0ull;
        const int clen = (int)__popcll(mine);
        const int maxcell = wave_max_i32(chead ? clen : 0);
        int c4 = chead ? (clen + 3) / 4 : 0;
        for (int o = 32; o > 0; o >>= 1) c4 += __shfl_xor(c4, o);
        const unsigned long long hl = __ballot(chead);
        const int ncell = (int)__popcll(hl);
        const int slots = ((3 * ncell + 63) >> 6) << 6, room = slots / 3 - ncell;
        int rounds = 0, iters = 0, iters_multi = 0;
        {
            const int Ls[5] = {4, 5, 6, 8, 12};
            unsigned long long sb = 0ull;
            for (int q = 0; q < 5; ++q) {
                const unsigned long long bq = __ballot(chead && clen > Ls[q]);
                if ((int)__popcll(bq) <= room) { sb = bq; break; }
            }
            const int C = ncell + (int)__popcll(sb);
            __shared__ int s_cnt[BLK_THREADS / 64][64];
            int* cl = s_cnt[tid >> 6];
            if (chead) {
                const int c = (int)__popcll(hl & lt) + (int)__popcll(sb & lt);
                const bool split = (sb >> lane) & 1ull;
                const int n0 = split ? (clen + 1) >> 1 : clen;
                cl[c] = n0;
                if (split) cl[c + 1] = clen - n0;
            }
            __builtin_amdgcn_wave_barrier();
            for (int t0 = 0; t0 < 3 * C; t0 += 64) {
                const int t = t0 + lane;
                const int cnt = t < 3 * C ? cl[t % C] : 0;
                rounds++;
                iters += max(wave_max_i32(cnt), 1);
            }
            __builtin_amdgcn_wave_barrier();
            // equal chunks
            int Lm = 64;
            for (int q = 0; q < 5; ++q) {
                int extra = chead ? (clen + Ls[q] - 1) / Ls[q] - 1 : 0;
                for (int o = 32; o > 0; o >>= 1) extra += __shfl_xor(extra, o);
                if (extra <= room) { Lm = Ls[q]; break; }
            }
            const int pieces = (clen + Lm - 1) / Lm;
            iters_multi = rounds * wave_max_i32(chead ? (clen + pieces - 1) / pieces : 0);   // (upper bound when rounds > 1)
        }
        const int nlive = (int)__popcll(__ballot(live));
        // wave maximum of |F^T F - I|_F (what selects the series path of the return map): bin k = first threshold of
        // {1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 3e-1, inf} above it
        int sbin = 0, lbin = 0, mbin = 0;
        {
            const float4 q1 = P.q[1][i], q2 = P.q[2][i];
            const M3 E = {{q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}};
            float S6[6];
            strain_metric(E, S6);
            float f = live ? sqrtf(sym_frob2(S6)) : 0.f;
            for (int o = 32; o > 0; o >>= 1) f = fmaxf(f, __shfl_xor(f, o));
            sbin = f < 1e-3f ? 0 : f < 3e-3f ? 1 : f < 1e-2f ? 2 : f < 3e-2f ? 3 : f < 1e-1f ? 4 : f < 3e-1f ? 5 : 6;
            // same row, bins 16.. : wave maximum of the pending |log_vol_gain|; bins 32.. : of max(|S|, |lvg|)
            float g = live ? fabsf(P.d[i].y) : 0.f;
            for (int o = 32; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o));
            lbin = g < 1e-3f ? 0 : g < 3e-3f ? 1 : g < 1e-2f ? 2 : g < 3e-2f ? 3 : g < 1e-1f ? 4 : g < 3e-1f ? 5 : 6;
            const float m = fmaxf(f, g);
            mbin = m < 1e-3f ? 0 : m < 3e-3f ? 1 : m < 1e-2f ? 2 : m < 3e-2f ? 3 : m < 1e-1f ? 4 : m < 3e-1f ? 5 : 6;
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
            atomicAdd(&out[RS_STRAIN * RS_BINS + 16 + lbin], 1ull);
            atomicAdd(&out[RS_STRAIN * RS_BINS + 32 + mbin], 1ull);
            atomicAdd(&out[RS_MAXCELL * RS_BINS + min(maxcell, 64)], 1ull);
            atomicAdd(&out[RS_ITERS_MULTI * RS_BINS + min(iters_multi, 64)], 1ull);
        }
    }
}


// sparkl_hip_debug_row_reduce: row_reduce_scatter4x3 on known inputs (tests/test_gpu_parity.py): lane l, node k, value c carry (l + 1) (k + 1) + 100 c -- exact in
// f32 -- and every lane writes the three values it ends with: quad q of a row must hold the row's totals of node q.
__global__ void k_test_row_reduce(float* out) {
    const int lane = threadIdx.x & 63;
    float X[12];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) X[3 * k + c] = (float)((lane + 1) * (k + 1) + 100 * c);
    row_reduce_scatter4x3(X);
    out[3 * threadIdx.x] = X[0]; out[3 * threadIdx.x + 1] = X[1]; out[3 * threadIdx.x + 2] = X[2];
}
}  // namespace spk
