// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under sparkl_amd/ may include, link or call this.
//
// CPU restatement of sparkl's *intended* CPU dynamics path (dim3), templated on the scalar
// type: R = float follows the reference's f32 arithmetic operation by operation; R = double
// is the same algorithm with f64 state/accumulation, used to bound rounding noise.
//
// Parity status: the reference holds NO test that pins any dynamics result; the only
// reference-pinned part is the SPGrid addressing (pack/unpack/packed_add/neighbour tables), which
// tests/test_oracle_spgrid.py checks against the constants of
// /root/reference/src/geometry/sp_grid.rs:13-16,162-165,411-555.  Everything else is pinned by
// analytic known-answer tests (tests/test_oracle_kats.py).  => dynamics: "parity unpinned".
//
// Two deliberate deviations from the reference as shipped (both documented in SURVEY F4/F5):
//   * SpGrid::new maps `2 ^ 30` (= 28) bytes (sp_grid.rs:175, XOR typo); we map the intended 2^30
//     (configurable 2^30..2^39 to fit larger synthetic clouds -- it is only an addressing scheme).
//   * a particle flagged out-of-domain (grid_index = u64::MAX) makes the reference's G2P index out of
//     bounds (panic); here such particles are `failed` and skipped by P2G/G2P.
// Third-party arithmetic that is NOT in /root/reference:
//   * nalgebra 0.32 `Matrix3::svd_unordered` (crates-ptx/Cargo.lock:347-349): replaced by a one-sided
//     Jacobi SVD computed in double and rounded to R.  All call sites use SVD-invariant quantities.
//   * parry3d 0.13 `Cuboid::project_point` (grid_update.rs:78-82): restated from its published
//     algorithm (clamp when outside; closest face when inside, solid = false).
#pragma once
#include <sys/mman.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <array>
#include <limits>
#include <map>
#include <memory>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../include/sparkl_hip.h"

namespace oracle {

// ---------------------------------------------------------------- SPGrid addressing (sp_grid.rs)
// src_core/utils/morton3.rs:3-15 (the 0x010f... constant is the reference's, kept on purpose)
inline uint64_t spread(uint64_t w) {
    w &= 0x00000000001fffffULL;
    w = (w | w << 32) & 0x001f00000000ffffULL;
    w = (w | w << 16) & 0x001f0000ff0000ffULL;
    w = (w | w << 8) & 0x010f00f00f00f00fULL;
    w = (w | w << 4) & 0x10c30c30c30c30c3ULL;
    w = (w | w << 2) & 0x1249249249249249ULL;
    return w;
}
inline uint64_t morton_encode3(uint32_t x, uint32_t y, uint32_t z) {
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}
// morton3.rs:19-31
inline uint32_t compact(uint64_t w) {
    w &= 0x1249249249249249ULL;
    w = (w ^ (w >> 2)) & 0x30c30c30c30c30c3ULL;
    w = (w ^ (w >> 4)) & 0xf00f00f00f00f00fULL;
    w = (w ^ (w >> 8)) & 0x00ff0000ff0000ffULL;
    w = (w ^ (w >> 16)) & 0x00ff00000000ffffULL;
    w = (w ^ (w >> 32)) & 0x00000000001fffffULL;
    return (uint32_t)w;
}
constexpr int PACK_HEADER = 12;  // sp_grid.rs:157
constexpr int PACK_ALIGN = 6;    // sp_grid.rs:158
constexpr uint64_t REGION_ID_MASK = ~0xFFFULL;  // sp_grid.rs:159
// sp_grid.rs:149-155
constexpr uint64_t BLOCK_MX = 0b0001001001001001001001001001001001001001001001001001000011000000ULL;
constexpr uint64_t BLOCK_MY = 0b0010010010010010010010010010010010010010010010010010001100000000ULL;
constexpr uint64_t BLOCK_MZ = 0b0100100100100100100100100100100100100100100100100100110000000000ULL;

// sp_grid.rs:362-369
inline uint64_t pack(uint32_t x, uint32_t y, uint32_t z) {
    uint64_t res = morton_encode3(x >> 2, y >> 2, z >> 2);
    res <<= PACK_HEADER;
    res |= ((uint64_t)x & 3) << PACK_ALIGN;
    res |= ((uint64_t)y & 3) << (PACK_ALIGN + 2);
    res |= ((uint64_t)z & 3) << (PACK_ALIGN + 4);
    return res;
}
// sp_grid.rs:384-393
inline void unpack(uint64_t xyz, uint32_t out[3]) {
    uint64_t code = xyz >> PACK_HEADER;
    uint32_t x = compact(code), y = compact(code >> 1), z = compact(code >> 2);
    out[0] = (x << 2) | (((uint32_t)xyz >> PACK_ALIGN) & 3);
    out[1] = (y << 2) | (((uint32_t)xyz >> (PACK_ALIGN + 2)) & 3);
    out[2] = (z << 2) | (((uint32_t)xyz >> (PACK_ALIGN + 4)) & 3);
}
// sp_grid.rs:339-351
inline uint64_t masked_add(uint64_t mask, uint64_t i, uint64_t j) { return ((i & mask) + (j | ~mask)) & mask; }
inline uint64_t packed_add(uint64_t i, uint64_t j) {
    return masked_add(BLOCK_MX, i, j) | masked_add(BLOCK_MY, i, j) | masked_add(BLOCK_MZ, i, j);
}
// src/geometry/mod.rs:22-50 (iteration order of the 27 neighbours)
static const int NBH_SHIFTS[27][3] = {
    {2, 2, 2}, {2, 0, 2}, {2, 1, 2}, {0, 2, 2}, {0, 0, 2}, {0, 1, 2}, {1, 2, 2}, {1, 0, 2}, {1, 1, 2},
    {2, 2, 0}, {2, 0, 0}, {2, 1, 0}, {0, 2, 0}, {0, 0, 0}, {0, 1, 0}, {1, 2, 0}, {1, 0, 0}, {1, 1, 0},
    {2, 2, 1}, {2, 0, 1}, {2, 1, 1}, {0, 2, 1}, {0, 0, 1}, {0, 1, 1}, {1, 2, 1}, {1, 0, 1}, {1, 1, 1}};
// sp_grid.rs:21-31
static const uint32_t NBH_REGION_SHIFTS[7][3] = {{0, 0, 4}, {0, 4, 0}, {0, 4, 4}, {4, 0, 0},
                                                 {4, 0, 4}, {4, 4, 0}, {4, 4, 4}};
inline uint64_t packed_nbh_shift(int k) {
    return pack((uint32_t)NBH_SHIFTS[k][0], (uint32_t)NBH_SHIFTS[k][1], (uint32_t)NBH_SHIFTS[k][2]);
}
inline uint64_t packed_plus_five() { return pack(5, 5, 5); }  // sp_grid.rs:165,433-435

// ------------------------------------------------------------------------------ small 3x3 algebra
// column-major like nalgebra: (i, j) at i + 3 j
template <class R>
struct Mat3 {
    R m[9];
    R& operator()(int i, int j) { return m[i + 3 * j]; }
    R operator()(int i, int j) const { return m[i + 3 * j]; }
    static Mat3 zeros() { Mat3 a; for (auto& e : a.m) e = R(0); return a; }
    static Mat3 identity() { Mat3 a = zeros(); a.m[0] = a.m[4] = a.m[8] = R(1); return a; }
};
template <class R> struct Vec3 { R v[3]; R& operator[](int i) { return v[i]; } R operator[](int i) const { return v[i]; } };

template <class R> Mat3<R> operator*(R s, const Mat3<R>& a) { Mat3<R> r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] * s; return r; }
template <class R> Mat3<R> operator*(const Mat3<R>& a, R s) { return s * a; }
template <class R> Mat3<R> operator+(const Mat3<R>& a, const Mat3<R>& b) { Mat3<R> r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
template <class R> Mat3<R> operator-(const Mat3<R>& a, const Mat3<R>& b) { Mat3<R> r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
template <class R> Mat3<R> operator*(const Mat3<R>& a, const Mat3<R>& b) {
    Mat3<R> r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            R s = a(i, 0) * b(0, j);
            s += a(i, 1) * b(1, j);
            s += a(i, 2) * b(2, j);
            r(i, j) = s;
        }
    return r;
}
template <class R> Vec3<R> operator*(const Mat3<R>& a, const Vec3<R>& x) {
    Vec3<R> r;
    for (int i = 0; i < 3; ++i) {
        R s = a(i, 0) * x[0];
        s += a(i, 1) * x[1];
        s += a(i, 2) * x[2];
        r[i] = s;
    }
    return r;
}
template <class R> Mat3<R> transpose(const Mat3<R>& a) { Mat3<R> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i); return r; }
// nalgebra Matrix3::determinant (cofactor expansion along the first row)
template <class R> R determinant(const Mat3<R>& a) {
    R m11 = a(0, 0), m12 = a(0, 1), m13 = a(0, 2), m21 = a(1, 0), m22 = a(1, 1), m23 = a(1, 2), m31 = a(2, 0), m32 = a(2, 1), m33 = a(2, 2);
    R minor_m12_m23 = m22 * m33 - m32 * m23;
    R minor_m11_m23 = m21 * m33 - m31 * m23;
    R minor_m11_m22 = m21 * m32 - m31 * m22;
    return m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
}
template <class R> R trace(const Mat3<R>& a) { return a.m[0] + a.m[4] + a.m[8]; }
template <class R> R frob_norm(const Mat3<R>& a) { R s = 0; for (int i = 0; i < 9; ++i) s += a.m[i] * a.m[i]; return std::sqrt(s); }
template <class R> R norm(const Vec3<R>& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
template <class R> R dot(const Vec3<R>& a, const Vec3<R>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// One-sided (Hestenes) Jacobi SVD in double: A = U diag(S) V^T, S >= 0, unordered.
// Stands in for nalgebra's svd_unordered (not available here).
inline void svd3_f64(const double A[9], double U[9], double S[3], double V[9]) {
    double W[9];
    std::memcpy(W, A, sizeof(W));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    static const int PAIRS[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (auto& pq : PAIRS) {
            int p = pq[0], q = pq[1];
            double* wp = W + 3 * p; double* wq = W + 3 * q;
            double alpha = wp[0] * wp[0] + wp[1] * wp[1] + wp[2] * wp[2];
            double beta = wq[0] * wq[0] + wq[1] * wq[1] + wq[2] * wq[2];
            double gamma = wp[0] * wq[0] + wp[1] * wq[1] + wp[2] * wq[2];
            if (std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta) || gamma == 0.0) continue;
            rotated = true;
            double zeta = (beta - alpha) / (2.0 * gamma);
            double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
            double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
            for (int i = 0; i < 3; ++i) {
                double a = wp[i], b = wq[i];
                wp[i] = c * a - s * b; wq[i] = s * a + c * b;
                double va = V[3 * p + i], vb = V[3 * q + i];
                V[3 * p + i] = c * va - s * vb; V[3 * q + i] = s * va + c * vb;
            }
        }
        if (!rotated) break;
    }
    for (int j = 0; j < 3; ++j) {
        double* w = W + 3 * j;
        S[j] = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    }
    // U columns; complete the basis for (near-)zero singular values
    double smax = std::max(S[0], std::max(S[1], S[2]));
    bool ok[3];
    for (int j = 0; j < 3; ++j) {
        ok[j] = S[j] > 1e-300 && S[j] > 1e-14 * smax;
        if (ok[j]) for (int i = 0; i < 3; ++i) U[3 * j + i] = W[3 * j + i] / S[j];
    }
    for (int j = 0; j < 3; ++j) {
        if (ok[j]) continue;
        int a = (j + 1) % 3, b = (j + 2) % 3;
        double ua[3], ub[3];
        auto pick = [&](int col, double out[3], const double* avoid) {
            if (ok[col]) { for (int i = 0; i < 3; ++i) out[i] = U[3 * col + i]; return; }
            // any unit vector orthogonal to `avoid` (or e0 if none)
            double e[3] = {1, 0, 0};
            if (avoid && std::fabs(avoid[0]) > 0.9) { e[0] = 0; e[1] = 1; }
            if (avoid) { double d = e[0] * avoid[0] + e[1] * avoid[1] + e[2] * avoid[2]; for (int i = 0; i < 3; ++i) e[i] -= d * avoid[i]; }
            double n = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            for (int i = 0; i < 3; ++i) out[i] = e[i] / n;
        };
        pick(a, ua, nullptr);
        if (!ok[a]) { for (int i = 0; i < 3; ++i) U[3 * a + i] = ua[i]; ok[a] = true; }
        pick(b, ub, ua);
        if (!ok[b]) { for (int i = 0; i < 3; ++i) U[3 * b + i] = ub[i]; ok[b] = true; }
        U[3 * j + 0] = ua[1] * ub[2] - ua[2] * ub[1];
        U[3 * j + 1] = ua[2] * ub[0] - ua[0] * ub[2];
        U[3 * j + 2] = ua[0] * ub[1] - ua[1] * ub[0];
        ok[j] = true;
    }
}

template <class R> struct Svd { Mat3<R> u; Vec3<R> s; Mat3<R> vt; };
template <class R> Svd<R> svd_unordered(const Mat3<R>& a) {
    double A[9], U[9], S[3], V[9];
    for (int i = 0; i < 9; ++i) A[i] = (double)a.m[i];
    svd3_f64(A, U, S, V);
    Svd<R> r;
    for (int i = 0; i < 9; ++i) r.u.m[i] = (R)U[i];
    for (int i = 0; i < 3; ++i) r.s[i] = (R)S[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.vt(i, j) = (R)V[3 * i + j];  // V^T(i,j) = V(j,i)
    return r;
}
// nalgebra SVD::recompose: U * diag(S) * V^T  (U columns scaled, then multiplied)
template <class R> Mat3<R> recompose(const Svd<R>& d) {
    Mat3<R> us = d.u;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) us(i, j) = us(i, j) * d.s[j];
    return us * d.vt;
}

// ------------------------------------------------------------------------------------- utilities
template <class R> R inv_exact(R e) { return e == R(0) ? R(0) : R(1) / e; }  // src_core/utils/physics.rs:7-14
// kernel.rs:10-21
template <class R> R inv_d(R h) { return R(4) / (h * h); }
template <class R> void eval_all(R x, R out[3]) {
    R a = R(1.5) - x, b = x - R(1), c = x - R(0.5);
    out[0] = R(0.5) * (a * a);
    out[1] = R(0.75) - (b * b);
    out[2] = R(0.5) * (c * c);
}
// Rust f32::round = half away from zero; `as u32` saturates (NaN -> 0)
template <class R> uint32_t round_as_u32(R e) {
    R r = std::round(e);
    if (!(r > R(0))) return 0;
    if (r >= R(4294967296.0)) return 0xFFFFFFFFu;
    return (uint32_t)r;
}

// ------------------------------------------------------------------------------- parallel helpers
// Allocator whose default construction leaves trivially constructible elements untouched: `resize(n)` then does not write the pages, so the
// parallel fill that follows places them on the NUMA nodes of the threads that will work on them (first touch) instead of all on the node of
// the loading thread.
template <class T>
struct no_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = no_init_allocator<U>; };
    no_init_allocator() = default;
    template <class U> no_init_allocator(const no_init_allocator<U>&) {}
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

// Stable parallel merge sort of (key, value) pairs by key -- the order rayon's `par_sort_by_key` (a stable parallel merge sort,
// particle_set.rs:142) produces, which IS the sequential stable order: ties keep their previous relative order.  T chunks sorted with
// std::stable_sort in parallel, then log2(T) rounds of pairwise merges, every merge cut into T pieces by co-ranking (merge path), so all
// threads work in every round.  10 M keys: ~0.1 s on 32 threads where the serial std::stable_sort of round 4 took 1.6 s.
struct KeyIdx { uint64_t key; size_t idx; };
inline size_t merge_co_rank(size_t d, const KeyIdx* A, size_t m, const KeyIdx* B, size_t n) {
    // smallest i (elements taken from A among the first d outputs, j = d - i from B) with  j == 0 || i == m || B[j-1].key < A[i].key :
    // equal keys come from A first (stability)
    size_t lo = d > n ? d - n : 0, hi = std::min(d, m);
    while (lo < hi) {
        const size_t i = (lo + hi) / 2, j = d - i;
        if (j > 0 && i < m && !(B[j - 1].key < A[i].key)) lo = i + 1; else hi = i;
    }
    return lo;
}
inline void parallel_stable_sort_pairs(std::vector<KeyIdx, no_init_allocator<KeyIdx>>& a, std::vector<KeyIdx, no_init_allocator<KeyIdx>>& tmp, int threads) {
    const size_t n = a.size();
    auto less = [](const KeyIdx& x, const KeyIdx& y) { return x.key < y.key; };
    int T = 1;
    while (T * 2 <= threads && (size_t)T * 2 * 4096 <= n) T *= 2;
    if (T == 1) { std::stable_sort(a.begin(), a.end(), less); return; }
    tmp.resize(n);
    auto bound = [&](int c, int parts) { return (size_t)((unsigned __int128)n * (unsigned)c / (unsigned)parts); };
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int c = 0; c < T; ++c) std::stable_sort(a.begin() + bound(c, T), a.begin() + bound(c + 1, T), less);
    KeyIdx* src = a.data();
    KeyIdx* dst = tmp.data();
    for (int width = 1; width < T; width *= 2) {
        const int pairs = T / (2 * width), pieces = 2 * width;   // every pair's merge is cut into `pieces` parts: T tasks per round
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int task = 0; task < pairs * pieces; ++task) {
            const int pr = task / pieces, pc = task % pieces;
            const size_t lo = bound(2 * width * pr, T), mid = bound(2 * width * pr + width, T), hi = bound(2 * width * (pr + 1), T);
            const KeyIdx* A = src + lo; const size_t m = mid - lo;
            const KeyIdx* B = src + mid; const size_t nb = hi - mid;
            const size_t d0 = (size_t)((unsigned __int128)(m + nb) * (unsigned)pc / (unsigned)pieces), d1 = (size_t)((unsigned __int128)(m + nb) * (unsigned)(pc + 1) / (unsigned)pieces);
            const size_t i0 = merge_co_rank(d0, A, m, B, nb), i1 = merge_co_rank(d1, A, m, B, nb);
            std::merge(A + i0, A + i1, B + (d0 - i0), B + (d1 - i1), dst + lo + d0, less);   // std::merge takes from the first range on ties
        }
        std::swap(src, dst);
    }
    if (src != a.data()) {
#pragma omp parallel for schedule(static) num_threads(T)
        for (size_t i = 0; i < n; ++i) a[i] = src[i];
    }
}

// -------------------------------------------------------------------------------------- particle
// src/dynamics/particle.rs:7-51 (fields used by the in-scope models)
template <class R>
struct Particle {
    uint64_t model;
    R mass, volume0, radius0;
    Vec3<R> position, velocity;
    bool kinematic_vel_enabled; Vec3<R> kinematic_vel;
    Mat3<R> velocity_gradient, deformation_gradient;
    R plastic_deformation_gradient_det;
    uint64_t grid_index;
    bool failed, is_static;
    R psi_pos, parameter1, parameter2, crack_propagation_factor, crack_threshold;
    R phase, nacc_alpha, plastic_hardening, elastic_hardening, log_vol_gain;

    R density0() const { return mass / volume0; }
    R density_fluid() const { return density0() / deformation_gradient(0, 0); }
    R density_def_grad() const { return density0() / determinant(deformation_gradient); }
    // particle.rs:110-120
    Vec3<R> dir_to_associated_grid_node(R h) const {
        Vec3<R> r;
        for (int a = 0; a < 3; ++a) r[a] = (std::round(position[a] / h) - R(1)) * h - position[a];
        return r;
    }
};

// grid_node.rs:13-21 (48 B payload for f32)
template <class R>
struct GridNode {
    R mass; R momentum[3]; R velocity[3];
    uint32_t particles[2]; uint32_t flags;
    R psi_momentum, psi_mass;
};

// ---------------------------------------------------------------------------------------- models
// elasticity_sound_speed_timestep_bound.rs:22-34
template <class R> R sound_speed_bound(R alpha, R bulk, R shear, R density0, const Vec3<R>& vel, R h) {
    R c_dir = std::sqrt((bulk + R(4.0) / R(3.0) * shear) * R(1.0) / density0);
    R max_denom = std::max(norm(vel), c_dir);
    return alpha * h / max_denom;
}

// elasticity_corotated_linear.rs:31-78
template <class R> Mat3<R> corotated_kirchhoff(const SparklCorotatedLinearElasticity& m, R phase, R hardening, const Mat3<R>& F) {
    R mu = (R)m.mu, lambda = (R)m.lambda;
    R j = determinant(F);
    Svd<R> svd = svd_unordered(F);
    for (int i = 0; i < 3; ++i) svd.s[i] = svd.s[i] - R(1);
    Mat3<R> I = Mat3<R>::identity();
    if (phase == R(1)) {
        return (R(2) * mu * hardening) * recompose(svd) * transpose(F) + (lambda * hardening * (j - R(1)) * j) * I;
    }
    Svd<R> pos = svd, neg = svd;
    for (int i = 0; i < 3; ++i) { pos.s[i] = std::max(pos.s[i], R(0)); neg.s[i] = std::min(neg.s[i], R(0)); }
    Mat3<R> pos_dev = R(2) * mu * hardening * recompose(pos) * transpose(F);
    Mat3<R> neg_dev = R(2) * mu * hardening * recompose(neg) * transpose(F);
    Mat3<R> sph = (lambda * hardening * (j - R(1)) * j) * I;
    Mat3<R> pos_part, neg_part;
    if (j < R(1)) { pos_part = pos_dev; neg_part = neg_dev + sph; } else { pos_part = pos_dev + sph; neg_part = neg_dev; }
    R phase_coeff = (m.split_stress_on_failure && phase == R(0)) ? R(0) : R(1);
    return pos_part * phase_coeff + neg_part;
}
// elasticity_corotated_linear.rs:105-118
template <class R> R corotated_pos_energy(const SparklCorotatedLinearElasticity& m, const Mat3<R>& F, R hardening) {
    R j = determinant(F);
    Svd<R> d = svd_unordered(F);
    R nsq = 0;
    for (int i = 0; i < 3; ++i) { R e = std::max(d.s[i] - R(1), R(0)); nsq += e * e; }
    R pos_dev = (R)m.mu * hardening * nsq;
    R sph = (R)m.lambda * hardening / R(2) * (j - R(1)) * (j - R(1));
    return j < R(1) ? pos_dev : pos_dev + sph;
}
// lib.rs:30-45 + physics.rs:75-77
template <class R> Mat3<R> deviatoric_part(const Mat3<R>& t) {
    R sph = trace(t) / R(3);
    Mat3<R> d = t;
    for (int i = 0; i < 3; ++i) d(i, i) -= sph;
    return d;
}
template <class R> R nh_phase_coeff(R phase) { const R r = R(0.001); return (R(1) - r) * phase * phase + r; }  // elasticity_neo_hookean.rs:70-73
// elasticity_neo_hookean.rs:43-67
template <class R> Mat3<R> neo_hookean_kirchhoff(const SparklNeoHookeanElasticity& m, R phase, R hardening, const Mat3<R>& F) {
    R mu = (R)m.mu, lambda = (R)m.lambda;
    R pc = nh_phase_coeff(phase);
    R j = determinant(F);
    R k = R(2.0) / R(3.0) * mu * hardening + lambda * hardening;
    Mat3<R> cg = F * transpose(F);
    Mat3<R> dev = ((mu * hardening) * std::pow(j, R(-2.0) / R(3))) * deviatoric_part(cg);
    Mat3<R> vol = (k / R(2) * (j * j - R(1))) * Mat3<R>::identity();
    Mat3<R> pos_part, neg_part;
    if (j >= R(1)) { pos_part = dev + vol; neg_part = Mat3<R>::zeros(); } else { pos_part = dev; neg_part = vol; }
    return pos_part * pc + neg_part;
}
// elasticity_neo_hookean.rs:118-141
template <class R> R neo_hookean_pos_energy(const SparklNeoHookeanElasticity& m, R phase, R hardening, const Mat3<R>& F) {
    R pc = nh_phase_coeff(phase);
    R j = determinant(F);
    R k = R(2.0) / R(3.0) * (R)m.mu * hardening + (R)m.lambda * hardening;
    R dev = hardening * (R)m.mu / R(2) * (trace(F * transpose(F)) * std::pow(j, R(-2.0) / R(3)) - R(3));
    R vol = k / R(2) * ((j * j - R(1)) / R(2) - std::log(j));
    return j < R(1) ? dev * pc : (dev + vol) * phase;
}
// eos_monaghan_sph.rs:28-37 (powi).  Rust's f32::powi is the llvm.powi intrinsic, lowered to compiler-rt's __powisf2 / __powidf2 for a
// run-time exponent: multiplication by repeated squaring, low bit first (compiler-rt/lib/builtins/powisf2.c) -- restated, so that the
// rounding sequence is the reference's (gamma = 7: x, x^2, x^4 -> x * x^2 * x^4; a plain loop rounds differently in the last bits, and
// the equation of state multiplies those bits by p0 gamma = 7e6).
template <class R> R powi(R x, int n) {
    const bool recip = n < 0;
    R r = 1;
    while (true) {
        if (n & 1) r *= x;
        n /= 2;
        if (n == 0) break;
        x *= x;
    }
    return recip ? R(1) / r : r;
}
template <class R> R monaghan_pressure(const SparklMonaghanSphEos& m, R mass, R volume0, R density_fluid) {
    R density0 = mass / volume0;
    return std::max((R)m.pressure0 * (powi(density_fluid / density0, (int)m.gamma) - R(1)), -(R)m.max_neg_pressure);
}
// eos_monaghan_sph.rs:43-63
template <class R> Mat3<R> monaghan_kirchhoff(const SparklMonaghanSphEos& m, R mass, R volume0, R density_fluid, R fluid_j, const Mat3<R>& C) {
    Mat3<R> stress = (-monaghan_pressure(m, mass, volume0, density_fluid) * fluid_j) * Mat3<R>::identity();
    if (m.viscosity != 0.0f) {
        Mat3<R> sr = (C + transpose(C)) * R(0.5);  // physics.rs:71-73
        stress = stress + (R(2) * (R)m.viscosity * fluid_j) * deviatoric_part(sr);
    }
    return stress;
}
// eos_monaghan_sph.rs:72-99
template <class R> R monaghan_timestep_bound(const SparklMonaghanSphEos& m, R fluid_j, R mass, R volume0, R density_fluid, const Vec3<R>& vel, R h) {
    R j = fluid_j, density0 = mass / volume0, k = R(6), d = R(3);
    R pressure = -monaghan_pressure(m, mass, volume0, density_fluid);
    R single = (h / j) * std::sqrt(density0 * (j - R(1)) / (k * pressure * d));
    R c_square = std::max(dot(vel, vel), R(1)) / R(0.1);
    R cfl = h / std::sqrt(c_square);
    return std::min(single, cfl);  // f32::min semantics: NaN-ignoring
}
template <class R> R rust_min(R a, R b) { return std::fmin(a, b); }  // f32::min ignores NaN like fmin

// plasticity_drucker_prager.rs:61-66
template <class R> R dp_alpha(const SparklDruckerPragerPlasticity& m, R q) {
    R angle = (R)m.h0 + ((R)m.h1 * q - (R)m.h3) * std::exp(-(R)m.h2 * q);
    R s = std::sin(angle);
    return std::sqrt(R(2.0) / R(3.0)) * (R(2) * s) / (R(3) - s);
}
// plasticity_drucker_prager.rs:36-59.  returns: 0 = None, 1 = Some
template <class R> int dp_project(const SparklDruckerPragerPlasticity& m, const Vec3<R>& sv, R log_vol_gain, R alpha, Vec3<R>& out, R& dq) {
    R d = R(3);
    Vec3<R> strain;
    for (int i = 0; i < 3; ++i) strain[i] = std::log(sv[i]) + log_vol_gain / d;
    R tr = strain[0] + strain[1] + strain[2];
    Vec3<R> dev;
    for (int i = 0; i < 3; ++i) dev[i] = strain[i] - tr / d;
    if ((dev[0] == R(0) && dev[1] == R(0) && dev[2] == R(0)) || tr > R(0)) {
        out[0] = out[1] = out[2] = R(1);
        dq = norm(strain);
        return 1;
    }
    R gamma = norm(dev) + (d * (R)m.lambda + R(2) * (R)m.mu) / (R(2) * (R)m.mu) * tr * alpha;
    if (gamma <= R(0)) return 0;
    R dn = norm(dev);
    for (int i = 0; i < 3; ++i) out[i] = std::exp(strain[i] - gamma * (dev[i] / dn));
    dq = gamma;
    return 1;
}
// plasticity_drucker_prager.rs:68-104
template <class R> void dp_update_particle(const SparklDruckerPragerPlasticity& m, R phase, Mat3<R>& F, R& Jp, R& q, R& lvg) {
    if (m.only_active_when_failed && phase != R(0)) return;
    Svd<R> svd = svd_unordered(F);
    R alpha = dp_alpha(m, q);
    Vec3<R> nsv; R dq;
    if (dp_project(m, svd.s, lvg, alpha, nsv, dq)) {
        R prev_det = svd.s[0] * svd.s[1] * svd.s[2];
        R new_det = nsv[0] * nsv[1] * nsv[2];
        R diff = new_det - prev_det;
        new_det = diff > R(0) ? new_det : prev_det + diff * (R)m.volume_correction;
        Jp *= prev_det / new_det;
        lvg += std::log(prev_det) - std::log(new_det);
        svd.s = nsv;
        q += dq;
        F = recompose(svd);
    }
}

// plasticity_snow.rs:26-48
template <class R> void snow_update_particle(const SparklSnowPlasticity& m, Mat3<R>& F, R& elastic_hardening, R& Jp) {
    Svd<R> svd = svd_unordered(F);
    Vec3<R> nsv;
    for (int i = 0; i < 3; ++i) nsv[i] = std::fmin(std::fmax(svd.s[i], R(1) - (R)m.min_epsilon), R(1) + (R)m.max_epsilon);
    Jp *= (svd.s[0] * svd.s[1] * svd.s[2]) / (nsv[0] * nsv[1] * nsv[2]);
    elastic_hardening = std::exp((R)m.hardening_coeff * (R(1) - Jp));
    svd.s = nsv;
    F = recompose(svd);
}
// plasticity_rankine.rs:34-78
template <class R> void rankine_update_particle(const SparklRankinePlasticity& m, Mat3<R>& F, R& plastic_hardening) {
    const R lambda = m.lambda, mu = m.mu;
    Svd<R> svd = svd_unordered(F);
    Vec3<R> eigv, prev;
    for (int i = 0; i < 3; ++i) eigv[i] = prev[i] = std::log(svd.s[i]);
    int idx[3] = {0, 1, 2};
    std::sort(idx, idx + 3, [&](int a, int b) { return eigv[a] < eigv[b]; });
    const int e3 = idx[0], e2 = idx[1], e1 = idx[2];
    const R sum = eigv[0] + eigv[1] + eigv[2];
    const R soft = (R)m.tensile_strength - (plastic_hardening - R(1));
    if (lambda * sum + R(2) * mu * eigv[e1] <= soft) return;
    if ((R(2) * mu + lambda) * eigv[e2] + lambda * (sum - eigv[e1]) <= soft) {
        eigv[e1] = (soft - lambda * (sum - eigv[e1])) / (R(2) * mu + lambda);
    } else if ((R(2) * mu + R(3) * lambda) * eigv[e3] <= soft) {
        const R ne = (soft - lambda * (sum - eigv[e1] - eigv[e2])) / (R(2) * mu + R(2) * lambda);
        eigv[e1] = ne; eigv[e2] = ne;
    } else {
        const R ne = soft / (R(2) * mu + R(3) * lambda);
        eigv[0] = eigv[1] = eigv[2] = ne;
    }
    Vec3<R> dd{{prev[0] - eigv[0], prev[1] - eigv[1], prev[2] - eigv[2]}};
    plastic_hardening += (R)m.softening_rate * norm(dd);
    plastic_hardening = std::fmin(plastic_hardening, (R)m.tensile_strength);
    for (int i = 0; i < 3; ++i) svd.s[i] = std::exp(eigv[i]);
    F = recompose(svd);
}
// plasticity_nacc.rs:62-152 (dim3: d = 3)
template <class R> void nacc_update_particle(const SparklNaccPlasticity& m, Mat3<R>& F, R& alpha) {
    const R xi = m.hardening_factor, beta = m.cohesion, M = m.friction, d = R(3), mu = m.mu, kappa = m.kappa;
    Svd<R> svd = svd_unordered(F);
    Vec3<R> sq{{svd.s[0] * svd.s[0], svd.s[1] * svd.s[1], svd.s[2] * svd.s[2]}};
    const R sq_tr = sq[0] + sq[1] + sq[2];
    const R p0 = kappa * (R(1.0e-5) + std::sinh(xi * std::fmax(-alpha, R(0))));
    const R j_e_tr = svd.s[0] * svd.s[1] * svd.s[2];
    const R sc = mu * std::pow(j_e_tr, R(-2) / d);
    Vec3<R> s_tr{{sc * (sq[0] - sq_tr / d), sc * (sq[1] - sq_tr / d), sc * (sq[2] - sq_tr / d)}};
    const R psi_kappa = kappa / R(2) * (j_e_tr - R(1) / j_e_tr);
    const R p_tr = -psi_kappa * j_e_tr;
    if (p_tr > p0) {
        const R j_e_n1 = std::sqrt(R(-2) * p0 / kappa + R(1));
        const R v = std::pow(j_e_n1, R(1) / d);
        svd.s = Vec3<R>{{v, v, v}};
        if (m.hardening_enabled) alpha += std::log(j_e_tr / j_e_n1);
        F = recompose(svd);
        return;
    }
    if (p_tr < -beta * p0) {
        const R j_e_n1 = std::sqrt(R(2) * beta * p0 / kappa + R(1));
        const R v = std::pow(j_e_n1, R(1) / d);
        svd.s = Vec3<R>{{v, v, v}};
        if (m.hardening_enabled) alpha += std::log(j_e_tr / j_e_n1);
        F = recompose(svd);
        return;
    }
    const R y0 = (R(1) + R(2) * beta) * ((R(6) - d) / R(2));
    const R y1 = M * M * (p_tr + beta * p0) * (p_tr - p0);
    const R s_n2 = s_tr[0] * s_tr[0] + s_tr[1] * s_tr[1] + s_tr[2] * s_tr[2];
    const R y = y0 * s_n2 + y1;
    if (y < R(1.0e-4)) return;
    if (m.hardening_enabled && p0 > R(1.0e-4) && p_tr < p0 - R(1.0e-4) && p_tr > -beta * p0 + R(1.0e-4)) {
        const R p_c = (R(1) - beta) * p0 / R(2);
        const R q_tr = std::sqrt((R(6) - d) / R(2)) * std::sqrt(s_n2);
        R dx = p_c - p_tr, dy = R(0) - q_tr;
        const R dn = std::sqrt(dx * dx + dy * dy);
        dx /= dn; dy /= dn;
        const R c = M * M * (p_c + beta * p0) * (p_c - p0);
        const R b = M * M * dx * (R(2) * p_c - p0 + beta * p0);
        const R a = M * M * dx * dx + (R(1) + R(2) * beta) * dy * dy;
        const R discr = std::sqrt(b * b - R(4) * a * c);
        const R l1 = (-b + discr) / (R(2) * a), l2 = (-b - discr) / (R(2) * a);
        const R p1 = p_c + l1 * dx, p2 = p_c + l2 * dx;
        const R p_x = (p_tr - p_c) * (p1 - p_c) > R(0) ? p1 : p2;
        const R j_e_x = std::sqrt(std::fabs(R(-2) * p_x / kappa + R(1)));
        if (j_e_x > R(1.0e-4)) alpha += std::log(j_e_tr / j_e_x);
    }
    const R sn = std::sqrt(s_n2);
    const R k = std::sqrt(-y1 / y0) * (std::pow(j_e_tr, R(2) / d) / mu);
    for (int i = 0; i < 3; ++i) svd.s[i] = std::sqrt(k * (s_tr[i] / sn) + sq_tr / d);
    F = recompose(svd);
}
// failure_maximum_stress.rs:20-28; nalgebra try_symmetric_eigen (third party) -> Jacobi eigenvalues in double here
template <class R> bool maximum_stress_failed(const SparklMaximumStressFailure& m, const Mat3<R>& stress) {
    double a[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = 0.5 * ((double)stress(i, j) + (double)stress(j, i));
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (a[p][q] == 0.0) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
        }
    }
    const double mn = std::min(a[0][0], std::min(a[1][1], a[2][2])), mx = std::max(a[0][0], std::max(a[1][1], a[2][2]));
    return mx > (double)m.max_principal_stress || (mx - mn) / 2.0 > (double)m.max_shear_stress;
}

// --------------------------------------------------------------------------- cuboid projection
// parry3d 0.13 Cuboid (= local Aabb [-he, he]) project_local_point, restated; unpinned.
template <class R> struct Projection { bool is_inside; Vec3<R> point; };
// CudaColliderOptions (src/cuda/cuda_collider_set.rs:22-38) of one collider: own boundary handling (-1 = the solver parameters'), penalty
// stiffness (src_kernels/cuda/particle_updater.rs:200-225, compiled out in the reference: the intended rule), flip_interior (gpu_collider.rs:58-92)
struct ColliderOpts { int handling = -1; double penalty = 0.0; bool flip_interior = false; };
template <class R> struct Cuboid { Vec3<R> he; R q[4]; Vec3<R> t; R friction; ColliderOpts opts; };
template <class R> Vec3<R> quat_rotate(const R q[4], const Vec3<R>& v, bool inverse) {
    // q = (i, j, k, w); v' = v + 2 w (u x v) + 2 u x (u x v)
    R s = inverse ? R(-1) : R(1);
    R ux = s * q[0], uy = s * q[1], uz = s * q[2], w = q[3];
    R cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
    R dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
    return Vec3<R>{{v[0] + R(2) * (w * cx + dx), v[1] + R(2) * (w * cy + dy), v[2] + R(2) * (w * cz + dz)}};
}
template <class R> Projection<R> cuboid_project_point(const Cuboid<R>& c, const Vec3<R>& pt_world) {
    Vec3<R> rel{{pt_world[0] - c.t[0], pt_world[1] - c.t[1], pt_world[2] - c.t[2]}};
    Vec3<R> pt = quat_rotate(c.q, rel, true);
    Vec3<R> mins_pt, pt_maxs, shift;
    bool inside = true;
    for (int i = 0; i < 3; ++i) {
        mins_pt[i] = -c.he[i] - pt[i];
        pt_maxs[i] = pt[i] - c.he[i];
        shift[i] = std::max(mins_pt[i], R(0)) - std::max(pt_maxs[i], R(0));
        if (shift[i] != R(0)) inside = false;
    }
    Vec3<R> proj;
    if (!inside) {
        for (int i = 0; i < 3; ++i) proj[i] = pt[i] + shift[i];
    } else {
        R best = -std::numeric_limits<R>::max();
        bool is_mins = false; int best_id = 0;
        for (int i = 0; i < 3; ++i) {
            if (mins_pt[i] < pt_maxs[i]) {
                if (pt_maxs[i] > best) { best_id = i; is_mins = false; best = pt_maxs[i]; }
            } else if (mins_pt[i] > best) { best_id = i; is_mins = true; best = mins_pt[i]; }
        }
        Vec3<R> sh{{R(0), R(0), R(0)}};
        sh[best_id] = is_mins ? best : -best;
        for (int i = 0; i < 3; ++i) proj[i] = pt[i] + sh[i];
    }
    Vec3<R> w = quat_rotate(c.q, proj, false);
    return Projection<R>{inside, Vec3<R>{{w[0] + c.t[0], w[1] + c.t[1], w[2] + c.t[2]}}};
}

// parry3d 0.13 Triangle::project_local_point = Ericson's closest point on a triangle (unpinned third party)
template <class R> Vec3<R> closest_on_triangle(const Vec3<R>& p, const Vec3<R>& a, const Vec3<R>& b, const Vec3<R>& c) {
    auto sub = [](const Vec3<R>& x, const Vec3<R>& y) { return Vec3<R>{{x[0] - y[0], x[1] - y[1], x[2] - y[2]}}; };
    auto lin = [](const Vec3<R>& x, const Vec3<R>& d, R t) { return Vec3<R>{{x[0] + t * d[0], x[1] + t * d[1], x[2] + t * d[2]}}; };
    Vec3<R> ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
    R d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) return a;
    Vec3<R> bp = sub(p, b);
    R d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) return b;
    R vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) return lin(a, ab, d1 / (d1 - d3));
    Vec3<R> cp = sub(p, c);
    R d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) return c;
    R vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) return lin(a, ac, d2 / (d2 - d6));
    R va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) return lin(b, sub(c, b), (d4 - d3) / ((d4 - d3) + (d5 - d6)));
    R denom = R(1) / (va + vb + vc), v = vb * denom, w = vc * denom;
    return Vec3<R>{{a[0] + ab[0] * v + ac[0] * w, a[1] + ab[1] * v + ac[1] * w, a[2] + ab[2] * v + ac[2] * w}};
}
// parry3d 0.13 HeightField (rows along z, columns along x, cell = triangles (p00,p10,p01) + (p10,p11,p01));
// project_local_point brute-forces every triangle; is_inside is never set on the CPU path.  `below_is_solid` adds the
// CUDA path's rule (src_kernels/gpu_collider.rs:69-91).
template <class R> struct HeightField { int nrows, ncols; std::vector<R> heights; Vec3<R> scale; R q[4]; Vec3<R> t; R friction; bool below_is_solid; ColliderOpts opts; };
template <class R> Projection<R> heightfield_project_point(const HeightField<R>& hf, const Vec3<R>& pt_world) {
    Vec3<R> rel{{pt_world[0] - hf.t[0], pt_world[1] - hf.t[1], pt_world[2] - hf.t[2]}};
    Vec3<R> lp = quat_rotate(hf.q, rel, true);
    R cw = R(1) / R(hf.ncols - 1), ch = R(1) / R(hf.nrows - 1);
    R best = std::numeric_limits<R>::max();
    Vec3<R> bp = lp;
    for (int i = 0; i + 1 < hf.nrows; ++i)
        for (int j = 0; j + 1 < hf.ncols; ++j) {
            R z0 = (R(-0.5) + ch * R(i)) * hf.scale[2], z1 = (R(-0.5) + ch * R(i + 1)) * hf.scale[2];
            R x0 = (R(-0.5) + cw * R(j)) * hf.scale[0], x1 = (R(-0.5) + cw * R(j + 1)) * hf.scale[0];
            Vec3<R> p00{{x0, hf.heights[i + hf.nrows * j] * hf.scale[1], z0}}, p10{{x0, hf.heights[i + 1 + hf.nrows * j] * hf.scale[1], z1}};
            Vec3<R> p01{{x1, hf.heights[i + hf.nrows * (j + 1)] * hf.scale[1], z0}}, p11{{x1, hf.heights[i + 1 + hf.nrows * (j + 1)] * hf.scale[1], z1}};
            for (int tri = 0; tri < 2; ++tri) {
                Vec3<R> c = tri == 0 ? closest_on_triangle(lp, p00, p10, p01) : closest_on_triangle(lp, p10, p11, p01);
                R d = (c[0] - lp[0]) * (c[0] - lp[0]) + (c[1] - lp[1]) * (c[1] - lp[1]) + (c[2] - lp[2]) * (c[2] - lp[2]);
                if (d < best) { best = d; bp = c; }
            }
        }
    bool inside = false;
    if (hf.below_is_solid) {
        R ux = lp[0] / hf.scale[0] + R(0.5), uz = lp[2] / hf.scale[2] + R(0.5);
        inside = ux >= 0 && ux <= 1 && uz >= 0 && uz <= 1 && (hf.opts.flip_interior ? lp[1] >= bp[1] : lp[1] <= bp[1]);   // gpu_collider.rs:81-89
    }
    Vec3<R> w = quat_rotate(hf.q, bp, false);
    return Projection<R>{inside, Vec3<R>{{w[0] + hf.t[0], w[1] + hf.t[1], w[2] + hf.t[2]}}};
}

// parry3d 0.13 TriMesh::project_local_point: closest point over all triangles (a BVH there, brute force here); is_inside only
// with TriMeshFlags::ORIENTED, from the pseudo-normal of the closest feature (TriMesh::compute_pseudo_normals: angle-weighted
// sum of the incident face normals on a vertex, sum of the two face normals on an edge).  Third party, unpinned.
template <class R> struct TriMesh {
    std::vector<Vec3<R>> v;
    std::vector<std::array<uint32_t, 3>> idx;
    std::vector<Vec3<double>> vn;                                 // per vertex
    std::vector<std::array<Vec3<double>, 3>> en;                  // per triangle: edges ab, ac, bc
    std::vector<Vec3<double>> fn;
    R q[4]; Vec3<R> t; R friction; bool oriented, flip_interior; ColliderOpts opts;
    void compute_pseudo_normals() {
        vn.assign(v.size(), Vec3<double>{{0, 0, 0}});
        fn.assign(idx.size(), Vec3<double>{{0, 0, 0}});
        en.assign(idx.size(), std::array<Vec3<double>, 3>{{Vec3<double>{{0, 0, 0}}, Vec3<double>{{0, 0, 0}}, Vec3<double>{{0, 0, 0}}}});
        std::map<std::pair<uint32_t, uint32_t>, Vec3<double>> acc;
        auto d = [&](uint32_t a, uint32_t b) { return Vec3<double>{{(double)v[a][0] - (double)v[b][0], (double)v[a][1] - (double)v[b][1], (double)v[a][2] - (double)v[b][2]}}; };
        auto ang = [](const Vec3<double>& x, const Vec3<double>& y) {
            const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), ny = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
            if (nx == 0 || ny == 0) return 0.0;
            return std::acos(std::max(-1.0, std::min(1.0, (x[0] * y[0] + x[1] * y[1] + x[2] * y[2]) / (nx * ny))));
        };
        for (size_t k = 0; k < idx.size(); ++k) {
            const uint32_t a = idx[k][0], b = idx[k][1], c = idx[k][2];
            const Vec3<double> ab = d(b, a), ac = d(c, a);
            Vec3<double> n{{ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]}};
            const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (l == 0) continue;
            for (int e = 0; e < 3; ++e) n[e] /= l;
            fn[k] = n;
            const double w[3] = {ang(ab, ac), ang(d(a, b), d(c, b)), ang(d(a, c), d(b, c))};
            const uint32_t vid[3] = {a, b, c};
            for (int j = 0; j < 3; ++j) for (int e = 0; e < 3; ++e) vn[vid[j]][e] += n[e] * w[j];
            const uint32_t ed[3][2] = {{a, b}, {a, c}, {b, c}};
            for (int j = 0; j < 3; ++j) {
                auto& s = acc.emplace(std::make_pair(std::min(ed[j][0], ed[j][1]), std::max(ed[j][0], ed[j][1])), Vec3<double>{{0, 0, 0}}).first->second;
                for (int e = 0; e < 3; ++e) s[e] += n[e];
            }
        }
        for (size_t k = 0; k < idx.size(); ++k) {
            const uint32_t a = idx[k][0], b = idx[k][1], c = idx[k][2];
            const uint32_t ed[3][2] = {{a, b}, {a, c}, {b, c}};
            for (int j = 0; j < 3; ++j) en[k][j] = acc[std::make_pair(std::min(ed[j][0], ed[j][1]), std::max(ed[j][0], ed[j][1]))];
        }
    }
};
// which feature of (a, b, c) the closest point sits on: 0 1 2 vertices, 3 4 5 edges ab ac bc, 6 face (same regions as closest_on_triangle)
template <class R> int triangle_feature(const Vec3<R>& p, const Vec3<R>& a, const Vec3<R>& b, const Vec3<R>& c) {
    auto sub = [](const Vec3<R>& x, const Vec3<R>& y) { return Vec3<R>{{x[0] - y[0], x[1] - y[1], x[2] - y[2]}}; };
    Vec3<R> ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
    R d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) return 0;
    Vec3<R> bp = sub(p, b);
    R d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) return 1;
    if (d1 * d4 - d3 * d2 <= 0 && d1 >= 0 && d3 <= 0) return 3;
    Vec3<R> cp = sub(p, c);
    R d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) return 2;
    if (d5 * d2 - d1 * d6 <= 0 && d2 >= 0 && d6 <= 0) return 4;
    if (d3 * d6 - d5 * d4 <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) return 5;
    return 6;
}
// src_kernels/gpu_collider.rs:99-201 (`polyline_project_point`): a closed counter-clockwise polygon in the local x-y plane, extruded along z.
// Closest segment in 3-D (brute force, as the reference), the projection keeps the point's local z; inside / outside from the segment's planar
// normal (Segment::planar_normal(2) = (dir.y, -dir.x, 0) normalised), at a vertex from the adjacent segment's direction unless the two are
// collinear within 1e-3.  Only the reference's GPU path has this shape (the CPU path's parry Polyline is open and never inside).  Deviation:
// the reference dots (WORLD point - LOCAL projection) with the normal -- right for an identity pose only; the local point is used.
template <class R> struct Polyline { std::vector<Vec3<R>> v; R q[4]; Vec3<R> t; R friction; ColliderOpts opts; };
template <class R> Projection<R> polyline_project_point(const Polyline<R>& pl, const Vec3<R>& pt_world) {
    Vec3<R> rel{{pt_world[0] - pl.t[0], pt_world[1] - pl.t[1], pt_world[2] - pl.t[2]}};
    Vec3<R> lp = quat_rotate(pl.q, rel, true);
    const size_t n = pl.v.size();
    R best = std::numeric_limits<R>::max(), bt = 0;
    Vec3<R> bp = lp;
    size_t bi = 0;
    for (size_t i = 0; i < n; ++i) {
        const Vec3<R>& A = pl.v[i];
        const Vec3<R>& B = pl.v[(i + 1) % n];
        Vec3<R> ab{{B[0] - A[0], B[1] - A[1], B[2] - A[2]}}, ap{{lp[0] - A[0], lp[1] - A[1], lp[2] - A[2]}};
        R ab2 = dot(ab, ab);
        R tt = ab2 > R(0) ? dot(ap, ab) / ab2 : R(0);
        tt = std::min(std::max(tt, R(0)), R(1));
        Vec3<R> c{{A[0] + tt * ab[0], A[1] + tt * ab[1], A[2] + tt * ab[2]}};
        Vec3<R> dd{{c[0] - lp[0], c[1] - lp[1], c[2] - lp[2]}};
        R d = norm(dd);
        if (d < best) { best = d; bp = c; bt = tt; bi = i; }
    }
    bool inside = false;
    {
        const Vec3<R>& A = pl.v[bi];
        const Vec3<R>& B = pl.v[(bi + 1) % n];
        R dx = B[0] - A[0], dy = B[1] - A[1];
        R nl = std::sqrt(dx * dx + dy * dy);
        if (nl > R(0)) {
            R n1[2] = {dy / nl, -dx / nl};
            R side = (lp[0] - bp[0]) * n1[0] + (lp[1] - bp[1]) * n1[1];
            if (bt > R(0) && bt < R(1)) inside = side <= R(0);
            else {
                Vec3<R> d2;
                if (bt <= R(0)) { const Vec3<R>& P = pl.v[(bi + n - 1) % n]; d2 = Vec3<R>{{-(A[0] - P[0]), -(A[1] - P[1]), -(A[2] - P[2])}}; }
                else { const Vec3<R>& Nn = pl.v[(bi + 2) % n]; d2 = Vec3<R>{{Nn[0] - B[0], Nn[1] - B[1], Nn[2] - B[2]}}; }
                R dt = n1[0] * d2[0] + n1[1] * d2[1];
                R thr = R(1.0e-3) * norm(d2);
                inside = std::fabs(dt) > thr ? dt >= R(0) : side <= R(0);
            }
        }
    }
    bp[2] = lp[2];
    if (pl.opts.flip_interior) inside = !inside;
    Vec3<R> w = quat_rotate(pl.q, bp, false);
    return Projection<R>{inside, Vec3<R>{{w[0] + pl.t[0], w[1] + pl.t[1], w[2] + pl.t[2]}}};
}

template <class R> Projection<R> trimesh_project_point(const TriMesh<R>& m, const Vec3<R>& pt_world) {
    Vec3<R> rel{{pt_world[0] - m.t[0], pt_world[1] - m.t[1], pt_world[2] - m.t[2]}};
    Vec3<R> lp = quat_rotate(m.q, rel, true);
    R best = std::numeric_limits<R>::max();
    Vec3<R> bp = lp;
    size_t bt = 0;
    for (size_t k = 0; k < m.idx.size(); ++k) {
        Vec3<R> c = closest_on_triangle(lp, m.v[m.idx[k][0]], m.v[m.idx[k][1]], m.v[m.idx[k][2]]);
        R d = (c[0] - lp[0]) * (c[0] - lp[0]) + (c[1] - lp[1]) * (c[1] - lp[1]) + (c[2] - lp[2]) * (c[2] - lp[2]);
        if (d < best) { best = d; bp = c; bt = k; }
    }
    bool inside = false;
    if (m.oriented && !m.idx.empty()) {
        const int f = triangle_feature(lp, m.v[m.idx[bt][0]], m.v[m.idx[bt][1]], m.v[m.idx[bt][2]]);
        const Vec3<double>& n = f < 3 ? m.vn[m.idx[bt][f]] : (f < 6 ? m.en[bt][f - 3] : m.fn[bt]);
        inside = ((double)(lp[0] - bp[0]) * n[0] + (double)(lp[1] - bp[1]) * n[1] + (double)(lp[2] - bp[2]) * n[2]) <= 0.0;
        if (m.flip_interior) inside = !inside;
    }
    Vec3<R> w = quat_rotate(m.q, bp, false);
    return Projection<R>{inside, Vec3<R>{{w[0] + m.t[0], w[1] + m.t[1], w[2] + m.t[2]}}};
}

// ---------------------------------------------------------------------------------------- solver
struct Region { uint64_t id; size_t begin, end; };

template <class R>
struct Oracle {
    R h;
    uint32_t log2_bytes;
    uint64_t mem_len;        // the reference's `memory.len()` (bytes at 64-B slots)
    size_t slot;             // bytes per node slot in OUR map (64 for f32, 128 for f64)
    uint8_t* mem = nullptr;
    size_t map_bytes = 0;
    std::vector<Particle<R>, no_init_allocator<Particle<R>>> particles;   // filled in parallel by set_particles (first touch)
    std::vector<size_t> order;
    std::vector<KeyIdx, no_init_allocator<KeyIdx>> sort_pairs, sort_tmp;
    std::vector<Region> regions;
    std::vector<std::array<size_t, 5>> bins;
    std::vector<SparklParticleModel> models;
    std::vector<Cuboid<R>> cuboids;
    std::vector<HeightField<R>> heightfields;
    std::vector<TriMesh<R>> trimeshes;
    std::vector<Polyline<R>> polylines;
    int last_num_substeps = 0;
    std::vector<double> last_dts;
    int num_threads = 1;

    Oracle(R cell_width, uint32_t log2b) : h(cell_width), log2_bytes(log2b) {
        mem_len = 1ULL << log2b;
        slot = sizeof(GridNode<R>) <= 64 ? 64 : 128;
        map_bytes = (mem_len / 64) * slot;
        void* m = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { std::perror("oracle mmap"); mem = nullptr; } else mem = (uint8_t*)m;
    }
    ~Oracle() { if (mem) munmap(mem, map_bytes); }

    GridNode<R>& node(uint64_t cell_id) { return *(GridNode<R>*)(mem + (cell_id >> PACK_ALIGN) * slot); }
    // sp_grid.rs:252-254, 276-279 with size_of::<GridNode>() == 48
    bool is_cell_index_valid(uint64_t id) const { return id <= mem_len - 48 && id + 48 <= mem_len; }
    bool is_neighborhood_valid(uint64_t id) const { return is_cell_index_valid(id) && is_cell_index_valid(packed_add(id, packed_plus_five())); }
    // sp_grid.rs:216-226
    uint64_t cell_associated_to_point(const Vec3<R>& pt) const {
        uint32_t c[3];
        for (int a = 0; a < 3; ++a) c[a] = round_as_u32(pt[a] / h) - 1u;  // wrapping
        return pack(c[0], c[1], c[2]);
    }
    Vec3<R> cell_center(uint64_t id) const {  // sp_grid.rs:256-258
        uint32_t c[3]; unpack(id, c);
        return Vec3<R>{{(R)c[0] * h, (R)c[1] * h, (R)c[2] * h}};
    }

    const SparklParticleModel& model_of(const Particle<R>& p) const { return models[p.model]; }
    static bool is_fluid(const SparklParticleModel& m) { return m.constitutive_model.tag == SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH; }

    // src/dynamics/models/*.rs glue (update_particle_stress)
    Mat3<R> stress_of(const Particle<R>& p) const {
        const auto& cm = model_of(p).constitutive_model;
        switch (cm.tag) {
            case SPARKL_CONSTITUTIVE_COROTATED_LINEAR: return corotated_kirchhoff(cm.u.corotated, p.phase, p.elastic_hardening, p.deformation_gradient);
            case SPARKL_CONSTITUTIVE_NEO_HOOKEAN: return neo_hookean_kirchhoff(cm.u.neo_hookean, p.phase, p.elastic_hardening, p.deformation_gradient);
            case SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH:
                // NB: the glue passes plastic_deformation_gradient_det as the "fluid J" (src/dynamics/models/eos_monaghan_sph.rs:12-20)
                return monaghan_kirchhoff(cm.u.monaghan, p.mass, p.volume0, p.density_fluid(), p.plastic_deformation_gradient_det, p.velocity_gradient);
            default: return Mat3<R>::zeros();
        }
    }
    R pos_energy_of(const Particle<R>& p) const {
        const auto& cm = model_of(p).constitutive_model;
        switch (cm.tag) {
            case SPARKL_CONSTITUTIVE_COROTATED_LINEAR: return corotated_pos_energy(cm.u.corotated, p.deformation_gradient, p.elastic_hardening);
            case SPARKL_CONSTITUTIVE_NEO_HOOKEAN: return neo_hookean_pos_energy(cm.u.neo_hookean, p.phase, p.elastic_hardening, p.deformation_gradient);
            default: return R(0);  // trait default, constitutive_model.rs:12-14
        }
    }
    R model_timestep_bound(const Particle<R>& p) const {
        const auto& cm = model_of(p).constitutive_model;
        switch (cm.tag) {
            case SPARKL_CONSTITUTIVE_COROTATED_LINEAR: {  // elasticity_corotated_linear.rs:126-142
                const auto& m = cm.u.corotated;
                R bulk = (R)m.lambda + R(2) * (R)m.mu / R(3), shear = (R)m.mu;
                return sound_speed_bound((R)m.cfl_coeff, bulk * p.elastic_hardening, shear * p.elastic_hardening, p.density0(), p.velocity, h);
            }
            case SPARKL_CONSTITUTIVE_NEO_HOOKEAN: {  // elasticity_neo_hookean.rs:149-165
                const auto& m = cm.u.neo_hookean;
                R bulk = (R)m.lambda + R(2) * (R)m.mu / R(3), shear = (R)m.mu;
                return sound_speed_bound((R)m.cfl_coeff, bulk * p.elastic_hardening, shear * p.elastic_hardening, p.density0(), p.velocity, h);
            }
            case SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH:
                return monaghan_timestep_bound(cm.u.monaghan, p.deformation_gradient(0, 0), p.mass, p.volume0, p.density_fluid(), p.velocity, h);
            default: return std::numeric_limits<R>::max();
        }
    }

    // ---- ParticleSet::sort, src/dynamics/particle_set.rs:127-277 (update_particle_range = false)
    void sort() {
        const size_t n = particles.size();
        if (order.size() != n) { order.resize(n); for (size_t i = 0; i < n; ++i) order[i] = i; }
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t i = 0; i < n; ++i) {
            auto& p = particles[i];
            p.grid_index = cell_associated_to_point(p.position);
            if (!is_neighborhood_valid(p.grid_index)) { p.grid_index = ~0ULL; p.failed = true; }
        }
        // `order.par_sort_by_key(|i| particles[*i].grid_index)` (particle_set.rs:142): stable, so the result is the sequential stable order
        sort_pairs.resize(n);
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t i = 0; i < n; ++i) sort_pairs[i] = KeyIdx{particles[order[i]].grid_index, order[i]};
        parallel_stable_sort_pairs(sort_pairs, sort_tmp, num_threads);
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t i = 0; i < n; ++i) order[i] = sort_pairs[i].idx;
        // reset previously active regions (:151-160)
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t r = 0; r < regions.size(); ++r)
            for (uint64_t i = 0; i < 64; ++i) std::memset(&node(regions[r].id | (i << PACK_ALIGN)), 0, sizeof(GridNode<R>));
        regions.clear(); bins.clear();
        if (n == 0) return;
        std::unordered_set<uint64_t> active;
        size_t region_start = 0;
        uint64_t curr_cell = particles[order[0]].grid_index;
        std::array<size_t, 5> bin; bin.fill(SIZE_MAX);
        int bin_len = 0;
        uint64_t curr_model = particles[order[0]].model;
        for (size_t i = 0; i < n; ++i) {  // the serial "big loop" (:224-247)
            const auto& p = particles[order[i]];
            uint64_t new_cell = p.grid_index, new_model = p.model;
            uint64_t new_region = new_cell & REGION_ID_MASK, curr_region = curr_cell & REGION_ID_MASK;
            if (new_region != curr_region) {
                regions.push_back({curr_region, region_start, i});
                active.insert(curr_region);
                region_start = i;
            }
            if (new_cell != curr_cell || new_model != curr_model || bin_len == 4) {
                bins.push_back(bin); bin.fill(SIZE_MAX); bin_len = 0;
            }
            bin[bin_len++] = order[i];
            curr_cell = new_cell; curr_model = new_model;
        }
        if (curr_cell != ~0ULL) {
            uint64_t curr_region = curr_cell & REGION_ID_MASK;
            regions.push_back({curr_region, region_start, n});
            active.insert(curr_region);
            if (bin_len > 0) bins.push_back(bin);
        }
        // region expansion (:268-275, sp_grid.rs:299-309)
        size_t nreg = regions.size();
        for (size_t i = 0; i < nreg; ++i)
            for (auto& s : NBH_REGION_SHIFTS) {
                uint64_t nb = packed_add(regions[i].id, pack(s[0], s[1], s[2]));
                if (active.insert(nb).second) regions.push_back({nb, 0, 0});
            }
    }

    // ---- timestep_estimator.rs:10-51.  Returns NaN if any particle's bound is NaN (reference panics).
    R adaptive_timestep_length(R max_dt) const {
        R d = (h * h) / R(4);
        R best = std::numeric_limits<R>::infinity();
        bool nan = false;
#pragma omp parallel for schedule(static) num_threads(num_threads) reduction(min : best) reduction(|| : nan)
        for (size_t ii = 0; ii < order.size(); ++ii) {
            const auto& p = particles[order[ii]];
            // every in-scope model has PARTICLE_VELOCITY_BOUND and CONSTITUTIVE_MODEL_BOUND active
            R dt = max_dt;
            R norm_b = d * frob_norm(p.velocity_gradient);
            R apic_v = norm_b * R(6) * std::sqrt(R(3)) / h;
            R v = norm(p.velocity) + apic_v;
            dt = rust_min(dt, h / v);
            if (!p.failed) dt = rust_min(dt, model_timestep_bound(p));
            if (dt != dt) nan = true;
            if (dt < best) best = dt;
        }
        return nan ? std::numeric_limits<R>::quiet_NaN() : best;
    }

    static bool region_has_color(uint64_t region_id, uint64_t color) {  // sp_grid.rs:184-190
        return (region_id & (0x7ULL << PACK_HEADER)) == (color << PACK_HEADER);
    }

    // ---- particle_to_grid.rs:10-95
    void p2g(R dt, const Vec3<R>& gravity) {
        R invd = inv_d(h);
        uint64_t pshift[27];
        for (int k = 0; k < 27; ++k) pshift[k] = packed_nbh_shift(k);
        for (uint64_t color = 0; color < 8; ++color) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(num_threads)
            for (size_t r = 0; r < regions.size(); ++r) {
                const Region& reg = regions[r];
                if (!region_has_color(reg.id, color)) continue;
                for (size_t oi = reg.begin; oi < reg.end; ++oi) {
                    const Particle<R>& p = particles[order[oi]];
                    if (p.grid_index == ~0ULL) continue;  // deviation 2 (reference would panic later)
                    Mat3<R> stress = !p.failed ? stress_of(p) : Mat3<R>::zeros();
                    Vec3<R> d0 = p.dir_to_associated_grid_node(h);
                    R w[3][3];
                    for (int a = 0; a < 3; ++a) eval_all(-d0[a] / h, w[a]);
                    Mat3<R> affine = p.mass * p.velocity_gradient - (p.volume0 * invd * dt) * stress;
                    Vec3<R> mom{{p.mass * p.velocity[0], p.mass * p.velocity[1], p.mass * p.velocity[2]}};
                    R psi_mass = (p.phase > R(0) && p.crack_propagation_factor != R(0) && !p.failed) ? p.mass : R(0);
                    R psi_pos_momentum = psi_mass * p.psi_pos;
                    for (int k = 0; k < 27; ++k) {
                        GridNode<R>& cell = node(packed_add(p.grid_index, pshift[k]));
                        const int* s = NBH_SHIFTS[k];
                        Vec3<R> dpt{{d0[0] + (R)s[0] * h, d0[1] + (R)s[1] * h, d0[2] + (R)s[2] * h}};
                        R weight = w[0][s[0]] * w[1][s[1]] * w[2][s[2]];
                        cell.mass += weight * p.mass;
                        Vec3<R> ad = affine * dpt;
                        for (int a = 0; a < 3; ++a) cell.momentum[a] += weight * (ad[a] + mom[a]);
                        cell.psi_momentum += weight * psi_pos_momentum;
                        cell.psi_mass += weight * psi_mass;
                        R im = inv_exact(cell.mass);
                        for (int a = 0; a < 3; ++a) cell.velocity[a] = (cell.momentum[a] + cell.mass * gravity[a] * dt) * im;
                    }
                }
            }
        }
    }

    // ---- the reference's GPU grid update (src_kernels/cuda/grid_update.rs:54-190), the only place where colliders carry options of their own
    // (CudaColliderOptions): ONE collider acts on a node -- the one whose projection of the node is closest -- and the collision normal is the
    // normalised central-difference gradient of the signed distance to ALL colliders, not the direction of that projection.
    // project_point_with_max_dist (src_kernels/gpu_collider.rs:31-92): cuboids and polylines always answer, triangle meshes too (the
    // max_dist argument is commented out there: Real::MAX), heightfields only within max_dist = 2 h.
    template <class F> void for_each_projection(const Vec3<R>& q, int params_handling, F&& f) const {
        auto handling_of = [&](const ColliderOpts& o) { return o.handling >= 0 ? o.handling : params_handling; };
        const R max_dist = R(2) * h;
        for (const auto& col : cuboids) if (handling_of(col.opts) != SPARKL_BOUNDARY_NONE) f(cuboid_project_point(col, q), col.friction, handling_of(col.opts));
        for (const auto& hf : heightfields) if (handling_of(hf.opts) != SPARKL_BOUNDARY_NONE) {
            const Projection<R> pr = heightfield_project_point(hf, q);
            Vec3<R> d{{pr.point[0] - q[0], pr.point[1] - q[1], pr.point[2] - q[2]}};
            if (norm(d) <= max_dist) f(pr, hf.friction, handling_of(hf.opts));
        }
        for (const auto& tm : trimeshes) if (handling_of(tm.opts) != SPARKL_BOUNDARY_NONE && !tm.idx.empty()) f(trimesh_project_point(tm, q), tm.friction, handling_of(tm.opts));
        for (const auto& pl : polylines) if (handling_of(pl.opts) != SPARKL_BOUNDARY_NONE && pl.v.size() >= 2) f(polyline_project_point(pl, q), pl.friction, handling_of(pl.opts));
    }
    // grid_update.rs:54-70
    bool sdf(const Vec3<R>& q, int params_handling, R& out) const {
        bool any = false;
        R best = 0;
        for_each_projection(q, params_handling, [&](const Projection<R>& pr, R, int) {
            Vec3<R> d{{pr.point[0] - q[0], pr.point[1] - q[1], pr.point[2] - q[2]}};
            const R sd = norm(d) * (pr.is_inside ? R(-1) : R(1));
            if (!any || sd < best) { best = sd; any = true; }
        });
        out = best;
        return any;
    }
    // grid_update.rs:72-95
    Vec3<R> sdf_gradient(const Vec3<R>& q, int params_handling) const {
        const R FACTOR = R(0.1);
        Vec3<R> g{{0, 0, 0}};
        for (int a = 0; a < 3; ++a) {
            Vec3<R> qp = q, qn = q;
            qp[a] += h * FACTOR; qn[a] -= h * FACTOR;
            R sp, sn;
            if (sdf(qp, params_handling, sp) && sdf(qn, params_handling, sn)) g[a] = (sp - sn) / h / FACTOR / R(2);
        }
        const R n = norm(g);
        if (!(n > R(1.0e-5))) return Vec3<R>{{0, 0, 0}};   // try_normalize(1.0e-5).unwrap_or(zeros)
        return Vec3<R>{{g[0] / n, g[1] / n, g[2] / n}};
    }
    bool any_collider_options() const {
        for (const auto& c : cuboids) if (c.opts.handling >= 0) return true;
        for (const auto& c : heightfields) if (c.opts.handling >= 0) return true;
        for (const auto& c : trimeshes) if (c.opts.handling >= 0) return true;
        for (const auto& c : polylines) if (c.opts.handling >= 0) return true;
        return false;
    }
    // grid_update.rs:97-190 (update_single_cell) on a node whose velocity already holds (momentum + m g dt) / m
    void gpu_rule_cell(GridNode<R>& cell, const Vec3<R>& cell_pos, R dt, int params_handling) {
        bool have = false, inside = false;
        Vec3<R> scaled_dir{{0, 0, 0}};
        R best_dist = std::numeric_limits<R>::max(), friction = 0;
        int handling = SPARKL_BOUNDARY_NONE;
        for_each_projection(cell_pos, params_handling, [&](const Projection<R>& pr, R fr, int hd) {
            Vec3<R> d{{cell_pos[0] - pr.point[0], cell_pos[1] - pr.point[1], cell_pos[2] - pr.point[2]}};
            const R dist = norm(d);
            if (dist < best_dist) { best_dist = dist; scaled_dir = d; inside = pr.is_inside; friction = fr; handling = hd; have = true; }
        });
        if (!have) return;
        if (inside) cell.flags |= 2u;
        if (handling == SPARKL_BOUNDARY_STICK) {
            if (inside) cell.velocity[0] = cell.velocity[1] = cell.velocity[2] = 0;
            return;
        }
        if (handling != SPARKL_BOUNDARY_FRICTION && handling != SPARKL_BOUNDARY_FRICTION_Z_UP) return;
        const Vec3<R> normal = sdf_gradient(cell_pos, params_handling);
        if (normal[0] == R(0) && normal[1] == R(0) && normal[2] == R(0)) return;
        const R dist = norm(scaled_dir);
        if (handling == SPARKL_BOUNDARY_FRICTION_Z_UP && !(normal[2] >= R(0))) return;
        Vec3<R> vel{{cell.velocity[0], cell.velocity[1], cell.velocity[2]}};
        const R normal_vel = dot(vel, normal);
        if (normal_vel < R(0)) {
            const R dist_with_margin = dist - h;
            if (inside || dist_with_margin <= R(0)) {
                Vec3<R> tv{{vel[0] - normal_vel * normal[0], vel[1] - normal_vel * normal[1], vel[2] - normal_vel * normal[2]}};
                const R tvn = norm(tv);
                for (int a = 0; a < 3; ++a) cell.velocity[a] = tv[a];
                if (tvn > R(1.0e-10)) {
                    const R f = std::max(tvn + normal_vel * friction, R(0));
                    for (int a = 0; a < 3; ++a) cell.velocity[a] = tv[a] / tvn * f;
                }
            } else if (-normal_vel * dt > dist_with_margin) {
                const R k = dist_with_margin / dt + normal_vel;
                for (int a = 0; a < 3; ++a) cell.velocity[a] -= k * normal[a];
            }
        }
    }

    // ---- grid_update.rs:43-132 (the CPU path: every collider in turn, SolverParameters::boundary_handling for all of them).  As soon as ONE
    // collider carries a boundary handling of its own -- an option only the reference's GPU path has -- that path's rule applies to the whole
    // collider set (gpu_rule_cell above).
    void grid_update(R dt, int boundary_handling, uint32_t dofs) {
        const bool gpu_rule = any_collider_options();
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t r = 0; r < regions.size(); ++r) {
            for (uint64_t i = 0; i < 64; ++i) {
                uint64_t cell_id = regions[r].id | (i << PACK_ALIGN);
                Vec3<R> cell_pos = cell_center(cell_id);
                GridNode<R>& cell = node(cell_id);
                for (int a = 0; a < 3; ++a)
                    if (dofs & (1u << a)) { cell.momentum[a] = 0; cell.velocity[a] = 0; }
                if (gpu_rule) { gpu_rule_cell(cell, cell_pos, dt, boundary_handling); continue; }
                auto apply = [&](const Projection<R>& proj, R friction) {
                    if (!(cell.flags & 2u) && proj.is_inside) cell.flags |= 2u;
                    if (boundary_handling == SPARKL_BOUNDARY_NONE) return;
                    if (boundary_handling == SPARKL_BOUNDARY_STICK) {
                        if (proj.is_inside) cell.velocity[0] = cell.velocity[1] = cell.velocity[2] = 0;
                        return;
                    }
                    Vec3<R> dv{{cell_pos[0] - proj.point[0], cell_pos[1] - proj.point[1], cell_pos[2] - proj.point[2]}};
                    R dist = norm(dv);
                    if (!(dist > R(1.0e-5))) return;  // Unit::try_new_and_get(.., 1e-5)
                    Vec3<R> normal{{dv[0] / dist, dv[1] / dist, dv[2] / dist}};
                    if (proj.is_inside) for (int a = 0; a < 3; ++a) normal[a] = -normal[a];
                    Vec3<R> vel{{cell.velocity[0], cell.velocity[1], cell.velocity[2]}};
                    R normal_vel = dot(vel, normal);
                    if (normal_vel < R(0)) {
                        R dist_with_margin = dist - h;
                        if (proj.is_inside || dist_with_margin <= R(0)) {
                            Vec3<R> tv{{vel[0] - normal_vel * normal[0], vel[1] - normal_vel * normal[1], vel[2] - normal_vel * normal[2]}};
                            R tvn = norm(tv);
                            for (int a = 0; a < 3; ++a) cell.velocity[a] = tv[a];
                            if (tvn > R(1.0e-10)) {
                                R f = std::max(tvn + normal_vel * friction, R(0));
                                for (int a = 0; a < 3; ++a) cell.velocity[a] = tv[a] / tvn * f;
                            }
                        } else if (-normal_vel * dt > dist_with_margin) {
                            R k = dist_with_margin / dt + normal_vel;
                            for (int a = 0; a < 3; ++a) cell.velocity[a] -= k * normal[a];
                        }
                    }
                };
                for (const auto& col : cuboids) apply(cuboid_project_point(col, cell_pos), col.friction);
                for (const auto& hf : heightfields) apply(heightfield_project_point(hf, cell_pos), hf.friction);
                for (const auto& tm : trimeshes) apply(trimesh_project_point(tm, cell_pos), tm.friction);
                for (const auto& pl : polylines) apply(polyline_project_point(pl, cell_pos), pl.friction);
            }
        }
    }

    // ---- grid_to_particle.rs:10-166.  damage_model: DamageModel::ModifiedEigenerosion (grid_to_particle.rs:65-77) or None; the
    // other two (CdMpm, Eigenerosion) are separate passes of mls_solver.rs:68-83 that this restatement does not hold.
    int damage_model = SPARKL_DAMAGE_NONE;
    bool project_particles = false;   // SolverParameters::enable_boundary_particle_projection (set by substep / oracle_g2p_project)
    bool penalty_force = false;       // opt-in: apply the penalty force the reference has under `if false` (see g2p)
    void g2p(R dt) {
        R invd = inv_d(h);
        uint64_t pshift[27];
        for (int k = 0; k < 27; ++k) pshift[k] = packed_nbh_shift(k);
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t b = 0; b < bins.size(); ++b) {
            for (int ii = 0; ii < 4; ++ii) {
                if (bins[b][ii] == SIZE_MAX) break;
                Particle<R>& p = particles[bins[b][ii]];
                if (p.grid_index == ~0ULL) continue;  // deviation 2
                const SparklParticleModel& model = model_of(p);
                bool fluid = is_fluid(model);
                Vec3<R> d0 = p.dir_to_associated_grid_node(h);
                Vec3<R> vel{{0, 0, 0}};
                Mat3<R> vg = Mat3<R>::zeros();
                R vg_det = 0;
                R psi_pos_momentum = 0;
                R w[3][3];
                for (int a = 0; a < 3; ++a) eval_all(-d0[a] / h, w[a]);
                for (int k = 0; k < 27; ++k) {
                    const GridNode<R>& cell = node(packed_add(p.grid_index, pshift[k]));
                    const int* s = NBH_SHIFTS[k];
                    Vec3<R> dpt{{d0[0] + (R)s[0] * h, d0[1] + (R)s[1] * h, d0[2] + (R)s[2] * h}};
                    R weight = w[0][s[0]] * w[1][s[1]] * w[2][s[2]];
                    Vec3<R> cv{{cell.velocity[0], cell.velocity[1], cell.velocity[2]}};
                    for (int a = 0; a < 3; ++a) vel[a] += weight * cv[a];
                    R wi = weight * invd;
                    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) vg(i, j) += (wi * cv[i]) * dpt[j];
                    vg_det += weight * dot(cv, dpt) * invd;
                    psi_pos_momentum += weight * cell.psi_momentum * inv_exact(cell.psi_mass);   // grid_to_particle.rs:57-58
                }
                p.velocity = vel;
                p.velocity_gradient = vg;
                // Modified Eigenerosion (grid_to_particle.rs:65-77)
                if (damage_model == SPARKL_DAMAGE_MODIFIED_EIGENEROSION && p.crack_propagation_factor != R(0) && p.phase > R(0)) {
                    R crack_energy = p.crack_propagation_factor * h * psi_pos_momentum;
                    if (crack_energy > p.crack_threshold) p.phase = R(0);
                }
                if (p.kinematic_vel_enabled) p.velocity = p.kinematic_vel;
                for (int a = 0; a < 3; ++a) p.position[a] += p.velocity[a] * dt;
                if (!fluid) {
                    p.deformation_gradient = p.deformation_gradient + (p.velocity_gradient * dt) * p.deformation_gradient;
                } else {
                    p.deformation_gradient(0, 0) += (vg_det * dt) * p.deformation_gradient(0, 0);
                }
                if (model.has_plastic_model) {   // plastic_model.update_particle (src/dynamics/models/plasticity_*.rs glue)
                    const auto& pm = model.plastic_model;
                    if (pm.tag == SPARKL_PLASTIC_DRUCKER_PRAGER)
                        dp_update_particle(pm.u.drucker_prager, p.phase, p.deformation_gradient, p.plastic_deformation_gradient_det, p.plastic_hardening, p.log_vol_gain);
                    else if (pm.tag == SPARKL_PLASTIC_SNOW)
                        snow_update_particle(pm.u.snow, p.deformation_gradient, p.elastic_hardening, p.plastic_deformation_gradient_det);
                    else if (pm.tag == SPARKL_PLASTIC_RANKINE)
                        rankine_update_particle(pm.u.rankine, p.deformation_gradient, p.plastic_hardening);
                    else if (pm.tag == SPARKL_PLASTIC_NACC)
                        nacc_update_particle(pm.u.nacc, p.deformation_gradient, p.nacc_alpha);
                }
                if (p.is_static) { p.velocity = Vec3<R>{{0, 0, 0}}; p.velocity_gradient = Mat3<R>::zeros(); }
                if (p.density_def_grad() == R(0) || p.failed || (!fluid && std::fabs(p.deformation_gradient(0, 0)) > R(1.0e4))) {
                    p.failed = true;
                    p.deformation_gradient = Mat3<R>::identity();
                    p.velocity_gradient = Mat3<R>::zeros();
                }
                R energy = pos_energy_of(p);
                p.psi_pos = std::fmax(p.psi_pos, energy);
                p.parameter1 = p.psi_pos * p.mass;
                p.parameter2 = p.mass;
                // grid_to_particle.rs:134-144, failure_maximum_stress.rs (glue): stress of the updated particle
                if (model.has_failure_model && model.failure_model.tag == SPARKL_FAILURE_MAXIMUM_STRESS &&
                    maximum_stress_failed(model.failure_model.u.maximum_stress, stress_of(p)))
                    p.phase = R(0);
                // grid_to_particle.rs:146-163: particles found inside a collider are put back on its surface
                if (project_particles) {
                    auto fix = [&](const Projection<R>& proj) {
                        if (!proj.is_inside) return;
                        for (int a = 0; a < 3; ++a) { p.velocity[a] += (proj.point[a] - p.position[a]) / dt; p.position[a] = proj.point[a]; }
                    };
                    for (const auto& col : cuboids) fix(cuboid_project_point(col, p.position));
                    for (const auto& hf : heightfields) fix(heightfield_project_point(hf, p.position));
                    for (const auto& tm : trimeshes) fix(trimesh_project_point(tm, p.position));
                    for (const auto& pl : polylines) fix(polyline_project_point(pl, p.position));
                }
                // penalty force of colliders with a stiffness (particle_updater.rs:200-225).  The reference compiles it OUT (`if false`,
                // particle_updater.rs:204): a collider's penalty_stiffness changes nothing there, and nothing here unless `penalty_force` is
                // switched on (the intended rule: a particle inside feels f = (proj - x) k, and the momentum it scatters next is
                // m v + f dt, g2p2g.rs:286-287 -- the kick on the stored velocity, which nothing but the next scatter and the dt estimate
                // reads before G2P overwrites it)
                if (penalty_force) {
                    Vec3<R> f{{0, 0, 0}};
                    bool any = false;
                    auto push = [&](const Projection<R>& proj, const ColliderOpts& o) {
                        if (!proj.is_inside) return;
                        any = true;
                        for (int a = 0; a < 3; ++a) f[a] += (proj.point[a] - p.position[a]) * (R)o.penalty;
                    };
                    for (const auto& col : cuboids) if (col.opts.penalty > 0) push(cuboid_project_point(col, p.position), col.opts);
                    for (const auto& hf : heightfields) if (hf.opts.penalty > 0) push(heightfield_project_point(hf, p.position), hf.opts);
                    for (const auto& tm : trimeshes) if (tm.opts.penalty > 0) push(trimesh_project_point(tm, p.position), tm.opts);
                    for (const auto& pl : polylines) if (pl.opts.penalty > 0) push(polyline_project_point(pl, p.position), pl.opts);
                    if (any) { const R k = dt * inv_exact(p.mass); for (int a = 0; a < 3; ++a) p.velocity[a] += f[a] * k; }
                }
            }
        }
    }

    // ---- fluids_volume.rs:9-115
    void recompute_fluids_volumes() {
        uint64_t pshift[27];
        for (int k = 0; k < 27; ++k) pshift[k] = packed_nbh_shift(k);
        for (uint64_t color = 0; color < 8; ++color) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(num_threads)
            for (size_t r = 0; r < regions.size(); ++r) {
                const Region& reg = regions[r];
                if (!region_has_color(reg.id, color)) continue;
                for (size_t oi = reg.begin; oi < reg.end; ++oi) {
                    const Particle<R>& p = particles[order[oi]];
                    if (p.grid_index == ~0ULL) continue;
                    Vec3<R> d0 = p.dir_to_associated_grid_node(h);
                    R w[3][3];
                    for (int a = 0; a < 3; ++a) eval_all(-d0[a] / h, w[a]);
                    for (int k = 0; k < 27; ++k) {
                        const int* s = NBH_SHIFTS[k];
                        R weight = w[0][s[0]] * w[1][s[1]] * w[2][s[2]];
                        node(packed_add(p.grid_index, pshift[k])).mass += weight * p.mass;
                    }
                }
            }
        }
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (size_t b = 0; b < bins.size(); ++b) {
            for (int ii = 0; ii < 4; ++ii) {
                if (bins[b][ii] == SIZE_MAX) break;
                Particle<R>& p = particles[bins[b][ii]];
                if (p.grid_index == ~0ULL) continue;
                if (!is_fluid(model_of(p))) continue;
                R new_mass = 0;
                Vec3<R> d0 = p.dir_to_associated_grid_node(h);
                R w[3][3];
                for (int a = 0; a < 3; ++a) eval_all(-d0[a] / h, w[a]);
                for (int k = 0; k < 27; ++k) {
                    const int* s = NBH_SHIFTS[k];
                    R weight = w[0][s[0]] * w[1][s[1]] * w[2][s[2]];
                    new_mass += weight * node(packed_add(p.grid_index, pshift[k])).mass;
                }
                R new_density = new_mass / (h * h * h);
                R new_volume = p.mass / new_density;
                p.deformation_gradient(0, 0) = new_volume / p.volume0;
            }
        }
        for (size_t r = 0; r < regions.size(); ++r)
            for (uint64_t i = 0; i < 64; ++i) node(regions[r].id | (i << PACK_ALIGN)).mass = 0;
    }

    // one substep with the CPU ordering; fixed_dt > 0 skips the estimator.  returns dt used.
    R substep(const SparklSolverParams& params, const Vec3<R>& gravity, R remaining_time, R min_dt, R fixed_dt) {
        sort();
        if (params.force_fluids_volume_recomputation) recompute_fluids_volumes();
        R dt;
        if (fixed_dt > R(0)) dt = fixed_dt;
        else {
            dt = adaptive_timestep_length(rust_min(remaining_time, (R)params.max_substep_dt));
            if (dt < min_dt && remaining_time > min_dt) dt = min_dt;
        }
        p2g(dt, gravity);
        grid_update(dt, params.boundary_handling, params.simulation_dofs);
        project_particles = params.enable_boundary_particle_projection != 0;
        damage_model = params.damage_model;
        g2p(dt);
        return dt;
    }

    // ---- MlsSolver::step, mls_solver.rs:15-139
    R step(const SparklSolverParams& params, const Vec3<R>& gravity) {
        last_num_substeps = 0; last_dts.clear();
        if (particles.empty() || params.dt == 0.0f) return (R)params.dt;
        R min_dt = (R)params.dt / (R)params.max_num_substeps;
        R remaining = (R)params.dt;
        while (remaining > R(0)) {
            R dt = substep(params, gravity, remaining, min_dt, R(0));
            ++last_num_substeps; last_dts.push_back((double)dt);
            if (dt != dt) return dt;  // NaN: the reference panics (timestep_estimator.rs:46)
            remaining -= dt;
            if (params.stop_after_one_substep) return dt;
        }
        return (R)params.dt;
    }
};

}  // namespace oracle
