#!/bin/bash
# Builds sparkl_amd/lib/ab_ablate.so: a COPY of csrc/ (under /tmp) with runtime switches patched into k_g2p2g for tools/ablate.py.
# The product sources carry none of this.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
w=/tmp/spk_ablate; rm -rf $w; mkdir -p $w/x/y; cp -r $root/sparkl_amd/csrc $w/x/y/csrc; ln -sfn $root/include $w/x/include
cd $w/x/y/csrc
python3 - <<'EOF'
k = open("kernels.h").read()
def rep(a, b):
    global k
    assert k.count(a) == 1, a
    k = k.replace(a, b)
rep("        const int maxcnt = wave_max_i32(cnt);", "        const int abl = spk_ablate;\n        const int maxcnt = (abl & 4) ? 1 : wave_max_i32(cnt);")
rep("        if (deposit) {\n            const int p = (run_base >> 6) + z;\n            const int n0 = run_base + TILE_W * TILE_W * z + 8 * p - 4 * (p & 1);      // acc_slot of the plane's (0, 0) node",
    "        if (deposit && !(abl & 1)) {\n            const int p = (run_base >> 6) + z;\n            int n0 = run_base + TILE_W * TILE_W * z + 8 * p - 4 * (p & 1);\n            const int sxm = (abl & 2) ? 0 : 1, sym = (abl & 2) ? 0 : TILE_W; if (abl & 2) n0 = lane * 9;")
rep("                    const int n = n0 + sx + TILE_W * sy;\n                    atomicAdd(&tile[0][n], (double)acc01[3 * sy + sx].x);",
    "                    const int n = n0 + sx * sxm + sy * sym + ((abl & 2) ? 3 * sy + sx : 0);\n                    atomicAdd(&tile[0][n], (double)acc01[3 * sy + sx].x);")
rep("            const unsigned long long u0 = vq[row], u1 = vq[row + 1], u2 = vq[row + 2];",
    "            unsigned long long u0, u1, u2; float z0, z1, z2;\n            if (spk_ablate & 8) { u0 = u1 = u2 = 0x3f8000003f800000ull; z0 = z1 = z2 = 0.5f; } else {\n            u0 = vq[row]; u1 = vq[row + 1]; u2 = vq[row + 2]; z0 = vt.z[row]; z1 = vt.z[row + 1]; z2 = vt.z[row + 2]; }")
rep("            const float z0 = vt.z[row], z1 = vt.z[row + 1], z2 = vt.z[row + 2];\n            const float wy = wgt[1][sy];", "            const float wy = wgt[1][sy];")
rep("        if (REDUCE_STEPS < 0) wave_heads += p2g_scatter_runs(tile, sc, live, tid, s_runs[tid >> 6], s_A[tid >> 6]);",
    "        if (REDUCE_STEPS < 0) { if (!(spk_ablate & 16)) wave_heads += p2g_scatter_runs(tile, sc, live, tid, s_runs[tid >> 6], s_A[tid >> 6]); }")
rep("    if (dp_e) {\n        SPK_MARK(\"dp_update\");", "    if (dp_e && (spk_ablate & 128)) { o.T.valid = true; o.T.jm1 = 0.f; sv_valid = true; sv[0] = sv[1] = sv[2] = 0.f; }\n    if (dp_e && !(spk_ablate & 128)) {\n        SPK_MARK(\"dp_update\");")
rep("    psi = fmaxf(psi, model_pos_energy(mo, phase, eh, F, E, sv, sv_valid, jm1));", "    if (!(spk_ablate & 256)) psi = fmaxf(psi, model_pos_energy(mo, phase, eh, F, E, sv, sv_valid, jm1));")
rep("    apic_gather(vt, px, org, S, inv_h, vel, C);\n    SPK_MARK(\"advect_F\");", "    if (spk_ablate & 512) { vel[0] = vel[1] = vel[2] = 0.f; C = m3_zero(); } else apic_gather(vt, px, org, S, inv_h, vel, C);\n    SPK_MARK(\"advect_F\");")
open("kernels.h", "w").write(k)
d = open("device_math.h").read()
assert d.count("namespace spk {") == 1
d = d.replace("namespace spk {", "namespace spk {\n__device__ int spk_ablate = 0;", 1)
n = d.count("__all(series_ok(")
assert n >= 2
d = d.replace("__all(series_ok(lin, lvg))", "(__all(series_ok(lin, lvg)) || (spk_ablate & 64))").replace("__all(series_ok(S))", "(__all(series_ok(S)) || (spk_ablate & 64))").replace("__all(series_ok(S6))", "(__all(series_ok(S6)) || (spk_ablate & 64))")
# counters of the return map's outcomes per wave (sparkl_hip_debug_map_counts): waves, all lanes tensile, any tensile, all inside, lanes moved
d = d.replace("__device__ int spk_ablate = 0;", "__device__ int spk_ablate = 0;\n__device__ unsigned long long spk_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};", 1)
x = "    if (moved) {\n        const float so = tre, sn = tre + 3.f * c;"
assert d.count(x) == 1
d = d.replace(x, """    {
        const bool tens = dn2 == 0.f || tr > 0.f;
        const unsigned long long bt = __ballot(tens), bm = __ballot(moved), ba = __ballot(true);
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
            atomicAdd(&spk_cnt[0], 1ull);
            if (bt == ba) atomicAdd(&spk_cnt[1], 1ull);
            if (bt) atomicAdd(&spk_cnt[2], 1ull);
            if (!bm) atomicAdd(&spk_cnt[3], 1ull);
            atomicAdd(&spk_cnt[4], (unsigned long long)__popcll(bm));
            atomicAdd(&spk_cnt[5], (unsigned long long)__popcll(bt));
            atomicAdd(&spk_cnt[6], (unsigned long long)__popcll(ba));
        }
    }
""" + x)
open("device_math.h", "w").write(d)
s = open("sparkl_hip.hip").read()
s += '\nextern "C" int sparkl_hip_debug_map_counts(unsigned long long* out, int reset) { unsigned long long z[8] = {0,0,0,0,0,0,0,0}; if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spk::spk_cnt), sizeof(z)) != hipSuccess) return -1; if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(spk::spk_cnt), z, sizeof(z)) == hipSuccess ? 0 : -1; return 0; }\n'
s += '\nextern "C" int sparkl_hip_debug_set_ablate(int mode) { return hipMemcpyToSymbol(HIP_SYMBOL(spk::spk_ablate), &mode, sizeof(int)) == hipSuccess ? 0 : -1; }\n'
open("sparkl_hip.hip", "w").write(s)
EOF
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result -o $root/sparkl_amd/lib/ab_ablate.so sparkl_hip.hip
echo built $root/sparkl_amd/lib/ab_ablate.so
