// mfma_moments.hip -- VERDICT r01 item 5: the ten exact moments of a point set through V_MFMA_I32_16X16X64_I8
// instead of 64-bit integer multiply-adds.
//
//   VALU path (what the fit kernels do, pwpp_common.hpp Moments::add_uncounted): per point 3 x (cvt, fma) + clamp
//   = the quantised coordinates, then 9 v_mad_i64_i32: 16 instructions per 64 points of a wave.
//   MFMA path: a point's three 32-bit quantised coordinates ARE its twelve 8-bit limbs; with a thirteenth feature
//   "1" the Gram matrix G = F^T F of the 64 x 16 feature matrix holds every limb cross-sum (exact in int32 for
//   2^10 issues of full-range limbs), i.e. n, S1 and S2 after a recombination with powers of 256.  One MFMA per 64 points -- but the
//   operands want, per lane, ONE feature of SIXTEEN points, while the plane test leaves a lane with ALL features
//   of ONE point: the transposition (3 ds_write_b32, 4 ds_read_b128, 12 v_perm_b32 per wave and 64 points) is
//   what this benchmark prices.  Both paths are checked against each other (exact totals).
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_moments mfma_moments.hip && ./mfma_moments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fxp_q(float v, double scale, double c) {
    return (int)(unsigned)(unsigned long long)__double_as_longlong(__builtin_fma((double)v, scale, c));
}

struct Pt { float x, y, z; };

__device__ __forceinline__ Pt make_pt(unsigned i, unsigned lane, unsigned wave) {
    // deterministic points inside +-30 m of the origin used below
    const unsigned h = (i * 2654435761u) ^ (lane * 40503u) ^ (wave * 2246822519u);
    Pt p;
    p.x = 12.5f + (float)((int)(h & 0xffff) - 32768) * (20.0f / 32768.0f);
    p.y = -7.25f + (float)((int)((h >> 8) & 0xffff) - 32768) * (20.0f / 32768.0f);
    p.z = -1.75f + (float)((int)((h >> 16) & 0xffff) - 32768) * (4.0f / 32768.0f);
    return p;
}

// ---- VALU path ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_valu(long long *out, int steps, float thr) {
    const double scale = 2097152.0, magic = 6755399441055744.0;
    const double cx = magic - 12.5 * scale, cy = magic + 7.25 * scale, cz = magic + 1.75 * scale;
    long long n = 0, s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int i = 0; i < steps; ++i) {
        const Pt p = make_pt((unsigned)i, lane, wave);
        if (p.z < thr) {  // membership (the plane test of the real kernels)
            const int qx = fxp_q(p.x, scale, cx), qy = fxp_q(p.y, scale, cy), qz = fxp_q(p.z, scale, cz);
            n += 1;
            s1[0] += qx; s1[1] += qy; s1[2] += qz;
            s2[0] += (long long)qx * qx; s2[1] += (long long)qx * qy; s2[2] += (long long)qx * qz;
            s2[3] += (long long)qy * qy; s2[4] += (long long)qy * qz; s2[5] += (long long)qz * qz;
        }
    }
    long long v[10] = {n, s1[0], s1[1], s1[2], s2[0], s2[1], s2[2], s2[3], s2[4], s2[5]};
    for (int k = 0; k < 10; ++k) {
        long long t = v[k];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) out[(size_t)wave * 10 + k] = t;
    }
}

// ---- MFMA path ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mfma(long long *out, int steps, float thr) {
    __shared__ int s_q[4][3][64];  // per wave: the quantised coordinates of the 64 points of a step, coordinate-major
    const double scale = 2097152.0, magic = 6755399441055744.0;
    const double cx = magic - 12.5 * scale, cy = magic + 7.25 * scale, cz = magic + 1.75 * scale;
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * 4 + wv;
    const unsigned f = lane & 15, g = lane >> 4;          // operand view: feature f, points 16 g .. 16 g + 15
    const unsigned coord = f >> 2, limb = f & 3;          // features 0..11 = limb `limb` of coordinate `coord`, 12 = ones
    // v_perm_b32 selectors that gather byte `limb` of four dwords: pairwise, then the pair of pairs
    const unsigned sel_pair = 0x0c0c0000u | ((4u + limb) << 8) | limb;      // D = {0, 0, hi.byte[limb], lo.byte[limb]} (S0 = hi dword, S1 = lo dword)
    v4i acc = {0, 0, 0, 0};
    long long total[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto flush = [&]() {
        // D[4 (l / 16) + i][l % 16] is in acc[i] of lane l: through LDS, then lane 0 recombines (rare: every 2048 steps)
        __shared__ int s_g[4][16][16];
        for (int i = 0; i < 4; ++i) s_g[wv][4 * g + i][f] = acc[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            const long long C = 128ll * 65793ll;  // the three low limbs are stored minus 128 (signed bytes): q = q' + C
            auto G = [&](int a, int b) { return (long long)s_g[wv][a][b]; };
            const long long n = G(12, 12);
            long long s1p[3];
            for (int a = 0; a < 3; ++a) { s1p[a] = 0; for (int j = 0; j < 4; ++j) s1p[a] += G(12, 4 * a + j) << (8 * j); }
            total[0] += n;
            for (int a = 0; a < 3; ++a) total[1 + a] += s1p[a] + n * C;
            int m = 0;
            for (int a = 0; a < 3; ++a)
                for (int b = a; b < 3; ++b, ++m) {
                    __int128 s = 0;
                    for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) s += (__int128)G(4 * a + j, 4 * b + k) << (8 * (j + k));
                    s += (__int128)C * (s1p[a] + s1p[b]) + (__int128)n * C * C;
                    total[4 + m] += (long long)s;
                }
        }
        __builtin_amdgcn_wave_barrier();
        acc = (v4i){0, 0, 0, 0};
    };
    for (int i = 0; i < steps; ++i) {
        const Pt p = make_pt((unsigned)i, lane, wave);
        const bool in = p.z < thr;
        // quantise as the VALU path does; the low three bytes minus 128 make every limb a signed byte
        int qx = fxp_q(p.x, scale, cx) ^ 0x00808080, qy = fxp_q(p.y, scale, cy) ^ 0x00808080, qz = fxp_q(p.z, scale, cz) ^ 0x00808080;
        // a point outside the set contributes nothing: all its limbs (and its "1") are zero -- but zero limbs mean q' = 0,
        // not q = 0, which is exactly what "no contribution" needs
        qx = in ? qx : 0; qy = in ? qy : 0; qz = in ? qz : 0;
        s_q[wv][0][lane] = qx; s_q[wv][1][lane] = qy; s_q[wv][2][lane] = qz;
        const unsigned long long member = __ballot(in);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        v4i op;
        if (f < 12) {
            const v4i *src = reinterpret_cast<const v4i *>(&s_q[wv][coord][16 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // sixteen dwords -> their byte `limb`, four per register
                const v4i d = src[r];
                const unsigned lo = __builtin_amdgcn_perm((unsigned)d.y, (unsigned)d.x, sel_pair);  // {0, 0, d1.b, d0.b}
                const unsigned hi = __builtin_amdgcn_perm((unsigned)d.w, (unsigned)d.z, sel_pair);  // {0, 0, d3.b, d2.b}
                op[r] = (int)(lo | (hi << 16));
            }
        } else {  // the ones feature (f == 12): 0x01 for the members among this lane's sixteen points; features 13-15: zero
            const unsigned bits = f == 12 ? (unsigned)(member >> (16 * g)) & 0xffffu : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned b4 = (bits >> (4 * r)) & 0xfu;
                op[r] = (int)(((b4 * 0x00204081u) & 0x01010101u));
            }
        }
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(op, op, acc, 0, 0, 0);
        if ((i & 1023) == 1023) flush();  // 1024 issues x 64 points x 128 x 128 < 2^31
        __builtin_amdgcn_wave_barrier();  // the staging tile is rewritten by the next step
    }
    flush();
    if (lane == 0) for (int k = 0; k < 10; ++k) out[(size_t)wave * 10 + k] = total[k];
}

int main() {
    const int blocks = 1024, waves = blocks * 4, steps = 4096;
    long long *a, *b;
    hipMalloc(&a, (size_t)waves * 10 * 8);
    hipMalloc(&b, (size_t)waves * 10 * 8);
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (float thr : {-1.0f, 10.0f}) {  // about 60 % / all of the points in the set
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, a, 64, thr);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, b, 64, thr);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, a, steps, thr);
        hipEventRecord(e1);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, b, steps, thr);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float ms_a = 0, ms_b = 0;
        hipEventElapsedTime(&ms_a, e0, e1);
        hipEventElapsedTime(&ms_b, e1, e2);
        std::vector<long long> ha((size_t)waves * 10), hb((size_t)waves * 10);
        hipMemcpy(ha.data(), a, ha.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, hb.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
        // 4 waves per SIMD everywhere: wave-steps per SIMD = 4 * steps
        const double per = 1e6 / (4.0 * steps);
        printf("membership %s: VALU path %.3f ms = %.1f ns per 64-point step per SIMD | MFMA path %.3f ms = %.1f ns | totals %s (n of wave 0: %lld, S2zz: %lld)\n",
               thr < 0 ? "~60 %" : "100 %", ms_a, ms_a * per, ms_b, ms_b * per, bad ? "DIFFER" : "identical", ha[0], ha[9]);
    }
    return 0;
}
