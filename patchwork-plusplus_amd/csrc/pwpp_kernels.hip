// pwpp_kernels.hip -- the Patchwork++ estimateGround() hot path as hand-written HIP for
// gfx950 (MI355X, CDNA4: wave64, 256 CUs in 8 XCDs, 160 KiB LDS/CU, HBM3E).
//
// One batch of F independent frames goes through six launches; every launch covers all
// frames (grid.y = frame), so a 1024-frame batch is 6 launches, not 6144:
//
//   K1 k_czm_bin      RNR + CZM code per point + per-frame bin histogram   (ref :377-400, :578-622)
//   K2 k_czm_scan     exclusive scan of the histogram -> bin offsets
//   K3 k_czm_scatter  points grouped by bin: {x,y,z,idx} 16 B records      (ref :602-614 emplace_back)
//   K4 k_patch_fit    per patch: LPR seeds, R-VPF, R-GPF, final plane       (ref :77-149, :47-75, :467-554)
//   K5 k_gle_tgr      per frame: GLE ladder, A-GLE history, TGR, thresholds (ref :211-309, :338-375, :402-464)
//   K6 k_emit         ground / non-ground index lists                       (ref :28-31, :18-26)
//
// All reference citations are /root/reference/cpp/patchworkpp/src/patchworkpp.cpp unless a
// header is named.  This is memory/latency-bound integer + scalar-float work: no MFMA
// anywhere (3x3 covariances), the levers are coalesced 16 B/lane traffic, LDS-staged
// atomics and keeping the per-patch iteration inside one workgroup.
//
// ARITHMETIC CONTRACT.  Everything the reference evaluates in its own
// float/double expressions is evaluated here with the same operations in the same order
// (this file is compiled with -ffp-contract=off; f32 / and sqrt are correctly rounded under
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; f64 always).  The one place the
// reference defers to Eigen -- the mean and covariance sums of estimate_plane (:56-60) -- is
// evaluated in order-independent fixed point: coordinates rounded to a 2^-s m grid, exact
// integer moment sums (int64 / int128), one rounding per output.  Integer sums commute, so
// the result does not depend on thread count, wave scheduling or the order the scatter
// atomics happened to produce, and oracle/pwpp_oracle.cpp (PWO_ARITH_FXP) reproduces it
// bit for bit on the CPU.
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "pwpp_dev.h"

#define PWPP_LAYOUT_ROW_MAJOR 0
#define PWPP_LAYOUT_COL_MAJOR 1

namespace {

constexpr int kBlock = 256;          // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kPtsPerBlock = 1024;   // K1/K3: 4 points per thread, 16 KiB of input per workgroup

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void load_point(const PwppFrameDesc &fd, int i, float &x, float &y, float &z, float &w) {
    if (fd.layout == PWPP_LAYOUT_ROW_MAJOR) {
        if (fd.cols == 4) {
            const float4 v = reinterpret_cast<const float4 *>(fd.pts)[i];  // 16 B/lane, 1 KiB per wave instruction
            x = v.x;
            y = v.y;
            z = v.z;
            w = v.w;
        } else {
            const float *p = fd.pts + (size_t)3 * (size_t)i;
            x = p[0];
            y = p[1];
            z = p[2];
            w = 0.0f;
        }
    } else {  // column-major planes (Eigen::MatrixXf storage)
        const size_t n = (size_t)fd.n;
        x = fd.pts[i];
        y = fd.pts[n + i];
        z = fd.pts[2 * n + i];
        w = fd.cols == 4 ? fd.pts[3 * n + i] : 0.0f;
    }
}

// atan2 for the sector angle (ref xy2theta :568-571).  The CPU reference calls glibc's
// atan2; ocml's differs from it by at most an ulp or two, which can only change
// static_cast<int>(theta / sector_size) when theta/sector_size sits within ~1e-15 of an
// integer.  For float inputs that happens with non-negligible probability only where
// atan2 is an exact rational multiple of pi -- on the axes and the diagonals (Niven) --
// and those do occur in real scans (y == 0, |x| == |y|).  They are answered with the
// correctly rounded constants glibc returns (checked in tests/test_oracle.py).
__device__ __forceinline__ double czm_atan2(double y, double x) {
    const double kPi = 3.14159265358979323846;         // 0x400921FB54442D18
    const double kPi2 = 1.57079632679489661923;        // 0x3FF921FB54442D18
    const double kPi4 = 0.78539816339744830962;        // 0x3FE921FB54442D18
    const double k3Pi4 = 2.35619449019234492885;       // 0x4002D97C7F3321D2
    if (y == 0.0) return signbit(x) ? copysign(kPi, y) : copysign(0.0, y);
    if (x == 0.0) return copysign(kPi2, y);
    if (fabs(x) == fabs(y)) return copysign(x > 0.0 ? kPi4 : k3Pi4, y);
    return atan2(y, x);
}

// ------------------------------------------------------------------------------------------
// K1  RNR + CZM code + histogram
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned czm_code(const PwppDevParams &P, float x, float y, float z, float inten,
                                             bool has_intensity, double sensor_height) {
    const unsigned B = (unsigned)P.num_bins;
    // Reflected Noise Removal, ref :385-396.  r is FLOAT there (:387), the rest double.
    if (P.enable_RNR && has_intensity) {
        const double zd = z;
        // the three conjuncts of :391 are pure; evaluate the cheap two first
        if (zd < -sensor_height - 0.8 && inten < P.RNR_intensity_thr) {
            const float rf = sqrtf(x * x + y * y);
            const double r = rf;
            const double ver_angle_in_deg = atan2(zd, r) * 180 / 3.14159265358979323846;
            if (ver_angle_in_deg < P.RNR_ver_angle_thr) return PWPP_CODE_RNR(B);
        }
    }
    if (z == FLT_MIN) return PWPP_CODE_DROP;  // ref :591 (tombstone value in the input itself)
    // pc2czm, ref :593-615, all double
    const double xd = x, yd = y;
    const double r = sqrt(xd * xd + yd * yd);
    if (!((r <= P.max_range) && (r > P.min_range))) return PWPP_CODE_OOR(B);
    double theta = czm_atan2(yd, xd);
    theta = theta > 0 ? theta : 2 * 3.14159265358979323846 + theta;
    int k;
    if (r < P.min_ranges[1])
        k = 0;
    else if (r < P.min_ranges[2])
        k = 1;
    else if (r < P.min_ranges[3])
        k = 2;
    else
        k = 3;
    const int ring = min(static_cast<int>((r - P.min_ranges[k]) / P.ring_sizes[k]), P.rings[k] - 1);
    const int sector = min(static_cast<int>(theta / P.sector_sizes[k]), P.sectors[k] - 1);
    return (unsigned)(P.bin_base[k] + ring * P.sectors[k] + sector);
}

__global__ __launch_bounds__(kBlock) void k_czm_bin(PwppBatch Bt) {
    __shared__ unsigned s_hist[PWPP_MAX_BINS + 2];
    const int f = blockIdx.y;
    const PwppFrameDesc fd = Bt.frames[f];
    const int first = blockIdx.x * kPtsPerBlock;
    if (first >= fd.n) return;
    const PwppDevParams &P = Bt.P;
    const int NB = P.num_bins + 2;
    for (int b = threadIdx.x; b < NB; b += kBlock) s_hist[b] = 0;
    __syncthreads();
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    uint16_t *codes = Bt.codes + fd.base;
    unsigned dropped = 0;
#pragma unroll
    for (int j = 0; j < kPtsPerBlock / kBlock; ++j) {
        const int i = first + j * kBlock + threadIdx.x;
        if (i < fd.n) {
            float x, y, z, w;
            load_point(fd, i, x, y, z, w);
            const unsigned code = czm_code(P, x, y, z, w, fd.cols >= 4, sensor_height);
            codes[i] = (uint16_t)code;
            if (code == PWPP_CODE_DROP)
                ++dropped;
            else
                atomicAdd(&s_hist[code], 1u);
        }
    }
    __syncthreads();
    unsigned *gcount = Bt.bin_count + (size_t)f * NB;
    for (int b = threadIdx.x; b < NB; b += kBlock) {
        const unsigned c = s_hist[b];
        if (c) atomicAdd(&gcount[b], c);  // one global atomic per non-empty bin per 1024 points
    }
    dropped = wave_sum_u32(dropped);
    if (lane_id() == 0 && dropped) atomicAdd((unsigned *)&Bt.results[f].n_dropped, dropped);
}

// ------------------------------------------------------------------------------------------
// K2  exclusive scan of the per-frame histogram
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_czm_scan(PwppBatch Bt) {
    __shared__ unsigned s_part[kBlock];
    const int f = blockIdx.x;
    const int NB = Bt.P.num_bins + 2;
    const unsigned *cnt = Bt.bin_count + (size_t)f * NB;
    unsigned *off = Bt.bin_off + (size_t)f * NB;
    constexpr int kPer = (PWPP_MAX_BINS + 2 + kBlock - 1) / kBlock;
    unsigned local[kPer];
    unsigned sum = 0;
    const int b0 = threadIdx.x * kPer;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int b = b0 + j;
        local[j] = b < NB ? cnt[b] : 0u;
        sum += local[j];
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele over 256 partials
    for (int o = 1; o < kBlock; o <<= 1) {
        const unsigned v = threadIdx.x >= (unsigned)o ? s_part[threadIdx.x - o] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = s_part[threadIdx.x] - sum;  // exclusive
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int b = b0 + j;
        if (b < NB) off[b] = run;
        run += local[j];
    }
    if (threadIdx.x == 0) {
        Bt.results[f].n_rnr = (int)cnt[Bt.P.num_bins];
        Bt.results[f].n_oor = (int)cnt[Bt.P.num_bins + 1];
    }
}

// ------------------------------------------------------------------------------------------
// K3  scatter into bin order
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_czm_scatter(PwppBatch Bt) {
    __shared__ unsigned s_cnt[PWPP_MAX_BINS + 2];   // points of this block per bin
    __shared__ unsigned s_base[PWPP_MAX_BINS + 2];  // first slot reserved for this block in that bin
    const int f = blockIdx.y;
    const PwppFrameDesc fd = Bt.frames[f];
    const int first = blockIdx.x * kPtsPerBlock;
    if (first >= fd.n) return;
    const int NB = Bt.P.num_bins + 2;
    for (int b = threadIdx.x; b < NB; b += kBlock) s_cnt[b] = 0;
    __syncthreads();
    const uint16_t *codes = Bt.codes + fd.base;
    constexpr int kPer = kPtsPerBlock / kBlock;
    unsigned code[kPer], rank[kPer];
    float4 pt[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = first + j * kBlock + threadIdx.x;
        code[j] = PWPP_CODE_DROP;
        if (i < fd.n) {
            code[j] = codes[i];
            float x, y, z, w;
            load_point(fd, i, x, y, z, w);
            pt[j] = make_float4(x, y, z, __int_as_float(i));
            if (code[j] != PWPP_CODE_DROP) rank[j] = atomicAdd(&s_cnt[code[j]], 1u);
        }
    }
    __syncthreads();
    unsigned *cursor = Bt.bin_cursor + (size_t)f * NB;
    for (int b = threadIdx.x; b < NB; b += kBlock) {
        const unsigned c = s_cnt[b];
        if (c) s_base[b] = atomicAdd(&cursor[b], c);  // reserve a contiguous range in the bin
    }
    __syncthreads();
    const unsigned *off = Bt.bin_off + (size_t)f * NB;
    float4 *sorted = Bt.sorted + fd.base;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        if (code[j] != PWPP_CODE_DROP) sorted[off[code[j]] + s_base[code[j]] + rank[j]] = pt[j];
    }
}

// ------------------------------------------------------------------------------------------
// K4  per-patch plane fitting
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double i128_to_double(__int128 v) {  // one rounding, to nearest even
    const bool neg = v < 0;
    const unsigned __int128 a = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const unsigned long long hi = (unsigned long long)(a >> 64), lo = (unsigned long long)a;
    double r;
    if (hi == 0) {
        r = (double)lo;
    } else {
        const int sh = 64 - __clzll((long long)hi);  // bits above bit 63
        unsigned long long top = (unsigned long long)(a >> sh);
        const unsigned __int128 rest = a & ((((unsigned __int128)1) << sh) - 1);
        top |= (rest != 0) ? 1ull : 0ull;  // sticky bit, far below the 53-bit mantissa
        r = ldexp((double)top, sh);
    }
    return neg ? -r : r;
}

__device__ __forceinline__ float f_abs(float v) { return v < 0.0f ? -v : v; }
__device__ __forceinline__ float f_max(float a, float b) { return a < b ? b : a; }

// Eigen 3.4.0 JacobiSVD<MatrixX3f>(cov, ComputeFullU) as used at ref :62 -- two-sided Jacobi,
// real square case, float.  a: row-major symmetric 3x3.  Outputs U (row-major) and the
// singular values sorted descending.  Same operation sequence as oracle/pwpp_oracle.cpp
// jacobi_svd3 and oracle/eigen_shim (all three are compared bitwise by the tests).
__device__ void jacobi_svd3(const float a[9], float u[9], float sv[3]) {
    const float tiny = FLT_MIN, precision = 2.0f * FLT_EPSILON;
    float scale = 0.0f;
    bool invalid = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float v = f_abs(a[k]);
        if (!(v == v) || v > FLT_MAX) invalid = true;
        if (v > scale) scale = v;
    }
    if (invalid) {
        const float nanv = __uint_as_float(0x7fc00000u);
#pragma unroll
        for (int k = 0; k < 9; ++k) u[k] = nanv;
        sv[0] = sv[1] = sv[2] = nanv;
        return;
    }
    if (scale == 0.0f) scale = 1.0f;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a[k] / scale;
#pragma unroll
    for (int k = 0; k < 9; ++k) u[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    float max_diag = f_max(f_abs(w[0]), f_max(f_abs(w[4]), f_abs(w[8])));

    for (int sweep = 0; sweep < 1000; ++sweep) {
        bool finished = true;
#pragma unroll
        for (int p = 1; p < 3; ++p) {
#pragma unroll
            for (int q = 0; q < p; ++q) {
                const float thr = f_max(tiny, precision * max_diag);
                if (f_abs(w[p * 3 + q]) > thr || f_abs(w[q * 3 + p]) > thr) {
                    finished = false;
                    const float m00 = w[p * 3 + p], m01 = w[p * 3 + q], m10 = w[q * 3 + p], m11 = w[q * 3 + q];
                    const float t = m00 + m11, d = m10 - m01;
                    float c1, s1;
                    if (f_abs(d) < tiny) {
                        s1 = 0.0f;
                        c1 = 1.0f;
                    } else {
                        const float r = t / d;
                        const float h = sqrtf(1.0f + r * r);
                        s1 = 1.0f / h;
                        c1 = r / h;
                    }
                    const float b00 = c1 * m00 + s1 * m10;
                    const float b01 = c1 * m01 + s1 * m11;
                    const float b11 = -s1 * m01 + c1 * m11;
                    float cr, sr;
                    const float deno = 2.0f * f_abs(b01);
                    if (deno < tiny) {
                        cr = 1.0f;
                        sr = 0.0f;
                    } else {
                        const float tau = (b00 - b11) / deno;
                        const float ww = sqrtf(tau * tau + 1.0f);
                        const float tt = (tau > 0.0f) ? 1.0f / (tau + ww) : 1.0f / (tau - ww);
                        const float sign_t = tt > 0.0f ? 1.0f : -1.0f;
                        const float nn = 1.0f / sqrtf(tt * tt + 1.0f);
                        sr = -sign_t * (b01 / f_abs(b01)) * f_abs(tt) * nn;
                        cr = nn;
                    }
                    const float cl = c1 * cr - s1 * (-sr);
                    const float sl = c1 * (-sr) + s1 * cr;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float x = w[p * 3 + k], y = w[q * 3 + k];
                        w[p * 3 + k] = cl * x + sl * y;
                        w[q * 3 + k] = -sl * x + cl * y;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float x = u[k * 3 + p], y = u[k * 3 + q];
                        u[k * 3 + p] = cl * x - (-sl) * y;
                        u[k * 3 + q] = (-sl) * x + cl * y;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float x = w[k * 3 + p], y = w[k * 3 + q];
                        w[k * 3 + p] = cr * x - sr * y;
                        w[k * 3 + q] = sr * x + cr * y;
                    }
                    max_diag = f_max(max_diag, f_max(f_abs(w[p * 3 + p]), f_abs(w[q * 3 + q])));
                }
            }
        }
        if (finished) break;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float dgl = w[i * 3 + i];
        sv[i] = f_abs(dgl);
        if (dgl < 0.0f) {
#pragma unroll
            for (int k = 0; k < 3; ++k) u[k * 3 + i] = -u[k * 3 + i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    bool stop = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!stop) {
            int pos = i;
#pragma unroll
            for (int j = i + 1; j < 3; ++j)
                if (sv[j] > sv[pos]) pos = j;
            if (sv[pos] == 0.0f) {
                stop = true;
            } else if (pos != i) {
                const float ts = sv[i];
                sv[i] = sv[pos];
                sv[pos] = ts;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float tu = u[k * 3 + i];
                    u[k * 3 + i] = u[k * 3 + pos];
                    u[k * 3 + pos] = tu;
                }
            }
        }
    }
}

// Q(v)
__device__ __forceinline__ int fxp_quantise(float v, float scale) {
    float t = v * scale;
    if (!(t == t)) return 0;
    t = rintf(t);
    if (t > 8388607.0f) t = 8388607.0f;
    if (t < -8388607.0f) t = -8388607.0f;
    return (int)t;
}

// order-preserving map float -> uint32 (for the lowest-point selection)
__device__ __forceinline__ unsigned z_key(float z) {
    const unsigned b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_z(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct Moments {  // per-lane partial sums of the quantised coordinates
    long long n, s1[3], s2[6];
    __device__ __forceinline__ void clear() {
        n = 0;
        s1[0] = s1[1] = s1[2] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s2[k] = 0;
    }
    __device__ __forceinline__ void add(float x, float y, float z, float scale) {
        const int qx = fxp_quantise(x, scale), qy = fxp_quantise(y, scale), qz = fxp_quantise(z, scale);
        n += 1;
        s1[0] += qx;
        s1[1] += qy;
        s1[2] += qz;
        s2[0] += (long long)qx * qx;
        s2[1] += (long long)qx * qy;
        s2[2] += (long long)qx * qz;
        s2[3] += (long long)qy * qy;
        s2[4] += (long long)qy * qz;
        s2[5] += (long long)qz * qz;
    }
};

struct FitShared {
    long long part[kWaves][16];
    float normal[3];
    float mean[3];
    float sv[3];
    float pad_;
    double d;
    double lpr;
    unsigned hist[256];
    unsigned sel_keys[PWPP_MAX_LPR];
    unsigned sel_sorted[PWPP_MAX_LPR];
    unsigned sel_count;
    unsigned prefix;
    unsigned krem;
    unsigned keff;
    unsigned cnt_g;
    unsigned cnt_ng;
};

// Block-wide sum of the moments and, if the set is non-empty, the plane of ref :47-75.
// An empty set leaves the previous plane in force, as ref :49 does.
// `wide`: bins above 65536 points could overflow an int64 second moment in the cross-lane
// sum; they are reduced as two 32-bit limbs and recombined in 128 bits (exact either way).
__device__ void reduce_and_fit(FitShared &sh, const Moments &m, bool wide, int shift, int debug = 0) {
    long long v[16];
    v[0] = m.n;
    v[1] = m.s1[0];
    v[2] = m.s1[1];
    v[3] = m.s1[2];
    const int nv = wide ? 16 : 10;
    if (!wide) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[4 + k] = m.s2[k];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            v[4 + k] = m.s2[k] & 0xffffffffll;
            v[10 + k] = m.s2[k] >> 32;
        }
    }
    const int wv = wave_id(), ln = lane_id();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < nv) {
            const long long t = wave_sum_i64(v[k]);
            if (ln == 0) sh.part[wv][k] = t;
        }
    }
    __syncthreads();
    if (wv == 0) {
        long long t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            t[k] = 0;
            if (k < nv) {
#pragma unroll
                for (int q = 0; q < kWaves; ++q) t[k] += sh.part[q][k];
            }
        }
        const long long n = t[0];
        if (n > 0) {
            __int128 s2[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) s2[k] = wide ? ((__int128)t[10 + k] * (__int128)4294967296ll + (__int128)t[4 + k]) : (__int128)t[4 + k];
            const long long s1[3] = {t[1], t[2], t[3]};
            const double inv = 1.0 / (double)(1 << shift);
            const double den = (double)n * (double)(n - 1);
            float mean[3], cov[9];
#pragma unroll
            for (int a = 0; a < 3; ++a) mean[a] = (float)(((double)s1[a] / (double)n) * inv);
            const int map[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = a; b < 3; ++b) {
                    const __int128 num = (__int128)n * s2[map[a * 3 + b]] - (__int128)s1[a] * (__int128)s1[b];
                    const float c = (float)((i128_to_double(num) / den) * (inv * inv));
                    cov[a * 3 + b] = c;
                    cov[b * 3 + a] = c;
                }
            }
            float u[9], sv[3];
            if (debug & 1) {  // ablation: no eigen-solve
                for (int k = 0; k < 9; ++k) u[k] = cov[k];
                sv[0] = cov[0]; sv[1] = cov[4]; sv[2] = cov[8];
                u[2] = 0.01f; u[5] = 0.01f; u[8] = 0.9999f;
            } else {
                jacobi_svd3(cov, u, sv);
            }
            float nx = u[2], ny = u[5], nz = u[8];  // U.col(2), ref :66
            if (nz < 0) {                           // ref :68
                nx *= -1;
                ny *= -1;
                nz *= -1;
            }
            const float dot = nx * mean[0] + ny * mean[1] + nz * mean[2];  // ref :74, float dot
            if (ln == 0) {
                sh.normal[0] = nx;
                sh.normal[1] = ny;
                sh.normal[2] = nz;
                sh.mean[0] = mean[0];
                sh.mean[1] = mean[1];
                sh.mean[2] = mean[2];
                sh.sv[0] = sv[0];
                sh.sv[1] = sv[1];
                sh.sv[2] = sv[2];
                sh.d = -dot;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ bool pt_stripped(const float4 &p) { return (__float_as_uint(p.w) & 0x80000000u) != 0; }

// Lowest-point representative height, ref :84-103, without sorting the bin: the reference
// needs (a) how many points lie below the adaptive cut-off (zone 0 only, :88-96), (b) the
// num_lpr smallest z among the others, summed in ascending order in double (:99-102).
// A 4-pass 8-bit radix select finds the k-th smallest key; the elements below its 24-bit
// prefix are gathered in the last pass, the rest is implied by the last histogram.
__device__ double block_lpr(FitShared &sh, const float4 *pts, unsigned n, bool use_cutoff, double cutoff, int num_lpr) {
    const int ln = lane_id(), wv = wave_id();
    unsigned prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int bits = 24 - 8 * pass;
        sh.hist[threadIdx.x] = 0;  // kBlock == 256 counters
        if (threadIdx.x == 0 && pass == 3) sh.sel_count = 0;
        __syncthreads();
        for (unsigned i = threadIdx.x; i < n; i += kBlock) {
            const float4 p = pts[i];
            if (pt_stripped(p)) continue;
            if (use_cutoff && (double)p.z < cutoff) continue;  // init_idx prefix, ref :88-96
            const unsigned key = z_key(p.z);
            if (pass > 0) {
                const unsigned hp = key >> (bits + 8);
                if (hp != prefix) {
                    if (pass == 3 && hp < prefix) {
                        const unsigned s = atomicAdd(&sh.sel_count, 1u);
                        if (s < PWPP_MAX_LPR) sh.sel_keys[s] = key;
                    }
                    continue;
                }
            }
            atomicAdd(&sh.hist[(key >> bits) & 255u], 1u);
        }
        __syncthreads();
        if (wv == 0) {
            const unsigned c0 = sh.hist[4 * ln], c1 = sh.hist[4 * ln + 1], c2 = sh.hist[4 * ln + 2], c3 = sh.hist[4 * ln + 3];
            const unsigned s = c0 + c1 + c2 + c3;
            unsigned incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o, 64);
                if (ln >= o) incl += t;
            }
            unsigned kk;
            if (pass == 0) {
                const unsigned total = __shfl(incl, 63, 64);
                const unsigned keff = total < (unsigned)num_lpr ? total : (unsigned)num_lpr;
                if (ln == 0) sh.keff = keff;
                kk = keff;
            } else {
                kk = sh.krem;
            }
            const unsigned excl = incl - s;
            if (kk >= 1 && excl < kk && kk <= incl) {  // exactly one lane
                unsigned run = excl, dgt = 4 * ln;
                if (run + c0 >= kk) {
                } else {
                    run += c0;
                    ++dgt;
                    if (run + c1 >= kk) {
                    } else {
                        run += c1;
                        ++dgt;
                        if (run + c2 >= kk) {
                        } else {
                            run += c2;
                            ++dgt;
                        }
                    }
                }
                sh.prefix = (prefix << 8) | dgt;
                sh.krem = kk - run;  // rank inside the chosen bucket, 1-based
            }
        }
        __syncthreads();
        if (sh.keff == 0) return 0.0;  // ref :103 "in case divide by 0"
        prefix = sh.prefix;
    }
    // wave 0: order the gathered keys (all below the last bucket) and add up, ascending
    if (wv == 0) {
        const unsigned c = sh.sel_count;  // < keff <= PWPP_MAX_LPR
        if ((unsigned)ln < c) {
            const unsigned mine = sh.sel_keys[ln];
            unsigned rank = 0;
            for (unsigned j = 0; j < c; ++j) {
                const unsigned o = sh.sel_keys[j];
                rank += (o < mine || (o == mine && j < (unsigned)ln)) ? 1u : 0u;
            }
            sh.sel_sorted[rank] = mine;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ln == 0) {
            const unsigned keff = sh.keff;
            double sum = 0;
            for (unsigned j = 0; j < c; ++j) sum += key_z(sh.sel_sorted[j]);
            unsigned r = keff - c;
            const unsigned p24 = prefix >> 8;
            for (unsigned dgt = 0; dgt < 256 && r > 0; ++dgt) {
                unsigned m = sh.hist[dgt];
                if (m > r) m = r;
                const double z = key_z((p24 << 8) | dgt);
                for (unsigned t = 0; t < m; ++t) sum += z;
                r -= m;
            }
            sh.lpr = sum / (int)keff;  // ref :103
        }
    }
    __syncthreads();
    return sh.lpr;
}

// ref :551-554  (float products, float adds left to right, one double add)
__device__ __forceinline__ double point_to_plane(float nx, float ny, float nz, double d, const float4 &p) {
    return nx * p.x + ny * p.y + nz * p.z + d;
}

__global__ __launch_bounds__(kBlock) void k_patch_fit(PwppBatch Bt) {
    __shared__ FitShared sh;
    const int f = blockIdx.y, bin = blockIdx.x;
    const PwppDevParams &P = Bt.P;
    const int NB = P.num_bins + 2;
    const unsigned n = Bt.bin_count[(size_t)f * NB + bin];
    if ((uint64_t)n < P.min_pts) return;  // small bin: all non-ground, handled by K6 (ref :191-195)
    PwppPatchRec *rec = Bt.recs + (size_t)f * P.num_bins + bin;
    if (n == 0) {  // only reachable with num_min_pts <= 0: no fit runs (ref :49), K5 inherits the previous plane
        if (threadIdx.x == 0) {
            rec->valid = 0;
            rec->n_points = 0;
            rec->n_ground = 0;
            rec->n_nonground = 0;
        }
        return;
    }
    const PwppFrameDesc fd = Bt.frames[f];
    const unsigned off = Bt.bin_off[(size_t)f * NB + bin];
    float4 *pts = Bt.sorted + fd.base + off;
    int *plist = Bt.plist + fd.base + off;
    const int zone = bin < P.bin_base[1] ? 0 : (bin < P.bin_base[2] ? 1 : (bin < P.bin_base[3] ? 2 : 3));
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const double cutoff = P.margin * sensor_height;  // ref :90
    const bool use_cutoff = zone == 0;
    const float qscale = (float)(1 << P.fxp_shift);
    const bool wide = n > 65536u;

    if (threadIdx.x == 0) {
        sh.normal[0] = sh.normal[1] = sh.normal[2] = 0.0f;
        sh.mean[0] = sh.mean[1] = sh.mean[2] = 0.0f;
        sh.sv[0] = sh.sv[1] = sh.sv[2] = 0.0f;
        sh.d = 0.0;
        sh.cnt_g = 0;
        sh.cnt_ng = 0;
    }
    __syncthreads();

    double lpr = 0.0;
    bool lpr_valid = false;
    Moments m;

    // ---- R-VPF, ref :482-508
    if (P.enable_RVPF) {
        for (int it = 0; it < P.num_iter; ++it) {
            if (!lpr_valid) {
                lpr = (Bt.debug & 2) ? -1.8 : block_lpr(sh, pts, n, use_cutoff, cutoff, P.num_lpr);
                lpr_valid = true;
            }
            const double thr = lpr + P.th_seeds_v;  // ref :108
            m.clear();
            for (unsigned i = threadIdx.x; i < n; i += kBlock) {
                const float4 p = pts[i];
                if (!pt_stripped(p) && (double)p.z < thr) m.add(p.x, p.y, p.z, qscale);
            }
            reduce_and_fit(sh, m, wide, P.fxp_shift, Bt.debug);
            const float nx = sh.normal[0], ny = sh.normal[1], nz = sh.normal[2];
            const double d = sh.d;
            if (zone == 0 && (double)nz < P.uprightness_thr) {  // ref :489
                int any = 0;
                for (unsigned i = threadIdx.x; i < n; i += kBlock) {
                    float4 p = pts[i];
                    if (pt_stripped(p)) continue;
                    const double dist = point_to_plane(nx, ny, nz, d, p);
                    if (fabs(dist) < P.th_dist_v) {  // ref :499 -> non_ground_dst
                        reinterpret_cast<unsigned *>(pts)[(size_t)i * 4 + 3] = __float_as_uint(p.w) | 0x80000000u;
                        any = 1;
                    }
                }
                if (__syncthreads_or(any)) lpr_valid = false;  // the working set changed
            } else {
                break;  // ref :506
            }
        }
    }

    // ---- R-GPF, ref :513-543
    if (!lpr_valid) lpr = (Bt.debug & 2) ? -1.8 : block_lpr(sh, pts, n, use_cutoff, cutoff, P.num_lpr);
    {
        const double thr = lpr + P.th_seeds;  // ref :145
        m.clear();
        for (unsigned i = threadIdx.x; i < n; i += kBlock) {
            const float4 p = pts[i];
            if (!pt_stripped(p) && (double)p.z < thr) m.add(p.x, p.y, p.z, qscale);
        }
        reduce_and_fit(sh, m, wide, P.fxp_shift, Bt.debug);
    }
    const int ln = lane_id();
    for (int it = 0; it < P.num_iter; ++it) {
        const bool last = it == P.num_iter - 1;
        const float nx = sh.normal[0], ny = sh.normal[1], nz = sh.normal[2];
        const double d = sh.d;
        m.clear();
        for (unsigned i0 = 0; i0 < n; i0 += kBlock) {
            const unsigned i = i0 + threadIdx.x;
            const bool in = i < n;
            float4 p = make_float4(0, 0, 0, 0);
            if (in) p = pts[i];
            const bool stripped = in && pt_stripped(p);
            const bool active = in && !stripped;
            bool g = false;
            if (active) {
                const double dist = point_to_plane(nx, ny, nz, d, p);
                g = dist < P.th_dist;  // ref :525,529 (one-sided)
            }
            if (g) m.add(p.x, p.y, p.z, qscale);
            if (last) {
                // regionwise_ground_ from the front, regionwise_nonground_ (R-VPF strips included,
                // ref :500,532) from the back of this patch's slot range
                const int idx = (int)(__float_as_uint(p.w) & 0x7fffffffu);
                const unsigned long long mg = __ballot(g);
                const unsigned long long mn = __ballot(in && !g);
                const unsigned long long lt = (1ull << ln) - 1ull;
                unsigned bg = 0, bn = 0;
                if (ln == 0) {
                    if (mg) bg = atomicAdd(&sh.cnt_g, (unsigned)__popcll(mg));
                    if (mn) bn = atomicAdd(&sh.cnt_ng, (unsigned)__popcll(mn));
                }
                bg = __shfl(bg, 0, 64);
                bn = __shfl(bn, 0, 64);
                if (g)
                    plist[bg + (unsigned)__popcll(mg & lt)] = idx;
                else if (in)
                    plist[n - 1u - (bn + (unsigned)__popcll(mn & lt))] = idx;
            }
        }
        reduce_and_fit(sh, m, wide, P.fxp_shift, Bt.debug);  // ref :537-542
    }

    if (threadIdx.x == 0) {
        rec->mean[0] = sh.mean[0];
        rec->mean[1] = sh.mean[1];
        rec->mean[2] = sh.mean[2];
        rec->normal[0] = sh.normal[0];
        rec->normal[1] = sh.normal[1];
        rec->normal[2] = sh.normal[2];
        rec->sv[0] = sh.sv[0];
        rec->sv[1] = sh.sv[1];
        rec->sv[2] = sh.sv[2];
        rec->d = sh.d;
        rec->n_points = (int)n;
        rec->n_ground = (int)sh.cnt_g;
        rec->n_nonground = (int)(n - sh.cnt_g);
        rec->decision = 0;
        rec->valid = 1;
    }
}

// ------------------------------------------------------------------------------------------
// K5  GLE + A-GLE history + TGR + adaptive thresholds, one workgroup (one wave) per frame.
// v1: the reference's sequential loop (ref :184-311) executed by lane 0 in the reference's
// own order, so every double sum has the reference's summation order.
// ------------------------------------------------------------------------------------------
__device__ void mean_stdev(const double *v, int n, double &mean, double &stdev) {  // ref :557-566
    if (n <= 1) return;
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += v[i];
    mean = acc / n;
    for (int i = 0; i < n; ++i) stdev += (v[i] - mean) * (v[i] - mean);
    stdev /= n - 1;
    stdev = sqrt(stdev);
}

__global__ __launch_bounds__(64) void k_gle_tgr(PwppBatch Bt) {
    __shared__ double s_ring_flat[PWPP_MAX_NEAR_BINS];
    __shared__ uint8_t s_dec[PWPP_MAX_BINS];
    const int f = blockIdx.x;
    if (threadIdx.x != 0) return;
    const PwppDevParams &P = Bt.P;
    const int B = P.num_bins, NB = B + 2;
    const PwppFrameDesc fd = Bt.frames[f];
    const unsigned *cnt = Bt.bin_count + (size_t)f * NB;
    PwppPatchRec *recs = Bt.recs + (size_t)f * B;
    unsigned *dst_a = Bt.dst_a + (size_t)f * NB;
    unsigned *dst_b = Bt.dst_b + (size_t)f * NB;
    float *centers = Bt.centers + (size_t)f * B * 3;
    float *normals = Bt.normals + (size_t)f * B * 3;

    // adaptive state this frame reads (ref: params_ members mutated by :347-350,368)
    PwppStateScalar st;
    if (fd.state_in >= 0) {
        st = Bt.st_scalar[fd.state_in];
    } else {
        st.sensor_height = P.sensor_height;
        for (int k = 0; k < 4; ++k) {
            st.elevation_thr[k] = P.elevation_thr0[k];
            st.flatness_thr[k] = P.flatness_thr0[k];
            st.elev_len[k] = 0;
            st.flat_len[k] = 0;
        }
    }
    double *hist_out = Bt.st_hist + (size_t)fd.state_out * 8 * P.hist_cap;
    if (fd.state_in >= 0 && fd.state_in != fd.state_out) {
        const double *hist_in = Bt.st_hist + (size_t)fd.state_in * 8 * P.hist_cap;
        for (int k = 0; k < 4; ++k) {
            for (int i = 0; i < st.elev_len[k]; ++i) hist_out[(0 * 4 + k) * P.hist_cap + i] = hist_in[(0 * 4 + k) * P.hist_cap + i];
            for (int i = 0; i < st.flat_len[k]; ++i) hist_out[(1 * 4 + k) * P.hist_cap + i] = hist_in[(1 * 4 + k) * P.hist_cap + i];
        }
    }

    // ---- pass 1: decisions in traversal order (ref :184-311)
    int concentric_idx = 0;
    int n_ring_flat = 0;  // ringwise_flatness: only cleared when a ring had candidates (ref :292-304)
    int n_patches = 0;
    PwppPatchRec prev;    // plane members persist across bins in the reference (stale-plane quirk)
    prev.mean[0] = prev.mean[1] = prev.mean[2] = 0.0f;
    prev.normal[0] = prev.normal[1] = prev.normal[2] = 0.0f;
    prev.sv[0] = prev.sv[1] = prev.sv[2] = 0.0f;
    prev.d = 0.0;
    unsigned total_ground = 0;
    for (int zone = 0; zone < 4; ++zone) {
        for (int ring = 0; ring < P.rings[zone]; ++ring) {
            const int b0 = P.bin_base[zone] + ring * P.sectors[zone];
            int n_cand = 0;
            for (int sector = 0; sector < P.sectors[zone]; ++sector) {
                const int bin = b0 + sector;
                const unsigned n = cnt[bin];
                if ((uint64_t)n < P.min_pts) {
                    s_dec[bin] = 0;  // small bin
                    continue;
                }
                PwppPatchRec r = recs[bin];
                if (!r.valid) {  // no fit ran: the previous plane is still in the members
                    for (int i = 0; i < 3; ++i) {
                        r.mean[i] = prev.mean[i];
                        r.normal[i] = prev.normal[i];
                        r.sv[i] = prev.sv[i];
                    }
                    r.d = prev.d;
                }
                prev = r;
                for (int i = 0; i < 3; ++i) {  // ref :211-212
                    centers[n_patches * 3 + i] = r.mean[i];
                    normals[n_patches * 3 + i] = r.normal[i];
                }
                ++n_patches;
                // ref :217-223
                const double uprightness = r.normal[2];
                const double elevation = r.mean[2];
                float fmin3 = r.sv[0];  // minCoeff(): first minimum, NaN-transparent like std::min_element
                if (r.sv[1] < fmin3) fmin3 = r.sv[1];
                if (r.sv[2] < fmin3) fmin3 = r.sv[2];
                const double flatness = fmin3;
                double heading = 0.0;
                for (int i = 0; i < 3; ++i) heading += r.mean[i] * r.normal[i];  // float product, double sum
                const bool is_upright = uprightness > P.uprightness_thr;
                const bool is_near = concentric_idx < P.num_rings_of_interest;
                const bool heading_outside = heading < 0.0;
                bool not_elevated = false, is_flat = false;
                if (is_near) {
                    not_elevated = elevation < st.elevation_thr[concentric_idx];
                    is_flat = flatness < st.flatness_thr[concentric_idx];
                }
                if (is_upright && not_elevated && is_near) {  // ref :253-259
                    // (the reference's vectors are unbounded; the slabs are sized so that this guard
                    // only trips in the pathological un-trimmed case of ref :363-364, then flagged)
                    if (st.elev_len[concentric_idx] < P.hist_cap && st.flat_len[concentric_idx] < P.hist_cap) {
                        hist_out[(0 * 4 + concentric_idx) * P.hist_cap + st.elev_len[concentric_idx]++] = elevation;
                        hist_out[(1 * 4 + concentric_idx) * P.hist_cap + st.flat_len[concentric_idx]++] = flatness;
                    } else {
                        Bt.results[f].pad0 = 1;
                    }
                    s_ring_flat[n_ring_flat++] = flatness;
                }
                int dec;
                if (!is_upright)
                    dec = 1;
                else if (!is_near)
                    dec = 2;
                else if (!heading_outside)
                    dec = 3;
                else if (not_elevated || is_flat)
                    dec = 4;
                else {
                    dec = 5;  // candidate; settled at the end of the ring
                    ++n_cand;
                }
                s_dec[bin] = (uint8_t)dec;
                recs[bin].decision = dec;
                recs[bin].mean[0] = r.mean[0];  // (only differs for inherited planes)
                recs[bin].mean[1] = r.mean[1];
                recs[bin].mean[2] = r.mean[2];
                recs[bin].normal[0] = r.normal[0];
                recs[bin].normal[1] = r.normal[1];
                recs[bin].normal[2] = r.normal[2];
                recs[bin].sv[0] = r.sv[0];
                recs[bin].sv[1] = r.sv[1];
                recs[bin].sv[2] = r.sv[2];
                recs[bin].d = r.d;
                if (dec == 2 || dec == 4) total_ground += (unsigned)r.n_ground;
            }
            if (n_cand > 0) {  // ref :292-304
                if (P.enable_TGR) {
                    double mean_flatness = 0.0, stdev_flatness = 0.0;  // ref :407-408
                    mean_stdev(s_ring_flat, n_ring_flat, mean_flatness, stdev_flatness);
                    for (int sector = 0; sector < P.sectors[zone]; ++sector) {
                        const int bin = b0 + sector;
                        if (s_dec[bin] != 5) continue;
                        const PwppPatchRec r = recs[bin];
                        float fmin3 = r.sv[0];
                        if (r.sv[1] < fmin3) fmin3 = r.sv[1];
                        if (r.sv[2] < fmin3) fmin3 = r.sv[2];
                        const double flatness = fmin3;
                        const double line_variable = r.sv[1] != 0 ? (double)(r.sv[0] / r.sv[1]) : DBL_MAX;
                        const double mu = mean_flatness + 1.5 * stdev_flatness;              // ref :428
                        double prob_flatness = 1 / (1 + exp((flatness - mu) / (mu / 10)));  // ref :429
                        if (r.n_ground > 1500 && flatness < P.th_dist * P.th_dist) prob_flatness = 1.0;  // ref :431
                        double prob_line = 1.0;
                        if (line_variable > 8.0) prob_line = 0.0;  // ref :434-438
                        const bool revert = prob_line * prob_flatness > 0.5;
                        // ref :442 guard is always true for candidates (they only arise in near rings)
                        if (revert) {
                            s_dec[bin] = 6;
                            recs[bin].decision = 6;
                            total_ground += (unsigned)r.n_ground;
                        }
                    }
                }
                n_ring_flat = 0;
            }
            ++concentric_idx;
        }
    }

    // ---- pass 2: where each sub-list goes, in the reference's append order
    const unsigned n_rnr = cnt[B], n_oor = cnt[B + 1];
    unsigned g_cur = 0;
    unsigned ng_cur = total_ground;  // the non-ground list follows the ground list in out_idx
    dst_a[B] = ng_cur;               // RNR hits first (ref :393), then out-of-range points (ref :618)
    ng_cur += n_rnr;
    dst_a[B + 1] = ng_cur;
    ng_cur += n_oor;
    dst_b[B] = dst_b[B + 1] = 0;
    for (int zone = 0; zone < 4; ++zone) {
        for (int ring = 0; ring < P.rings[zone]; ++ring) {
            const int b0 = P.bin_base[zone] + ring * P.sectors[zone];
            bool ring_has_cand = false;
            for (int sector = 0; sector < P.sectors[zone]; ++sector) {
                const int bin = b0 + sector;
                const unsigned n = cnt[bin];
                const int dec = s_dec[bin];
                if (dec == 0) {  // small bin, whole (ref :193)
                    dst_a[bin] = ng_cur;
                    dst_b[bin] = 0;
                    ng_cur += n;
                    continue;
                }
                const unsigned ng = (unsigned)recs[bin].n_ground;
                if (dec == 1 || dec == 3) {  // ref :264, :272
                    dst_a[bin] = ng_cur;
                    ng_cur += ng;
                } else if (dec == 2 || dec == 4) {  // ref :268, :276
                    dst_a[bin] = g_cur;
                    g_cur += ng;
                } else {
                    ring_has_cand = true;  // placed at the end of the ring
                }
                dst_b[bin] = ng_cur;  // ref :284
                ng_cur += n - ng;
            }
            if (ring_has_cand) {
                for (int sector = 0; sector < P.sectors[zone]; ++sector) {
                    const int bin = b0 + sector;
                    const int dec = s_dec[bin];
                    if (dec == 6) {  // ref :450
                        dst_a[bin] = g_cur;
                        g_cur += (unsigned)recs[bin].n_ground;
                    } else if (dec == 5) {  // ref :458 / :298
                        dst_a[bin] = ng_cur;
                        ng_cur += (unsigned)recs[bin].n_ground;
                    }
                }
            }
        }
    }
    PwppFrameResult *res = Bt.results + f;
    res->n_ground = (int)g_cur;
    res->n_nonground = (int)(ng_cur - total_ground);
    res->n_patches = n_patches;

    // ---- adaptive thresholds for the next frame of this stream (ref :338-375)
    for (int i = 0; i < P.num_rings_of_interest; ++i) {  // update_elevation_thr
        const int len = st.elev_len[i];
        if (len == 0) continue;
        double *h = hist_out + (0 * 4 + i) * P.hist_cap;
        double m = 0.0, s = 0.0;
        mean_stdev(h, len, m, s);
        if (i == 0) {
            st.elevation_thr[i] = m + 3 * s;
            st.sensor_height = -m;
        } else {
            st.elevation_thr[i] = m + 2 * s;
        }
        const int exceed = len - P.max_elev_storage;
        if (exceed > 0) {
            for (int j = 0; j + exceed < len; ++j) h[j] = h[j + exceed];
            st.elev_len[i] = len - exceed;
        }
    }
    for (int i = 0; i < P.num_rings_of_interest; ++i) {  // update_flatness_thr ("break", not "continue")
        const int len = st.flat_len[i];
        if (len <= 1) break;
        double *h = hist_out + (1 * 4 + i) * P.hist_cap;
        double m = 0.0, s = 0.0;
        mean_stdev(h, len, m, s);
        st.flatness_thr[i] = m + s;
        const int exceed = len - P.max_flat_storage;
        if (exceed > 0) {
            for (int j = 0; j + exceed < len; ++j) h[j] = h[j + exceed];
            st.flat_len[i] = len - exceed;
        }
    }
    Bt.st_scalar[fd.state_out] = st;
}

// ------------------------------------------------------------------------------------------
// K6  write the index lists
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_emit(PwppBatch Bt) {
    const int f = blockIdx.y, seg = blockIdx.x;
    const PwppDevParams &P = Bt.P;
    const int B = P.num_bins, NB = B + 2;
    const unsigned n = Bt.bin_count[(size_t)f * NB + seg];
    if (n == 0) return;
    const PwppFrameDesc fd = Bt.frames[f];
    const unsigned off = Bt.bin_off[(size_t)f * NB + seg];
    int *out = Bt.out_idx + fd.base;
    const unsigned da = Bt.dst_a[(size_t)f * NB + seg];
    const bool whole = seg >= B || (uint64_t)n < P.min_pts;
    if (whole) {
        const float4 *src = Bt.sorted + fd.base + off;
        for (unsigned i = threadIdx.x; i < n; i += kBlock) out[da + i] = (int)(__float_as_uint(src[i].w) & 0x7fffffffu);
        return;
    }
    const int *src = Bt.plist + fd.base + off;
    const unsigned ng = (unsigned)Bt.recs[(size_t)f * B + seg].n_ground;
    const unsigned db = Bt.dst_b[(size_t)f * NB + seg];
    for (unsigned i = threadIdx.x; i < n; i += kBlock) {
        const int v = src[i];
        if (i < ng)
            out[da + i] = v;
        else
            out[db + (i - ng)] = v;
    }
}

// getGround()/getNonground() rows (ref :8-16): xyz of the listed points, gathered on the device
__global__ __launch_bounds__(kBlock) void k_gather_xyz(PwppFrameDesc fd, const int *idx, int count, float *out) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    float x, y, z, w;
    load_point(fd, idx[j], x, y, z, w);
    out[(size_t)j * 3] = x;
    out[(size_t)j * 3 + 1] = y;
    out[(size_t)j * 3 + 2] = z;
}

}  // namespace

extern "C" int pwpp_launch_gather_xyz(const PwppFrameDesc *fd, const int *idx, int count, float *out, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_gather_xyz, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, *fd, idx, count, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host-side launcher used by pwpp_capi.cpp
// ------------------------------------------------------------------------------------------
extern "C" int pwpp_launch_pipeline(const PwppBatch *batch, hipStream_t stream, hipEvent_t *ev /* 7 events or null */) {
    const PwppBatch &B = *batch;
    const int F = B.num_frames;
    if (F <= 0) return 0;
    const int NB = B.P.num_bins + 2;
    const unsigned gx = (unsigned)((B.max_n + kPtsPerBlock - 1) / kPtsPerBlock);
    if (ev) (void)hipEventRecord(ev[0], stream);
    if (gx > 0) hipLaunchKernelGGL(k_czm_bin, dim3(gx, F), dim3(kBlock), 0, stream, B);
    if (ev) (void)hipEventRecord(ev[1], stream);
    hipLaunchKernelGGL(k_czm_scan, dim3(F), dim3(kBlock), 0, stream, B);
    if (ev) (void)hipEventRecord(ev[2], stream);
    if (gx > 0) hipLaunchKernelGGL(k_czm_scatter, dim3(gx, F), dim3(kBlock), 0, stream, B);
    if (ev) (void)hipEventRecord(ev[3], stream);
    hipLaunchKernelGGL(k_patch_fit, dim3(B.P.num_bins, F), dim3(kBlock), 0, stream, B);
    if (ev) (void)hipEventRecord(ev[4], stream);
    hipLaunchKernelGGL(k_gle_tgr, dim3(F), dim3(64), 0, stream, B);
    if (ev) (void)hipEventRecord(ev[5], stream);
    hipLaunchKernelGGL(k_emit, dim3(NB, F), dim3(kBlock), 0, stream, B);
    if (ev) (void)hipEventRecord(ev[6], stream);
    return (int)hipGetLastError();
}
