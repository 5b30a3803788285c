// pwpp_common.hpp -- device helpers shared by the gfx950 kernels of this library.
// The arithmetic helpers implement the contract of DESIGN.md section 3.4; citations are
// /root/reference/cpp/patchworkpp/src/patchworkpp.cpp unless a header is named.
#ifndef PWPP_COMMON_HPP
#define PWPP_COMMON_HPP

#include <float.h>
#include <math.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "../../include/pwpp.h"
#include "pwpp_dev.h"

// (PWPP_LAYOUT_ROW_MAJOR / _COL_MAJOR / _FIELDS: include/pwpp.h)

namespace {

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
    } else if (fd.layout == PWPP_LAYOUT_FIELDS) {
        // sensor_msgs/PointCloud2 as the reference's ROS wrapper reads it (ros/src/Utils.hpp:158-172: one float
        // iterator per field): the fields are fetched where they lie, no repack on the host
        const char *rec = reinterpret_cast<const char *>(fd.pts) + (size_t)i * (size_t)fd.step;
        x = *reinterpret_cast<const float *>(rec + fd.off[0]);
        y = *reinterpret_cast<const float *>(rec + fd.off[1]);
        z = *reinterpret_cast<const float *>(rec + fd.off[2]);
        w = fd.off[3] >= 0 ? *reinterpret_cast<const float *>(rec + fd.off[3]) : 0.0f;
    } else {  // column-major planes (Eigen::MatrixXf storage)
        const size_t n = (size_t)fd.n;
        x = fd.pts[i];
        y = fd.pts[n + i];
        z = fd.pts[2 * n + i];
        w = fd.cols == 4 ? fd.pts[3 * n + i] : 0.0f;
    }
}

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
                        // b01 / |b01| (Eigen's sign factor) is exactly +-1 for every finite non-zero b01 and NaN
                        // otherwise (inf / inf): no need for the ten instructions of an IEEE division
                        const float unit = f_abs(b01) <= FLT_MAX ? __uint_as_float(0x3f800000u | (__float_as_uint(b01) & 0x80000000u))
                                                                 : __uint_as_float(0x7fc00000u);
                        sr = -sign_t * unit * f_abs(tt) * nn;
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

// ------------------------------------------------------------------------------------------
// The fixed-point contract of the plane-fit sums (restated for the checker
// in oracle/pwpp_oracle.cpp):
//   Q_x(v) = rint(double(v) * 2^s - ox * 2^s), Q_y alike        (ox, oy: the bin's origin, multiples of 1/8 m)
//   Q_z(v) = the same around z0 after clamping v to [z0 - ZR, z0 + ZR] in float
//            (z0: the patch's first lowest-point representative rounded to 1/8 m, ZR = 2^(35-s) m; 2^(26-s) on the narrow grid)
// |Q| <= 2^35 (2^26) by construction of s (pwpp_capi.cpp, fxp_geometry): s = 30 (21) with the default CZM.
// One v_fma_f64 does the scaling, the subtraction and the rounding: added to 2^52 + 2^51 the exact
// value double(v) * 2^s - O is rounded to an integer (ties to even, as rint) by the FMA itself, and
// the low bits of the result are that integer in two's complement (32 of them on the narrow grid, 51 on the wide one).
// ------------------------------------------------------------------------------------------
struct FxpOrg {
    double cx, cy, cz;  // 2^52 + 2^51 - origin * 2^s
    float zlo, zhi;     // z0 -+ ZR (exact in float: multiples of 1/8 m below 2^13)
};
__device__ __forceinline__ FxpOrg fxp_org(float ox, float oy, float z0, double scale, float zr) {
    const double magic = 6755399441055744.0;
    FxpOrg o;
    o.cx = magic - (double)ox * scale;
    o.cy = magic - (double)oy * scale;
    o.cz = magic - (double)z0 * scale;
    o.zlo = z0 - zr;
    o.zhi = z0 + zr;
    return o;
}
// v * scale + c as ONE three-operand v_fma_f64.  Left to itself the compiler picks the two-operand v_fmac_f64 for about half of these
// (shorter encoding), which accumulates into its destination and therefore needs the addend -- a per-pass constant -- copied into a
// fresh register pair first: ten v_mov_b64 per chunk of the fit kernels' points loop.
__device__ __forceinline__ double fxp_fma(double v, double scale, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(scale), "v"(c));
    return r;
}
__device__ __forceinline__ int fxp_q(float v, double scale, double c) {
    return (int)(unsigned)(unsigned long long)__double_as_longlong(fxp_fma((double)v, scale, c));
}
// z origin of a patch from its first lowest-point representative (ref :103)
__device__ __forceinline__ float fxp_z_origin(double lpr) {
    if (!(fabs(lpr) <= DBL_MAX)) return 0.0f;  // NaN, +-inf
    double t = rint(lpr * 8.0) * 0.125;
    t = t > 4096.0 ? 4096.0 : t;
    t = t < -4096.0 ? -4096.0 : t;
    return (float)t;
}

// order-preserving map float -> uint32 (for the lowest-point selection)
__device__ __forceinline__ unsigned z_key(float z) {
    const unsigned b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_z(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Per-lane partial sums of the quantised coordinates, in the two widths of the contract (PwppDevParams.fxp_wide):
//
// MomentsT<false>  |Q| <= 2^26 (the 2^-21 m grid of rounds 3-5, option "exact_moments" = 0).  Every product and every sum
//   is a 64-bit integer multiply-add (v_mad_i64_i32, full rate on gfx950 like v_fma_f64: tools/ubench/valu_rates.hip): exact
//   whatever the order, no conversion back and forth, no chunk-wise flush.  A lane may add up 2^11 - 1 points before a second
//   moment could leave int64; the kernels deal at most that many to a lane (pwpp_launch_fit caps the classes) and k_fit_stream,
//   which takes whatever is larger, sums in 128 bits.
//
// MomentsT<true>   |Q| <= 2^35 (contract v4, the default: a 2^-30 m grid, on which every float of magnitude >= 2^-7 m lies --
//   the sums are those of exact arithmetic on the reference's own floats).  Q = H * 2^9 + L with H = Q >> 9 (|H| <= 2^26) and
//   L = Q & 511, both cut out of the FMA's bit pattern (v_alignbit_b32, v_and_b32), and
//       Q_a Q_b = 2^18 H_a H_b + 2^9 (H_a L_b + L_a H_b) + L_a L_b
//   is added up term by term: hh (v_mad_i64_i32, as above), hl (v_mad_i64_i32; |H L| < 2^35: 2^11 points leave 17 bits of
//   head-room), ll (v_mad_u32_u24; < 2^18 each, 2^29 after 2^11 points).  First moments: the FMA result is the double
//   2^52 + 2^51 + Q, whose BIT PATTERN is 0x4338000000000000 + Q as an integer -- the patterns are added up as they are
//   (one 64-bit add per coordinate) and n times the constant is taken off at the end (mod 2^64: exact).
//   21 multiply-adds + 6 bit operations + 3 adds per point against 9 + 0 + 0: the price of 36-bit values on a 32-bit multiplier.
template <bool WIDE>
struct MomentsT;

template <>
struct MomentsT<false> {
    long long n, s1[3], s2[6];
    __device__ __forceinline__ void clear() {
        n = 0;
        s1[0] = s1[1] = s1[2] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s2[k] = 0;
    }
    // (the caller adds the number of points itself, e.g. the population count of a chunk's mask, and hands over the height already
    // clamped to [zlo, zhi]: fxp_clamp_z -- it wants to know whether the clamp acted)
    __device__ __forceinline__ void add_uncounted(float x, float y, float zc, double scale, const FxpOrg &o) {
        const int qx = fxp_q(x, scale, o.cx), qy = fxp_q(y, scale, o.cy);
        const int qz = fxp_q(zc, scale, o.cz);  // (a NaN z never gets here: every test on it fails)
        // first moments as multiply-adds by a 1 the compiler cannot see through: ONE v_mad_i64_i32 each, where the plain
        // 64-bit add of a sign-extended int is a shift and an add
        int one = 1;
        asm volatile("" : "+s"(one));
        s1[0] += (long long)qx * one;
        s1[1] += (long long)qy * one;
        s1[2] += (long long)qz * one;
        s2[0] += (long long)qx * qx;
        s2[1] += (long long)qx * qy;
        s2[2] += (long long)qx * qz;
        s2[3] += (long long)qy * qy;
        s2[4] += (long long)qy * qz;
        s2[5] += (long long)qz * qz;
    }
    __device__ __forceinline__ void add(float x, float y, float z, double scale, const FxpOrg &o) {
        n += 1;
        add_uncounted(x, y, __builtin_amdgcn_fmed3f(z, o.zlo, o.zhi), scale, o);
    }
    // this lane's sums once its pass is over (n final)
    __device__ __forceinline__ long long first(int k) const { return s1[k]; }
    __device__ __forceinline__ __int128 second(int k) const { return (__int128)s2[k]; }
};

template <>
struct MomentsT<true> {
    long long n;
    unsigned long long s1b[3];  // sums of the bit patterns 0x4338000000000000 + Q
    long long hh[6], hl[6];     // pairs (0,0) (0,1) (0,2) (1,1) (1,2) (2,2); hl of a diagonal pair holds H L once (doubled in second())
    unsigned ll[6];
    __device__ __forceinline__ void clear() {
        n = 0;
        s1b[0] = s1b[1] = s1b[2] = 0ull;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            hh[k] = 0;
            hl[k] = 0;
            ll[k] = 0u;
        }
    }
    __device__ __forceinline__ void add_uncounted(float x, float y, float zc, double scale, const FxpOrg &o) {
        const unsigned long long tx = (unsigned long long)__double_as_longlong(fxp_fma((double)x, scale, o.cx));
        const unsigned long long ty = (unsigned long long)__double_as_longlong(fxp_fma((double)y, scale, o.cy));
        const unsigned long long tz = (unsigned long long)__double_as_longlong(fxp_fma((double)zc, scale, o.cz));
        s1b[0] += tx;
        s1b[1] += ty;
        s1b[2] += tz;
        const int hx = (int)__builtin_amdgcn_alignbit((unsigned)(tx >> 32), (unsigned)tx, 9), lx = (int)((unsigned)tx & 511u);
        const int hy = (int)__builtin_amdgcn_alignbit((unsigned)(ty >> 32), (unsigned)ty, 9), ly = (int)((unsigned)ty & 511u);
        const int hz = (int)__builtin_amdgcn_alignbit((unsigned)(tz >> 32), (unsigned)tz, 9), lz = (int)((unsigned)tz & 511u);
        hh[0] += (long long)hx * hx;
        hh[1] += (long long)hx * hy;
        hh[2] += (long long)hx * hz;
        hh[3] += (long long)hy * hy;
        hh[4] += (long long)hy * hz;
        hh[5] += (long long)hz * hz;
        // (two chained multiply-adds per mixed pair: the barrier keeps the compiler from forming a * b + c * d on the side and
        // adding that to the accumulator with a third instruction)
        hl[0] += (long long)hx * lx;
        hl[1] += (long long)hx * ly;
        asm volatile("" : "+v"(hl[1]));
        hl[1] += (long long)hy * lx;
        hl[2] += (long long)hx * lz;
        asm volatile("" : "+v"(hl[2]));
        hl[2] += (long long)hz * lx;
        hl[3] += (long long)hy * ly;
        hl[4] += (long long)hy * lz;
        asm volatile("" : "+v"(hl[4]));
        hl[4] += (long long)hz * ly;
        hl[5] += (long long)hz * lz;
        ll[0] += __umul24((unsigned)lx, (unsigned)lx);
        ll[1] += __umul24((unsigned)lx, (unsigned)ly);
        ll[2] += __umul24((unsigned)lx, (unsigned)lz);
        ll[3] += __umul24((unsigned)ly, (unsigned)ly);
        ll[4] += __umul24((unsigned)ly, (unsigned)lz);
        ll[5] += __umul24((unsigned)lz, (unsigned)lz);
    }
    __device__ __forceinline__ void add(float x, float y, float z, double scale, const FxpOrg &o) {
        n += 1;
        add_uncounted(x, y, __builtin_amdgcn_fmed3f(z, o.zlo, o.zhi), scale, o);
    }
    __device__ __forceinline__ long long first(int k) const {
        const unsigned long long b = k == 0 ? s1b[0] : (k == 1 ? s1b[1] : s1b[2]);
        return (long long)(b - (unsigned long long)n * 0x4338000000000000ull);
    }
    __device__ __forceinline__ __int128 second(int k) const {
        const bool diag = k == 0 || k == 3 || k == 5;
        return (__int128)hh[k] * 262144 + (__int128)hl[k] * (diag ? 1024 : 512) + (__int128)ll[k];
    }
    // The sixteen values a 16-lane row adds up for a patch below 1024 points (k_fit_w64<16, .>), none of which leaves int64 there:
    // n, the three raw bit-pattern sums (the row total minus n times the constant is S1: wide_first), the six hh sums (<= 2^52 per
    // point), and the six sums 2^9 (H_a L_b + L_a H_b) + L_a L_b (<= 2^46 per point) -- two shifts and an add per pair where the 128-bit
    // second moment cost twelve instructions per lane and pass; the owner of the patch puts S2 = 2^18 hh + the rest together.
    __device__ __forceinline__ void to_row16(long long (&v)[16]) const {
        v[0] = n;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[1 + k] = (long long)s1b[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool diag = k == 0 || k == 3 || k == 5;
            v[4 + k] = hh[k];
            v[10 + k] = hl[k] * (diag ? 1024 : 512) + (long long)ll[k];
        }
    }
};
__device__ __forceinline__ long long wide_first(long long raw_total, long long n) { return (long long)((unsigned long long)raw_total - (unsigned long long)n * 0x4338000000000000ull); }
__device__ __forceinline__ __int128 wide_second(long long hh_total, long long rest_total) { return (__int128)hh_total * 262144 + (__int128)rest_total; }
// the 36-bit Q itself (the kernels that walk a patch point by point: k_fit_stream)
__device__ __forceinline__ long long fxp_q_wide(float v, double scale, double c) {
    const long long t = __double_as_longlong(fxp_fma((double)v, scale, c));
    return (t << 13) >> 13;  // bits 0..50 of the pattern = Q mod 2^51, sign-extended
}

// ------------------------------------------------------------------------------------------
// plane of ref :47-75 from the exact integer moments of a point set:
//   mean_a = float( (S1_a * (1/n)) * 2^-s + origin_a )
//   cov_ab = float( ((n*S2_ab - S1_a*S1_b) * (1/(n*(n-1)))) * 2^-2s )     numerator exact in 128 bits; both reciprocals in double
// then Eigen's JacobiSVD on the float covariance, normal = U.col(2) flipped to z >= 0 (:66-68),
// d = -(normal . mean) as a float dot product widened to double (:74).
// ------------------------------------------------------------------------------------------
struct PlaneFit {
    float nx, ny, nz;
    float mean[3];
    float sv[3];
    double d;
};

// (mean, covariance) -> plane: Eigen's JacobiSVD on the float covariance, normal = U.col(2) flipped to z >= 0, d
__device__ __forceinline__ void plane_from_cov(const float mean[3], const float cov[9], int debug, PlaneFit &out) {
    float u[9], sv[3];
#ifdef PWPP_ABLATE_NO_JACOBI  // (timing experiments only: a horizontal plane through the mean instead of the eigen-solve)
    for (int k = 0; k < 9; ++k) u[k] = (k == 8) ? 1.0f : 0.0f;
    sv[0] = cov[0];
    sv[1] = cov[4];
    sv[2] = cov[8] * 1e-3f;
#else
    jacobi_svd3(cov, u, sv);
#endif
    float nx = u[2], ny = u[5], nz = u[8];  // U.col(2), ref :66
    if (nz < 0) {                           // ref :68
        nx *= -1;
        ny *= -1;
        nz *= -1;
    }
    const float dot = nx * mean[0] + ny * mean[1] + nz * mean[2];  // ref :74, float dot
    out.nx = nx;
    out.ny = ny;
    out.nz = nz;
    out.mean[0] = mean[0];
    out.mean[1] = mean[1];
    out.mean[2] = mean[2];
    out.sv[0] = sv[0];
    out.sv[1] = sv[1];
    out.sv[2] = sv[2];
    out.d = -dot;
}

// output o of the moments -> (mean, covariance) step: o = 0..2 mean[o], o = 3..8 cov of the pair (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
//   mean_a = float( S1_a * (1/n) * 2^-s + origin_a )                 cov_ab = float( (n*S2_ab - S1_a*S1_b) * (1/(n*(n-1))) * 2^-2s )
// with the two reciprocals formed ONCE per fit in double (contract v3: nine IEEE double divisions were 270 of a solve's
// 2 100 instructions; a product with the correctly rounded reciprocal differs from the quotient by at most one ulp of a
// double, 2^-29 of a float ulp, before the single rounding to float).  rn = 1/n, rd = 1/(n*(n-1)) (n = 1: 1/0 = inf, and
// 0 * inf = NaN as the reference's 0/0).
__device__ __forceinline__ float moment_output(int o, long long n, const long long s1[3], const __int128 s2[6], int shift, const double org[3],
                                               double rn, double rd) {
    const double inv = __longlong_as_double((long long)(1023 - shift) << 52);  // 2^-shift, exactly what 1.0 / (1 << shift) gives, without the division
    if (o < 3) {
        const long long v = o == 0 ? s1[0] : (o == 1 ? s1[1] : s1[2]);
        const double g = o == 0 ? org[0] : (o == 1 ? org[1] : org[2]);
        return (float)(((double)v * rn) * inv + g);
    }
    const int k = o - 3;  // pair index
    const int a = k < 3 ? 0 : (k < 5 ? 1 : 2), b = k < 3 ? k : (k < 5 ? k - 2 : 2);
    const long long sa = a == 0 ? s1[0] : (a == 1 ? s1[1] : s1[2]), sb = b == 0 ? s1[0] : (b == 1 ? s1[1] : s1[2]);
    __int128 m2 = s2[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) m2 = k == q ? s2[q] : m2;
    const __int128 num = (__int128)n * m2 - (__int128)sa * (__int128)sb;
    return (float)((i128_to_double(num) * rd) * (inv * inv));
}
__device__ __forceinline__ void fit_reciprocals(long long n, double &rn, double &rd) {
    rn = 1.0 / (double)n;
    rd = 1.0 / ((double)n * (double)(n - 1));
}

// the moments -> (mean, covariance) step alone: mean[3] and the six distinct covariance entries (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
__device__ __forceinline__ void mean_cov_from_totals(long long n, const long long s1[3], const __int128 s2[6], int shift,
                                                     float ox, float oy, float z0, float mean[3], float c6[6]) {
    const double org[3] = {(double)ox, (double)oy, (double)z0};
    double rn, rd;
    fit_reciprocals(n, rn, rd);
#pragma unroll
    for (int a = 0; a < 3; ++a) mean[a] = moment_output(a, n, s1, s2, shift, org, rn, rd);
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = moment_output(3 + k, n, s1, s2, shift, org, rn, rd);
}
__device__ __forceinline__ void plane_from_mean_c6(const float mean[3], const float c6[6], int debug, PlaneFit &out) {
    const int map[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    float cov[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cov[k] = c6[map[k]];
    plane_from_cov(mean, cov, debug, out);
}
__device__ __forceinline__ void plane_from_totals(long long n, const long long s1[3], const __int128 s2[6], int shift,
                                                  float ox, float oy, float z0, int debug, PlaneFit &out) {
    float mean[3], c6[6];
    mean_cov_from_totals(n, s1, s2, shift, ox, oy, z0, mean, c6);
    plane_from_mean_c6(mean, c6, debug, out);
}

// ------------------------------------------------------------------------------------------
// Contract v3: a fit set of ONE, TWO or THREE points.  There the reference's float sums (ref :56-60) are determinate --
// Eigen reduces fewer elements than one SIMD packet sequentially in storage order (the column redux of colwise().mean()
// and the coefficient-based product it takes for 3 x n times n x 3 with n this small), and two terms commute -- so these
// sets follow the reference's own float arithmetic instead of the fixed-point grid: mean = float sum / n, centred rows,
// products summed in float, / (n - 1).  Order: the reference's z-sorted bin (ref :199), equal heights in cloud order
// (the kernels hand the points over sorted).  The covariance of a float sum is symmetric term by term (a * b == b * a),
// so six entries suffice.  Restated for the checker in oracle/pwpp_oracle.cpp (estimate_plane, n <= 3).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void mean_cov_tiny(int n, const float px[3], const float py[3], const float pz[3], float mean[3], float c6[6]) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < n) {
            ax += px[i];
            ay += py[i];
            az += pz[i];
        }
    }
    const float fn = (float)n;
    mean[0] = ax / fn;
    mean[1] = ay / fn;
    mean[2] = az / fn;
    float acc[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < n) {
            const float cx = px[i] - mean[0], cy = py[i] - mean[1], cz = pz[i] - mean[2];
            acc[0] += cx * cx;
            acc[1] += cx * cy;
            acc[2] += cx * cz;
            acc[3] += cy * cy;
            acc[4] += cy * cz;
            acc[5] += cz * cz;
        }
    }
    const float den = (float)(n - 1);
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = acc[k] / den;
}

// The same when every lane of the wave holds the SAME totals (the four-waves-per-patch kernel): the nine outputs --
// nine IEEE double divisions and six 128-bit products, a third of the solve's instructions -- are computed by nine
// lanes side by side and handed round with v_readlane; only the Jacobi iteration stays serial.
__device__ __forceinline__ void mean_cov_from_totals_uniform(long long n, const long long s1[3], const __int128 s2[6], int shift,
                                                             float ox, float oy, float z0, float mean[3], float c6[6]) {
    const double org[3] = {(double)ox, (double)oy, (double)z0};
    const int o = lane_id() & 15;
    double rn, rd;
    fit_reciprocals(n, rn, rd);
    const float mine = moment_output(o < 9 ? o : 0, n, s1, s2, shift, org, rn, rd);
#pragma unroll
    for (int a = 0; a < 3; ++a) mean[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), a));
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), 3 + k));
}
__device__ __forceinline__ void plane_from_totals_uniform(long long n, const long long s1[3], const __int128 s2[6], int shift,
                                                          float ox, float oy, float z0, int debug, PlaneFit &out) {
    float mean[3], c6[6];
    mean_cov_from_totals_uniform(n, s1, s2, shift, ox, oy, z0, mean, c6);
    plane_from_mean_c6(mean, c6, debug, out);
}

// The height that separates the two parts of a frame's bins (pwpp_dev.h): k_czm_bin / k_czm_bin_scatter compare every z
// with it, the fit kernels build their skip tests on it.  sensor_height = the frame's state BEFORE this call (K5 updates
// the state after the fits).
__device__ __forceinline__ float hi_split_z(const PwppDevParams &P, double sensor_height) { return (float)(-sensor_height + (double)P.hi_split); }

// ref :551-554  (float products, float adds left to right, one double add)
__device__ __forceinline__ double plane_dist(float nx, float ny, float nz, double d, float x, float y, float z) {
    return nx * x + ny * y + nz * z + d;
}
// the float part of that expression: what the streamed passes evaluate per point
__device__ __forceinline__ float plane_s(float nx, float ny, float nz, float x, float y, float z) { return nx * x + ny * y + nz * z; }

// The test of ref :525 / :108 / :499 per point is  double(s) + d < thr  with s the float above and d, thr fixed for a whole
// pass.  double(s) + d is a correctly rounded, hence monotone function of s, so the set of floats that pass is a down-set
// {s < T} (NaN passes neither form): T = the smallest float that FAILS.  The passes compare s with T in float -- one
// v_cmp_f32 per point instead of a conversion, a double add and a double compare -- and the owner of the patch finds T
// once per pass: from the real-number boundary thr - d, corrected by walking the ordered float keys until
// "T fails and its predecessor passes" holds (a binary search over all floats backs that up; the result is exact by
// construction whatever d and thr are: infinities and NaN included).
__device__ __forceinline__ float plane_test_threshold(double d, double thr) {
    const unsigned kmin = 0x007fffffu /* z_key(-inf) */, kmax = 0xff800000u /* z_key(+inf) */;
    auto pass_key = [&](unsigned k) { return (double)key_z(k) + d < thr; };
    if (!pass_key(kmin)) return key_z(kmin);  // nothing passes (d or thr NaN, d = +inf, thr = -inf): s < -inf never holds
    float t = (float)(thr - d);
    unsigned k = (t == t) ? z_key(t) : kmax;
    k = k < kmin ? kmin : (k > kmax ? kmax : k);
    for (int i = 0; i < 3 && k < kmax && pass_key(k); ++i) ++k;
    for (int i = 0; i < 3 && k > kmin && !pass_key(k - 1u); ++i) --k;
    if (pass_key(k) || !pass_key(k - 1u)) {  // (k > kmin here: kmin passes)  not bracketed by the short walk: search all floats
        unsigned lo = kmin, hi = kmax;       // pass_key(lo), !pass_key(hi): +inf never passes (inf + d is inf or NaN)
        while (hi - lo > 1u) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            if (pass_key(mid))
                lo = mid;
            else
                hi = mid;
        }
        k = hi;
    }
    return key_z(k);
}

// Patches are sorted by size into quarter-octave buckets (k_czm_scan); the fit kernels
// take contiguous bucket ranges, so the rows of one wave have similar point counts.
// bucket(n) = 4*floor(log2 n) + next two mantissa bits, n >= 1; monotone in n.
__host__ __device__ __forceinline__ int pwpp_size_bucket(unsigned n) {
    if (n < 4u) return (int)n;  // 1,2,3 -> 1,2,3 (0 unused)
    const int e = 31 - __builtin_clz(n);  // (n >= 4)
    const int b = 4 * e + (int)((n >> (e - 2)) & 3u) - 4;  // n=4 -> 4
    return b < PWPP_NUM_BUCKETS - 1 ? b : PWPP_NUM_BUCKETS - 1;
}
// smallest n that lands in bucket b (b >= 4)
__host__ __device__ __forceinline__ unsigned pwpp_bucket_floor(int b) {
    if (b < 4) return (unsigned)(b < 1 ? 1 : b);
    const int e = (b + 4) / 4, q = (b + 4) % 4;
    return (1u << e) + ((unsigned)q << (e - 2));
}

}  // namespace

#endif
