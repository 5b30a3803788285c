// oracle/pwpp_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference hot path patchwork::PatchWorkpp::estimateGround()
// (/root/reference/cpp/patchworkpp/src/patchworkpp.cpp:151-336 and everything it
// calls), written from the reference's behaviour, function by function, with the
// reference file:line each step follows.  It is the checker the HIP path is compared
// with; nothing in the product links, imports or executes it.
//
// PARITY STATUS: "parity unpinned" at the Eigen boundary.  The reference holds no test,
// golden vector or known-answer fixture for this path, and the
// arithmetic of PatchWorkpp::estimate_plane lives in Eigen 3.4.0
// (cpp/cmake/eigen.cmake:31), which is absent from /root/reference and from this image.
// What pins this file instead: tests/test_oracle.py requires it to agree BIT FOR BIT --
// index lists in the reference's own output order, centres, normals, adaptive thresholds
// and histories -- with the reference's unmodified patchworkpp.cpp compiled against
// oracle/eigen_shim (oracle/_ref/libpwpp_ref*.so), on the six KITTI sample frames and on
// synthetic clouds, in every flavour the shim has, stateless and as a sequence.
//
// Arithmetic flavours for the plane fit (the only place where the reference defers to
// Eigen); everything else follows the reference's own float/double expressions:
//   PWO_ARITH_EIGEN_F32  float accumulators in storage order (plain reading of Eigen)
//   PWO_ARITH_FXP        the product's contract v4: fit sets of 1-3 points in the
//                        reference's own float arithmetic (determinate there), larger ones as EXACT integer
//                        moments on a 2^-30 m grid -- every float of magnitude >= 2^-7 m enters the sums
//                        unquantised, so the sums are those of exact arithmetic on the reference's own
//                        floats, whatever the order; this is the flavour the HIP kernels must match bit for bit.
//   PWO_ARITH_FXP21      contract v3 of rounds 3-5 (the same on a 2^-21 m grid), kept as a witness: the
//                        grid was coarser than the float ulp of the data, which cost 0.2 % of the frames
//                        a few indices against the reference (profiles/r05_parity_statistics_10k.json).
//   PWO_ARITH_EXACT_F64  reference-neutral arbiter: what :56-60 give in (near-)exact
//                        arithmetic -- double accumulation of the unquantised floats, one
//                        rounding to float per output.  Neither the product nor the
//                        reference computes this; both are measured against it.
//   PWO_ARITH_F32_PACKET4 float again, four partial sums: a witness of how far float
//                        results move with the summation order alone.
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#include "oracle_api.h"
#include "oracle_ext.h"

namespace {

struct Pt {  // reference patchworkpp.h:20-27 (PointXYZ)
    float x, y, z;
    int idx;
};

// reference patchworkpp.cpp:6 -- by-value comparator; std::sort (introsort) with it is
// unstable, the permutation depends only on the comparison outcomes, so sorting this
// struct reproduces the reference's order exactly, ties included.
bool z_less(Pt a, Pt b) { return a.z < b.z; }

struct Plane {      // scratch members of the reference class, patchworkpp.h:177-182
    float normal[3];
    float mean[3];
    float sv[3];
    double d;
};

struct Candidate {  // reference patchworkpp.h:29-40 (RevertCandidate)
    int concentric_idx, sector_idx;
    double flatness, line_variable;
    std::vector<Pt> ground;
    int patch_slot;
};

float f_abs(float v) { return v < 0.0f ? -v : v; }
float f_max(float a, float b) { return a < b ? b : a; }

// ---------------------------------------------------------------------------------
// The fixed-point contract of the plane-fit sums, version 4 (v3 in brackets).
//   * every CZM bin has an ORIGIN (ox, oy): its polar centre rounded to 1/8 m; R = the largest
//     distance of any bin corner from its origin;
//   * s = the largest shift <= 30 [21] with (R + 0.01) * 2^s <= 2^35 [2^26]; ZR = 2^(35 - s) [2^(26 - s)] metres;
//   * every visit of a bin (ref :206) has a z origin z0: the first lowest-point representative
//     the visit computes (ref :103), rounded to 1/8 m (0 if it is not finite, +-4096 at most);
//   * Q_x(v) = the integer nearest to the EXACT value v * 2^s - ox * 2^s (ties to even), Q_y alike, Q_z(v) the
//     same around z0 after clamping v to [z0 - ZR, z0 + ZR] in float (fmaxf, then fminf);
//     with s = 30 a float of magnitude >= 2^-7 m is a multiple of 2^-30 m: Q is then exact, not rounded;
//   * exact integer moments n, S1_a = sum Q_a, S2_ab = sum Q_a Q_b;
//   * mean_a = float(double(S1_a) * (1 / double(n)) * 2^-s + origin_a);
//     cov_ab = float(double(n S2_ab - S1_a S1_b) * (1 / (double(n) double(n-1))) * 2^-2s), numerator exact,
//     the two reciprocals formed once per fit in double (contract v3);
//   * fit sets of 1-3 points: the reference's own float sums in (z, cloud index) order (estimate_plane below).
// ---------------------------------------------------------------------------------
struct FxpGrid {
    int max_shift;
    double qmax;
};
constexpr FxpGrid kFxpV4 = {30, 34359738368.0};  // 2^-30 m, |Q| <= 2^35
constexpr FxpGrid kFxpV3 = {21, 67108864.0};     // 2^-21 m, |Q| <= 2^26

struct FxpGeom {
    int shift = 0;
    double scale = 1.0;  // 2^shift
    double zr = 0.0;     // qmax / 2^shift
    std::vector<float> ox, oy;
};

// geometry in double, exactly as the reference's constructor computes it (patchworkpp.h:122-134)
FxpGeom fxp_geometry(const double min_ranges[4], const double ring_sizes[4], const double sector_sizes[4],
                     const int rings[4], const int sectors[4], double max_range, const FxpGrid &grid) {
    FxpGeom g;
    double rmax = 0.0;
    for (int z = 0; z < 4; ++z)
        for (int r = 0; r < rings[z]; ++r)
            for (int k = 0; k < sectors[z]; ++k) {
                const double r0 = min_ranges[z] + r * ring_sizes[z];
                const double r1 = (z == 3 && r == rings[z] - 1) ? max_range : r0 + ring_sizes[z];
                const double t0 = k * sector_sizes[z], t1 = (k + 1) * sector_sizes[z];
                double cx = 0.0, cy = 0.0;
                if (sectors[z] >= 4) {  // (a sector wider than a quarter turn keeps the sensor as origin)
                    const double rc = 0.5 * (r0 + r1), tc = 0.5 * (t0 + t1);
                    cx = std::rint(rc * std::cos(tc) * 8.0) / 8.0;
                    cy = std::rint(rc * std::sin(tc) * 8.0) / 8.0;
                }
                g.ox.push_back((float)cx);
                g.oy.push_back((float)cy);
                double far = 0.0;
                if (sectors[z] >= 4) {
                    const double cr[2] = {r0, r1}, ct[2] = {t0, t1};
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            const double dx = cr[a] * std::cos(ct[b]) - cx, dy = cr[a] * std::sin(ct[b]) - cy;
                            far = std::max(far, std::sqrt(dx * dx + dy * dy));
                        }
                } else {
                    far = r1;
                }
                rmax = std::max(rmax, far);
            }
    int s = grid.max_shift;
    while (s > 0 && (rmax + 0.01) * (double)(1 << s) > grid.qmax) --s;
    g.shift = s;
    g.scale = (double)(1 << s);
    g.zr = grid.qmax / g.scale;
    return g;
}

double fxp_z_origin(double lpr) {
    if (!(std::fabs(lpr) <= DBL_MAX)) return 0.0;  // NaN, +-inf
    double t = std::rint(lpr * 8.0) / 8.0;
    if (t > 4096.0) t = 4096.0;
    if (t < -4096.0) t = -4096.0;
    return t;
}

// Q(v) = round-half-even of the exact value v * 2^shift - origin * 2^shift, in integers (the device forms it with ONE
// fused multiply-add, a single rounding; two roundings in double could differ on a tie for |v| < 2^-shift).
// v finite; origin * 2^shift is an integer (origins are multiples of 1/8 m, shift >= 3).
int64_t fxp_quantise(float v, double origin, int shift) {  // Q_x, Q_y
    const int64_t O = (int64_t)std::ldexp(origin, shift);
    int e = 0;
    const double fr = std::frexp((double)v, &e);        // v = fr * 2^e, 0.5 <= |fr| < 1 (or 0)
    const int64_t m = (int64_t)std::ldexp(fr, 24);      // 24-bit signed mantissa, exact
    const int sh = e - 24 + shift;                      // v * 2^shift = m * 2^sh
    if (m == 0) return -O;
    if (sh >= 0) return (int64_t)((__int128)m << sh) - O;  // (|v * scale| < 2^63 for every value the clamps let through)
    const int k = -sh;
    if (k > 26) return -O;                               // |m * 2^sh| < 1/4: nearest integer of (tiny - O) is -O
    const __int128 num = (__int128)m - ((__int128)O << k);  // value = num / 2^k
    const __int128 one = (__int128)1 << k, half = one >> 1;
    __int128 q = num >> k;                               // floor
    const __int128 rem = num - (q << k);                 // 0 <= rem < 2^k
    if (rem > half || (rem == half && (q & 1))) ++q;
    return (int64_t)q;
}
int64_t fxp_quantise_z(float v, double z0, double zr, int shift) {
    const float lo = (float)(z0 - zr), hi = (float)(z0 + zr);
    const float c = std::fmin(std::fmax(v, lo), hi);  // NaN -> lo, as v_max_f32 / v_min_f32 do
    return fxp_quantise(c, z0, shift);
}

// ---------------------------------------------------------------------------------
// 3x3 two-sided Jacobi SVD in float: Eigen 3.4.0 JacobiSVD<MatrixX3f>(cov, ComputeFullU)
// as called at reference patchworkpp.cpp:62 (square input -> no QR preconditioner).
// Published algorithm: Eigen/src/SVD/JacobiSVD.h compute() + real_2x2_jacobi_svd(),
// Eigen/src/Jacobi/Jacobi.h makeJacobi() / apply_rotation_in_the_plane().
// a is row-major 3x3; u gets the left singular vectors in columns (row-major 3x3).
// ---------------------------------------------------------------------------------
thread_local long g_max_sweeps = 0;

void jacobi_svd3(const float a[9], float u[9], float sv[3], long *sweeps) {
    const float tiny = FLT_MIN, precision = 2.0f * FLT_EPSILON;
    float scale = 0.0f;
    bool invalid = false;
    for (int k = 0; k < 9; ++k) {
        const float v = f_abs(a[k]);
        if (!(v == v) || v > FLT_MAX) invalid = true;
        if (v > scale) scale = v;
    }
    if (invalid) {  // Eigen: info() == InvalidInput, outputs unspecified; oracle defines NaN
        for (int k = 0; k < 9; ++k) u[k] = NAN;
        sv[0] = sv[1] = sv[2] = NAN;
        return;
    }
    if (scale == 0.0f) scale = 1.0f;
    float w[9];
    for (int k = 0; k < 9; ++k) w[k] = a[k] / scale;
    for (int k = 0; k < 9; ++k) u[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    float max_diag = f_max(f_abs(w[0]), f_max(f_abs(w[4]), f_abs(w[8])));

    for (int sweep = 0; sweep < 1000; ++sweep) {
        bool finished = true;
        if (sweeps) ++*sweeps;
        if (sweep + 1 > g_max_sweeps) g_max_sweeps = sweep + 1;
        for (int p = 1; p < 3; ++p) {
            for (int q = 0; q < p; ++q) {
                const float thr = f_max(tiny, precision * max_diag);
                if (!(f_abs(w[p * 3 + q]) > thr || f_abs(w[q * 3 + p]) > thr)) continue;
                finished = false;
                // real_2x2_jacobi_svd on [[w_pp w_pq],[w_qp w_qq]]
                const float m00 = w[p * 3 + p], m01 = w[p * 3 + q], m10 = w[q * 3 + p], m11 = w[q * 3 + q];
                const float t = m00 + m11, d = m10 - m01;
                float c1, s1;
                if (f_abs(d) < tiny) {
                    s1 = 0.0f;
                    c1 = 1.0f;
                } else {
                    const float r = t / d;
                    const float h = std::sqrt(1.0f + r * r);
                    s1 = 1.0f / h;
                    c1 = r / h;
                }
                const float b00 = c1 * m00 + s1 * m10;
                const float b01 = c1 * m01 + s1 * m11;
                const float b11 = -s1 * m01 + c1 * m11;
                // makeJacobi(b00, b01, b11)
                float cr, sr;
                const float deno = 2.0f * f_abs(b01);
                if (deno < tiny) {
                    cr = 1.0f;
                    sr = 0.0f;
                } else {
                    const float tau = (b00 - b11) / deno;
                    const float ww = std::sqrt(tau * tau + 1.0f);
                    const float tt = (tau > 0.0f) ? 1.0f / (tau + ww) : 1.0f / (tau - ww);
                    const float sign_t = tt > 0.0f ? 1.0f : -1.0f;
                    const float nn = 1.0f / std::sqrt(tt * tt + 1.0f);
                    sr = -sign_t * (b01 / f_abs(b01)) * f_abs(tt) * nn;
                    cr = nn;
                }
                // j_left = rot1 * j_right^T
                const float cl = c1 * cr - s1 * (-sr);
                const float sl = c1 * (-sr) + s1 * cr;
                for (int k = 0; k < 3; ++k) {  // W.applyOnTheLeft(p, q, j_left)
                    const float x = w[p * 3 + k], y = w[q * 3 + k];
                    w[p * 3 + k] = cl * x + sl * y;
                    w[q * 3 + k] = -sl * x + cl * y;
                }
                for (int k = 0; k < 3; ++k) {  // U.applyOnTheRight(p, q, j_left^T)
                    const float x = u[k * 3 + p], y = u[k * 3 + q];
                    u[k * 3 + p] = cl * x - (-sl) * y;
                    u[k * 3 + q] = (-sl) * x + cl * y;
                }
                for (int k = 0; k < 3; ++k) {  // W.applyOnTheRight(p, q, j_right)
                    const float x = w[k * 3 + p], y = w[k * 3 + q];
                    w[k * 3 + p] = cr * x - sr * y;
                    w[k * 3 + q] = sr * x + cr * y;
                }
                max_diag = f_max(max_diag, f_max(f_abs(w[p * 3 + p]), f_abs(w[q * 3 + q])));
            }
        }
        if (finished) break;
    }
    for (int i = 0; i < 3; ++i) {
        const float dgl = w[i * 3 + i];
        sv[i] = f_abs(dgl);
        if (dgl < 0.0f)
            for (int k = 0; k < 3; ++k) u[k * 3 + i] = -u[k * 3 + i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        for (int j = i + 1; j < 3; ++j)
            if (sv[j] > sv[pos]) pos = j;
        if (sv[pos] == 0.0f) break;
        if (pos != i) {
            std::swap(sv[i], sv[pos]);
            for (int k = 0; k < 3; ++k) std::swap(u[k * 3 + i], u[k * 3 + pos]);
        }
    }
}

class Oracle {
public:
    Oracle(const pwo_params &p, int arith) : prm(p), arith_(arith) {
        // reference patchworkpp.h:120-150 (constructor: CZM geometry in double)
        const double mn = prm.min_range, mx = prm.max_range;
        const double z2 = (7 * mn + mx) / 8.0, z3 = (3 * mn + mx) / 4.0, z4 = (mn + mx) / 2.0;
        min_ranges[0] = mn;
        min_ranges[1] = z2;
        min_ranges[2] = z3;
        min_ranges[3] = z4;
        ring_sizes[0] = (z2 - mn) / prm.num_rings_each_zone[0];
        ring_sizes[1] = (z3 - z2) / prm.num_rings_each_zone[1];
        ring_sizes[2] = (z4 - z3) / prm.num_rings_each_zone[2];
        ring_sizes[3] = (mx - z4) / prm.num_rings_each_zone[3];
        for (int k = 0; k < 4; ++k) sector_sizes[k] = 2 * M_PI / prm.num_sectors_each_zone[k];
        int base = 0;
        for (int k = 0; k < 4; ++k) {
            bin_base[k] = base;
            base += prm.num_rings_each_zone[k] * prm.num_sectors_each_zone[k];
        }
        num_bins = base;
        bins.resize((size_t)num_bins);
        std::memset(&plane, 0, sizeof(plane));
        fxp = fxp_geometry(min_ranges, ring_sizes, sector_sizes, prm.num_rings_each_zone, prm.num_sectors_each_zone,
                           prm.max_range, arith == PWO_ARITH_FXP21 ? kFxpV3 : kFxpV4);
    }

    pwo_params prm;  // params_ of the reference; sensor_height / thresholds mutate (:347-350,368)
    std::vector<double> hist_elev[4], hist_flat[4];  // update_elevation_/update_flatness_, patchworkpp.h:174-175
    std::vector<Pt> cloud_ground, cloud_nonground, centers, normals;
    std::vector<pwo_patch_record> records;
    long time_taken = 0;
    long fits = 0, sweeps = 0;
    void get_fxp(int *shift, double *zr, float *ox, float *oy) const {
        *shift = fxp.shift;
        *zr = fxp.zr;
        for (int b = 0; b < num_bins; ++b) {
            if (ox) ox[b] = fxp.ox[(size_t)b];
            if (oy) oy[b] = fxp.oy[(size_t)b];
        }
    }

    // ------------------------------------------------------------------ estimateGround
    void estimate_ground(const float *pts, int n, int cols) {  // reference :151-336
        cloud_ground.clear();
        cloud_nonground.clear();
        records.clear();
        clock_t beg = clock();

        zwork.resize((size_t)n);
        for (int i = 0; i < n; ++i) zwork[(size_t)i] = pts[(size_t)i * cols + 2];
        if (prm.enable_RNR) reflected_noise_removal(pts, n, cols);  // :161
        for (auto &b : bins) b.clear();                                // flush_patches :33-45
        pc2czm(pts, n, cols);                                          // :170

        int concentric_idx = 0;
        centers.clear();
        normals.clear();
        std::vector<Candidate> candidates;
        std::vector<double> ringwise_flatness;
        std::vector<Pt> rg, rng;  // regionwise_ground_, regionwise_nonground_

        for (int zone = 0; zone < prm.num_zones; ++zone) {
            for (int ring = 0; ring < prm.num_rings_each_zone[zone]; ++ring) {
                for (int sector = 0; sector < prm.num_sectors_each_zone[zone]; ++sector) {
                    const int bin = bin_base[zone] + ring * prm.num_sectors_each_zone[zone] + sector;
                    std::vector<Pt> &cell = bins[(size_t)bin];
                    // :191 compares size_t with int -> the int is converted to size_t
                    if (cell.size() < (size_t)prm.num_min_pts) {
                        cloud_nonground.insert(cloud_nonground.end(), cell.begin(), cell.end());
                        continue;
                    }
                    std::sort(cell.begin(), cell.end(), z_less);  // :199
                    cur_bin = bin;
                    z0_set = false;
                    extract_piecewiseground(zone, cell, rg, rng);  // :206

                    centers.push_back(Pt{plane.mean[0], plane.mean[1], plane.mean[2], -1});     // :211
                    normals.push_back(Pt{plane.normal[0], plane.normal[1], plane.normal[2], -1});  // :212

                    // ---- GLE, :217-246
                    const double uprightness = plane.normal[2];
                    const double elevation = plane.mean[2];
                    const double flatness = std::min(plane.sv[0], std::min(plane.sv[1], plane.sv[2]));
                    const double line_variable =
                        plane.sv[1] != 0 ? plane.sv[0] / plane.sv[1] : std::numeric_limits<double>::max();
                    double heading = 0.0;
                    for (int i = 0; i < 3; ++i) heading += plane.mean[i] * plane.normal[i];  // float product, :223

                    const bool is_upright = uprightness > prm.uprightness_thr;
                    const bool is_near = concentric_idx < prm.num_rings_of_interest;
                    const bool heading_outside = heading < 0.0;
                    bool not_elevated = false, is_flat = false;
                    if (concentric_idx < prm.num_rings_of_interest) {
                        not_elevated = elevation < prm.elevation_thr[concentric_idx];
                        is_flat = flatness < prm.flatness_thr[concentric_idx];
                    }
                    if (is_upright && not_elevated && is_near) {  // :253-259
                        hist_elev[concentric_idx].push_back(elevation);
                        hist_flat[concentric_idx].push_back(flatness);
                        ringwise_flatness.push_back(flatness);
                    }

                    pwo_patch_record rec;
                    std::memset(&rec, 0, sizeof(rec));
                    rec.bin = bin;
                    rec.concentric_idx = concentric_idx;
                    rec.n_points = (int)cell.size();
                    rec.n_ground = (int)rg.size();
                    rec.n_nonground = (int)rng.size();
                    for (int i = 0; i < 3; ++i) {
                        rec.mean[i] = plane.mean[i];
                        rec.normal[i] = plane.normal[i];
                        rec.sv[i] = plane.sv[i];
                    }
                    rec.d = plane.d;

                    // ---- decision ladder, :262-284
                    if (!is_upright) {
                        append(cloud_nonground, rg);
                        rec.decision = PWO_DEC_NOT_UPRIGHT;
                    } else if (!is_near) {
                        append(cloud_ground, rg);
                        rec.decision = PWO_DEC_FAR_GROUND;
                    } else if (!heading_outside) {
                        append(cloud_nonground, rg);
                        rec.decision = PWO_DEC_HEADING;
                    } else if (not_elevated || is_flat) {
                        append(cloud_ground, rg);
                        rec.decision = PWO_DEC_GROUND;
                    } else {
                        Candidate c;
                        c.concentric_idx = concentric_idx;
                        c.sector_idx = sector;
                        c.flatness = flatness;
                        c.line_variable = line_variable;
                        c.ground = rg;
                        c.patch_slot = (int)records.size();
                        candidates.push_back(c);
                        rec.decision = PWO_DEC_TGR_REJECT;  // finalised at ring end
                    }
                    append(cloud_nonground, rng);  // :284
                    records.push_back(rec);
                }
                // ---- end of ring, :291-304
                if (!candidates.empty()) {
                    if (prm.enable_TGR) {
                        temporal_ground_revert(ringwise_flatness, candidates, concentric_idx);
                    } else {
                        for (auto &c : candidates) append(cloud_nonground, c.ground);
                    }
                    candidates.clear();
                    ringwise_flatness.clear();
                }
                concentric_idx++;  // :309
            }
        }
        update_elevation_thr();  // :314
        update_flatness_thr();   // :315
        time_taken = clock() - beg;  // :320-321
    }

private:
    int arith_;
    FxpGeom fxp;
    int cur_bin = 0;          // the bin extract_piecewiseground is working on
    bool z0_set = false;      // fxp: z origin of this visit (first LPR, ref :103)
    double z0 = 0.0;
    double min_ranges[4], ring_sizes[4], sector_sizes[4];
    int bin_base[4];
    int num_bins;
    std::vector<std::vector<Pt>> bins;  // ConcentricZoneModel_, flattened zone->ring->sector
    std::vector<float> zwork;           // z column of the by-value copy (tombstones, :394)
    Plane plane;                        // normal_, pc_mean_, singular_values_, d_ (persist across bins/frames)
    std::vector<Pt> ground_pc;          // ground_pc_

    static void append(std::vector<Pt> &dst, const std::vector<Pt> &src) {  // addCloud, :28-31
        dst.insert(dst.end(), src.begin(), src.end());
    }

    // ------------------------------------------------------------------ RNR, :377-400
    void reflected_noise_removal(const float *pts, int n, int cols) {
        if (cols < 4) return;  // ":380 RNR requires intensity information !"
        for (int i = 0; i < n; ++i) {
            const float x = pts[(size_t)i * cols], y = pts[(size_t)i * cols + 1];
            const float zf = zwork[(size_t)i];
            const double r = std::sqrt(x * x + y * y);  // FLOAT products, sum and sqrt (:387), then widened
            const double z = zf;
            const double ver_angle_in_deg = std::atan2(z, r) * 180 / M_PI;  // :389
            if (ver_angle_in_deg < prm.RNR_ver_angle_thr && z < -prm.sensor_height - 0.8 &&
                pts[(size_t)i * cols + 3] < prm.RNR_intensity_thr) {
                cloud_nonground.push_back(Pt{x, y, zf, i});            // :393 original z
                zwork[(size_t)i] = std::numeric_limits<float>::min();  // :394 tombstone
            }
        }
    }

    // ------------------------------------------------------------------ pc2czm, :578-622
    void pc2czm(const float *pts, int n, int cols) {
        const double max_range = prm.max_range, min_range = prm.min_range;
        for (int i = 0; i < n; ++i) {
            const float x = pts[(size_t)i * cols], y = pts[(size_t)i * cols + 1], z = zwork[(size_t)i];
            if (z == std::numeric_limits<float>::min()) continue;  // :591
            const double xd = x, yd = y;
            const double r = std::sqrt(xd * xd + yd * yd);  // xy2radius :573-576, double
            if ((r <= max_range) && (r > min_range)) {
                double theta = std::atan2(yd, xd);  // xy2theta :568-571
                theta = theta > 0 ? theta : 2 * M_PI + theta;
                int zone;
                if (r < min_ranges[1])
                    zone = 0;
                else if (r < min_ranges[2])
                    zone = 1;
                else if (r < min_ranges[3])
                    zone = 2;
                else
                    zone = 3;
                const int ring = std::min(static_cast<int>((r - min_ranges[zone]) / ring_sizes[zone]),
                                          prm.num_rings_each_zone[zone] - 1);
                const int sector = std::min(static_cast<int>(theta / sector_sizes[zone]),
                                            prm.num_sectors_each_zone[zone] - 1);
                bins[(size_t)(bin_base[zone] + ring * prm.num_sectors_each_zone[zone] + sector)].push_back(
                    Pt{x, y, z, i});
            } else {
                cloud_nonground.push_back(Pt{x, y, z, i});  // :618
            }
        }
    }

    // ------------------------------------------------------------------ seeds, :77-149
    void extract_initial_seeds(int zone, const std::vector<Pt> &sorted, std::vector<Pt> &seeds, double th_seed) {
        seeds.clear();
        double sum = 0;
        int cnt = 0;
        size_t init_idx = 0;
        if (zone == 0) {
            for (size_t i = 0; i < sorted.size(); ++i) {
                if (sorted[i].z < prm.adaptive_seed_selection_margin * prm.sensor_height)
                    ++init_idx;
                else
                    break;
            }
        }
        for (size_t i = init_idx; i < sorted.size() && cnt < prm.num_lpr; ++i) {
            sum += sorted[i].z;
            cnt++;
        }
        const double lpr_height = cnt != 0 ? sum / cnt : 0;
        if (!z0_set) {  // fxp contract: the visit's z origin
            z0 = fxp_z_origin(lpr_height);
            z0_set = true;
        }
        for (size_t i = 0; i < sorted.size(); ++i)
            if (sorted[i].z < lpr_height + th_seed) seeds.push_back(sorted[i]);
    }

    // ------------------------------------------------------------------ plane, :47-75
    void estimate_plane(const std::vector<Pt> &g) {
        if (g.empty()) return;  // :49 -> the previous plane stays in force
        ++fits;
        const int n = (int)g.size();
        float mean[3], cov[9];
        if (arith_ == PWO_ARITH_EIGEN_F32) {
            // colwise().mean(): float sum in storage order / float(n)           (:56,60)
            for (int j = 0; j < 3; ++j) {
                float acc = 0.0f;
                for (int i = 0; i < n; ++i) acc += coord(g[(size_t)i], j);
                mean[j] = acc / (float)n;
            }
            // centred^T * centred, float, then / float(double(n-1))               (:56-57)
            const float den = (float)(double)(n - 1);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    float acc = 0.0f;
                    for (int i = 0; i < n; ++i)
                        acc += (coord(g[(size_t)i], a) - mean[a]) * (coord(g[(size_t)i], b) - mean[b]);
                    cov[a * 3 + b] = acc / den;
                }
        } else if (arith_ == PWO_ARITH_EXACT_F64) {
            // reference-neutral arbiter: double accumulation of the unquantised floats (two passes,
            // the products are centred on the double mean), one rounding to float per output
            double md[3];
            for (int j = 0; j < 3; ++j) {
                double acc = 0.0;
                for (int i = 0; i < n; ++i) acc += (double)coord(g[(size_t)i], j);
                md[j] = acc / (double)n;
                mean[j] = (float)md[j];
            }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double acc = 0.0;
                    for (int i = 0; i < n; ++i)
                        acc += ((double)coord(g[(size_t)i], a) - md[a]) * ((double)coord(g[(size_t)i], b) - md[b]);
                    cov[a * 3 + b] = (float)(acc / (double)(n - 1));
                }
        } else if (arith_ == PWO_ARITH_F32_PACKET4) {
            // float, four partial sums (lane = row mod 4, combined pairwise, then the tail): what a
            // 4-wide SIMD reduction forms.  Not a claim about real Eigen's order.
            auto sum4 = [&](auto term) {
                float lane[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const int n4 = n & ~3;
                for (int i = 0; i < n4; i += 4)
                    for (int l = 0; l < 4; ++l) lane[l] += term(i + l);
                float acc = (lane[0] + lane[2]) + (lane[1] + lane[3]);
                for (int i = n4; i < n; ++i) acc += term(i);
                return acc;
            };
            for (int j = 0; j < 3; ++j) mean[j] = sum4([&](int i) { return coord(g[(size_t)i], j); }) / (float)n;
            const float den = (float)(double)(n - 1);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    cov[a * 3 + b] = sum4([&](int i) {
                                         return (coord(g[(size_t)i], a) - mean[a]) * (coord(g[(size_t)i], b) - mean[b]);
                                     }) / den;
        } else if (n <= 3) {
            // contract v3: a fit set of one, two or three points.  The reference's float sums are DETERMINATE there --
            // Eigen reduces fewer elements than one packet sequentially, in storage order (redux of a column, and the
            // coefficient-based lazy product it takes for 3 x n times n x 3 with n this small), and two terms commute --
            // so the contract follows the reference's own float arithmetic (:56-60) instead of the grid: mean = float
            // sum / n, centred rows, products summed in float, / (n - 1).  Order: the reference's z-sorted bin (:199);
            // equal heights (std::sort leaves them to libstdc++) in cloud order.
            Pt q[3];
            for (int i = 0; i < n; ++i) q[i] = g[(size_t)i];
            for (int i = 1; i < n; ++i)
                for (int j = i; j > 0 && (q[j].z < q[j - 1].z || (!(q[j - 1].z < q[j].z) && q[j].idx < q[j - 1].idx)); --j)
                    std::swap(q[j], q[j - 1]);
            for (int j = 0; j < 3; ++j) {
                float acc = 0.0f;
                for (int i = 0; i < n; ++i) acc += coord(q[i], j);
                mean[j] = acc / (float)n;
            }
            const float den = (float)(double)(n - 1);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    float acc = 0.0f;
                    for (int i = 0; i < n; ++i) acc += (coord(q[i], a) - mean[a]) * (coord(q[i], b) - mean[b]);
                    cov[a * 3 + b] = acc / den;
                }
        } else {
            // the fixed-point contract (top of this file): exact integer moments around the bin's origin
            const double org[3] = {(double)fxp.ox[(size_t)cur_bin], (double)fxp.oy[(size_t)cur_bin], z0};
            int64_t s1[3] = {0, 0, 0};
            __int128 s2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < n; ++i) {
                const Pt &p = g[(size_t)i];
                const int64_t q[3] = {fxp_quantise(p.x, org[0], fxp.shift), fxp_quantise(p.y, org[1], fxp.shift),
                                      fxp_quantise_z(p.z, z0, fxp.zr, fxp.shift)};
                for (int a = 0; a < 3; ++a) {
                    s1[a] += q[a];
                    for (int b = 0; b < 3; ++b) s2[a * 3 + b] += (__int128)q[a] * (__int128)q[b];
                }
            }
            const double inv = 1.0 / fxp.scale;
            // (contract v3: ONE reciprocal for the means and one for the covariance entries, both in double -- the product
            // with a correctly rounded reciprocal is within an ulp of a double of the quotient, 2^-29 of a float ulp)
            const double rn = 1.0 / (double)n, rd = 1.0 / ((double)n * (double)(n - 1));
            for (int a = 0; a < 3; ++a) {
                mean[a] = (float)(((double)s1[a] * rn) * inv + org[a]);
                for (int b = 0; b < 3; ++b) {
                    const __int128 num = (__int128)n * s2[a * 3 + b] - (__int128)s1[a] * (__int128)s1[b];
                    cov[a * 3 + b] = (float)(((double)num * rd) * (inv * inv));
                }
            }
        }
        for (int j = 0; j < 3; ++j) plane.mean[j] = mean[j];  // pc_mean_, :59-60

        float u[9];
        jacobi_svd3(cov, u, plane.sv, &sweeps);  // :62-63
        for (int j = 0; j < 3; ++j) plane.normal[j] = u[j * 3 + 2];  // U.col(2), :66
        if (plane.normal[2] < 0)
            for (int j = 0; j < 3; ++j) plane.normal[j] *= -1;  // :68
        // d_ = -(normal^T * mean)(0,0): float dot product widened to double, :74
        const float dot = plane.normal[0] * mean[0] + plane.normal[1] * mean[1] + plane.normal[2] * mean[2];
        plane.d = -dot;
    }
    static float coord(const Pt &p, int j) { return j == 0 ? p.x : (j == 1 ? p.y : p.z); }

    // :551-554 -- three float products, two float adds, one double add
    double point_to_plane(const Pt &p) const {
        return plane.normal[0] * p.x + plane.normal[1] * p.y + plane.normal[2] * p.z + plane.d;
    }

    // ------------------------------------------------------------------ R-VPF + R-GPF, :467-549
    void extract_piecewiseground(int zone, const std::vector<Pt> &src, std::vector<Pt> &dst,
                                 std::vector<Pt> &non_ground_dst) {
        ground_pc.clear();
        dst.clear();
        non_ground_dst.clear();
        std::vector<Pt> work = src;  // src_wo_verticals

        if (prm.enable_RVPF) {  // :482-508
            for (int i = 0; i < prm.num_iter; ++i) {
                extract_initial_seeds(zone, work, ground_pc, prm.th_seeds_v);
                estimate_plane(ground_pc);
                if (zone == 0 && plane.normal[2] < prm.uprightness_thr) {
                    std::vector<Pt> tmp;
                    tmp.swap(work);
                    for (const Pt &p : tmp) {
                        const double dist = point_to_plane(p);
                        if (std::abs(dist) < prm.th_dist_v)
                            non_ground_dst.push_back(p);
                        else
                            work.push_back(p);
                    }
                } else {
                    break;
                }
            }
        }
        extract_initial_seeds(zone, work, ground_pc, prm.th_seeds);  // :513
        estimate_plane(ground_pc);                                   // :514
        for (int i = 0; i < prm.num_iter; ++i) {                     // :516-543
            ground_pc.clear();
            for (const Pt &p : work) {
                const double dist = point_to_plane(p);
                if (i < prm.num_iter - 1) {
                    if (dist < prm.th_dist) ground_pc.push_back(p);
                } else {
                    if (dist < prm.th_dist)
                        dst.push_back(p);
                    else
                        non_ground_dst.push_back(p);
                }
            }
            if (i < prm.num_iter - 1)
                estimate_plane(ground_pc);
            else
                estimate_plane(dst);
        }
    }

    // ------------------------------------------------------------------ :557-566
    static void calc_mean_stdev(const std::vector<double> &v, double &mean, double &stdev) {
        if (v.size() <= 1) return;
        mean = std::accumulate(v.begin(), v.end(), 0.0) / v.size();
        for (size_t i = 0; i < v.size(); ++i) stdev += (v[i] - mean) * (v[i] - mean);
        stdev /= v.size() - 1;
        stdev = std::sqrt(stdev);
    }

    // ------------------------------------------------------------------ TGR, :402-464
    void temporal_ground_revert(const std::vector<double> &ring_flatness, const std::vector<Candidate> &cands,
                                int concentric_idx) {
        double mean_flatness = 0.0, stdev_flatness = 0.0;
        calc_mean_stdev(ring_flatness, mean_flatness, stdev_flatness);
        for (const Candidate &c : cands) {
            const double mu = mean_flatness + 1.5 * stdev_flatness;
            double prob_flatness = 1 / (1 + std::exp((c.flatness - mu) / (mu / 10)));
            if (c.ground.size() > 1500 && c.flatness < prm.th_dist * prm.th_dist) prob_flatness = 1.0;
            double prob_line = 1.0;
            if (c.line_variable > 8.0) prob_line = 0.0;
            const bool revert = prob_line * prob_flatness > 0.5;
            if (concentric_idx < prm.num_rings_of_interest) {
                if (revert) {
                    append(cloud_ground, c.ground);
                    records[(size_t)c.patch_slot].decision = PWO_DEC_TGR_REVERT;
                } else {
                    append(cloud_nonground, c.ground);
                }
            }
        }
    }

    // ------------------------------------------------------------------ :338-357
    void update_elevation_thr() {
        for (int i = 0; i < prm.num_rings_of_interest; ++i) {
            if (hist_elev[i].empty()) continue;
            double m = 0.0, s = 0.0;
            calc_mean_stdev(hist_elev[i], m, s);
            if (i == 0) {
                prm.elevation_thr[i] = m + 3 * s;
                prm.sensor_height = -m;
            } else {
                prm.elevation_thr[i] = m + 2 * s;
            }
            const int exceed = (int)hist_elev[i].size() - prm.max_elevation_storage;
            if (exceed > 0) hist_elev[i].erase(hist_elev[i].begin(), hist_elev[i].begin() + exceed);
        }
    }
    // ------------------------------------------------------------------ :359-375
    void update_flatness_thr() {
        for (int i = 0; i < prm.num_rings_of_interest; ++i) {
            if (hist_flat[i].empty()) break;
            if (hist_flat[i].size() <= 1) break;
            double m = 0.0, s = 0.0;
            calc_mean_stdev(hist_flat[i], m, s);
            prm.flatness_thr[i] = m + s;
            const int exceed = (int)hist_flat[i].size() - prm.max_flatness_storage;
            if (exceed > 0) hist_flat[i].erase(hist_flat[i].begin(), hist_flat[i].begin() + exceed);
        }
    }
};

thread_local long g_fits = 0, g_sweeps = 0;

void copy_xyz(const std::vector<Pt> &v, float *out) {
    for (size_t i = 0; i < v.size(); ++i) {
        out[i * 3] = v[i].x;
        out[i * 3 + 1] = v[i].y;
        out[i * 3 + 2] = v[i].z;
    }
}

}  // namespace

extern "C" {

void pwo_default_params(pwo_params *p) {  // reference patchworkpp.h:79-111
    std::memset(p, 0, sizeof(*p));
    p->verbose = 0;
    p->enable_RNR = 1;
    p->enable_RVPF = 1;
    p->enable_TGR = 1;
    p->num_iter = 3;
    p->num_lpr = 20;
    p->num_min_pts = 10;
    p->num_zones = 4;
    p->num_rings_of_interest = 4;
    p->RNR_ver_angle_thr = -15.0;
    p->RNR_intensity_thr = 0.2;
    p->sensor_height = 1.723;
    p->th_seeds = 0.125;
    p->th_dist = 0.125;
    p->th_seeds_v = 0.25;
    p->th_dist_v = 0.1;
    p->max_range = 80.0;
    p->min_range = 2.7;
    p->uprightness_thr = 0.707;
    p->adaptive_seed_selection_margin = -1.2;
    const int sectors[4] = {16, 32, 54, 32}, rings[4] = {2, 4, 4, 4};
    for (int k = 0; k < 4; ++k) {
        p->num_sectors_each_zone[k] = sectors[k];
        p->num_rings_each_zone[k] = rings[k];
        p->elevation_thr[k] = 0;
        p->flatness_thr[k] = 0;
    }
    p->max_flatness_storage = 1000;
    p->max_elevation_storage = 1000;
}

int pwo_arith_supported(int arith) { return arith >= PWO_ARITH_EIGEN_F32 && arith <= PWO_ARITH_FXP21; }

void *pwo_create(const pwo_params *p, int arith) {
    if (!pwo_arith_supported(arith)) return nullptr;
    return new Oracle(*p, arith);
}
void pwo_destroy(void *h) { delete (Oracle *)h; }

int pwo_estimate_ground(void *h, const float *pts, int n, int cols) {
    Oracle *o = (Oracle *)h;
    const long f0 = o->fits, s0 = o->sweeps;
    o->estimate_ground(pts, n, cols);
    g_fits += o->fits - f0;
    g_sweeps += o->sweeps - s0;
    return 0;
}

int pwo_num_ground(void *h) { return (int)((Oracle *)h)->cloud_ground.size(); }
int pwo_num_nonground(void *h) { return (int)((Oracle *)h)->cloud_nonground.size(); }
int pwo_num_patches(void *h) { return (int)((Oracle *)h)->centers.size(); }
void pwo_get_ground_indices(void *h, int32_t *out) {
    const auto &v = ((Oracle *)h)->cloud_ground;
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i].idx;
}
void pwo_get_nonground_indices(void *h, int32_t *out) {
    const auto &v = ((Oracle *)h)->cloud_nonground;
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i].idx;
}
void pwo_get_ground(void *h, float *out) { copy_xyz(((Oracle *)h)->cloud_ground, out); }
void pwo_get_nonground(void *h, float *out) { copy_xyz(((Oracle *)h)->cloud_nonground, out); }
void pwo_get_centers(void *h, float *out) { copy_xyz(((Oracle *)h)->centers, out); }
void pwo_get_normals(void *h, float *out) { copy_xyz(((Oracle *)h)->normals, out); }
double pwo_get_height(void *h) { return ((Oracle *)h)->prm.sensor_height; }
double pwo_get_time_taken(void *h) { return (double)((Oracle *)h)->time_taken; }

void pwo_get_thresholds(void *h, double *sensor_height, double *elev, double *flat) {
    Oracle *o = (Oracle *)h;
    *sensor_height = o->prm.sensor_height;
    for (int k = 0; k < 4; ++k) {
        elev[k] = o->prm.elevation_thr[k];
        flat[k] = o->prm.flatness_thr[k];
    }
}
int pwo_get_history_len(void *h, int which, int ring) {
    Oracle *o = (Oracle *)h;
    return (int)(which == 0 ? o->hist_elev[ring].size() : o->hist_flat[ring].size());
}
void pwo_get_history(void *h, int which, int ring, double *out) {
    Oracle *o = (Oracle *)h;
    const std::vector<double> &v = which == 0 ? o->hist_elev[ring] : o->hist_flat[ring];
    std::copy(v.begin(), v.end(), out);
}
void pwo_get_counters(long *plane_fits, long *jacobi_sweeps) {
    *plane_fits = g_fits;
    *jacobi_sweeps = g_sweeps;
}

int pwo_ext_num_records(void *h) { return (int)((Oracle *)h)->records.size(); }
void pwo_ext_get_records(void *h, pwo_patch_record *out) {
    const auto &r = ((Oracle *)h)->records;
    std::copy(r.begin(), r.end(), out);
}
void pwo_ext_set_state(void *h, double sensor_height, const double *elev, const double *flat) {
    Oracle *o = (Oracle *)h;
    o->prm.sensor_height = sensor_height;
    for (int k = 0; k < 4; ++k) {
        o->prm.elevation_thr[k] = elev[k];
        o->prm.flatness_thr[k] = flat[k];
    }
}
void pwo_ext_jacobi(const float *cov9, float *u9, float *sv3) { jacobi_svd3(cov9, u9, sv3, nullptr); }
void pwo_ext_fxp_geometry(void *h, int *shift, double *zr, float *ox, float *oy) {
    ((Oracle *)h)->get_fxp(shift, zr, ox, oy);
}
long pwo_ext_max_sweeps(int reset) {
    const long v = g_max_sweeps;
    if (reset) g_max_sweeps = 0;
    return v;
}
long long pwo_ext_quantise(float v, double origin, int shift) { return fxp_quantise(v, origin, shift); }
long long pwo_ext_quantise_z(float v, double z0, int shift, double zr) {
    return fxp_quantise_z(v, z0, zr, shift);
}
double pwo_ext_z_origin(double lpr) { return fxp_z_origin(lpr); }

double pwo_bench(const pwo_params *p, int arith, const float *const *frames, const int *n_points, int cols,
                 int num_distinct, int total, int threads, double *sum_call_seconds) {
    if (!pwo_arith_supported(arith) || threads < 1) return -1.0;
    std::vector<double> call_s((size_t)threads, 0.0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            for (int i = t; i < total; i += threads) {
                Oracle o(*p, arith);  // fresh state per frame
                const int k = i % num_distinct;
                auto a = std::chrono::steady_clock::now();
                o.estimate_ground(frames[k], n_points[k], cols);
                auto b = std::chrono::steady_clock::now();
                call_s[(size_t)t] += std::chrono::duration<double>(b - a).count();
            }
        });
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    if (sum_call_seconds) *sum_call_seconds = std::accumulate(call_s.begin(), call_s.end(), 0.0);
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
