// pwpp_kernels.hip -- the Patchwork++ estimateGround() hot path as hand-written HIP for
// gfx950 (MI355X, CDNA4: wave64, 256 CUs in 8 XCDs, 160 KiB LDS/CU, HBM3E).
//
// One batch of F independent frames goes through seven or eight launches; every launch covers
// all frames, so a 1024-frame batch is 7 launches, not 7168, and nothing else is enqueued per call:
//
//   K0  k_clear            zeroes the histogram / cursor slabs and the frame counters
//   K1' k_czm_bin_scatter  RNR + CZM code per point, straight into the bin's FIXED segment   (ref :377-400, :578-622)
//       (or K1 k_czm_bin + K3 k_czm_scatter: histogram, then scatter -- the exact two-pass path
//        for <= 4 frames and for the redo after a segment overflow)
//   K2  k_czm_scan         part / bin offsets (a near-zone bin = a low and a high part, pwpp_dev.h), patches sorted into size buckets
//   K4  k_fit_*            per patch: LPR seeds, R-VPF, R-GPF, final plane       (ref :77-149, :47-75, :467-554)
//                          one or two launches by patch size, see pwpp_fit.hip
//   K5  k_gle_tgr          per frame: GLE ladder, A-GLE history, TGR, thresholds (ref :211-309, :338-375, :402-464);
//                          the object's plane members after the frame (PwppPlaneState)
//   K6  k_emit             ground / non-ground index lists                       (ref :28-31, :18-26);
//                          mirrors the frame counters into pinned host memory
//   K7  k_order_sublists   optional: the reference's order inside every part of the lists
//
// All reference citations are /root/reference/cpp/patchworkpp/src/patchworkpp.cpp unless a
// header is named.  This is integer + scalar-float work bound by HBM traffic (binning, emit) and by
// VALU issue (plane fits): no MFMA anywhere (3x3 covariances); the levers are coalesced 16 B/lane
// traffic, LDS-staged atomics, XCD-aware grids and as few passes over a patch as the chain allows.
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
#include <type_traits>

#include "pwpp_common.hpp"

namespace {

constexpr int kBlock = 256;          // 4 waves

constexpr int kPtsPerBlock = 1024;   // K1/K3: 4 points per thread, 16 KiB of input per workgroup

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

// Cross-lane moves on the VALU (DPP) instead of through the LDS crossbar (ds_bpermute): lane i reads lane
// i - 1 of the wave (lane 0 gets `first`), and the inclusive prefix sum of a wave in six adds
// (row_shr 1/2/4/8 inside the 16-lane rows, then the rows' last lanes broadcast into the rows behind).
__device__ __forceinline__ unsigned wave_prev_lane(unsigned x, unsigned first) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31
    return (unsigned)x;
}

// Consecutive points of a scan mostly fall into the same bin, so a wave's LDS atomics pile up
// on one address (PMC: 78 % of the LDS cycles of k_czm_bin were bank-conflict cycles).  Lanes
// are grouped into runs of equal code; only the first lane of a run touches the LDS counter,
// with the run length.  Returns the old counter value for the run (valid in every lane of the
// run) and this lane's position inside the run.
__device__ __forceinline__ unsigned wave_run_add(unsigned *counters, unsigned code, bool active, unsigned &pos_in_run) {
    const int ln = lane_id();
    const unsigned key = active ? code : (0x80000000u | (unsigned)ln);  // inactive lanes never join a run
    const unsigned prev = wave_prev_lane(key, ~key);
    const bool head = key != prev;  // (lane 0 reads ~key: always a head)
    const unsigned long long heads = __ballot(head);
    const unsigned long long le = (ln == 63) ? ~0ull : ((1ull << (ln + 1)) - 1ull);
    const int h = 63 - __clzll((long long)(heads & le));            // first lane of my run
    const unsigned long long above = (h == 63) ? 0ull : (heads >> (h + 1));
    const int len = above ? (__ffsll((long long)above)) : (64 - h);  // distance to the next head
    unsigned old = 0;
    if (head && active) old = atomicAdd(&counters[code], (unsigned)len);
    old = (unsigned)__shfl((int)old, h, 64);
    pos_in_run = (unsigned)(ln - h);
    return old;
}

// Histogram flavour of wave_run_add (K1): only the first lane of a run needs anything -- the run
// length, i.e. the distance to the next run head.
__device__ __forceinline__ void wave_run_count(unsigned *counters, unsigned code, bool active) {
    const int ln = lane_id();
    const unsigned key = active ? code : (0x80000000u | (unsigned)ln);
    const unsigned prev = wave_prev_lane(key, ~key);
    const bool head = key != prev;  // (lane 0 reads ~key: always a head)
    const unsigned long long heads = __ballot(head);
    if (head && active) {
        const unsigned long long above = (heads >> ln) >> 1;  // run heads after this lane
        const int len = above ? __ffsll((long long)above) : 64 - ln;
        atomicAdd(&counters[code], (unsigned)len);
    }
}

// ------------------------------------------------------------------------------------------
// K1  RNR + CZM code + histogram
// ------------------------------------------------------------------------------------------
// pc2czm, ref :593-615, exactly as the reference evaluates it: all double
__device__ __forceinline__ unsigned bin_code_exact(const PwppDevParams &P, float x, float y) {
    const unsigned B = (unsigned)P.num_bins;
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

// The same bin in ~60 float instructions, for the points that are provably not near any
// decision boundary of the double computation (>99.9 % of a scan).  The exact path costs ~220
// mostly double instructions per point and made k_czm_bin VALU-bound at twice its HBM time.
//   radius: v_sqrt_f32 of a float sum of squares, relative error < 3e-7 (1.6e-5 m at 80 m);
//   angle : octant reduction + a degree-13 odd polynomial for atan on [0,1] (max error 3.3e-7 rad
//           in float arithmetic, coefficients fitted for this file) + v_rcp_f32, < 1e-6 rad in all.
// A point further than f_margin_r (>= 5x the radius error) from every range / zone / ring boundary
// and further than f_margin_t (>= 8x the angle error) from every sector boundary lands in the same
// bin in double.  Everything else -- and the exact directions of czm_atan2 -- returns false and
// takes bin_code_exact.
__device__ __forceinline__ bool bin_code_fast(const PwppDevParams &P, const float4 *zt, float x, float y, unsigned &code) {
    const float ax = fabsf(x), ay = fabsf(y);
    if (!(ax > 0.0f) || !(ay > 0.0f) || ax == ay) return false;  // axes, diagonals, NaN
    const float rf = __builtin_amdgcn_sqrtf(__builtin_fmaf(x, x, y * y));
    const float lo = P.f_min_range, hi = P.f_max_range, mr = P.f_margin_r;
    if (!(rf > lo) || !(rf <= hi)) {
        code = PWPP_CODE_OOR((unsigned)P.num_bins);
        return (rf <= lo - mr) || (rf >= hi + mr);  // NaN: not sure
    }
    const int k = (rf >= P.f_zone[1] ? 1 : 0) + (rf >= P.f_zone[2] ? 1 : 0) + (rf >= P.f_zone[3] ? 1 : 0);
    // the zone's constants come from a 128-byte LDS table (two ds_read_b128): as selects on the kernel
    // arguments they cost ~45 VALU instructions and five per-lane global loads per point
    const float4 za = zt[2 * k], zb = zt[2 * k + 1];
    const float zmin = za.x, inv_ring = za.y, inv_sec = za.z;
    const int nring = __float_as_int(za.w), nsec = __float_as_int(zb.x), base = __float_as_int(zb.y);
    // ring (zone and range boundaries are ring boundaries too)
    const float rq = (rf - zmin) * inv_ring;
    const float rfl = floorf(rq);
    const float rfrac = rq - rfl;
    const float eps_r = mr * inv_ring;
    bool unsure = rfrac < eps_r || rfrac > 1.0f - eps_r;
    // angle in (0, 2 pi)
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float t2 = t * t;
    float pz = 0.006811772007495165f;
    pz = __builtin_fmaf(pz, t2, -0.03360415995121002f);
    pz = __builtin_fmaf(pz, t2, 0.07962360978126526f);
    pz = __builtin_fmaf(pz, t2, -0.13233338296413422f);
    pz = __builtin_fmaf(pz, t2, 0.19807815551757812f);
    pz = __builtin_fmaf(pz, t2, -0.3331736922264099f);
    pz = __builtin_fmaf(pz, t2, 0.9999961256980896f);
    float a = pz * t;                                   // atan(mn / mx) in [0, pi/4]
    a = ay > ax ? 1.57079632679489661923f - a : a;      // first quadrant
    a = x < 0.0f ? 3.14159265358979323846f - a : a;     // upper half plane
    a = y < 0.0f ? 6.28318530717958647692f - a : a;     // ref :570: negative atan2 + 2 pi
    const float sq = a * inv_sec;
    const float sfl = floorf(sq);
    const float sfrac = sq - sfl;
    const float eps_s = __builtin_fmaf(sq, 2.5e-7f, P.f_margin_t * inv_sec);
    unsure = unsure || sfrac < eps_s || sfrac > 1.0f - eps_s;
    const int ring = min((int)rfl, nring - 1);
    const int sector = min((int)sfl, nsec - 1);
    code = (unsigned)(base + ring * nsec + sector);
    return !unsure;
}

// zone table of bin_code_fast: {zmin, 1/ring size, 1/sector size, rings}, {sectors, first bin, -, -} per zone
__device__ __forceinline__ void fill_zone_table(const PwppDevParams &P, float4 *zt) {
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        zt[2 * k] = make_float4(P.f_zone[k], P.f_inv_ring[k], P.f_inv_sector[k], __int_as_float(P.rings[k]));
        zt[2 * k + 1] = make_float4(__int_as_float(P.sectors[k]), __int_as_float(P.bin_base[k]), 0.0f, 0.0f);
    }
}

// z as the bin-ordered plane holds it: a NaN of the cloud becomes THE quiet NaN 0x7fc00000, because the fit
// kernels mark the points an R-VPF round removes with 0x7fc00000 | round (pwpp_fit.hip, strip_point).  A NaN z
// never enters a seed or ground set either way (every test it takes part in is false).
__device__ __forceinline__ float binned_z(float z) { return z != z ? __uint_as_float(0x7fc00000u) : z; }

__device__ __forceinline__ unsigned czm_code(const PwppDevParams &P, const float4 *zt, float x, float y, float z, float inten,
                                             bool has_intensity, double sensor_height, float rnr_z_guard, bool exact_only) {
    const unsigned B = (unsigned)P.num_bins;
    // Reflected Noise Removal, ref :385-396.  r is FLOAT there (:387), the rest double.
    // rnr_z_guard = float(-sensor_height - 0.8) + 1e-3: a float pre-test that can only say "no"
    if (P.enable_RNR && has_intensity && z < rnr_z_guard) {
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
    unsigned code = 0;
    if (!exact_only && bin_code_fast(P, zt, x, y, code)) return code;
    return bin_code_exact(P, x, y);
}

// the PART a point goes to (pwpp_dev.h): the low or the high part of its bin, by its height over the ground level
// the frame's adaptive state reports (hi_split_z, pwpp_common.hpp: the fit kernels compute the same value)
__device__ __forceinline__ unsigned part_of(const PwppDevParams &P, unsigned code, float z, float zs) {
    if (code == PWPP_CODE_DROP) return code;
    const unsigned B = (unsigned)P.num_bins;
    // Only the first split_end bins are split (the near zone: the big bins, where the fit passes are bound by the points
    // they stream -- the small patches further out are bound by their solves, and twice as many non-empty parts cost the
    // one-pass binning more than they save).  A NaN z goes to the high part: every key of a low part is then below every
    // key of its high part (srow_lpr, pwpp_fit.hip).
    if (code < (unsigned)P.split_end) return 2u * code + (z < zs ? 0u : 1u);
    return code < B ? 2u * code : code + B;
}

__global__ __launch_bounds__(kBlock) void k_czm_bin(PwppBatch Bt) {
    extern __shared__ unsigned s_dyn[];  // [parts of this model] -- sized at launch (binning_lds_bytes): 4 KB for the default
    unsigned *s_hist = s_dyn;            // model instead of the 16 KB of the largest one
    __shared__ float4 s_zt[8];
    const int f = blockIdx.y;
    const PwppFrameDesc fd = Bt.frames[f];
    const int first = blockIdx.x * kPtsPerBlock;
    if (first >= fd.n) return;
    const PwppDevParams &P = Bt.P;
    const int NB = PWPP_NUM_PARTS(P.num_bins);
    for (int b = threadIdx.x; b < NB; b += kBlock) s_hist[b] = 0;
    fill_zone_table(P, s_zt);
    __syncthreads();
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const float rnr_z_guard = (float)(-sensor_height - 0.8) + 1e-3f;
    const float zs = hi_split_z(P, sensor_height);
    uint16_t *codes = Bt.codes + fd.base;
    unsigned dropped = 0;
    unsigned pcode[kPtsPerBlock / kBlock];
#pragma unroll
    for (int j = 0; j < kPtsPerBlock / kBlock; ++j) {
        const int i = first + j * kBlock + threadIdx.x;
        pcode[j] = PWPP_CODE_DROP;
        if (i < fd.n) {
            float x, y, z, w;
            load_point(fd, i, x, y, z, w);
            const unsigned code = part_of(P, czm_code(P, s_zt, x, y, z, w, fd.cols >= 4, sensor_height, rnr_z_guard, (Bt.debug & 16) != 0), z, zs);
            codes[i] = (uint16_t)code;
            if (code == PWPP_CODE_DROP) ++dropped;
            pcode[j] = code;
        }
    }
#pragma unroll
    for (int j = 0; j < kPtsPerBlock / kBlock; ++j) {
        wave_run_count(s_hist, pcode[j], pcode[j] != PWPP_CODE_DROP);
    }
    __syncthreads();
    unsigned *gcount = Bt.part_count + (size_t)f * NB;
    for (int b = threadIdx.x; b < NB; b += kBlock) {
        const unsigned c = s_hist[b];
        if (c) atomicAdd(&gcount[b], c);  // one global atomic per non-empty part per 1024 points
    }
    dropped = wave_sum_u32(dropped);
    if (lane_id() == 0 && dropped) atomicAdd((unsigned *)&Bt.results[f].n_dropped, dropped);
}

// ------------------------------------------------------------------------------------------
// K2  exclusive scan of the per-frame histogram
// ------------------------------------------------------------------------------------------
// The body of K2 for frame f (yblock 0: the scan; 1..8: the snapshot workgroups of a stream frame).  A kernel of its own
// (k_czm_scan, below K1') for batches; for fewer than eight frames the LAST workgroup of K1' to finish a frame runs it in place
// (FUSED: no kernel boundary between binning and scan, 2.5-3 us of a single frame's chain).
template <bool FUSED>
__device__ __forceinline__ void czm_scan_frame(const PwppBatch &Bt, const int f, const int yblock) {
    __shared__ unsigned s_part[kBlock];
    // the frame's part counts and offsets stay in LDS for the second half of the kernel (reading back what other
    // threads just wrote to global memory is a round trip of its own, and a single frame waits for this chain)
    __shared__ unsigned s_pc[PWPP_NUM_PARTS(PWPP_MAX_BINS)], s_po[PWPP_NUM_PARTS(PWPP_MAX_BINS)];
    if (yblock >= 1) {
        // One-pass binning of stateful streams (pwpp_dev.h, snap_*): eight more workgroups per frame copy the stream's state as
        // it is before this call -- what a redo after a segment overflow starts from -- beside the scan, off its chain: one
        // history each (a single workgroup took 9.5 us for the eight 1000-entry histories, the scan takes 5.8), the first of
        // them the scalars and the plane members as well.
        if (!Bt.snap_scalar) return;
        const int st = Bt.frames[f].state_in;
        if (st < 0) return;
        static_assert(sizeof(PwppStateScalar) % 4 == 0 && sizeof(PwppPlaneState) % 4 == 0 &&
                      offsetof(PwppStateScalar, flat_len) == offsetof(PwppStateScalar, elev_len) + 16, "copied word by word; lengths read as one array");
        const unsigned *ss = reinterpret_cast<const unsigned *>(Bt.st_scalar + st), *ps = reinterpret_cast<const unsigned *>(Bt.st_plane + st);
        const int w = yblock - 1;  // this workgroup's history: elevation of ring 0-3, flatness of ring 0-3
        if (w == 0) {
            if (threadIdx.x < sizeof(PwppStateScalar) / 4) reinterpret_cast<unsigned *>(Bt.snap_scalar + st)[threadIdx.x] = ss[threadIdx.x];
            if (threadIdx.x < sizeof(PwppPlaneState) / 4) reinterpret_cast<unsigned *>(Bt.snap_plane + st)[threadIdx.x] = ps[threadIdx.x];
        }
        const int *lens = Bt.st_scalar[st].elev_len;  // elev_len[4], flat_len[4]
        int len = lens[w];
        len = len < Bt.P.hist_cap ? len : Bt.P.hist_cap;
        const size_t row = ((size_t)st * 8 + (size_t)w) * (size_t)Bt.P.hist_cap;
        for (int i0 = 0; i0 < len; i0 += 4 * kBlock) {  // (four entries per thread in flight)
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * kBlock + (int)threadIdx.x;
                v[q] = i < len ? Bt.st_hist[row + i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * kBlock + (int)threadIdx.x;
                if (i < len) Bt.snap_hist[row + i] = v[q];
            }
        }
        return;
    }
    const int B = Bt.P.num_bins, NB = B + 2, NP = PWPP_NUM_PARTS(B);
    int probe_i = 16;  // timing probes (debug_flags & 8): slots 16.. of the probe array (tools/k5_chain.py)
    auto probe = [&]() {
        if ((Bt.debug & 8) && f == 0 && threadIdx.x == 0 && probe_i < 32) Bt.dbg[probe_i++] = wall_clock64();
    };
    probe();
    unsigned *pcnt = Bt.part_count + (size_t)f * NP;
    unsigned *poff = Bt.part_off + (size_t)f * NP;
    // the parts that outgrew their segments (overflow arena, pwpp_dev.h): up to PWPP_MAX_RELOC of them are moved into the arena
    __shared__ unsigned s_nov, s_reloc_fail;
    __shared__ unsigned short s_ovp[PWPP_MAX_RELOC];
    __shared__ unsigned s_ovoff[PWPP_MAX_RELOC];
    const bool arena = !FUSED && Bt.cap_off && Bt.arena_slots > 0u;
    if (threadIdx.x == 0) s_nov = s_reloc_fail = 0u;
    if (arena) __syncthreads();
    if (Bt.cap_off) {  // one-pass binning: fixed segments; a part never reports more points than its segment holds
        // four parts per thread at a time, their loads (segment table, count, observed maximum) all in flight before the first
        // is used: as a plain loop every iteration was an L2 round trip of its own (4 x 0.3 us of a single frame's chain, and
        // as much again for the maxima, which are now part of this loop)
        for (int p0 = 0; p0 < NP; p0 += 4 * kBlock) {
            unsigned seg[4], nxt[4], c[4], mx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = p0 + q * kBlock + (int)threadIdx.x;
                seg[q] = nxt[q] = c[q] = mx[q] = 0u;
                if (p < NP) {
                    seg[q] = Bt.cap_off[p];
                    nxt[q] = Bt.cap_off[p + 1];
                    // (FUSED: the counts are other workgroups' device-scope atomics of THIS kernel -- read them where those were performed)
                    c[q] = FUSED ? __hip_atomic_load(&pcnt[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pcnt[p];
                    mx[q] = Bt.bin_max[p];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = p0 + q * kBlock + (int)threadIdx.x;
                if (p >= NP) continue;
                const unsigned cap = nxt[q] - seg[q];
                poff[p] = seg[q];
                // what the host sizes the segments of the NEXT batches from: the part's TRUE count (the binning kernel counts every
                // point it meets, stored or not), so the table is right after ONE overflow, redo or not
                if (c[q] > mx[q]) atomicMax(&Bt.bin_max[p], c[q]);
                if (c[q] > cap) {
                    if (arena) {  // its points beyond the segment are in the arena: the part is moved there as a whole below
                        const unsigned k = atomicAdd(&s_nov, 1u);
                        if (k < PWPP_MAX_RELOC) s_ovp[k] = (unsigned short)p;
                    } else {
                        c[q] = cap;
                        pcnt[p] = cap;
                        atomicOr((unsigned *)&Bt.results[f].overflow, 1u);  // (other workgroups of a fused launch touch this word with atomics)
                    }
                }
                s_pc[p] = c[q];
                s_po[p] = seg[q];
            }
        }
    } else {
        // as few consecutive parts per thread as cover the model (4 for the default 1010 parts): all four waves busy
        constexpr int kPer = (PWPP_NUM_PARTS(PWPP_MAX_BINS) + kBlock - 1) / kBlock;
        const int per = (NP + kBlock - 1) / kBlock;
        unsigned local[kPer];
        unsigned sum = 0;
        const int p0 = threadIdx.x * per;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int p = p0 + j;
            const unsigned c = (j < per && p < NP) ? pcnt[p] : 0u;
            if (j < per && p < NP) s_pc[p] = c;
            local[j] = (c + (PWPP_SLOT_ALIGN - 1u)) & ~(PWPP_SLOT_ALIGN - 1u);  // every part starts at a multiple of PWPP_SLOT_ALIGN slots: the fit
            sum += local[j];  // kernels fetch four points (16 / 32 bytes) per lane and load, and whole bytes of the membership plane
        }
        // inclusive scan inside the wave (DPP), then the four wave totals through LDS: one barrier
        // instead of the sixteen of a Hillis-Steele scan over 256 partials (a single frame waits for this)
        const unsigned incl = wave_incl_scan(sum);
        if (lane_id() == 63) s_part[wave_id()] = incl;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wave_id(); ++w) before += s_part[w];
        unsigned run = before + incl - sum;  // exclusive
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int p = p0 + j;
            if (j < per && p < NP) {
                poff[p] = run;
                s_po[p] = run;
            }
            run += local[j];
        }
    }
    __syncthreads();
    if (arena) {
        const unsigned word = (unsigned)Bt.results[f].overflow;
        const unsigned spilled = word >> 8, nov = s_nov;  // (workgroup-uniform)
        if (nov > 0u || spilled > 0u) {
            float *sz = Bt.sorted_z + Bt.frames[f].sbase;
            float2 *sxy = Bt.sorted_xy + Bt.frames[f].sbase;
            int *sidx = Bt.sorted_idx + Bt.frames[f].sbase;
            const unsigned real_parts = (unsigned)(2 * B);  // (pseudo-bins: cloud indices only)
            if (threadIdx.x == 0) {
                bool fail = nov > PWPP_MAX_RELOC || spilled > Bt.arena_spill || (word & 1u) != 0u;
                if (!fail) {
                    // in PART order (pwpp_member_offset: the bits of the moved parts must follow each other as the parts do)
                    for (unsigned i = 1; i < nov; ++i) {
                        const unsigned short v = s_ovp[i];
                        unsigned j = i;
                        for (; j > 0u && s_ovp[j - 1u] > v; --j) s_ovp[j] = s_ovp[j - 1u];
                        s_ovp[j] = v;
                    }
                    unsigned long long run = (spilled + (PWPP_SLOT_ALIGN - 1u)) & ~(unsigned long long)(PWPP_SLOT_ALIGN - 1u);
                    for (unsigned i = 0; i < nov; ++i) {
                        s_ovoff[i] = Bt.arena_base + (unsigned)run;
                        run += ((unsigned long long)s_pc[s_ovp[i]] + (PWPP_SLOT_ALIGN - 1u)) & ~(unsigned long long)(PWPP_SLOT_ALIGN - 1u);
                    }
                    fail = run > (unsigned long long)Bt.arena_slots;
                }
                s_reloc_fail = fail ? 1u : 0u;
            }
            __syncthreads();
            if (s_reloc_fail) {  // the frame goes back to the host: clamp the counts to what the segments hold, as without an arena
                for (int p = threadIdx.x; p < NP; p += kBlock) {
                    const unsigned cap = Bt.cap_off[p + 1] - Bt.cap_off[p];
                    if (s_pc[p] > cap) {
                        s_pc[p] = cap;
                        pcnt[p] = cap;
                    }
                }
                if (threadIdx.x == 0) atomicOr((unsigned *)&Bt.results[f].overflow, 1u);
            } else {
                for (unsigned i = 0; i < nov; ++i) {  // the part's first `capacity` points: segment -> its new place
                    const unsigned p = s_ovp[i], src = s_po[p], dst = s_ovoff[i], cap = Bt.cap_off[p + 1] - Bt.cap_off[p];
                    for (unsigned k = threadIdx.x; k < cap; k += kBlock) {
                        if (p < real_parts) {
                            sz[dst + k] = sz[src + k];
                            sxy[dst + k] = sxy[src + k];
                        }
                        sidx[dst + k] = sidx[src + k];
                    }
                }
                __syncthreads();
                for (unsigned i = threadIdx.x; i < nov; i += kBlock) {
                    s_po[s_ovp[i]] = s_ovoff[i];
                    poff[s_ovp[i]] = s_ovoff[i];
                }
                __syncthreads();
                const uint2 *tag = Bt.arena_tag + (size_t)f * Bt.arena_spill;
                for (unsigned a = threadIdx.x; a < spilled; a += kBlock) {  // the spilled points: arena record -> rank `y` of part `x`
                    const uint2 t = tag[a];
                    const unsigned src = Bt.arena_base + a, dst = s_po[t.x] + t.y;
                    if (t.x < real_parts) {
                        sz[dst] = sz[src];
                        sxy[dst] = sxy[src];
                    }
                    sidx[dst] = sidx[src];
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                atomicAnd((unsigned *)&Bt.results[f].overflow, 255u);  // the cursor has served
                if (!s_reloc_fail) atomicOr((unsigned *)&Bt.results[f].overflow, 8u);  // (statistics: parts of this frame were moved, pwpp_get_arena_stats)
            }
        }
    }
    probe();  // 1: part counts and offsets
    if (!Bt.cap_off)  // the exact path's counts enter the observed maxima too (one-pass: done above)
        for (int p = threadIdx.x; p < NP; p += kBlock)
            if (s_pc[p] > Bt.bin_max[p]) atomicMax(&Bt.bin_max[p], s_pc[p]);
    // the bins: a bin's points = its two parts, its slots begin where its low part begins
    unsigned *cnt = Bt.bin_count + (size_t)f * NB;
    unsigned *off = Bt.bin_off + (size_t)f * NB;
    __shared__ unsigned s_bc[PWPP_MAX_BINS + 2];
    for (int b = threadIdx.x; b < NB; b += kBlock) {
        const unsigned c = b < B ? s_pc[PWPP_PART_LO(b)] + s_pc[PWPP_PART_HI(b)] : s_pc[B + b];
        cnt[b] = c;
        s_bc[b] = c;
        off[b] = b < B ? s_po[PWPP_PART_LO(b)] : s_po[B + b];
    }
    if (threadIdx.x == 0) {
        Bt.results[f].n_rnr = (int)s_pc[2 * B];
        Bt.results[f].n_oor = (int)s_pc[2 * B + 1];
    }
    __syncthreads();
    probe();  // 2: observed maxima, bin counts
    // patches of this frame sorted by size bucket (work lists of the K4 kernels)
    // One round of LDS atomics places a patch: the lanes of a wave that hold the same bucket find each other with seven
    // ballots (one per bit of the bucket number), the first of them adds the group's size to the bucket's counter and hands
    // the old value to the others, a lane's place in the bucket is that value + its rank in the group.  (Round 3 counted
    // with one atomic per patch and placed with a second one: ~60 patches of a frame share a bucket, and the atomics on
    // one counter are served one after the other -- 2.4 + 2.4 us of a single frame's chain.  A loop over the wave's
    // distinct buckets instead of the ballots was slower still, profiles/r04_latency_trace.txt.)
    __shared__ unsigned s_cnt[PWPP_NUM_BUCKETS], s_start[PWPP_NUM_BUCKETS + 1];
    __shared__ unsigned s_place[PWPP_MAX_BINS];  // bucket << 16 | place in the bucket; ~0u: no patch
    static_assert(PWPP_NUM_BUCKETS <= 128, "seven ballots; two buckets per lane in the prefix below");
    if (threadIdx.x < PWPP_NUM_BUCKETS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += kBlock) {  // (workgroup-uniform trip count: every lane takes part in the ballots)
        const int b = b0 + (int)threadIdx.x;
        bool live = false;
        unsigned n = 0;
        if (b < B) {
            n = s_bc[b];
            if ((uint64_t)n >= Bt.P.min_pts) {  // else: small bin (ref :191-195)
                PwppPatchRec *rec = Bt.recs + (size_t)f * B + b;
                rec->valid = 0;
                if (n == 0) {  // only with num_min_pts <= 0: no fit runs (ref :49); K5 inherits the previous plane
                    rec->n_points = 0;
                    rec->n_ground = 0;
                    rec->n_nonground = 0;
                } else {
                    live = true;
                }
            }
        }
        const unsigned c = live ? (unsigned)pwpp_size_bucket(n) : 0u;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 7; ++bit) {
            const bool one = (c >> bit & 1u) != 0u;
            const unsigned long long m = __ballot(one);
            peers &= one ? m : ~m;
        }
        unsigned place = ~0u;
        if (live) {  // (peers holds this lane)
            const int ln = lane_id(), leader = __ffsll((long long)peers) - 1;
            unsigned base = 0;
            if (ln == leader) base = atomicAdd(&s_cnt[c], (unsigned)__popcll(peers));
            base = __shfl(base, leader);
            place = c << 16 | (base + (unsigned)__popcll(peers & ((1ull << ln) - 1ull)));
        }
        if (b < B) s_place[b] = place;
    }
    __syncthreads();
    probe();  // 3: bucket histogram, places
    if (threadIdx.x < 64) {  // exclusive prefix over the 96 buckets by one wave, two buckets per lane (a serial loop of 96 LDS
                             // round trips by one thread was 1.4 us of a single frame's chain)
        const int c0 = 2 * (int)threadIdx.x, c1 = c0 + 1;
        const unsigned a = c0 < PWPP_NUM_BUCKETS ? s_cnt[c0] : 0u, b = c1 < PWPP_NUM_BUCKETS ? s_cnt[c1] : 0u;
        const unsigned incl = wave_incl_scan(a + b);
        const unsigned excl = incl - (a + b);
        if (c0 < PWPP_NUM_BUCKETS) s_start[c0] = excl;
        if (c1 < PWPP_NUM_BUCKETS) s_start[c1] = excl + a;
        if (threadIdx.x == 63) s_start[PWPP_NUM_BUCKETS] = incl;
    }
    __syncthreads();
    for (int c = threadIdx.x; c <= PWPP_NUM_BUCKETS; c += kBlock) Bt.cls_start[(size_t)f * PWPP_CLS_STRIDE + c] = s_start[c];
    for (int b = threadIdx.x; b < B; b += kBlock) {
        const unsigned place = s_place[b];
        if (place != ~0u) Bt.cls_list[(size_t)f * B + s_start[place >> 16] + (place & 0xffffu)] = (uint16_t)b;
    }
    probe();  // 4: end
}

// ------------------------------------------------------------------------------------------
// K1'  one-pass binning (k_czm_bin_scatter): code AND scatter in the same kernel.
// The two-pass path streams the cloud twice (K1: 18 B/pt, K3: 34 B/pt) only because a point's
// slot needs the frame's complete histogram.  With 288 GB of HBM the bins can have FIXED segments
// instead: bin b of every frame owns cap_off[b+1] - cap_off[b] slots (a multiple of its expected
// share of the largest frame, sized on the host), so the slot is  segment start + the range this
// workgroup reserves with one global atomic per bin + the rank inside the workgroup  -- no scan
// in between, 16 bytes read and 16 written per point.  A bin that outgrows its segment raises the
// frame's overflow flag (points beyond the segment are not written); the host then redoes the
// batch on the exact two-pass path, so the result never depends on the capacities.
// ------------------------------------------------------------------------------------------
// BLOCK threads, four points each.  The per-workgroup set-up (zeroing the counters, fetching the segment table, the
// reservation atomics) does not depend on the tile size, so a larger workgroup halves it per point.
template <int BLOCK, bool FUSE = false>
__global__ __launch_bounds__(BLOCK, FUSE ? 2 : 8) void k_czm_bin_scatter(PwppBatch Bt, int tiles_per_frame) {
    constexpr int kBlock = BLOCK;
    constexpr int kOnePassPts = 4 * BLOCK;  // points per workgroup
    extern __shared__ unsigned s_dyn[];  // sized at launch (binning_lds_bytes): 8 KB for the default model
    const int NB = PWPP_NUM_PARTS(Bt.P.num_bins);
    unsigned *s_cnt = s_dyn;             // [parts] points of this workgroup per part, then its first slot in the part
    unsigned *s_seg = s_dyn + NB;        // [parts + 1] segment starts
    __shared__ float4 s_zt[8];
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and the
    // scattered 12-byte / 4-byte records of a bin merge into full lines only if the workgroups that write
    // that bin share an L2.  So XCD k takes the frames k, k + 8, ..., all tiles of a frame in a row.
    // Fewer than 8 frames (tiles_per_frame < 0): latency, not write merging, is what counts, and a frame confined to one
    // XCD would have 32 of the 256 CUs -- its tiles are dealt over all XCDs instead.
    const bool spread = tiles_per_frame < 0;
    if (spread) tiles_per_frame = -tiles_per_frame;
    const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
    if constexpr (FUSE) {  // (few frames only: spread) the workgroups behind the tiles copy the streams' state (K2's snapshot workgroups)
        static_assert(BLOCK == kBlock, "czm_scan_frame is written for kBlock threads");
        const int tiles = tiles_per_frame * Bt.num_frames;
        if (lin >= tiles) {
            czm_scan_frame<true>(Bt, (lin - tiles) / 8, 1 + (lin - tiles) % 8);
            return;
        }
    }
    const int f = spread ? lin / tiles_per_frame : xcd + 8 * (slot / tiles_per_frame);
    if (f >= Bt.num_frames) return;
    const PwppFrameDesc fd = Bt.frames[f];
    const int first = ((spread ? lin : slot) % tiles_per_frame) * kOnePassPts;
    if (first >= fd.n) {
        if (FUSE && fd.n == 0 && first == 0) czm_scan_frame<true>(Bt, f, 0);  // an empty frame: nobody takes a ticket
        return;
    }
    const PwppDevParams &P = Bt.P;
    constexpr int kPer = kOnePassPts / kBlock;
    unsigned pc[kPer];  // code | rank inside the workgroup << 16
    float px[kPer], py[kPer], pz[kPer], pw[kPer];
    int probe_i = 0;  // timing probes (debug_flags & 8): slots 0.. of the probe array, workgroup 0; slot 8 = the latest end of any workgroup
    auto probe = [&]() {
        if ((Bt.debug & 8) && blockIdx.x == 0 && threadIdx.x == 0 && probe_i < 8) Bt.dbg[probe_i++] = wall_clock64();
    };
    probe();
    // the points first: their loads are under way while the tables are set up (the kernel is a latency chain per
    // workgroup -- at half its occupancy it takes 1.4 x as long -- and the barrier below would otherwise stand
    // between the table loads and these)
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = first + j * kBlock + threadIdx.x;
        px[j] = py[j] = pz[j] = pw[j] = 0.0f;
        if (i < fd.n) load_point(fd, i, px[j], py[j], pz[j], pw[j]);
    }
    for (int b = threadIdx.x; b < NB; b += kBlock) s_cnt[b] = 0;
    for (int b0 = 0; b0 <= NB; b0 += 4 * kBlock) {  // (four table entries per thread in flight: not one L2 round trip per iteration)
        unsigned sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + q * kBlock + (int)threadIdx.x;
            sv[q] = b <= NB ? Bt.cap_off[b] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + q * kBlock + (int)threadIdx.x;
            if (b <= NB) s_seg[b] = sv[q];
        }
    }
    fill_zone_table(P, s_zt);
    __syncthreads();
    probe();  // 1: tables in LDS
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const float rnr_z_guard = (float)(-sensor_height - 0.8) + 1e-3f;
    const float zs = hi_split_z(P, sensor_height);
    unsigned dropped = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = first + j * kBlock + threadIdx.x;
        unsigned code = PWPP_CODE_DROP;
        if (i < fd.n) {
            const float x = px[j], y = py[j], z = pz[j], w = pw[j];
            code = part_of(P, czm_code(P, s_zt, x, y, z, w, fd.cols >= 4, sensor_height, rnr_z_guard, (Bt.debug & 16) != 0), z, zs);
            if (code == PWPP_CODE_DROP) ++dropped;
            pz[j] = binned_z(z);
        }
        unsigned pos;
        const unsigned old = wave_run_add(s_cnt, code, code != PWPP_CODE_DROP, pos);
        pc[j] = code | ((old + pos) << 16);
    }
    __syncthreads();
    probe();  // 2: points in, codes and ranks
    unsigned *gcount = Bt.part_count + (size_t)f * NB;
    // histogram and range reservation in one: a global atomic per non-empty part.  Four parts per thread at a time, all
    // four atomics in flight before the first result is needed (as a plain loop every atomic waited for the one before)
    for (int b0 = 0; b0 < NB; b0 += 4 * kBlock) {
        unsigned c[4], base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + q * kBlock + (int)threadIdx.x;
            c[q] = b < NB ? s_cnt[b] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + q * kBlock + (int)threadIdx.x;
            base[q] = c[q] ? atomicAdd(&gcount[b], c[q]) : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + q * kBlock + (int)threadIdx.x;
            if (b < NB) s_cnt[b] = base[q];
        }
    }
    dropped = wave_sum_u32(dropped);
    if (lane_id() == 0 && dropped) atomicAdd((unsigned *)&Bt.results[f].n_dropped, dropped);
    __syncthreads();
    probe();  // 3: ranges reserved
    float *sorted_z = Bt.sorted_z + fd.sbase;
    float2 *sorted_xy = Bt.sorted_xy + fd.sbase;
    int *sorted_idx = Bt.sorted_idx + fd.sbase;
    bool over = false;
    unsigned over_mask = 0;  // the points of this thread whose part is full
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned code = pc[j] & 0xffffu;
        if (code != PWPP_CODE_DROP) {
            const unsigned seg = s_seg[code], cap = s_seg[code + 1] - seg;
            const unsigned r = s_cnt[code] + (pc[j] >> 16);
            if (r < cap) {
                if (code < (unsigned)NB - 2u) {  // (a pseudo-bin -- RNR hit, out of range -- is never fitted: only its cloud indices are read again)
                    sorted_z[seg + r] = pz[j];
                    sorted_xy[seg + r] = make_float2(px[j], py[j]);
                }
#ifndef PWPP_ABLATE_NO_IDX_STORE  // (timing experiment only: the step without a quarter of K1's stores -- how much of it is bytes?)
                sorted_idx[seg + r] = first + j * kBlock + (int)threadIdx.x;
#endif
            } else {
                over = true;
                over_mask |= 1u << j;
            }
        }
    }
    if (FUSE || Bt.arena_slots == 0u) {  // no arena (a few frames: generous segments, and the fused scan's tickets live where the arena's cursor would)
        if (__any(over) && lane_id() == 0) atomicOr((unsigned *)&Bt.results[f].overflow, 1u);
    } else if (__syncthreads_or(over ? 1 : 0)) {
        // The overflow arena (pwpp_dev.h): the workgroup's spilled points take a run of the frame's arena -- one atomic on the cursor
        // in bits 8.. of the overflow word -- and go there with {part, rank inside the part} beside them; k_czm_scan moves the parts
        // that outgrew their segments.  Rare by construction of the segments (none in a steady stream of similar frames).
        __shared__ unsigned s_spill_n, s_spill_base;
        if (threadIdx.x == 0) s_spill_n = 0u;
        __syncthreads();
        const unsigned mine = over_mask ? atomicAdd(&s_spill_n, (unsigned)__popc(over_mask)) : 0u;
        __syncthreads();
        if (threadIdx.x == 0) s_spill_base = atomicAdd((unsigned *)&Bt.results[f].overflow, s_spill_n << 8) >> 8;
        __syncthreads();
        unsigned a = s_spill_base + mine;
        uint2 *tag = Bt.arena_tag + (size_t)f * Bt.arena_spill;
        bool lost = false;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (over_mask >> j & 1u) {
                const unsigned code = pc[j] & 0xffffu;
                if (a < Bt.arena_spill) {
                    const unsigned sl = Bt.arena_base + a;
                    if (code < (unsigned)NB - 2u) {
                        sorted_z[sl] = pz[j];
                        sorted_xy[sl] = make_float2(px[j], py[j]);
                    }
                    sorted_idx[sl] = first + j * kBlock + (int)threadIdx.x;
                    tag[a] = make_uint2(code, s_cnt[code] + (pc[j] >> 16));
                } else {
                    lost = true;  // the arena is full too: the frame is binned again (k_czm_scan sees the cursor beyond the arena)
                }
                ++a;
            }
        }
        if (lost) atomicOr((unsigned *)&Bt.results[f].overflow, 1u);
    }
    probe();  // 4: stores issued
    if ((Bt.debug & 8) && threadIdx.x == 0) atomicMax(&Bt.dbg[8], wall_clock64());
    if constexpr (FUSE) {
        // K2 in place: every workgroup takes a ticket (bits 8.. of the frame's overflow word, zero when the kernel starts); the one
        // that takes the frame's last runs the scan.  Its inputs are the part counts -- device-scope atomics whose results this
        // workgroup's threads have all waited for (they needed the reserved ranges) before the barrier above, so they were
        // performed before the ticket is taken; the scan reads them with device-scope loads.  What the scan writes is read by
        // the NEXT kernel.
        __shared__ int s_last;
        if (threadIdx.x == 0) {
            const unsigned tiles = (unsigned)((fd.n + kOnePassPts - 1) / kOnePassPts);
            // (release what this workgroup did -- its count atomics, its overflow flag -- and acquire what the others did, at
            // agent scope, instead of relying on the order in which gfx950 happens to perform relaxed atomics)
            const unsigned old = __hip_atomic_fetch_add((unsigned *)&Bt.results[f].overflow, 256u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old >> 8) + 1u == tiles;
            if (s_last) (void)__hip_atomic_fetch_and((unsigned *)&Bt.results[f].overflow, 255u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last) czm_scan_frame<true>(Bt, f, 0);
    }
}

// ------------------------------------------------------------------------------------------
// K2 as a kernel of its own (batches; the two-pass path)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_czm_scan(PwppBatch Bt) { czm_scan_frame<false>(Bt, blockIdx.x, blockIdx.y); }

// ------------------------------------------------------------------------------------------
// K3  scatter into bin order
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_czm_scatter(PwppBatch Bt) {
    extern __shared__ unsigned s_dyn[];  // sized at launch (binning_lds_bytes)
    const int NB = PWPP_NUM_PARTS(Bt.P.num_bins);
    unsigned *s_cnt = s_dyn;             // [parts] points of this block per part
    unsigned *s_base = s_dyn + NB;       // [parts] first slot reserved for this block in that part
    const int f = blockIdx.y;
    const PwppFrameDesc fd = Bt.frames[f];
    const int first = blockIdx.x * kPtsPerBlock;
    if (first >= fd.n) return;
    for (int b = threadIdx.x; b < NB; b += kBlock) s_cnt[b] = 0;
    __syncthreads();
    const uint16_t *codes = Bt.codes + fd.base;
    constexpr int kPer = kPtsPerBlock / kBlock;
    unsigned code[kPer], rank[kPer];
    float px[kPer], py[kPer], pz[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = first + j * kBlock + threadIdx.x;
        code[j] = PWPP_CODE_DROP;
        if (i < fd.n) {
            code[j] = codes[i];
            float x, y, z, w;
            load_point(fd, i, x, y, z, w);
            px[j] = x;
            py[j] = y;
            pz[j] = binned_z(z);
            if (code[j] != PWPP_CODE_DROP) rank[j] = atomicAdd(&s_cnt[code[j]], 1u);
        }
    }
    __syncthreads();
    unsigned *cursor = Bt.part_cursor + (size_t)f * NB;
    for (int b = threadIdx.x; b < NB; b += kBlock) {
        const unsigned c = s_cnt[b];
        if (c) s_base[b] = atomicAdd(&cursor[b], c);  // reserve a contiguous range in the part
    }
    __syncthreads();
    const unsigned *off = Bt.part_off + (size_t)f * NB;
    float *sorted_z = Bt.sorted_z + fd.sbase;
    float2 *sorted_xy = Bt.sorted_xy + fd.sbase;
    int *sorted_idx = Bt.sorted_idx + fd.sbase;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        if (code[j] != PWPP_CODE_DROP) {
            const unsigned slot = off[code[j]] + s_base[code[j]] + rank[j];
            if (code[j] < (unsigned)NB - 2u) {  // (pseudo-bins: indices only, see k_czm_bin_scatter)
                sorted_z[slot] = pz[j];
                sorted_xy[slot] = make_float2(px[j], py[j]);
            }
            sorted_idx[slot] = first + j * kBlock + (int)threadIdx.x;
        }
    }
}

// A frame with a patch that needs the plane fitted before it (PwppPatchRec.valid == 4, pwpp_fit.hip) is left alone by
// K5 / K6 / K7 -- its stream's state must not advance on half-fitted patches -- until the host has run k_fit_fixup.  K5
// notices while it fetches the patch records it needs anyway (no extra round trip in front of its latency chain) and
// tells K6 / K7 through a value they read anyway: list offsets of ~0.
constexpr unsigned kAwaitsFixup = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------
// K5  GLE + A-GLE history + TGR + adaptive thresholds, one workgroup per frame.
// k_gle_tgr_seq: the reference's sequential loop (ref :184-311) executed by lane 0 in the
// reference's own order.  It is the executable specification and the path taken when
// num_min_pts <= 0 lets empty bins through (they inherit the previous bin's plane, a serial
// dependence).  k_gle_tgr below computes the same thing with the bins spread over 256 threads.
// ------------------------------------------------------------------------------------------
__device__ void mean_stdev(const double *v, int n, double &mean, double &stdev) {  // ref :557-566
    if (n <= 1) return;
    // The sums are sequential (the reference's order, one rounding per add); the LOADS are not: sixteen
    // values are fetched at once, so a 1000-entry history in global memory costs 63 round trips per
    // pass instead of 1000 (a stream's thresholds are updated by a single lane).
    double acc = 0.0;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = v[i + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += t[k];
    }
    for (; i < n; ++i) acc += v[i];
    mean = acc / n;
    i = 0;
    for (; i + 16 <= n; i += 16) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = v[i + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) stdev += (t[k] - mean) * (t[k] - mean);
    }
    for (; i < n; ++i) stdev += (v[i] - mean) * (v[i] - mean);
    stdev /= n - 1;
    stdev = sqrt(stdev);
}

// A sequential sum over `nb` batches of sixteen values with the READS taken off its chain: the reads of batch b + 1 are issued before
// the sixteen dependent adds of batch b.  load(b, t) fills t[16], add(t) is the chain's step.  (Pays for the short lists of one frame
// -- TGR's ring statistics of a stream 4.3 -> 3.1 us, the threshold statistics of a fresh frame 2.1 -> 1.5 us; NOT for a full
// 1000-entry history, whose pass is bound by the 1000 dependent f64 adds themselves: k_gle_tgr keeps its plain loop there.)
template <class Load, class Add>
__device__ __forceinline__ void chain_batches16(int nb, Load load, Add add) {
    double t[16], u[16];
    if (nb > 0) load(0, t);
    int b = 0;
    for (; b + 2 <= nb; b += 2) {
        load(b + 1, u);
        add(t);
        if (b + 2 < nb) load(b + 2, t);
        add(u);
    }
    if (b < nb) add(t);
}
// The same for values that already sit in LDS (stride in doubles): batches of sixteen reads, the
// tail masked instead of walked one dependent read at a time.
__device__ void mean_stdev_lds(const double *v, int stride, int n, double &mean, double &stdev) {  // ref :557-566
    if (n <= 1) return;
    const int nb = n >> 4, i0 = nb << 4;
    auto load = [&](int b, double (&t)[16]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = v[(16 * b + k) * stride];
    };
    double tl[16];  // the last, partial batch: fetched once, used by both passes
#pragma unroll
    for (int k = 0; k < 16; ++k) tl[k] = i0 + k < n ? v[(i0 + k) * stride] : 0.0;
    double acc = 0.0;
    chain_batches16(nb, load, [&](const double (&t)[16]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += t[k];
    });
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = i0 + k < n ? acc + tl[k] : acc;
    mean = acc / n;
    const double m = mean;
    double sq = stdev;
    chain_batches16(nb, load, [&](const double (&t)[16]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sq += (t[k] - m) * (t[k] - m);
    });
#pragma unroll
    for (int k = 0; k < 16; ++k) sq = i0 + k < n ? sq + (tl[k] - m) * (tl[k] - m) : sq;
    sq /= n - 1;
    stdev = sqrt(sq);
}

// K5 prepares the NEXT call's counters (pwpp_dev.h: next_part_count): this frame's share of the other copy, zeroed by the
// workgroup that finishes the frame anyway.  Nothing of the current call reads or writes that copy.
__device__ __forceinline__ void clear_next_counters(const PwppBatch &Bt, int f, int nthreads) {
    if (!Bt.next_part_count) return;
    const int NP = PWPP_NUM_PARTS(Bt.P.num_bins);
    for (int s = 0; s < Bt.next_slabs; ++s) {
        uint32_t *dst = Bt.next_part_count + (int64_t)s * Bt.next_slab_stride + (size_t)f * NP;
        for (int i = threadIdx.x; i < NP; i += nthreads) dst[i] = 0u;
    }
    if (threadIdx.x == 0) {
        PwppFrameResult z;
        z.n_ground = z.n_nonground = z.n_patches = z.n_rnr = z.n_oor = z.n_dropped = z.hist_state = z.overflow = 0;
        Bt.next_results[f] = z;
    }
}

__global__ __launch_bounds__(64) void k_gle_tgr_seq(PwppBatch Bt) {
    clear_next_counters(Bt, blockIdx.x, 64);
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
    if (!Bt.fixup_run && (Bt.results[f].overflow & 2)) {  // a patch awaits k_fit_fixup
        for (int b = 0; b < NB; ++b) dst_a[b] = kAwaitsFixup;
        return;
    }
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

    int dropped_push = 0;  // a history slab was full (never, unless the host failed to grow it: hist_state)
    // ---- pass 1: decisions in traversal order (ref :184-311)
    int concentric_idx = 0;
    int n_ring_flat = 0;  // ringwise_flatness: only cleared when a ring had candidates (ref :292-304)
    int n_patches = 0;
    PwppPatchRec prev;    // plane members persist across bins AND frames in the reference (stale-plane quirk)
    prev.mean[0] = prev.mean[1] = prev.mean[2] = 0.0f;
    prev.normal[0] = prev.normal[1] = prev.normal[2] = 0.0f;
    prev.sv[0] = prev.sv[1] = prev.sv[2] = 0.0f;
    prev.d = 0.0;
    if (fd.state_in >= 0) {  // what the stream's last frame left in the members
        const PwppPlaneState ps = Bt.st_plane[fd.state_in];
        for (int i = 0; i < 3; ++i) {
            prev.mean[i] = ps.mean[i];
            prev.normal[i] = ps.normal[i];
            prev.sv[i] = ps.sv[i];
        }
        prev.d = ps.d;
    }
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
                    // (the reference's vectors are unbounded -- update_flatness_thr stops trimming the higher rings
                    // while a lower one holds <= 1 entries, ref :363-364; the host grows the slabs in time
                    // (hist_state), so these guards never trip; each history has its own)
                    if (st.elev_len[concentric_idx] < P.hist_cap)
                        hist_out[(0 * 4 + concentric_idx) * P.hist_cap + st.elev_len[concentric_idx]++] = elevation;
                    else
                        dropped_push = 1;
                    if (st.flat_len[concentric_idx] < P.hist_cap)
                        hist_out[(1 * 4 + concentric_idx) * P.hist_cap + st.flat_len[concentric_idx]++] = flatness;
                    else
                        dropped_push = 1;
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
    {   // the plane members after this frame
        PwppPlaneState ps;
        for (int i = 0; i < 3; ++i) {
            ps.mean[i] = prev.mean[i];
            ps.normal[i] = prev.normal[i];
            ps.sv[i] = prev.sv[i];
        }
        ps.pad_ = 0.0f;
        ps.d = prev.d;
        Bt.st_plane[fd.state_out] = ps;
    }
    {   // how full the fullest history is: the host grows the slabs before they run out (pwpp_capi.cpp)
        int mx = 0;
        for (int k = 0; k < 4; ++k) {
            mx = st.elev_len[k] > mx ? st.elev_len[k] : mx;
            mx = st.flat_len[k] > mx ? st.flat_len[k] : mx;
        }
        res->hist_state = (mx << 1) | dropped_push;
    }
}

// ------------------------------------------------------------------------------------------
// K5 parallel.  The decisions of ref :217-282 are independent per patch; what the sequential
// loop adds is ORDER: history pushes in sector order, ring-wise flatness statistics that carry
// over rings without candidates (ref :292-304), and output lists appended bin by bin with the
// TGR candidates of a ring appended at its end.  All of that is prefix sums over the bins in
// traversal order plus a 4-iteration loop over the rings of interest.
// Every double sum keeps the reference's summation order (sequential over <= one ring).
// ------------------------------------------------------------------------------------------
constexpr int kGlePer = PWPP_MAX_BINS / kBlock;  // consecutive bins per thread, largest model
// the largest of the lanes' values in 0..255: a binary search over the bits with ballots (scalar work, no cross-lane data path)
__device__ __forceinline__ int wave_max_u8(int v) {
    int cur = 0;
#pragma unroll
    for (int bit = 7; bit >= 0; --bit) {
        const int cand = cur | (1 << bit);
        if (__ballot(v >= cand)) cur = cand;
    }
    return cur;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// Sum over the wave in butterfly order (quad_perm xor 1, xor 2, row_half_mirror, row_mirror, then the four rows): ONLY for sums
// that are exact in every order (k_gle_tgr's first pass over a history).  The result is wave-uniform.
__device__ __forceinline__ double wave_sum_f64_any_order(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 16 * k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 16 * k);
        r[k] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

template <int K>
__device__ __forceinline__ void block_excl_scan(unsigned v[K], unsigned (*s_wave)[K], unsigned total[K]) {
    // v[q]: this thread's sum of quantity q; returns the exclusive prefix over threads in v, block totals in total
    const int ln = lane_id(), wv = wave_id();
    unsigned incl[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        incl[q] = wave_incl_scan(v[q]);
        if (ln == 63) s_wave[wv][q] = incl[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < K; ++q) {
        unsigned before = 0, all = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            const unsigned t = s_wave[w][q];
            if (w < wv) before += t;
            all += t;
        }
        v[q] = incl[q] - v[q] + before;
        total[q] = all;
    }
    __syncthreads();
}

// LAT (a few dozen frames: every workgroup has a CU and its 160 KB of LDS to itself): the staging tile of
// the threshold statistics holds a whole 1000-entry history per row, so each of the two passes is ONE
// stage instead of three; otherwise the tile lives in the retired prefix arrays (32 KB).
// PER: consecutive bins per thread the kernel is compiled for (2 covers the default 504-bin model: every per-bin loop, and the
// registers of the patch records, four times shorter than for the largest model)
// PART (a single stream's chain, round 6): the kernel in two launches.  1 = everything the index lists wait for (decisions, history pushes,
// TGR, list offsets, the plane members) -- K6 follows it on the same stream; 2 = what only the stream's NEXT frame needs (the statistics of
// the eight histories, thresholds, sensor height, erase): launched on the handle's second stream, it runs under K6 and the host's turn-around
// instead of holding the lists back by its two chains of ~1000 dependent f64 adds (11 of 27 us).  Part 1 hands over through the stream's
// state (history lengths after the pushes) and results[f].hist_state (a slab was full).  0 = the whole kernel.
template <bool LAT, int PER, int PART = 0>
__global__ __launch_bounds__(kBlock) void k_gle_tgr(PwppBatch Bt) {
    // timing probes (debug_flags & 8): 100 MHz ticks along the chain of frame 0, slots 32.. of the probe array (tools/k5_chain.py)
    int probe_i = 32;
    auto probe = [&]() {
        if ((Bt.debug & 8) && blockIdx.x == 0 && threadIdx.x == 0 && probe_i < 60) Bt.dbg[probe_i++] = wall_clock64();
    };
    probe();
    probe();  // (slot 1 of the chain: the next call's counters used to be cleared here, in front of the first loads; now behind them)
    __shared__ uint8_t s_dec[PWPP_MAX_BINS];
    __shared__ __attribute__((aligned(16))) unsigned s_e[4][PWPP_MAX_BINS + 1];   // exclusive prefixes: gmain, gtail, nmain, ntail;
                                                                                // later the staging tile of the histories
    __shared__ unsigned s_epush[PWPP_MAX_NEAR_BINS + 1];
    __shared__ double s_pseq[PWPP_MAX_NEAR_BINS];     // flatness of the pushed patches, in push order
    __shared__ double s_pelev[LAT ? PWPP_MAX_NEAR_BINS : 1];  // their elevations (latency variant: a frame that starts from empty histories
                                                              // takes its threshold statistics from these two arrays, see below)
    __shared__ int s_dropped;                         // a history slab was full (never, unless the host failed to grow it)
    __shared__ unsigned s_wave[kBlock / 64][4];
    __shared__ PwppStateScalar s_st;
    __shared__ int s_ring_first[PWPP_MAX_ROI + 1];    // first bin of near ring ci; [roi] = end
    __shared__ unsigned s_ring_cand[PWPP_MAX_ROI];
    __shared__ double s_ring_mean[PWPP_MAX_ROI], s_ring_std[PWPP_MAX_ROI];
    __shared__ unsigned s_seg_begin[PWPP_MAX_ROI], s_seg_end[PWPP_MAX_ROI];  // slice of s_pseq behind the statistics of ring ci
    __shared__ int s_len0[2][PWPP_MAX_ROI];           // history lengths before this frame
    __shared__ int s_last_patch;                      // last bin of the frame that was fitted (its plane stays in the object's members)
    const int f = blockIdx.x;
    const PwppDevParams &P = Bt.P;
    const int B = P.num_bins, NB = B + 2;
    const PwppFrameDesc fd = Bt.frames[f];
    const unsigned *cnt = Bt.bin_count + (size_t)f * NB;
    PwppPatchRec *recs = Bt.recs + (size_t)f * B;
    unsigned *dst_a = Bt.dst_a + (size_t)f * NB;
    unsigned *dst_b = Bt.dst_b + (size_t)f * NB;
    float *centers = Bt.centers + (size_t)f * B * 3;
    float *normals = Bt.normals + (size_t)f * B * 3;
    double *hist_out = Bt.st_hist + (size_t)fd.state_out * 8 * P.hist_cap;
    const int roi = P.num_rings_of_interest;

    if (PART == 2 && threadIdx.x == 0) {  // the state as part 1 left it
        s_st = Bt.st_scalar[fd.state_out];
        s_dropped = Bt.results[f].hist_state & 1;
        s_last_patch = -1;
        for (int k = 0; k < PWPP_MAX_ROI; ++k) s_len0[0][k] = s_len0[1][k] = 1;  // (not a frame that starts from empty histories: they are read back)
    }
    if (PART != 2 && threadIdx.x == 0) {
        s_dropped = 0;
        s_last_patch = -1;
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
        s_st = st;
        for (int k = 0; k < PWPP_MAX_ROI; ++k) {
            s_len0[0][k] = st.elev_len[k];
            s_len0[1][k] = st.flat_len[k];
            s_ring_cand[k] = 0;
        }
        // first bin of every ring of interest (they are the first rings in traversal order)
        int ci = 0, bin = 0;
        for (int zone = 0; zone < 4 && ci <= roi; ++zone)
            for (int ring = 0; ring < P.rings[zone] && ci <= roi; ++ring) {
                if (ci <= PWPP_MAX_ROI) s_ring_first[ci] = bin;
                bin += P.sectors[zone];
                ++ci;
            }
        for (; ci <= roi; ++ci) s_ring_first[ci] = B;  // fewer rings than rings of interest
    }
    // Consecutive bins per thread: as few as cover the model (2 for the default 504 bins), so that a
    // single frame keeps all four waves busy instead of one thread walking eight bins.  Everything a
    // bin needs from global memory (its count, its patch record) is fetched once, up front and
    // unconditionally: the kernel is a latency chain, every dependent round trip costs ~1 us.
    const int per = (B + kBlock - 1) / kBlock;
    const int b0 = threadIdx.x * per;
    unsigned nn[PER];
    PwppPatchRec rr[PER];
    uint8_t dec[PER];
    int ci_of[PER];
    if constexpr (PART == 2) {
        if (dst_a[0] == kAwaitsFixup) return;  // (part 1 left the frame to k_fit_fixup; workgroup-uniform)
        __syncthreads();
    }
    if constexpr (PART != 2) {
    int awaits = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int bin = b0 + j;
        const bool have = j < per && bin < B;
        nn[j] = have ? cnt[bin] : 0u;
        rr[j] = recs[have ? bin : 0];
    }
    // the next call's counters (clear_next_counters): stores that nothing of this kernel waits for, issued while the loads above are on
    // their way -- in front of them they held the records back by the microsecond it takes to issue them
    clear_next_counters(Bt, blockIdx.x, kBlock);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int bin = b0 + j;
        const bool have = j < per && bin < B;
        if (have && nn[j] > 0u && (uint64_t)nn[j] >= P.min_pts && rr[j].valid == 4) awaits = 1;
    }
    if (__syncthreads_or(awaits && !Bt.fixup_run)) {  // a patch awaits k_fit_fixup: hands off (workgroup-uniform)
        for (int b = threadIdx.x; b < NB; b += kBlock) dst_a[b] = kAwaitsFixup;
        return;
    }
    probe();  // 2: records and counts are in
    if (fd.state_in >= 0 && fd.state_in != fd.state_out) {  // carry the histories over
        const double *hist_in = Bt.st_hist + (size_t)fd.state_in * 8 * P.hist_cap;
        for (int w = 0; w < 8; ++w) {
            const int len = w < 4 ? s_len0[0][w] : s_len0[1][w - 4];
            for (int i = threadIdx.x; i < len; i += kBlock) hist_out[w * P.hist_cap + i] = hist_in[w * P.hist_cap + i];
        }
    }
    const int near_end = s_ring_first[roi];  // bins [0, near_end) are "near" (concentric_idx < roi)

    // ---- pass 1: per-bin GLE (ref :217-282) ------------------------------------------------
    unsigned a_patch = 0, a_push = 0;
    int my_last = -1;  // this thread's last fitted bin -> s_last_patch: one LDS atomic per WAVE (256 on one address serialise: 1.5 us)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int bin = b0 + j;
        dec[j] = 0;
        ci_of[j] = 0;
        if (j >= per || bin >= B) continue;
        const unsigned n = nn[j];
        if ((uint64_t)n < P.min_pts) continue;  // small bin
        my_last = bin > my_last ? bin : my_last;  // (bins ascend with j)
        // concentric index of the bin
        const int zone = bin < P.bin_base[1] ? 0 : (bin < P.bin_base[2] ? 1 : (bin < P.bin_base[3] ? 2 : 3));
        int ci = (bin - P.bin_base[zone]) / P.sectors[zone];
        for (int z = 0; z < zone; ++z) ci += P.rings[z];
        ci_of[j] = ci;
        const PwppPatchRec &r = rr[j];
        const double uprightness = r.normal[2];
        const double elevation = r.mean[2];
        float fmin3 = r.sv[0];
        if (r.sv[1] < fmin3) fmin3 = r.sv[1];
        if (r.sv[2] < fmin3) fmin3 = r.sv[2];
        const double flatness = fmin3;
        double heading = 0.0;
        for (int i = 0; i < 3; ++i) heading += r.mean[i] * r.normal[i];
        const bool is_upright = uprightness > P.uprightness_thr;
        const bool is_near = ci < roi;
        const bool heading_outside = heading < 0.0;
        bool not_elevated = false, is_flat = false;
        if (is_near) {
            not_elevated = elevation < s_st.elevation_thr[ci];
            is_flat = flatness < s_st.flatness_thr[ci];
        }
        int d;
        if (!is_upright)
            d = 1;
        else if (!is_near)
            d = 2;
        else if (!heading_outside)
            d = 3;
        else if (not_elevated || is_flat)
            d = 4;
        else
            d = 5;
        if (is_upright && not_elevated && is_near) d |= 0x80;  // pushes to the A-GLE history (ref :253-259)
        dec[j] = (uint8_t)d;
        a_patch += 1;
        a_push += (d & 0x80) ? 1u : 0u;
        if ((d & 0x7f) == 5) s_ring_cand[ci] = 1u;  // benign race: all writers store 1
    }
    {
        const int wave_last = (int)wave_max_i32(my_last);
        if (lane_id() == 0 && wave_last >= 0) atomicMax(&s_last_patch, wave_last);
        unsigned v[2] = {a_patch, a_push}, tot[2];
        block_excl_scan<2>(v, reinterpret_cast<unsigned(*)[2]>(&s_wave[0][0]), tot);
        unsigned p_patch = v[0], p_push = v[1];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int bin = b0 + j;
            if (j >= per) continue;
            if (bin < near_end) s_epush[bin] = p_push;
            if (bin < B) s_dec[bin] = dec[j];
            if (dec[j] == 0) continue;
            const PwppPatchRec &r = rr[j];
            for (int i = 0; i < 3; ++i) {  // ref :211-212
                centers[p_patch * 3 + i] = r.mean[i];
                normals[p_patch * 3 + i] = r.normal[i];
            }
            ++p_patch;
            if (dec[j] & 0x80) {
                float fmin3 = r.sv[0];
                if (r.sv[1] < fmin3) fmin3 = r.sv[1];
                if (r.sv[2] < fmin3) fmin3 = r.sv[2];
                s_pseq[p_push] = (double)fmin3;
                if (LAT) s_pelev[p_push] = (double)r.mean[2];
                ++p_push;
            }
        }
        if (threadIdx.x == 0) {
            s_epush[near_end] = tot[1];
            Bt.results[f].n_patches = (int)tot[0];
        }
    }
    __syncthreads();
    probe();  // 3: decisions, first scan, centres / normals written
    // history pushes in sector order (ref :255-256): position inside the ring = prefix difference
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int bin = b0 + j;
        if (!(dec[j] & 0x80)) continue;
        const int ci = ci_of[j];
        const int pos = (int)(s_epush[bin] - s_epush[s_ring_first[ci]]);
        const int e = s_len0[0][ci] + pos, fl = s_len0[1][ci] + pos;
        const PwppPatchRec &r = rr[j];
        // (each history has its own guard; the host grows the slabs before one can fill: hist_state)
        if (e < P.hist_cap) hist_out[(0 * 4 + ci) * P.hist_cap + e] = (double)r.mean[2];
        if (fl < P.hist_cap) hist_out[(1 * 4 + ci) * P.hist_cap + fl] = s_pseq[s_epush[bin]];
    }
    // ring-wise flatness statistics for TGR: the list is only cleared by a ring that had
    // candidates (ref :292-304), so it may span several rings
    if (threadIdx.x == 0) {
        unsigned begin = 0;
        for (int ci = 0; ci < roi; ++ci) {
            const unsigned end = s_epush[s_ring_first[ci + 1]];
            const int pushed = (int)(end - s_epush[s_ring_first[ci]]);
            int ne = s_len0[0][ci] + pushed, nf = s_len0[1][ci] + pushed;
            if (ne > P.hist_cap) { ne = P.hist_cap; s_dropped = 1; }  // (the pushes beyond the slab were not written)
            if (nf > P.hist_cap) { nf = P.hist_cap; s_dropped = 1; }
            s_st.elev_len[ci] = ne;
            s_st.flat_len[ci] = nf;
            s_seg_begin[ci] = begin;
            s_seg_end[ci] = end;
            if (s_ring_cand[ci]) begin = end;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < roi && s_ring_cand[threadIdx.x]) {  // one lane per ring with candidates
        double m = 0.0, sd = 0.0;
        const unsigned begin = s_seg_begin[threadIdx.x];
        mean_stdev_lds(s_pseq + begin, 1, (int)(s_seg_end[threadIdx.x] - begin), m, sd);  // ref :407-408
        s_ring_mean[threadIdx.x] = m;
        s_ring_std[threadIdx.x] = sd;
    }
    __syncthreads();
    probe();  // 4: history pushes, ring statistics
    }  // (PART != 2)

    // The histories are complete now (this frame's pushes included): start fetching their first tile
    // for the threshold statistics at the end of the kernel, the loads fly while TGR and the list
    // offsets are worked out.
    constexpr int kHistTile = LAT ? 1056 : 496, kHistStride = kHistTile + 2;  // 8 rows, 16-byte aligned, on distinct LDS banks
    static_assert(kHistTile % 2 == 0 && kHistStride % 2 == 0, "pairs of entries");
    static_assert(LAT || sizeof(s_e) >= sizeof(double) * 8 * kHistStride, "history tile must fit into the retired s_e");
    __shared__ __attribute__((aligned(16))) double s_tile_lat[LAT ? 8 * kHistStride : 2];
    int len_w[8], maxlen = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        len_w[w] = (w & 3) < roi ? (w < 4 ? s_st.elev_len[w] : s_st.flat_len[w - 4]) : 0;
        if (len_w[w] <= 1) len_w[w] = 0;  // mean_stdev leaves (0, 0) behind, ref :558
        maxlen = len_w[w] > maxlen ? len_w[w] : maxlen;
    }
    const int my_len = threadIdx.x < 8 ? len_w[threadIdx.x & 7] : 0;
    const int ntiles = (maxlen + kHistTile - 1) / kHistTile;
    // A frame that starts from EMPTY histories (fresh state: a single frame, the first frame of a stream) has just written all
    // their entries itself, and they are still in LDS in push order (s_pelev, s_pseq): the threshold statistics take them from
    // there -- no read-back of the histories from global memory, no staging tiles (5 of 22 us of a single frame).
    bool fresh_hist = LAT && PART != 2 && s_dropped == 0;
#pragma unroll
    for (int k = 0; k < PWPP_MAX_ROI; ++k) fresh_hist = fresh_hist && s_len0[0][k] == 0 && s_len0[1][k] == 0;
    // (two entries per thread and load: a 256-thread workgroup issuing forty 8-byte loads per thread took 3.2 us to ISSUE them;
    // the slabs' rows start 16-byte aligned -- hist_cap is even -- and a pair that straddles the end of a history reads one
    // entry of slack, never beyond the row)
    constexpr int kHistPer2 = (kHistTile + 2 * kBlock - 1) / (2 * kBlock);
    double2 v[8][kHistPer2];
    auto fetch = [&](int base) {  // unconditional loads (clamped), all in flight together
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int q = 0; q < kHistPer2; ++q) {
                if (base + q * 2 * kBlock >= maxlen) continue;  // (uniform) nothing that far in any history
                const int i = base + 2 * (q * kBlock + (int)threadIdx.x);
                v[w][q] = *reinterpret_cast<const double2 *>(hist_out + (size_t)w * P.hist_cap + (i < len_w[w] && i + 1 < P.hist_cap ? i : 0));
            }
    };
    if (PART != 1 && ntiles > 0 && !fresh_hist) fetch(0);

    if constexpr (PART != 2) {
    // ---- pass 2: TGR (ref :416-461) and what each bin appends to which list -----------------
    unsigned q4[4] = {0, 0, 0, 0};  // gmain, gtail, nmain, ntail of this thread's bins
    unsigned gm[PER], gt[PER], nm[PER], nt[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int bin = b0 + j;
        gm[j] = gt[j] = nm[j] = nt[j] = 0;
        if (j >= per || bin >= B) continue;
        const unsigned n = nn[j];
        int d = dec[j] & 0x7f;
        if (d == 0) {
            nm[j] = n;  // small bin, whole (ref :193)
        } else {
            const PwppPatchRec &r = rr[j];
            const unsigned ng = (unsigned)r.n_ground;
            if (d == 5 && P.enable_TGR) {
                const int ci = ci_of[j];
                float fmin3 = r.sv[0];
                if (r.sv[1] < fmin3) fmin3 = r.sv[1];
                if (r.sv[2] < fmin3) fmin3 = r.sv[2];
                const double flatness = fmin3;
                const double line_variable = r.sv[1] != 0 ? (double)(r.sv[0] / r.sv[1]) : DBL_MAX;
                const double mu = s_ring_mean[ci] + 1.5 * s_ring_std[ci];                  // ref :428
                double prob_flatness = 1 / (1 + exp((flatness - mu) / (mu / 10)));        // ref :429
                if (r.n_ground > 1500 && flatness < P.th_dist * P.th_dist) prob_flatness = 1.0;  // ref :431
                double prob_line = 1.0;
                if (line_variable > 8.0) prob_line = 0.0;
                if (prob_line * prob_flatness > 0.5) d = 6;
            }
            recs[bin].decision = d;
            if (d == 1 || d == 3)
                nm[j] = n;  // candidates then non-ground part, both to the non-ground list (ref :264,272,284)
            else
                nm[j] = n - ng;  // ref :284
            if (d == 2 || d == 4) gm[j] = ng;
            if (d == 6) gt[j] = ng;
            if (d == 5) nt[j] = ng;
        }
        dec[j] = (uint8_t)d;
        q4[0] += gm[j];
        q4[1] += gt[j];
        q4[2] += nm[j];
        q4[3] += nt[j];
    }
    unsigned tot4[4];
    block_excl_scan<4>(q4, s_wave, tot4);
    {
        unsigned run[4] = {q4[0], q4[1], q4[2], q4[3]};
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int bin = b0 + j;
            if (j >= per || bin >= B) continue;
            for (int q = 0; q < 4; ++q) s_e[q][bin] = run[q];
            run[0] += gm[j];
            run[1] += gt[j];
            run[2] += nm[j];
            run[3] += nt[j];
        }
        if (threadIdx.x == 0)
            for (int q = 0; q < 4; ++q) s_e[q][B] = tot4[q];
    }
    __syncthreads();
    probe();  // 5: TGR, second scan
    const unsigned total_ground = tot4[0] + tot4[1];
    const unsigned n_rnr = cnt[B], n_oor = cnt[B + 1];
    const unsigned ng_base = total_ground + n_rnr + n_oor;  // non-ground list follows the ground list
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int bin = b0 + j;
        if (j >= per || bin >= B) continue;
        const int d = dec[j];
        // ring of this bin: its first bin and the first bin of the next ring
        const int zone = bin < P.bin_base[1] ? 0 : (bin < P.bin_base[2] ? 1 : (bin < P.bin_base[3] ? 2 : 3));
        const int rs = P.bin_base[zone] + ((bin - P.bin_base[zone]) / P.sectors[zone]) * P.sectors[zone];
        const int re = rs + P.sectors[zone];
        const unsigned nmain_at = ng_base + s_e[2][bin] + s_e[3][rs];
        if (d == 0) {
            dst_a[bin] = nmain_at;
            dst_b[bin] = 0;
        } else {
            const unsigned ng = (unsigned)rr[j].n_ground;
            if (d == 1 || d == 3) {
                dst_a[bin] = nmain_at;
                dst_b[bin] = nmain_at + ng;
            } else {
                dst_b[bin] = nmain_at;
                if (d == 2 || d == 4)
                    dst_a[bin] = s_e[0][bin] + s_e[1][rs];
                else if (d == 6)
                    dst_a[bin] = s_e[0][re] + s_e[1][bin];  // reverted at the end of its ring (ref :450)
                else
                    dst_a[bin] = ng_base + s_e[2][re] + s_e[3][bin];  // rejected at the end of its ring (ref :458)
            }
        }
    }
    if (threadIdx.x == 0) {
        dst_a[B] = total_ground;          // RNR hits first (ref :393) ...
        dst_a[B + 1] = total_ground + n_rnr;  // ... then the out-of-range points (ref :618)
        dst_b[B] = dst_b[B + 1] = 0;
        PwppFrameResult *res = Bt.results + f;
        res->n_ground = (int)total_ground;
        res->n_nonground = (int)(n_rnr + n_oor + tot4[2] + tot4[3]);
    }
    __threadfence_block();
    __syncthreads();
    probe();  // 6: list offsets written
    }  // (PART != 2)
    // the plane members after this frame: those of its last fitted bin, else what the stream's last frame left
    auto write_plane_state = [&]() {
        PwppPlaneState ps;
        if (s_last_patch >= 0) {
            const PwppPatchRec lr = recs[s_last_patch];
            for (int i = 0; i < 3; ++i) {
                ps.mean[i] = lr.mean[i];
                ps.normal[i] = lr.normal[i];
                ps.sv[i] = lr.sv[i];
            }
            ps.d = lr.d;
        } else if (fd.state_in >= 0) {
            ps = Bt.st_plane[fd.state_in];
        } else {
            for (int i = 0; i < 3; ++i) ps.mean[i] = ps.normal[i] = ps.sv[i] = 0.0f;
            ps.d = 0.0;
        }
        ps.pad_ = 0.0f;
        Bt.st_plane[fd.state_out] = ps;
    };
    if constexpr (PART == 1) {  // hand-over to part 2: the history lengths after this frame's pushes (thresholds and sensor height still the old ones)
        if (threadIdx.x == 0) {
            Bt.st_scalar[fd.state_out] = s_st;
            write_plane_state();
            Bt.results[f].hist_state = s_dropped;
        }
        return;
    }

    // ---- adaptive thresholds for the next frame of this stream (ref :338-375) ---------------
    // lanes 0..3: elevation history of ring i, lanes 4..7: flatness history; sequential sums
    __shared__ double s_mean[8], s_std[8], s_psum[8];
    __shared__ int s_pexact[8];
    // The eight histories (up to max_*_storage + one frame's pushes each) live in global memory, and the
    // reference's sums over them are sequential: one lane per history, ~2 x 1000 dependent f64 adds.
    // Everything else is taken off that chain: all threads stage the histories through LDS in tiles
    // (the next tile's loads are in flight while the current one is summed), and for the second pass
    // they also square the deviations, so the summing lanes execute one add per value.
    if (fresh_hist) {
        if (threadIdx.x < 8) {  // the same sequential sums (ref :557-566) over the same values, in push order
            const int ring = (int)threadIdx.x & 3;
            double m = 0.0, sd = 0.0;
            if (ring < roi) mean_stdev_lds((threadIdx.x < 4 ? s_pelev : s_pseq) + s_epush[s_ring_first[ring]], 1, my_len, m, sd);
            s_mean[threadIdx.x] = m;
            s_std[threadIdx.x] = sd;
        }
    } else {
        double *tile = LAT ? s_tile_lat : reinterpret_cast<double *>(&s_e[0][0]);
        double acc = 0.0, mean = 0.0;
        for (int step = 0; step < 2 * ntiles; ++step) {
            const int pass = step >= ntiles ? 1 : 0;
            const int base = (step - pass * ntiles) * kHistTile;
            __syncthreads();  // the previous tile has been summed; the means of pass 0 are in s_mean
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const double m = s_mean[w];
#pragma unroll
                for (int q = 0; q < kHistPer2; ++q) {
                    if (base + q * 2 * kBlock >= maxlen) continue;
                    const int i = 2 * (q * kBlock + (int)threadIdx.x);
                    if (i + 1 < kHistTile)  // (kHistTile is even)
                        *reinterpret_cast<double2 *>(tile + w * kHistStride + i) =
                            pass ? make_double2((v[w][q].x - m) * (v[w][q].x - m), (v[w][q].y - m) * (v[w][q].y - m)) : v[w][q];  // ref :564
                }
            }
            __syncthreads();
            if (step + 1 < 2 * ntiles && ntiles > 1) fetch(((step + 1) % ntiles) * kHistTile);  // (one tile: the second pass stages the same registers)
            // Pass 0 without the chain, where that is exact.  The entries of a history are float values held as doubles
            // (a patch's mean height / smallest singular value, ref :325-326).  If every entry IS a float and the exponents of
            // the non-zero ones span at most 18 binades, any sum of any of up to 2^11 of them is a multiple of the smallest
            // entry's last bit and below 2^11 times the largest: 24 + 18 + 11 = 53 bits -- it is exact in a double, so the
            // reference's sequential sum never rounds and equals the sum taken in any other order: two histories per wave,
            // seventeen adds per lane and a butterfly.  (A history restored from a checkpoint with arbitrary doubles, a
            // flatness of 1e-9 beside 1e-2: the lane keeps its sequential loop.)  Only where one tile holds the history (LAT).
            static_assert(kHistTile <= 2048, "the exactness argument below counts on at most 2^11 entries per history tile");
            bool fast0 = false;
            if (LAT && ntiles == 1 && pass == 0 && !(Bt.debug & 128)) {  // (debug flag 128: always the sequential sum, for the tests)
                const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int w = 2 * wv + hh;  // (kBlock = 256: four waves, eight histories)
                    int n = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) n = q == w ? len_w[q] : n;  // (len_w lives in registers)
                    // every load in flight at once, four independent partial sums (any order is exact here), the exponent
                    // range by ballots (eight steps of a binary search each, scalar work), the sum by DPP
                    constexpr int kPerLane = (kHistTile + 63) / 64;
                    double xs[kPerLane];
#pragma unroll
                    for (int q = 0; q < kPerLane; ++q) {
                        const int i = ln + 64 * q;
                        xs[q] = i < n ? tile[w * kHistStride + (i < kHistTile ? i : 0)] : 0.0;
                    }
                    double p4[4] = {0.0, 0.0, 0.0, 0.0};
                    int emax = 0, emin_inv = 0;  // largest exponent, 255 - the smallest one, over this lane's non-zero entries
                    bool rep = true;
#pragma unroll
                    for (int q = 0; q < kPerLane; ++q) {
                        const double x = xs[q];
                        p4[q & 3] += x;
                        const float xf = (float)x;
                        int e = (int)((__float_as_uint(xf) >> 23) & 0xffu);
                        rep = rep && (double)xf == x && e != 255;
                        e = e < 1 ? 1 : e;  // (a denormal's last bit is that of the smallest normal binade)
                        if (x != 0.0) {
                            emax = e > emax ? e : emax;
                            emin_inv = 255 - e > emin_inv ? 255 - e : emin_inv;
                        }
                    }
                    const double ps = wave_sum_f64_any_order((p4[0] + p4[1]) + (p4[2] + p4[3]));
                    emax = wave_max_u8(emax);
                    emin_inv = wave_max_u8(emin_inv);
                    const bool ok = __all(rep) && (emax == 0 || emax - (255 - emin_inv) <= 18);
                    if (ln == 0 && w < 8) {
                        s_psum[w] = ps;
                        s_pexact[w] = ok ? 1 : 0;
                    }
                }
                __syncthreads();
                fast0 = threadIdx.x < 8 && s_pexact[threadIdx.x & 7] != 0;
            }
            if (threadIdx.x < 8) {
                const int left = my_len - base;
                const int cnt_here = fast0 ? 0 : (left < 0 ? 0 : (left < kHistTile ? left : kHistTile));
                const double *row = tile + threadIdx.x * kHistStride;
                int i = 0;
                for (; i + 16 <= cnt_here; i += 16) {  // (the chain of dependent f64 adds bounds this loop, not its reads: with the reads of the next
                    double2 t[8];                       // batch issued ahead -- chain_batches16 -- the statistics of a full history took 13.4 instead of 11.3 us)
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) t[k2] = *reinterpret_cast<const double2 *>(row + i + 2 * k2);
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) {
                        acc += t[k2].x;
                        acc += t[k2].y;
                    }
                }
                if (i < cnt_here) {  // last, partial batch of the tile
                    double t[16];
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) t[k2] = row[i + k2];
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) acc = i + k2 < cnt_here ? acc + t[k2] : acc;
                }
                if (step == ntiles - 1) {  // ref :561
                    if (fast0) acc = s_psum[threadIdx.x];
                    mean = my_len > 0 ? acc / my_len : 0.0;
                    s_mean[threadIdx.x] = mean;
                    acc = 0.0;
                }
            }
        }
        if (threadIdx.x < 8) {
            double sd = 0.0;
            if (my_len > 0) {
                sd = acc / (my_len - 1);  // ref :565
                sd = sqrt(sd);
            }
            s_mean[threadIdx.x] = mean;
            s_std[threadIdx.x] = sd;
        }
    }
    __syncthreads();
    probe();  // 7: threshold statistics
    __shared__ int s_shift[8];
    if (threadIdx.x == 0) {
        for (int i = 0; i < 8; ++i) s_shift[i] = 0;
        for (int i = 0; i < roi; ++i) {  // update_elevation_thr: "continue" on an empty history
            const int len = s_st.elev_len[i];
            if (len == 0) continue;
            if (i == 0) {
                s_st.elevation_thr[i] = s_mean[i] + 3 * s_std[i];
                s_st.sensor_height = -s_mean[i];
            } else {
                s_st.elevation_thr[i] = s_mean[i] + 2 * s_std[i];
            }
            const int exceed = len - P.max_elev_storage;
            if (exceed > 0) {
                s_shift[i] = exceed;
                s_st.elev_len[i] = len - exceed;
            }
        }
        for (int i = 0; i < roi; ++i) {  // update_flatness_thr: "break" at the first history of <= 1
            const int len = s_st.flat_len[i];
            if (len <= 1) break;
            s_st.flatness_thr[i] = s_mean[4 + i] + s_std[4 + i];
            const int exceed = len - P.max_flat_storage;
            if (exceed > 0) {
                s_shift[4 + i] = exceed;
                s_st.flat_len[i] = len - exceed;
            }
        }
        Bt.st_scalar[fd.state_out] = s_st;
        if constexpr (PART == 0) write_plane_state();
        int mx = 0;
        for (int k = 0; k < 4; ++k) {
            mx = s_st.elev_len[k] > mx ? s_st.elev_len[k] : mx;
            mx = s_st.flat_len[k] > mx ? s_st.flat_len[k] : mx;
        }
        Bt.results[f].hist_state = (mx << 1) | s_dropped;  // the host grows the slabs before they run out (pwpp_capi.cpp)
        if constexpr (PART == 2) Bt.results_host[f].hist_state = (mx << 1) | s_dropped;  // (K6 copies the other fields: it may run before this)
    }
    __syncthreads();
    probe();  // 8: state written
    {  // erase(begin, begin + exceed), ref :354-355,372-373: every history at once, loads before stores
        constexpr int kErasePer = 4;
        int sh_w[8], new_w[8], maxnew = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            sh_w[w] = s_shift[w];
            new_w[w] = sh_w[w] > 0 ? (w < 4 ? s_st.elev_len[w] : s_st.flat_len[w - 4]) : 0;
            maxnew = new_w[w] > maxnew ? new_w[w] : maxnew;
        }
        for (int base = 0; base < maxnew; base += kErasePer * kBlock) {
            double e[8][kErasePer];
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int q = 0; q < kErasePer; ++q) {
                    const int i = base + q * kBlock + (int)threadIdx.x;
                    e[w][q] = hist_out[(size_t)w * P.hist_cap + (i < new_w[w] ? i + sh_w[w] : 0)];
                }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int q = 0; q < kErasePer; ++q) {
                    const int i = base + q * kBlock + (int)threadIdx.x;
                    if (i < new_w[w]) hist_out[(size_t)w * P.hist_cap + i] = e[w][q];
                }
            __syncthreads();
        }
    }
    probe();  // 9: end
}

// ------------------------------------------------------------------------------------------
// K6  write the index lists
// ------------------------------------------------------------------------------------------
// One WAVE per bin: a frame has ~500 bins of ~250 points, and with four-wave blocks three of four
// waves were launched only to find nothing to do (2 M waves per batch; the kernel was bound by wave launches).
constexpr int kEmitBlock = 64;
// EAGER (few frames: the kernel is a latency chain): everything a wave needs to know about its bin is fetched at once
// and unconditionally -- nine independent loads, one round trip; a load behind a branch is a round trip of its own
// (single frame 13.2 -> 4.8 us together with eight waves per bin).  Big batches keep the early exit of the empty bins
// in front of the other loads: with 500 k waves the loads of those that have nothing to do cost more (0.23 -> 0.27 ms).
template <bool EAGER, bool KEYS>
__global__ __launch_bounds__(kEmitBlock, 8) void k_emit(PwppBatch Bt, unsigned long long *keys /* reference-order mode: the sort keys of k_order_sublists, indexed like out_idx; else null */) {
    __shared__ int s_stage[512];  // (membership-plane path: the two lists of a block of 512 points, compacted before they are written)
    __shared__ unsigned s_zk[KEYS ? 512 : 1];  // reference-order mode: the height keys of the staged entries
    constexpr bool keep_cat = KEYS;  // (an instantiation of its own: the default path keeps its registers)
    const int f = blockIdx.y;
    // Which blocks of 512 entries of the bin's lists this wave copies: [first_block + part, end_block) in steps of `parts`.  Big
    // batches run one wave per bin; the bins the host has seen long lists in (Bt.emit_long) stop after PWPP_EMIT_LONG_BLOCKS blocks
    // and a second launch over those bins alone (emit_long_pass) deals out the rest.
    int seg = blockIdx.x;
    unsigned first_block = 0u, end_block = 0xffffffu;
    if (Bt.emit_long_pass) {
        seg = Bt.emit_long_list[blockIdx.x];
        first_block = PWPP_EMIT_LONG_BLOCKS;
    } else if (Bt.emit_long && Bt.emit_long[seg]) {
        end_block = PWPP_EMIT_LONG_BLOCKS;
    }
    // the frame's counters are final since K5: hand them to the host through its pinned mirror (eight posted
    // PCIe writes) instead of a copy command behind the pipeline (a dispatch of its own, ~9 us of a single frame)
    if (!Bt.emit_long_pass && seg == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        const PwppFrameResult r = Bt.results[f];
        if (Bt.k5_split) {  // every field but hist_state: K5's second part writes that one from its own stream, before or after this
            PwppFrameResult *d = Bt.results_host + f;
            d->n_ground = r.n_ground;
            d->n_nonground = r.n_nonground;
            d->n_patches = r.n_patches;
            d->n_rnr = r.n_rnr;
            d->n_oor = r.n_oor;
            d->n_dropped = r.n_dropped;
            d->overflow = r.overflow;
        } else {
            Bt.results_host[f] = r;
        }
    }
    const PwppDevParams &P = Bt.P;
    const int B = P.num_bins, NB = B + 2;
    // A pseudo-bin is one part -- its own count / offset stand in -- and has no patch record: that of bin 0 is read and ignored.
    const unsigned n = Bt.bin_count[(size_t)f * NB + seg];
    // (empty bin; or an extra wave of a list the main wave copies whole.  A bin of two parts is copied in blocks of 512 slots PER PART,
    // so it can have one block more than n / 512 says -- a 700 + 3300-point bin is nine blocks: the test leaves a block of slack and
    // the loops below decide exactly.  Found by tools/distinct_parity.py: such bins lost the entries of their last block.)
    if (!EAGER && (n == 0u || (first_block > 0u && n + 512u <= first_block * 512u))) return;
    const PwppFrameDesc fd = Bt.frames[f];
    const unsigned off = Bt.bin_off[(size_t)f * NB + seg];
    const unsigned da = Bt.dst_a[(size_t)f * NB + seg];
    const unsigned db = Bt.dst_b[(size_t)f * NB + seg];
    // (only the bins that ARE split have a high part: the others -- 94 % of the waves of a big batch -- skip the two loads)
    unsigned n_lo = n, off_hi = off;
    if (EAGER || seg < P.split_end) {
        n_lo = Bt.part_count[(size_t)f * PWPP_NUM_PARTS(B) + (seg < B ? PWPP_PART_LO(seg) : B + seg)];
        off_hi = Bt.part_off[(size_t)f * PWPP_NUM_PARTS(B) + (seg < B ? PWPP_PART_HI(seg) : B + seg)];
    }
    const PwppPatchRec *rec = Bt.recs + (size_t)f * B + (seg < B ? seg : 0);
    const int rec_valid = rec->valid;
    if (n == 0) return;
    if (da == kAwaitsFixup) return;  // K5 left the frame alone: the host sees the flag in the mirror and finishes it (k_fit_fixup)
    int *out = Bt.out_idx + fd.base;
    const bool whole = seg >= B || (uint64_t)n < P.min_pts;
    // the bin's high part (pwpp_dev.h): `n_lo` points at `off`, the others at `off_hi`
    const int *idx_lo = Bt.sorted_idx + fd.sbase + off, *idx_hi = Bt.sorted_idx + fd.sbase + off_hi;
    // blockIdx.z = part of the list this wave copies (long lists -- dense clouds have bins of 10^4 points -- are
    // dealt out in blocks of 512 entries to gridDim.z waves; eight loads in flight per lane)
    const unsigned part = blockIdx.z, parts = gridDim.z;
    constexpr int kU = 8;
    if (whole) {  // (the out-of-range pseudo-bin of a sensor that sees beyond max_range holds 10^5 points)
        static_assert(kU * kEmitBlock == 512, "first_block / end_block count blocks of 512 entries");
        const unsigned n_end = (uint64_t)end_block * 512u < n ? end_block * 512u : n;
        for (unsigned i0 = (first_block + part) * (kU * kEmitBlock) + threadIdx.x; i0 < n_end; i0 += parts * (kU * kEmitBlock)) {
            int v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const unsigned i = i0 + u * kEmitBlock;
                v[u] = i < n ? (i < n_lo ? idx_lo[i] : idx_hi[i - n_lo]) : 0;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (i0 + u * kEmitBlock < n) out[da + i0 + u * kEmitBlock] = v[u];
        }
        return;
    }
    const unsigned n_v = n, n_lo_v = n_lo, off_v = off, off_hi_v = off_hi, da_v = da, db_v = db;
    const int member_lg = __builtin_amdgcn_readfirstlane((rec_valid >> 3) & 7);
    if (member_lg == 0) return;  // (no fit kernel leaves a patch without the layout of its bits)
    {
        // (everything about the bin is the same in all lanes: said explicitly, so that counts, offsets and the pointers built
        // from them live in scalar registers -- the kernel must keep eight waves per SIMD)
        const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)n_v), n_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)n_lo_v);
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)off_v), off_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)off_hi_v);
        const unsigned da = (unsigned)__builtin_amdgcn_readfirstlane((int)da_v), db = (unsigned)__builtin_amdgcn_readfirstlane((int)db_v);
        const int *idx_lo = Bt.sorted_idx + fd.sbase + off, *idx_hi = Bt.sorted_idx + fd.sbase + off_hi;
        // The split of the patch is in the MEMBERSHIP PLANE (pwpp_dev.h, PWPP_SLOT_ALIGN): one bit per slot, stored the way
        // the fit rows of G = 2^member_lg lanes saw their chunks (byte c * G + j = the eight points of lane j in chunk c).
        // This wave compacts the two lists itself, in blocks of 512 points of one part: ground entries go to da + (ground
        // points before), the others to db + (points before - ground points before).  "Before" a block = the set bits of the
        // whole chunks in front of it (512 points are whole chunks for every G), counted from the plane; with one wave per
        // bin the counts simply run along.  A high part the last fit pass skipped (rec.valid bit 1) is non-ground unread.
        const uint8_t *mb = Bt.member + fd.mbase;
        const unsigned part_lo = seg < B ? PWPP_PART_LO(seg) : B + seg;
        const uint8_t *m_lo = mb + pwpp_member_offset(off, part_lo, Bt.arena_base, PWPP_NUM_PARTS(B));
        const uint8_t *m_hi = mb + pwpp_member_offset(off_hi, part_lo + 1, Bt.arena_base, PWPP_NUM_PARTS(B));
        const bool hi_bits = !(rec_valid & 2);
        const unsigned n_hi = n - n_lo;
        const unsigned nb_lo = (n_lo + 511u) >> 9, nb = nb_lo + ((n_hi + 511u) >> 9);
        const float *z_lo = Bt.sorted_z + fd.sbase + off, *z_hi2 = Bt.sorted_z + fd.sbase + off_hi;
        const unsigned G = 1u << member_lg, lgG = (unsigned)member_lg;
        auto count_bits = [&](const uint8_t *area, unsigned nbytes) -> unsigned {  // set bits of nbytes (a multiple of 4) bytes, wave-wide
            unsigned c = 0;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(area);
            for (unsigned i = threadIdx.x; i < (nbytes >> 2); i += kEmitBlock) c += (unsigned)__popc(w[i]);
            return (unsigned)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(c));
        };
        unsigned g_before = 0, next_b = 0;  // ground points in the blocks [0, next_b)
        const unsigned nb_end = nb < end_block ? nb : end_block;
        for (unsigned b = first_block + part; b < nb_end; b += parts) {
            const bool hi = b >= nb_lo;
            const unsigned bb = hi ? b - nb_lo : b;
            if (b != next_b) {  // (several waves per bin: count what the other waves' blocks hold)
                if (!hi) {
                    g_before = count_bits(m_lo, bb * 64u);
                } else {  // (all of the low part: its bits end with its last chunk -- beyond lies the pad, never written)
                    g_before = count_bits(m_lo, ((n_lo + (8u << lgG) - 1u) >> (lgG + 3)) << lgG);
                    if (hi_bits) g_before += count_bits(m_hi, bb * 64u);
                }
            }
            const unsigned pn = hi ? n_hi : n_lo, p_before = hi ? n_lo + bb * 512u : bb * 512u;
            const int *pidx = hi ? idx_hi : idx_lo;
            const uint8_t *pm = hi ? m_hi : m_lo;
            const float *pz = hi ? z_hi2 : z_lo;
            const bool bits = !hi || hi_bits;
            // Lane L takes the points 4 L ... 4 L + 3 of each half of the block (16-byte loads of their cloud indices) -- which
            // is how a 64-lane fit row holds them (one byte per lane and chunk: bit 4 q + b), and for the narrower rows four
            // points with consecutive lanes j, the same chunk c and the same bit k: ONE aligned word of the plane per half.
            const unsigned L = threadIdx.x;
            int4 v[2];
            unsigned zk[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};  // reference-order mode: z_key of every point's height
            unsigned mem4[2] = {0u, 0u};  // bit b = point b of the quad is ground
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned p0 = bb * 512u + 256u * q + 4u * L;
                v[q] = make_int4(0, 0, 0, 0);
                if (p0 < pn) {
                    v[q] = *reinterpret_cast<const int4 *>(pidx + p0);  // (a part's slots are padded to a multiple of PWPP_SLOT_ALIGN)
                    if (bits) {
                        if (G == 64u) {
                            mem4[q] = ((unsigned)pm[bb * 64u + L] >> (4 * q)) & 15u;
                        } else {
                            const unsigned c = p0 >> (lgG + 3), j0 = p0 & (G - 1u), k = (p0 >> lgG) & 7u;
                            const unsigned w = *reinterpret_cast<const uint32_t *>(pm + (c << lgG) + j0) >> k;
                            mem4[q] = (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u);
                        }
                    }
                    if (keep_cat) {
                        // Reference-order mode: this kernel hands k_order_sublists complete sort keys -- (R-VPF round, height, cloud
                        // index) -- from the z plane it can read in 16-byte pieces; gathered from the cloud entry by entry in the sort
                        // kernel they were most of its time.  A point R-VPF removed carries the round in place of its height
                        // (strip_point, pwpp_fit.hip): its true height is fetched from the cloud (rare).
                        const float4 zz = *reinterpret_cast<const float4 *>(pz + p0);
                        const float zf[4] = {zz.x, zz.y, zz.z, zz.w};
                        int *vv = reinterpret_cast<int *>(&v[q]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const unsigned zb = __float_as_uint(zf[t]);
                            float zt = zf[t];
                            if ((int)zb > 0x7fc00000) {
                                float x, y, w;
                                if (p0 + (unsigned)t < pn) load_point(fd, vv[t], x, y, zt, w);
                                vv[t] |= (int)((zb & 0xffu) << 24);
                            }
                            zk[q][t] = z_key(zt);
                        }
                    }
                }
            }
            // ranks inside the block: ground entries from the front of the staging tile, the others from its back
            const unsigned in0 = bb * 512u + 4u * L < pn ? (pn - (bb * 512u + 4u * L) < 4u ? pn - (bb * 512u + 4u * L) : 4u) : 0u;
            const unsigned in1 = bb * 512u + 256u + 4u * L < pn ? (pn - (bb * 512u + 256u + 4u * L) < 4u ? pn - (bb * 512u + 256u + 4u * L) : 4u) : 0u;
            mem4[0] &= (1u << in0) - 1u;
            mem4[1] &= (1u << in1) - 1u;
            const unsigned g0 = (unsigned)__popc(mem4[0]), g1 = (unsigned)__popc(mem4[1]);
            // running sums over the lanes, two per scan (ground points | points inside, 16 bits each: at most 256)
            unsigned eg0, eg1, ei0, ei1, tg0, tg1, ti0, ti1;
            {
                const unsigned a = g0 | (in0 << 16), bq = g1 | (in1 << 16);
                const unsigned ia = wave_incl_scan(a), ib = wave_incl_scan(bq);
                const unsigned ta = (unsigned)__builtin_amdgcn_readlane((int)ia, 63), tb = (unsigned)__builtin_amdgcn_readlane((int)ib, 63);
                eg0 = (ia - a) & 0xffffu;
                ei0 = (ia - a) >> 16;
                eg1 = (ib - bq) & 0xffffu;
                ei1 = (ib - bq) >> 16;
                tg0 = ta & 0xffffu;
                ti0 = ta >> 16;
                tg1 = tb & 0xffffu;
                ti1 = tb >> 16;
            }
            const unsigned gb = tg0 + tg1, nbk = ti0 + ti1;  // ground points / points of this block
            __syncthreads();  // (the tile of the block before has been copied out)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned gr = q == 0 ? eg0 : tg0 + eg1;                        // ground entries in front of this quad
                unsigned nr = q == 0 ? ei0 - eg0 : (ti0 - tg0) + (ei1 - eg1);  // other entries in front of it
                const int *vv = reinterpret_cast<const int *>(&v[q]);
                const unsigned inq = q == 0 ? in0 : in1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if ((unsigned)t < inq) {
                        if (mem4[q] >> t & 1u) {
                            if (keep_cat) s_zk[gr] = zk[q][t];
                            s_stage[gr++] = vv[t];
                        } else {
                            if (keep_cat) s_zk[511u - nr] = zk[q][t];
                            s_stage[511u - (nr++)] = vv[t];
                        }
                    }
                }
            }
            __syncthreads();
            int *og = out + (da + g_before), *on = out + (db + (p_before - g_before));  // (scalar bases, 32-bit lane offsets)
            const unsigned nn = nbk - gb;
            if (!keep_cat) {
#pragma unroll 2
                for (unsigned t = L; t < gb; t += kEmitBlock) og[t] = s_stage[t];
#pragma unroll 2
                for (unsigned t = L; t < nn; t += kEmitBlock) on[t] = s_stage[511u - t];
            } else {  // key = category (0 for ground; the R-VPF round, or 255 for the points of the final split) | height | cloud index
                unsigned long long *kg = keys + fd.base + (da + g_before), *kn = keys + fd.base + (db + (p_before - g_before));
                // (out_idx itself is written by k_order_sublists, which sorts every one of these lists)
                for (unsigned t = L; t < gb; t += kEmitBlock) kg[t] = ((unsigned long long)s_zk[t] << 24) | (unsigned long long)(unsigned)s_stage[t];
                for (unsigned t = L; t < nn; t += kEmitBlock) {
                    const unsigned e = (unsigned)s_stage[511u - t], round = e >> 24;
                    kn[t] = ((unsigned long long)(round ? round : 255u) << 56) | ((unsigned long long)s_zk[511u - t] << 24) | (unsigned long long)(e & 0x00ffffffu);
                }
            }
            g_before += gb;
            next_b = b + 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K7 (optional)  reference order inside the sub-lists
// k_gle_tgr / k_emit already put the SUB-LISTS where the reference appends them (bin traversal order,
// TGR candidates at the end of their ring).  Inside a sub-list the reference's order is that of the
// z-sorted bin (ref :199): ground candidates ascending in z; regionwise_nonground_ = the points R-VPF
// removed, round by round, each round ascending in z, then the rest ascending in z (ref :500,:532);
// small bins, RNR hits and out-of-range points in cloud order.  This kernel sorts every sub-list of
// out_idx by (R-VPF round, z, cloud index).  Equal z: the reference's std::sort is unstable, so its
// order among ties is an artefact of libstdc++; ties come out in cloud order here.
// One workgroup per (frame, bin); sub-lists up to 4096 entries are sorted in LDS (bitonic), longer
// ones tile by tile and then merged through two scratch arrays indexed like out_idx.
// ------------------------------------------------------------------------------------------
// The keys of a tile live in LDS with one slot of padding per sixteen (a thread's run of consecutive keys then starts
// in its own bank group: sixteen 8-byte keys are 128 bytes, and unpadded all lanes would hit the same two banks).
__device__ __forceinline__ int ord_at(int i) { return i + (i >> 4); }

// compare-exchange of a bitonic network
__device__ __forceinline__ void ord_ce(unsigned long long &a, unsigned long long &b, bool up) {
    const bool sw = (a > b) == up;
    const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}

// Bitonic sort of np = BLOCK * E keys (E a power of two up to 16): the stages whose partners lie within a thread's E
// consecutive keys run in REGISTERS -- every phase ends with log2(E) of them, the first log2(E) phases consist of
// nothing else -- and only the others go through LDS: 44 barriers and 45 passes over the tile instead of 78 and 78
// for 4096 keys (the sort was 1.0 of the 1.3 ms the reference-order mode adds to 256 frames).
template <int BLOCK, int E>
__device__ __forceinline__ void ord_sort_blocked(unsigned long long *s_key) {
    constexpr int np = BLOCK * E;
    const int t = threadIdx.x, base = t * E;
    unsigned long long r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = s_key[ord_at(base + e)];
#pragma unroll
    for (int k = 2; k <= E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & j) == 0) ord_ce(r[e], r[e | j], ((base + e) & k) == 0);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s_key[ord_at(base + e)] = r[e];
    __syncthreads();
    for (int k = 2 * E; k <= np; k <<= 1) {
        for (int j = k >> 1; j >= E; j >>= 1) {  // partners in different threads' runs: through LDS
#pragma unroll
            for (int q = 0; q < E / 2; ++q) {
                const int p = t + q * BLOCK;
                const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
                unsigned long long a = s_key[ord_at(lo)], b = s_key[ord_at(hi)];
                ord_ce(a, b, (lo & k) == 0);
                s_key[ord_at(lo)] = a;
                s_key[ord_at(hi)] = b;
            }
            __syncthreads();
        }
        const bool up = (base & k) == 0;  // (k >= 2 E: the same for the whole run)
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = s_key[ord_at(base + e)];
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & j) == 0) ord_ce(r[e], r[e | j], up);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) s_key[ord_at(base + e)] = r[e];
        __syncthreads();
    }
}

template <int BLOCK>
__device__ __forceinline__ void ord_tile_sort(unsigned long long *s_key, int len) {  // len <= tile size, padded with ~0
    int np = 1;
    while (np < len) np <<= 1;
    for (int i = len + (int)threadIdx.x; i < np; i += BLOCK) s_key[ord_at(i)] = ~0ull;
    __syncthreads();
    switch (np / BLOCK) {  // keys per thread
        case 16: ord_sort_blocked<BLOCK, 16>(s_key); return;
        case 8: ord_sort_blocked<BLOCK, 8>(s_key); return;
        case 4: ord_sort_blocked<BLOCK, 4>(s_key); return;
        case 2: ord_sort_blocked<BLOCK, 2>(s_key); return;
        default: break;
    }
    for (int k = 2; k <= np; k <<= 1) {  // fewer keys than threads (or a tile size this file does not use)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < np / 2; t += BLOCK) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = s_key[ord_at(lo)], b = s_key[ord_at(hi)];
                if ((a > b) == up) {
                    s_key[ord_at(lo)] = b;
                    s_key[ord_at(hi)] = a;
                }
            }
            __syncthreads();
        }
    }
}
// The lists above 256 entries (the ~100 ground / non-ground lists of a frame's big bins: 3.4 of the 3.9 ms the mode added to
// a 1024-frame batch while every one of them went through the bitonic network, padded to 2048 or 4096 keys) are first offered
// to a DISTRIBUTION sort: the heights of a list spread over a narrow range fairly evenly, so 1024 buckets by height --
// floor((z - zmin) * 1024 / (zmax - zmin)), a chain of monotone float operations, hence consistent with the key order -- hold a
// key or two each; count, scan, scatter, then every bucket is put in order by insertion on the full 64-bit key.  O(n) instead of
// O(n log^2 n) compare-exchanges.  Declined (the network takes the tile as before) when the list mixes R-VPF categories,
// holds a non-finite height, has one height only, or some bucket gets more than kOrdMaxBucket keys -- cloud-order lists
// (small bins, pseudo-bins: one "height") always are.
#ifndef PWPP_ORDER_BLOCK
#define PWPP_ORDER_BLOCK 256  // threads of the workgroup that sorts a long list
#endif
#ifndef PWPP_ORDER_BUCKETS
#define PWPP_ORDER_BUCKETS 1
#endif
constexpr int kOrdBuckets = 1024, kOrdMaxBucket = 32;
// load(i): key i of the tile (from global memory: read three times, coalesced); store(pos, key): the key's final place.
// lds_out[len], cnt / start[kOrdBuckets], red[4][4]: LDS.  Returns false (nothing stored) when the tile is declined.
template <int BLOCK, class Load, class Store>
__device__ __forceinline__ bool ord_bucket_sort(Load load, Store store, unsigned long long *lds_out, unsigned *cnt, unsigned *start, unsigned (*red)[4], int len) {
    constexpr int kPerT = kOrdBuckets / BLOCK;  // buckets per thread in the scan
    static_assert(kPerT * BLOCK == kOrdBuckets && BLOCK <= 512, "whole buckets per thread; the waves' partial results fit sixteen slots");
    const int t = threadIdx.x;
    unsigned zlo = 0xffffffffu, zhi = 0u, clo = 255u, chi = 0u;
#pragma unroll 4
    for (int i = t; i < len; i += BLOCK) {
        const unsigned long long k = load(i);
        const unsigned zk = (unsigned)(k >> 24), c = (unsigned)(k >> 56);
        zlo = zk < zlo ? zk : zlo;
        zhi = zk > zhi ? zk : zhi;
        clo = c < clo ? c : clo;
        chi = c > chi ? c : chi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)zlo, o, 64), b = (unsigned)__shfl_xor((int)zhi, o, 64);
        const unsigned c = (unsigned)__shfl_xor((int)clo, o, 64), d = (unsigned)__shfl_xor((int)chi, o, 64);
        zlo = a < zlo ? a : zlo;
        zhi = b > zhi ? b : zhi;
        clo = c < clo ? c : clo;
        chi = d > chi ? d : chi;
    }
    if (lane_id() == 0) {
        red[wave_id()][0] = zlo;
        red[wave_id()][1] = zhi;
        red[wave_id()][2] = clo;
        red[wave_id()][3] = chi;
    }
    for (int i = t; i < kOrdBuckets; i += BLOCK) cnt[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        zlo = red[w][0] < zlo ? red[w][0] : zlo;
        zhi = red[w][1] > zhi ? red[w][1] : zhi;
        clo = red[w][2] < clo ? red[w][2] : clo;
        chi = red[w][3] > chi ? red[w][3] : chi;
    }
    __syncthreads();  // (red is used again below)
    // (everything below is the same in every thread)
    if (clo != chi) return false;                                    // several R-VPF categories
    if (zlo <= 0x007fffffu || zhi >= 0xff800000u || zlo == zhi) return false;  // a non-finite height (z_key of -inf / +inf, NaN beyond) or one height
    const float fz0 = key_z(zlo), range = key_z(zhi) - fz0;
    const float scale = (float)kOrdBuckets / range;
    if (!(scale <= FLT_MAX) || !(scale > 0.0f)) return false;
    auto bucket = [&](unsigned long long k) -> int {  // monotone in the height: rounded subtraction, product with a positive constant, truncation
        const float v = (key_z((unsigned)(k >> 24)) - fz0) * scale;
        const int b = (int)v;
        return b < 0 ? 0 : (b > kOrdBuckets - 1 ? kOrdBuckets - 1 : b);
    };
#pragma unroll 4
    for (int i = t; i < len; i += BLOCK) atomicAdd(&cnt[bucket(load(i))], 1u);
    __syncthreads();
    unsigned cb[kPerT], mine = 0;
    int skew = 0;
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        cb[q] = cnt[kPerT * t + q];
        mine += cb[q];
        skew |= cb[q] > (unsigned)kOrdMaxBucket;
    }
    const unsigned incl = wave_incl_scan(mine);
    if (lane_id() == 63) red[wave_id()][0] = incl;
    if (__syncthreads_or(skew)) return false;
    unsigned before = 0;
    for (int w = 0; w < wave_id(); ++w) before += red[w][0];
    unsigned run = before + incl - mine;
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        start[kPerT * t + q] = run;
        run += cb[q];
        cnt[kPerT * t + q] = 0u;  // (now the bucket's cursor)
    }
    __syncthreads();
#pragma unroll 4
    for (int i = t; i < len; i += BLOCK) {
        const unsigned long long k = load(i);
        const int b = bucket(k);
        lds_out[start[b] + atomicAdd(&cnt[b], 1u)] = k;
    }
    __syncthreads();
    // order inside the buckets: every key counts the smaller keys of its bucket (independent reads, a thread per KEY -- an
    // insertion sort per bucket by one thread is a chain of dependent LDS round trips as long as the fullest bucket squared)
    // and goes straight to its final place.  Keys are distinct (they end in the cloud index).
#pragma unroll 2
    for (int i = t; i < len; i += BLOCK) {
        const unsigned long long k = lds_out[i];
        const int b = bucket(k);
        const unsigned lo = start[b], n = cnt[b];
        unsigned rank = 0;
        for (unsigned j = 0; j < n; ++j) rank += lds_out[lo + j] < k ? 1u : 0u;
        store(lo + rank, k);
    }
    __syncthreads();
    return true;
}
// Two instantiations: <64, 256, 0> one wave per bin for the short lists (most of them: ~250 points),
// <256, 4096, 256> a workgroup for the lists above 256 entries.  Each handles the lists in ITS range.
template <int BLOCK, int kOrdTile, int kMinLen>
__global__ __launch_bounds__(BLOCK) void k_order_sublists(PwppBatch Bt, unsigned long long *scr_a, unsigned long long *scr_b) {
    constexpr bool kBuckets = PWPP_ORDER_BUCKETS && BLOCK >= 256 && kOrdTile == 4096;  // the workgroup instantiation tries the distribution sort first
    // ONE piece of LDS for both sorts (40 KB: four workgroups per CU): the network's padded tile (ord_at), or the distribution
    // sort's scatter array + bucket counters + bucket starts -- it reads its keys from global memory
    constexpr int kOutSlots = 4096 - 48;  // (40 KB less the few hundred bytes the compiler's own LDS takes: FOUR workgroups per CU, not three)
    constexpr int kRaw = kBuckets ? kOutSlots + kOrdBuckets : kOrdTile + kOrdTile / 16;
    static_assert(kRaw >= kOrdTile + kOrdTile / 16, "the network's tile fits");
    __shared__ unsigned long long s_raw[kRaw];
    unsigned long long *s_key = s_raw;
    // (the distribution sort takes tiles of up to kBucketTile keys: the last sixteen slots of its scatter array hold the four waves'
    // partial results -- 40 KB exactly, not a byte more, or only three workgroups fit a CU)
    constexpr int kBucketTile = kOutSlots - 16;
    unsigned(*s_red)[4] = reinterpret_cast<unsigned(*)[4]>(s_raw + kBucketTile);
    const int f = blockIdx.y;
    const PwppDevParams &P = Bt.P;
    const int B = P.num_bins, NB = B + 2;
    // The wave instantiation (kMinLen == 0) takes bin blockIdx.x and both of its sub-lists.  The workgroup instantiation takes ONE
    // sub-list from the frame's work list (k_order_worklist: the sub-lists above kMinLen entries, ~100 of a KITTI frame's ~1000):
    // a workgroup without an item leaves after one load that its whole frame shares, instead of three dependent loads per bin.
    int seg = blockIdx.x, which_lo = 0, which_hi = 2;
    if constexpr (kMinLen > 0) {
        const uint32_t *work = Bt.order_work + (size_t)f * (size_t)(1 + 2 * NB);
        if (blockIdx.x >= work[0]) return;
        const uint32_t item = work[1 + blockIdx.x];
        seg = (int)(item & 0xffffu);
        which_lo = (int)(item >> 16);
        which_hi = which_lo + 1;
    }
    const unsigned n = Bt.bin_count[(size_t)f * NB + seg];
    if (n < 1) return;
    if (Bt.dst_a[(size_t)f * NB + seg] == kAwaitsFixup) return;  // (the frame awaits k_fit_fixup)
    const PwppFrameDesc fd = Bt.frames[f];
    int *out = Bt.out_idx + fd.base;
    const bool whole = seg >= B || (uint64_t)n < P.min_pts;
    const unsigned ng = whole ? n : (unsigned)Bt.recs[(size_t)f * B + seg].n_ground;
    for (int which = which_lo; which < which_hi; ++which) {
        const unsigned start = which == 0 ? Bt.dst_a[(size_t)f * NB + seg] : Bt.dst_b[(size_t)f * NB + seg];
        const int len = (int)(which == 0 ? ng : n - ng);
        if (len < 1 || len <= kMinLen || (kMinLen == 0 && len > kOrdTile)) continue;  // not this instantiation's range
        // the keys: k_emit left (category, height, cloud index) of every patch entry in scr_a, indexed like out_idx; a small bin or
        // pseudo-bin is in cloud order: its key is the index itself
        const unsigned long long *keys = scr_a + fd.base + start;
        auto load_key = [&](int i) -> unsigned long long { return whole ? (unsigned long long)(unsigned)(out[start + i] & 0x00ffffff) : keys[i]; };
        if (len <= kOrdTile) {
            bool done = false;
            if constexpr (kBuckets)
                if (len <= kBucketTile) done = ord_bucket_sort<BLOCK>(load_key, [&](unsigned pos, unsigned long long k) { out[start + pos] = (int)(k & 0x00ffffffull); }, s_raw,
                                            reinterpret_cast<unsigned *>(s_raw + kOutSlots), reinterpret_cast<unsigned *>(s_raw + kOutSlots) + kOrdBuckets, s_red, len);
            if (!done) {
                for (int i = threadIdx.x; i < len; i += BLOCK) s_key[ord_at(i)] = load_key(i);
                __syncthreads();
                ord_tile_sort<BLOCK>(s_key, len);
                for (int i = threadIdx.x; i < len; i += BLOCK) out[start + i] = (int)(s_key[ord_at(i)] & 0x00ffffffull);
            }
            __syncthreads();
            continue;
        }
        // long sub-list: sorted tiles, then merge passes a -> b -> a ...
        unsigned long long *a = scr_a + fd.base + start, *b = scr_b + fd.base + start;
        for (int t0 = 0; t0 < len; t0 += kOrdTile) {
            const int tl = len - t0 < kOrdTile ? len - t0 : kOrdTile;
            bool done = false;
            if constexpr (kBuckets)
                if (tl <= kBucketTile) done = ord_bucket_sort<BLOCK>([&](int i) { return load_key(t0 + i); }, [&](unsigned pos, unsigned long long k) { a[t0 + pos] = k; }, s_raw,
                                            reinterpret_cast<unsigned *>(s_raw + kOutSlots), reinterpret_cast<unsigned *>(s_raw + kOutSlots) + kOrdBuckets, s_red, tl);
            if (!done) {
                for (int i = threadIdx.x; i < tl; i += BLOCK) s_key[ord_at(i)] = load_key(t0 + i);
                __syncthreads();
                ord_tile_sort<BLOCK>(s_key, tl);
                for (int i = threadIdx.x; i < tl; i += BLOCK) a[t0 + i] = s_key[ord_at(i)];
            }
            __syncthreads();
        }
        for (int width = kOrdTile; width < len; width <<= 1) {
            __threadfence_block();
            __syncthreads();
            for (int i = threadIdx.x; i < len; i += BLOCK) {
                const int pair0 = (i / (2 * width)) * (2 * width);           // first element of the pair of runs
                const int mid = pair0 + width < len ? pair0 + width : len;  // [pair0, mid) and [mid, end)
                const int end = pair0 + 2 * width < len ? pair0 + 2 * width : len;
                const unsigned long long key = a[i];
                int lo, hi;
                const bool left = i < mid;
                if (left) { lo = mid; hi = end; } else { lo = pair0; hi = mid; }
                while (lo < hi) {  // keys are distinct (they end in the cloud index): plain lower bound
                    const int m = (lo + hi) >> 1;
                    if (a[m] < key) lo = m + 1; else hi = m;
                }
                const int rank_other = left ? lo - mid : lo - pair0;
                const int own = left ? i - pair0 : i - mid;
                b[pair0 + own + rank_other] = key;
            }
            unsigned long long *t = a;
            a = b;
            b = t;
        }
        __threadfence_block();
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += BLOCK) out[start + i] = (int)(a[i] & 0x00ffffffull);
        __syncthreads();
    }
}

// the sub-lists above `min_len` entries of every frame, for the workgroup instantiation of k_order_sublists
__global__ __launch_bounds__(kBlock) void k_order_worklist(PwppBatch Bt, int min_len) {
    __shared__ unsigned s_n;
    const int f = blockIdx.x;
    const PwppDevParams &P = Bt.P;
    const int B = P.num_bins, NB = B + 2;
    uint32_t *work = Bt.order_work + (size_t)f * (size_t)(1 + 2 * NB);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (Bt.dst_a[(size_t)f * NB] != kAwaitsFixup) {  // (a frame that awaits k_fit_fixup has no lists yet)
        for (int seg = threadIdx.x; seg < NB; seg += kBlock) {
            const unsigned n = Bt.bin_count[(size_t)f * NB + seg];
            if (n <= (unsigned)min_len) continue;
            const bool whole = seg >= B || (uint64_t)n < P.min_pts;
            const unsigned ng = whole ? n : (unsigned)Bt.recs[(size_t)f * B + seg].n_ground;
            if (ng > (unsigned)min_len) work[1 + atomicAdd(&s_n, 1u)] = (uint32_t)seg;
            if (n - ng > (unsigned)min_len) work[1 + atomicAdd(&s_n, 1u)] = (uint32_t)seg | (1u << 16);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) work[0] = s_n;
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
extern "C" int pwpp_launch_fit(const PwppBatch *batch, hipStream_t stream, hipEvent_t *ev, hipStream_t aux,
                               hipEvent_t aux_fork, hipEvent_t aux_join);
extern "C" int pwpp_launch_fixup(const PwppBatch *batch, hipStream_t stream);

// K0: the per-launch zeroing (histogram / cursor slabs and the frame counters) in ONE dispatch; two
// hipMemsetAsync calls were three fill kernels of the runtime, ~6 us apart on the queue
namespace {
__global__ __launch_bounds__(kBlock) void k_clear(uint4 *slabs, size_t n16, PwppFrameResult *results, int frames) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n16) slabs[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i < (size_t)frames) {
        PwppFrameResult z;
        z.n_ground = z.n_nonground = z.n_patches = z.n_rnr = z.n_oor = z.n_dropped = z.hist_state = z.overflow = 0;
        results[i] = z;
    }
}
}  // namespace

// part_count (+ part_off, part_cursor on the two-pass path: adjacent slabs) and the result counters start at zero
static void launch_clear(const PwppBatch &B, hipStream_t stream) {
    const int F = B.num_frames, NB = PWPP_NUM_PARTS(B.P.num_bins);
    const size_t words = (size_t)(B.cap_off ? 1 : 3) * (size_t)F * (size_t)NB;
    const size_t n16 = (words + 3) / 4;  // (the slabs are allocated with a few words to spare: rounding up is harmless)
    const size_t items = n16 > (size_t)F ? n16 : (size_t)F;
    hipLaunchKernelGGL(k_clear, dim3((unsigned)((items + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       reinterpret_cast<uint4 *>(B.part_count), n16, B.results, F);
}
extern "C" int pwpp_launch_clear(const PwppBatch *batch, hipStream_t stream) {
    if (batch->num_frames > 0) launch_clear(*batch, stream);
    return (int)hipGetLastError();
}

// K5's second part on its own (k5_split = 2): the caller has recorded aux_fork on the main stream where the part may start
extern "C" int pwpp_launch_k5_tail(const PwppBatch *batch, hipStream_t aux, hipEvent_t aux_fork, hipEvent_t aux_join) {
    const PwppBatch &B = *batch;
    (void)hipStreamWaitEvent(aux, aux_fork, 0);
    if (B.P.num_bins <= 2 * kBlock) hipLaunchKernelGGL((k_gle_tgr<true, 2, 2>), dim3(B.num_frames), dim3(kBlock), 0, aux, B);
    else hipLaunchKernelGGL((k_gle_tgr<true, kGlePer, 2>), dim3(B.num_frames), dim3(kBlock), 0, aux, B);
    (void)hipEventRecord(aux_join, aux);
    return (int)hipGetLastError();
}

// K0 + K1 + K2 only: the histogram of a few sample frames (sizes the one-pass segments before the first batch)
// dynamic LDS of the binning kernels: one (K1), two (K3) or two-and-a-bit (K1') words per part of this model
static size_t binning_lds_bytes(const PwppBatch &B, int words_per_part) {
    return ((size_t)words_per_part * (size_t)PWPP_NUM_PARTS(B.P.num_bins) + 4u) * sizeof(unsigned);
}
extern "C" int pwpp_launch_histogram(const PwppBatch *batch, hipStream_t stream) {
    const PwppBatch &B = *batch;
    const int F = B.num_frames;
    if (F <= 0) return 0;
    const unsigned gx = (unsigned)((B.max_n + kPtsPerBlock - 1) / kPtsPerBlock);
    launch_clear(B, stream);
    if (gx > 0) hipLaunchKernelGGL(k_czm_bin, dim3(gx, F), dim3(kBlock), binning_lds_bytes(B, 1), stream, B);
    hipLaunchKernelGGL(k_czm_scan, dim3(F), dim3(kBlock), 0, stream, B);
    return (int)hipGetLastError();
}

// stages: bit 0 = binning (K0-K3), bit 1 = plane fits + K5, bit 2 = index lists (K6, K7); the overlap schedule of
// pwpp_capi.cpp launches the stages of a frame range on different streams.  Bit 3 (with bit 1): k_fit_fixup instead of the
// fit kernels -- the host finishing a frame whose patches needed the plane fitted before them (batch->fixup_run set).
extern "C" int pwpp_launch_pipeline(const PwppBatch *batch, hipStream_t stream, hipEvent_t *ev /* PWPP_NUM_KERNELS + 1 events or null */,
                                    hipStream_t aux, hipEvent_t aux_fork, hipEvent_t aux_join,
                                    unsigned long long *order_a /* reference-order mode: two scratch arrays, else null */,
                                    unsigned long long *order_b, int stages) {
    const PwppBatch &B = *batch;
    const int F = B.num_frames;
    if (F <= 0) return 0;
    const int NB = B.P.num_bins + 2;
    const unsigned gx = (unsigned)((B.max_n + kPtsPerBlock - 1) / kPtsPerBlock);
    if (stages & 1) {
        if (!B.no_clear) launch_clear(B, stream);
        if (ev) (void)hipEventRecord(ev[0], stream);
        if (B.cap_off) {  // one-pass binning (fixed bin segments)
            const int bb = B.bin_block == 1024 ? 1024 : (B.bin_block == 512 ? 512 : (B.bin_block == 128 ? 128 : 256));
            const unsigned gx1 = (unsigned)((B.max_n + 4 * bb - 1) / (4 * bb));
            const bool spread = F < 8;  // (see the kernel: a few frames are dealt over all XCDs)
            const dim3 grid(gx1 * (unsigned)(spread ? F : (F + 7) / 8 * 8));
            const int tpf = spread ? -(int)gx1 : (int)gx1;
            // K2 inside K1' (option fuse_scan; debug 256 selects the OTHER variant; 128 is taken by K5's statistics).  Rounds 4-5: the default for
            // a few frames -- no kernel boundary in a single frame's chain.  Round 6: off -- since the ticket that elects the scanning
            // workgroup is an agent-scope acquire-release it costs what the boundary cost (the L2 write-back of a release:
            // ~3 us), and the two kernels are 1-2 us FASTER on every KITTI sample.
            const bool fused = spread && gx1 > 0 && bb == kBlock && ((B.fuse_scan != 0) != ((B.debug & 256) != 0));
            if (fused) {
                const dim3 fgrid(gx1 * (unsigned)F + (B.snap_scalar ? 8u * (unsigned)F : 0u));
                hipLaunchKernelGGL((k_czm_bin_scatter<kBlock, true>), fgrid, dim3(kBlock), binning_lds_bytes(B, 2), stream, B, tpf);
                if (ev) (void)hipEventRecord(ev[1], stream);
                if (ev) (void)hipEventRecord(ev[2], stream);
            } else {
            if (gx1 > 0) {
                if (bb == 128) hipLaunchKernelGGL(k_czm_bin_scatter<128>, grid, dim3(128), binning_lds_bytes(B, 2), stream, B, tpf);
                else if (bb == 256) hipLaunchKernelGGL(k_czm_bin_scatter<256>, grid, dim3(256), binning_lds_bytes(B, 2), stream, B, tpf);
                else if (bb == 512) hipLaunchKernelGGL(k_czm_bin_scatter<512>, grid, dim3(512), binning_lds_bytes(B, 2), stream, B, tpf);
                else hipLaunchKernelGGL(k_czm_bin_scatter<1024>, grid, dim3(1024), binning_lds_bytes(B, 2), stream, B, tpf);
            }
            if (ev) (void)hipEventRecord(ev[1], stream);
            hipLaunchKernelGGL(k_czm_scan, dim3(F, B.snap_scalar ? 9 : 1), dim3(kBlock), 0, stream, B);  // (+ eight snapshot workgroups per frame)
            if (ev) (void)hipEventRecord(ev[2], stream);
            }
        } else {
            if (gx > 0) hipLaunchKernelGGL(k_czm_bin, dim3(gx, F), dim3(kBlock), binning_lds_bytes(B, 1), stream, B);
            if (ev) (void)hipEventRecord(ev[1], stream);
            hipLaunchKernelGGL(k_czm_scan, dim3(F), dim3(kBlock), 0, stream, B);
            if (ev) (void)hipEventRecord(ev[2], stream);
            if (gx > 0) hipLaunchKernelGGL(k_czm_scatter, dim3(gx, F), dim3(kBlock), binning_lds_bytes(B, 2), stream, B);
        }
    }
    if (stages & 2) {
        const int frc = (stages & 8) ? pwpp_launch_fixup(batch, stream)
                                     : pwpp_launch_fit(batch, stream, ev ? ev + 3 : nullptr, aux, aux_fork, aux_join);  // records ev[3..9]
        if (frc) return frc;
        if (B.P.min_pts == 0)
            hipLaunchKernelGGL(k_gle_tgr_seq, dim3(F), dim3(64), 0, stream, B);  // empty bins inherit planes: serial
        else if (F <= 64 && B.k5_split && !ev && aux && aux_fork && aux_join && !(stages & 8)) {
            // a stream's chain: the lists wait for the first part only, the second runs on the other stream (joined by the caller: aux_join)
            if (B.P.num_bins <= 2 * kBlock) hipLaunchKernelGGL((k_gle_tgr<true, 2, 1>), dim3(F), dim3(kBlock), 0, stream, B);
            else hipLaunchKernelGGL((k_gle_tgr<true, kGlePer, 1>), dim3(F), dim3(kBlock), 0, stream, B);
            if (B.k5_split == 1) {  // (2: the caller launches the second part behind the lists, pwpp_launch_k5_tail)
                (void)hipEventRecord(aux_fork, stream);
                (void)hipStreamWaitEvent(aux, aux_fork, 0);
                if (B.P.num_bins <= 2 * kBlock) hipLaunchKernelGGL((k_gle_tgr<true, 2, 2>), dim3(F), dim3(kBlock), 0, aux, B);
                else hipLaunchKernelGGL((k_gle_tgr<true, kGlePer, 2>), dim3(F), dim3(kBlock), 0, aux, B);
                (void)hipEventRecord(aux_join, aux);
            }
        } else if (F <= 64) {  // every workgroup alone on a CU: the big-LDS variant
            if (B.P.num_bins <= 2 * kBlock) hipLaunchKernelGGL((k_gle_tgr<true, 2>), dim3(F), dim3(kBlock), 0, stream, B);
            else hipLaunchKernelGGL((k_gle_tgr<true, kGlePer>), dim3(F), dim3(kBlock), 0, stream, B);
        } else {
            if (B.P.num_bins <= 2 * kBlock) hipLaunchKernelGGL((k_gle_tgr<false, 2>), dim3(F), dim3(kBlock), 0, stream, B);
            else hipLaunchKernelGGL((k_gle_tgr<false, kGlePer>), dim3(F), dim3(kBlock), 0, stream, B);
        }
        if (ev) (void)hipEventRecord(ev[10], stream);
    }
    if (stages & 4) {
        const dim3 egrid(NB, F, B.emit_parts > 1 ? B.emit_parts : 1);
        if (F <= 64 && order_a) hipLaunchKernelGGL((k_emit<true, true>), egrid, dim3(kEmitBlock), 0, stream, B, order_a);
        else if (F <= 64) hipLaunchKernelGGL((k_emit<true, false>), egrid, dim3(kEmitBlock), 0, stream, B, order_a);
        else if (order_a) hipLaunchKernelGGL((k_emit<false, true>), egrid, dim3(kEmitBlock), 0, stream, B, order_a);
        else hipLaunchKernelGGL((k_emit<false, false>), egrid, dim3(kEmitBlock), 0, stream, B, order_a);
        if (B.emit_long && B.emit_long_n > 0) {  // the rest of the long lists: extra waves for the listed bins only
            PwppBatch L = B;
            L.emit_long_pass = 1;
            const dim3 lgrid(B.emit_long_n, F, B.emit_long_parts > 1 ? B.emit_long_parts : 1);
            if (order_a) hipLaunchKernelGGL((k_emit<false, true>), lgrid, dim3(kEmitBlock), 0, stream, L, order_a);
            else hipLaunchKernelGGL((k_emit<false, false>), lgrid, dim3(kEmitBlock), 0, stream, L, order_a);
        }
        if (ev) (void)hipEventRecord(ev[11], stream);
        if (order_a) {
            hipLaunchKernelGGL((k_order_sublists<64, 256, 0>), dim3(NB, F), dim3(64), 0, stream, B, order_a, order_b);
            // the lists above 256 entries: a work list per frame, then one workgroup per item (at most points / 257 + the two a bin can add)
            const unsigned max_items = (unsigned)(B.max_n / 257 + 2) < 2u * (unsigned)NB ? (unsigned)(B.max_n / 257 + 2) : 2u * (unsigned)NB;
            hipLaunchKernelGGL(k_order_worklist, dim3(F), dim3(kBlock), 0, stream, B, 256);
            hipLaunchKernelGGL((k_order_sublists<PWPP_ORDER_BLOCK, 4096, 256>), dim3(max_items, F), dim3(PWPP_ORDER_BLOCK), 0, stream, B, order_a, order_b);
        }
    }
    return (int)hipGetLastError();
}
