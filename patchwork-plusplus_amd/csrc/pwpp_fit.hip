// pwpp_fit.hip -- K4: per-patch plane fitting (LPR seeds, R-VPF, R-GPF, final plane), the
// 58 % of the reference's CPU time (ref :467-549 extract_piecewiseground, :77-149
// extract_initial_seeds, :47-75 estimate_plane, :551-554 calc_point_to_plane_d, plus the
// per-bin std::sort of :199 which this design does not need).
//
// A patch is a CZM bin with >= num_min_pts points: 10 ... ~30 000 points, median ~170.  Its
// work is a chain of 4-7 dependent plane fits, each = one pass over the points + a serial
// 3x3 eigen-solve (Eigen's Jacobi, ~1500 instructions).  k_czm_scan sorts the patches of a
// frame into quarter-octave size buckets; a PLAN (pwpp_launch_fit, bottom of this file) maps
// bucket ranges to kernels, chosen by the amount of work in the batch:
//
//   k_fit_w64<16,64|32|16>  64/32/16 small patches per wave: points phases in rows of 16 lanes
//                           (4 patches at a time), then ONE solve phase with a patch per lane
//   k_fit_w64<64,p>         big bins, p per wave: 64 lanes stream one patch at a time, the
//                           p solves share an instruction stream; dual seed pass
//   k_fit_srows<G>          one row of G lanes per patch, next chunk prefetched (mid-size batches)
//   k_fit_brows             four waves per patch; an R-VPF round and the R-GPF seed fit solved side by side
//   k_fit_hybrid            the single-frame kernel: k_fit_brows' body for the big patches, k_fit_srows<64>'s
//                           for the small ones (four per workgroup), one launch, one workgroup per CU
//   k_fit_stream            whatever exceeds the plan: workgroup per patch, 128-bit lane sums
//   k_fit_fixup             (run by the host for marked frames only) the patches whose first fit set is empty, in
//                           the reference's order: they start from the plane the object fitted last
//
// The bin-ordered records are planes (pwpp_dev.h): the lowest-point pass streams z alone (4 B per
// point), every other pass z and {x, y} (12 B); the cloud-index plane is k_emit's (and the tiny-fit gather's), not the passes'.
// The split of a patch is a bit per slot in the MEMBERSHIP PLANE (pwpp_dev.h), left by every R-GPF round that can be the last, and
// a patch whose integer totals repeat from one round to the next stops there (exact early termination: k_fit_w64, phase B).
// A near-zone bin is stored in two parts, below and above a split height: every pass skips the high part
// when it can prove that none of its points can enter (stage_needs_hi).
// DESIGN.md section 3 has the measurements that led here (and the variants that were dropped: points
// parked in LDS, the chain cut into phase kernels).
//
// All reductions are integer, so every variant produces bit-identical
// planes and the same index sets whatever the lane count.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pwpp_common.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kPPT = 8;  // points per lane held in registers
#ifndef PWPP_W64_OCC
#define PWPP_W64_OCC 3  // waves per SIMD the 64-lane fit kernel is compiled for
#endif
#ifndef PWPP_W16_OCC
#define PWPP_W16_OCC 4  // waves per SIMD the 16-lane fit kernel is compiled for (3: 2 spilled registers instead of 77, 18 % slower alone -- profiles/r04_experiments.txt)
#endif
#ifndef PWPP_W16_WIDE_OCC
#define PWPP_W16_WIDE_OCC 3  // ... on the wide grid (contract v4): sixteen totals per patch in LDS (12.4 KB per wave) allow three waves per SIMD anyway
#endif
#ifndef PWPP_FIT_PREFETCH
#define PWPP_FIT_PREFETCH 0
#endif

// ------------------------------------------------------------------------------------------
// row = G consecutive lanes of a wave working on one patch
// ------------------------------------------------------------------------------------------
// Cross-lane primitives.  The four steps inside a 16-lane DPP row are register-to-register
// (v_*_dpp quad_perm / row_half_mirror / row_mirror, ~8 cycles each); 16 <-> 16 goes through
// ds_swizzle(SWAP,16) and the two 32-lane halves are combined with v_readlane + scalar ALU.
// A 64-lane all-reduce is ~8 instructions instead of 6 dependent ds_bpermute round trips.
#define PWPP_DPP(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xF, 0xF, true)
#define PWPP_DPP_XOR1 0xB1   // quad_perm [1,0,3,2]
#define PWPP_DPP_XOR2 0x4E   // quad_perm [2,3,0,1]
#define PWPP_DPP_HMIR 0x141  // row_half_mirror
#define PWPP_DPP_MIR 0x140   // row_mirror
#define PWPP_SWZ16 0x401F    // ds_swizzle BITMASK_PERM xor 16

template <int G>
struct Row {
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "row width");
    static constexpr unsigned long long kMask = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
    __device__ static __forceinline__ int first_lane() { return lane_id() & ~(G - 1); }

    __device__ static __forceinline__ unsigned min_u32(unsigned v) {
        int x = (int)v, t;
        t = PWPP_DPP(x, PWPP_DPP_XOR1); x = (unsigned)t < (unsigned)x ? t : x;
        t = PWPP_DPP(x, PWPP_DPP_XOR2); x = (unsigned)t < (unsigned)x ? t : x;
        t = PWPP_DPP(x, PWPP_DPP_HMIR); x = (unsigned)t < (unsigned)x ? t : x;
        if (G >= 16) {
            t = PWPP_DPP(x, PWPP_DPP_MIR);
            x = (unsigned)t < (unsigned)x ? t : x;
        }
        if (G >= 32) {
            t = __builtin_amdgcn_ds_swizzle(x, PWPP_SWZ16);
            x = (unsigned)t < (unsigned)x ? t : x;
        }
        if (G == 64) {
            const unsigned a = (unsigned)__builtin_amdgcn_readlane(x, 0), b = (unsigned)__builtin_amdgcn_readlane(x, 32);
            x = (int)(a < b ? a : b);
        }
        return (unsigned)x;
    }
    __device__ static __forceinline__ int sum_i32(int x) {
        x += PWPP_DPP(x, PWPP_DPP_XOR1);
        x += PWPP_DPP(x, PWPP_DPP_XOR2);
        x += PWPP_DPP(x, PWPP_DPP_HMIR);
        if (G >= 16) x += PWPP_DPP(x, PWPP_DPP_MIR);
        if (G >= 32) x += __builtin_amdgcn_ds_swizzle(x, PWPP_SWZ16);
        if (G == 64) x = __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 32);
        return x;
    }
    __device__ static __forceinline__ long long step64(long long v, int ctrl_kind) {
        int lo = (int)(unsigned)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);
        int tl, th;
        switch (ctrl_kind) {
            case 0: tl = PWPP_DPP(lo, PWPP_DPP_XOR1); th = PWPP_DPP(hi, PWPP_DPP_XOR1); break;
            case 1: tl = PWPP_DPP(lo, PWPP_DPP_XOR2); th = PWPP_DPP(hi, PWPP_DPP_XOR2); break;
            case 2: tl = PWPP_DPP(lo, PWPP_DPP_HMIR); th = PWPP_DPP(hi, PWPP_DPP_HMIR); break;
            case 3: tl = PWPP_DPP(lo, PWPP_DPP_MIR); th = PWPP_DPP(hi, PWPP_DPP_MIR); break;
            default: tl = __builtin_amdgcn_ds_swizzle(lo, PWPP_SWZ16); th = __builtin_amdgcn_ds_swizzle(hi, PWPP_SWZ16); break;
        }
        // the two moved halves form the register pair of ONE 64-bit add (v_lshl_add_u64); written with
        // shifts and ORs the compiler splits it into two 64-bit adds plus a move
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i pair = {tl, th};
        return v + __builtin_bit_cast(long long, pair);
    }
    // ---- sixteen 64-bit row sums at once, as a reduce-SCATTER ---------------------------------------
    // sum_i64 is a butterfly per value: every lane ends up with every total (16 x log2(G) exchanges).
    // The kernels that park the totals in LDS only need each total ONCE, so the row is halved instead:
    // at every step a lane keeps one half of its values, sends the other half to its partner and adds
    // what the partner sends (8 + 4 + 2 + 1 exchanges for the first four steps), after which every lane
    // of a 16-lane group owns one value summed over the group; the groups of a 64-lane row are then
    // added up on that one value.  The partners must agree on every side bit already used, hence the
    // order: half-mirror (i <-> 7-i, side = bit 2), xor 1, xor 2, xor 8 (ds_swizzle), [xor 16, xor 32].
    // `slot` = which of the sixteen values this lane owns; integer sums, so the order is free.
    // The sixteen values of a patch: n, S1[3], the lower and the upper 32-bit halves of S2[6] -- the halves
    // because a second moment (|Q| <= 2^26: up to 2^52 per point) leaves int64 once 2^11 points are added up.
    template <int KIND>
    __device__ static __forceinline__ long long xchg64(long long v) {
        int lo = (int)(unsigned)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);
        int tl, th;
        if (KIND == 0) { tl = PWPP_DPP(lo, PWPP_DPP_HMIR); th = PWPP_DPP(hi, PWPP_DPP_HMIR); }
        else if (KIND == 1) { tl = PWPP_DPP(lo, PWPP_DPP_XOR1); th = PWPP_DPP(hi, PWPP_DPP_XOR1); }
        else if (KIND == 2) { tl = PWPP_DPP(lo, PWPP_DPP_XOR2); th = PWPP_DPP(hi, PWPP_DPP_XOR2); }
        else if (KIND == 3) { tl = __builtin_amdgcn_ds_swizzle(lo, 0x201F); th = __builtin_amdgcn_ds_swizzle(hi, 0x201F); }  // xor 8
        else if (KIND == 4) { tl = __builtin_amdgcn_ds_swizzle(lo, PWPP_SWZ16); th = __builtin_amdgcn_ds_swizzle(hi, PWPP_SWZ16); }
        else { const int a = (lane_id() ^ 32) << 2; tl = __builtin_amdgcn_ds_bpermute(a, lo); th = __builtin_amdgcn_ds_bpermute(a, hi); }
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i pair = {tl, th};
        return __builtin_bit_cast(long long, pair);
    }
    __device__ static __forceinline__ long long reduce16_scatter(const long long (&v)[16], int &slot) {
        static_assert(G == 64 || G == 16, "rows of 64 or 16 lanes");  // (16: the first four steps are the whole reduction)
        const int ln = lane_id();
        const bool s2 = (ln & 4) != 0, s0 = (ln & 1) != 0, s1 = (ln & 2) != 0, s3 = (ln & 8) != 0;
        long long a[8], b[4], c[2], d;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long keep = s2 ? v[8 + i] : v[i], send = s2 ? v[i] : v[8 + i];
            a[i] = keep + xchg64<0>(send);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long keep = s0 ? a[4 + i] : a[i], send = s0 ? a[i] : a[4 + i];
            b[i] = keep + xchg64<1>(send);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long keep = s1 ? b[2 + i] : b[i], send = s1 ? b[i] : b[2 + i];
            c[i] = keep + xchg64<2>(send);
        }
        {
            const long long keep = s3 ? c[1] : c[0], send = s3 ? c[0] : c[1];
            d = keep + xchg64<3>(send);
        }
        if (G == 64) {
            d += xchg64<4>(d);
            d += xchg64<5>(d);
        }
        slot = (s2 ? 8 : 0) + (s0 ? 4 : 0) + (s1 ? 2 : 0) + (s3 ? 1 : 0);
        return d;
    }
    __device__ static __forceinline__ long long sum_i64(long long v) {
        v = step64(v, 0);
        v = step64(v, 1);
        v = step64(v, 2);
        if (G >= 16) v = step64(v, 3);
        if (G >= 32) v = step64(v, 4);
        if (G == 64) {
            const int lo = (int)(unsigned)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);
            const unsigned long long a = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, 0) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane(lo, 0);
            const unsigned long long b = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, 32) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane(lo, 32);
            v = (long long)(a + b);
        }
        return v;
    }
    __device__ static __forceinline__ unsigned long long ballot(bool p) {
        return (__ballot(p) >> first_lane()) & kMask;
    }
    // exclusive prefix sum over the row; total = row sum
    __device__ static __forceinline__ unsigned excl_scan(unsigned v, unsigned &total) {
        if constexpr (G == 16 || G == 64) {
            // DPP scan: row_shr 1, 2, 4, 8 inside the 16-lane rows (lanes without a source add 0), then for a
            // 64-lane row the two broadcasts of a row's last lane into the rows behind it -- six adds and no LDS
            // round trips, where the shuffle version below waits for a ds_bpermute at every step
            int x = (int)v;
            x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
            x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
            x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
            x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
            if constexpr (G == 64) {
                x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
                x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
                total = (unsigned)__builtin_amdgcn_readlane(x, 63);
            } else {
                total = (unsigned)__shfl(x, G - 1, G);
            }
            return (unsigned)x - v;
        }
        const int j = lane_id() & (G - 1);
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o, G);
            if (j >= o) incl += t;
        }
        total = (unsigned)__shfl((int)incl, G - 1, G);
        return incl - v;
    }
};

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return ~Row<64>::min_u32(~v); }
// hand-over through LDS between the lanes of ONE wave (no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// compare-exchange of two keys, ascending
__device__ __forceinline__ void ce(unsigned &a, unsigned &b) {
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

// ------------------------------------------------------------------------------------------
// The fit chain of one patch as a small state machine, so that the (large) eigen-solve and
// the LPR selection are instantiated once per kernel and rows of one wave may be at different
// points of the chain:
//   ST_VPF   R-VPF round (zone 0 only): seeds z < lpr + th_seeds_v -> plane -> strip (ref :484-507)
//   ST_SEED  R-GPF seeds z < lpr + th_seeds -> plane E0                           (ref :513-514)
//   ST_LAZY  the R-VPF plane of a zone 1-3 bin, needed only if ST_SEED found no seed: ref :49
//            then leaves that plane in force.  (The reference always fits it, :486-487, and the
//            next fit overwrites it; skipping it changes nothing observable.)
//   ST_ITER  R-GPF round: keep dist < th_dist -> plane; from the second round on the set also goes to the membership plane,
//            and the final plane is fitted on the ground set                       (ref :516-543)
// ------------------------------------------------------------------------------------------
enum { ST_VPF = 0, ST_SEED = 1, ST_ITER = 2, ST_LAZY = 3, ST_DONE = 4 };

// hi_skipped: the pass that left the split did not read the high part (its points are non-ground, its bits were not written)
// member_lg: log2(G) of the rows that left the split in the membership plane (pwpp_dev.h)
__device__ __forceinline__ void write_record(PwppPatchRec *rec, const PlaneFit &pl, unsigned n, unsigned n_ground, bool hi_skipped, int member_lg, int rounds) {
    rec->mean[0] = pl.mean[0];
    rec->mean[1] = pl.mean[1];
    rec->mean[2] = pl.mean[2];
    rec->normal[0] = pl.nx;
    rec->normal[1] = pl.ny;
    rec->normal[2] = pl.nz;
    rec->sv[0] = pl.sv[0];
    rec->sv[1] = pl.sv[1];
    rec->sv[2] = pl.sv[2];
    rec->d = pl.d;
    rec->n_points = (int)n;
    rec->n_ground = (int)n_ground;
    rec->n_nonground = (int)(n - n_ground);
    rec->decision = 0;
    rec->valid = (hi_skipped ? 3 : 1) | (member_lg << 3) | ((rounds & 0xff) << 8);
}

// The reference object's plane members survive from patch to patch (and frame to frame): a patch whose FIRST fit set is
// empty consults whatever plane was fitted last (ref :49 -- the near-zone verticality test :489 or the first R-GPF round
// :525).  That is a serial dependence on the patch before, so the parallel kernels only RECOGNISE the case (it needs a
// lowest height of -inf, or one so large that th_seeds is absorbed, or num_lpr = 0: never seen in a real scan) and
// leave the patch untouched; K5 / K6 then leave the frame alone and the host runs k_fit_fixup for it (pwpp_capi.cpp).
//   no plane fitted yet and the stage's set is empty: an R-VPF round (its plane's z decides :489), the R-VPF fit of a
//   far-zone bin that was only evaluated because the R-GPF seeds were empty too, or the R-GPF seeds with no R-VPF
//   fit to fall back on -- the next thing would be a distance test against the stale plane.
__device__ __forceinline__ bool needs_previous_plane(const PwppDevParams &P, int kind, int zone, bool fitted) {
    return !fitted && (kind == ST_VPF || kind == ST_LAZY || (kind == ST_SEED && !(P.enable_RVPF != 0 && zone != 0)));
}
__device__ __forceinline__ void mark_needs_previous_plane(const PwppBatch &Bt, int f, PwppPatchRec *rec, unsigned n) {
    rec->n_points = (int)n;
    rec->n_ground = 0;
    rec->n_nonground = (int)n;
    rec->decision = 0;
    rec->valid = 4;
    atomicOr((unsigned *)&Bt.results[f].overflow, 2u);
}

// A patch in the part-ordered planes (pwpp_dev.h): the FRAME's planes z, {x, y}, cloud index (the frame is a grid
// dimension, so these are scalar registers and a load is base + 32-bit lane offset) and the slots of the bin's two
// parts in them -- the points below the split height zs and the others (z >= zs, or NaN).
struct PatchRef {
    float *z;
    const float2 *xy;
    const int *idx;
    unsigned off_lo, n_lo, off_hi, n_hi;
    int bin;  // (names the bits of its two parts in the membership plane: member_offset)
    unsigned arena_base;  // ... together with where the frame's overflow arena begins and the number of parts (pwpp_member_offset)
    int num_parts;
};
// Chunks are numbered through the low part and then through the high part; a pass that skips the high part
// (stage_needs_hi) simply finds no points in the chunks above the low part's.
struct PartSel {
    unsigned off, n, c;  // first slot and points of the part the chunk lies in, chunk index inside that part
    unsigned moff;       // first byte of the part's bits in the frame's share of the membership plane (only the passes that write them use it)
};
// membership plane (pwpp_dev.h, PWPP_SLOT_ALIGN): where the bits of a part begin, relative to the frame's first byte
__device__ __forceinline__ unsigned member_offset(const PatchRef &p, unsigned off, int part) { return pwpp_member_offset(off, part, p.arena_base, p.num_parts); }
// slot of the i-th point of the patch, low part first (the kernels that walk a patch point by point)
__device__ __forceinline__ unsigned patch_slot(const PatchRef &p, unsigned i) { return i < p.n_lo ? p.off_lo + i : p.off_hi + (i - p.n_lo); }
template <int G>
__device__ __forceinline__ unsigned part_chunks(unsigned n) { return (n + 8u * G - 1u) / (8u * G); }
template <int G>
__device__ __forceinline__ unsigned patch_chunks(const PatchRef &p, bool use_hi) {
    return part_chunks<G>(p.n_lo) + (use_hi ? part_chunks<G>(p.n_hi) : 0u);
}
template <int G>
__device__ __forceinline__ PartSel chunk_sel(const PatchRef &p, unsigned c, bool use_hi, bool on = true) {
    const unsigned nc_lo = part_chunks<G>(p.n_lo);
    const bool h = c >= nc_lo;
    PartSel s;
    s.off = h ? p.off_hi : p.off_lo;
    s.n = !on ? 0u : (h ? (use_hi ? p.n_hi : 0u) : p.n_lo);
    s.c = h ? c - nc_lo : c;
    s.moff = member_offset(p, s.off, h ? PWPP_PART_HI(p.bin) : PWPP_PART_LO(p.bin));
    return s;
}
// The membership bits of one chunk of a row of G lanes: lane j's byte = its eight points (bit k = point k of the chunk, set = the
// point is in the round's ground set; slots beyond the part's end are 0).  `wb` is row-uniform; a chunk beyond the part's last
// one (the wave runs to its longest row's count) stores nothing, so a part's bits end inside its own share of the plane.
template <int G>
__device__ __forceinline__ void store_member(uint8_t *frame_member, const PartSel &sel, unsigned gmask, bool wb) {
    const unsigned j = (unsigned)lane_id() & (G - 1);
    if (wb && sel.c * (8u * G) < sel.n) frame_member[sel.moff + sel.c * G + j] = (uint8_t)gmask;
}
// R-VPF removes a point from the patch's working set (ref :495-503) by overwriting its z with a NaN whose
// payload is the R-VPF round (1-based): the coordinates of a removed point are not needed again, and k_czm_*
// store a NaN z of the cloud as the payload-free 0x7fc00000, so the mark is unambiguous.  The round matters
// to the reference-order output mode: the reference appends the points a round removes to
// regionwise_nonground_ round by round (ref :500).
__device__ __forceinline__ void strip_point(const PatchRef &pr, unsigned slot, int round) {
    pr.z[slot] = __uint_as_float(0x7fc00000u | (unsigned)((round + 1) & 0xff));
}
__device__ __forceinline__ bool z_stripped(float z) { return (int)__float_as_uint(z) > 0x7fc00000; }
// One chunk = 8 points per lane of a row of G lanes.  Which lane sees which point is free (the sums are exact
// integers, the lists are written in scatter order anyway), so the mapping follows the loads:
//   G = 64 (big bins: every lane busy)  slot k of lane j = point c * 512 + (k / 4) * 256 + 4 j + (k % 4): a lane
//          fetches FOUR CONSECUTIVE POINTS per load (a patch starts at a multiple of four slots, k_czm_scan /
//          cap_off) -- two 16-byte loads for z, four for {x, y}, two for the cloud indices instead of 8 + 8 + 8;
//   G < 64 (small patches)              slot k of lane j = point c * 8G + k * G + j: a patch of n points fills
//          ceil(n / G) slots of every lane, and the slots above are skipped wave-wide (with four consecutive
//          points per lane a 20-point patch would keep 5 lanes busy for 4 slots: k_fit_w64<16,64> 0.84 -> 1.19 ms).
// The loads are unconditional (the first points of the patch stand in beyond its end -- NOT whatever follows in
// memory: a one-pass segment is mostly unwritten space, and fetching it cost k_fit_w64<16,64> 9 %): the compiler can
// keep a whole chunk in flight behind the arithmetic of the previous one and wait with a counted s_waitcnt.
// Which slots of a lane's chunk lie inside the part: slot k holds point first + k_off(k), so with
//   rem = points of the part - index of the lane's first point          (signed; <= 0: nothing of this lane's)
// slot k is valid iff k_off(k) < rem -- ONE compare with a constant per slot where it is needed at all: a chunk that is
// valid in every lane of the wave needs none, but a second code path for it costs more registers than the compare).
struct ChunkZ {  // what the lowest-point pass needs
    float z[kPPT];
    int rem;
};
struct ChunkPts {
    float x[kPPT], y[kPPT], z[kPPT];
    int rem;
};
template <int G>
__device__ __forceinline__ constexpr int k_off(int k) {
    return G == 64 ? (k >> 2) * 256 + (k & 3) : k * G;
}
template <int G>
__device__ __forceinline__ unsigned chunk_point(unsigned c, int k, unsigned j) {
    if constexpr (G == 64) return c * 512u + (unsigned)(k >> 2) * 256u + 4u * j + (unsigned)(k & 3);
    return c * (8u * G) + (unsigned)k * G + j;
}
template <int G>
__device__ __forceinline__ int chunk_rem(unsigned n, unsigned c, unsigned j) {
    return (int)n - (int)(G == 64 ? c * 512u + 4u * j : c * (8u * G) + j);
}
template <int G>
__device__ __forceinline__ bool chunk_interior(int rem) { return k_off<G>(kPPT - 1) < rem; }
template <int G>
__device__ __forceinline__ unsigned chunk_valid_bits(int rem) {
    unsigned valid = 0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k)
        if (k_off<G>(k) < rem) valid |= 1u << k;
    return valid;
}
// The lowest-point pass keeps the INTERLEAVED mapping (slot k of lane j = point c * 8G + k * G + j) for every row
// width: neighbouring slots of a bin are neighbouring points of a scan line with nearly the same z, and a lane that
// holds more than four of the patch's lowest points sends the selection to its slow exact path (with four
// consecutive points per lane the four-waves-per-patch kernel took it for the largest patch of KITTI frame 0:
// 6.6 -> 40 us).  Which lane sees which point is free per pass.
template <int G>
__device__ __forceinline__ void load_chunk_z(ChunkZ &cp, const PatchRef &pr, const PartSel &sel) {
    const unsigned j = (unsigned)lane_id() & (G - 1);
    cp.rem = (int)sel.n - (int)(sel.c * (8u * G) + j);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const unsigned i = sel.c * (8u * G) + (unsigned)k * G + j;
        cp.z[k] = pr.z[sel.off + (i < sel.n ? i : 0u)];
    }
}
// the slots of a ChunkZ that are inside the part and still in the patch's working set (not removed by R-VPF)
template <int G>
__device__ __forceinline__ unsigned chunk_act_z(const ChunkZ &cp) {
    unsigned act = 0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k)
        if (k * G < cp.rem && !z_stripped(cp.z[k])) act |= 1u << k;
    return act;
}
template <int G>
__device__ __forceinline__ void load_chunk(ChunkPts &cp, const PatchRef &pr, const PartSel &sel) {
    const unsigned j = (unsigned)lane_id() & (G - 1);
    cp.rem = chunk_rem<G>(sel.n, sel.c, j);
    if constexpr (G == 64) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned p1 = chunk_point<G>(sel.c, 4 * q, j), p0 = sel.off + (p1 < sel.n ? p1 : 0u);
            const float4 v = *reinterpret_cast<const float4 *>(pr.z + p0);
            const float4 a = *reinterpret_cast<const float4 *>(pr.xy + p0), b = *reinterpret_cast<const float4 *>(pr.xy + p0 + 2);
            cp.z[4 * q] = v.x;
            cp.z[4 * q + 1] = v.y;
            cp.z[4 * q + 2] = v.z;
            cp.z[4 * q + 3] = v.w;
            cp.x[4 * q] = a.x;
            cp.y[4 * q] = a.y;
            cp.x[4 * q + 1] = a.z;
            cp.y[4 * q + 1] = a.w;
            cp.x[4 * q + 2] = b.x;
            cp.y[4 * q + 2] = b.y;
            cp.x[4 * q + 3] = b.z;
            cp.y[4 * q + 3] = b.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const unsigned i1 = chunk_point<G>(sel.c, k, j), i = sel.off + (i1 < sel.n ? i1 : 0u);
            const float2 v = pr.xy[i];
            cp.z[k] = pr.z[i];
            cp.x[k] = v.x;
            cp.y[k] = v.y;
        }
    }
}
// the slot of point k of this lane (what strip_point marks)
template <int G>
__device__ __forceinline__ unsigned chunk_slot(const PartSel &sel, int k, unsigned j) { return sel.off + chunk_point<G>(sel.c, k, j); }
// per-lane part of one stage: adds the points of one chunk that enter this stage's fit to `m` (ground
// set of an R-GPF round, seeds of a seed stage) and returns their mask.
//   One test for every stage: a seed pass (ref :108,145, "z < lpr + th_seeds") is the plane test of ref :525 with normal
//   (0,0,1), d = 0: 0*x + 0*y + 1*z + 0.0 == z for the finite x, y that binning lets through, so the per-point code has
//   no branch on the stage.  The test itself is  s < T  in float, T = plane_test_threshold(d, thr) of the pass
//   (pwpp_common.hpp): bit for bit the reference's  double(s) + d < thr.  A point R-VPF removed has a NaN for its z
//   (strip_point) and fails like any NaN; a slot beyond the part's end holds a stand-in record and is masked by `rem`.
// `clamp_hit` (or-ed into): some point that entered the sums lay outside z0 +- ZR, i.e. its quantised height was clamped (flag_clamped
// below says what that means; found here, where the clamped height is formed anyway -- one compare per included point instead of four
// per slot in a loop of its own).
template <int G, class M>
__device__ __forceinline__ unsigned lane_stage_accum(const ChunkPts &cp, int kind, float T, float nx, float ny, float nz,
                                                     double scale, const FxpOrg &org, M &m, bool &clamp_hit) {
    const bool iter = kind == ST_ITER;
    const float tx = iter ? nx : 0.0f, ty = iter ? ny : 0.0f, tz = iter ? nz : 1.0f;
    unsigned gmask = 0;
    // (ONE body: a second copy without the validity compare for chunks that are valid in every lane doubled the unrolled code
    // and cost the kernels 40-110 spilled registers)
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        if ((int)(plane_s(tx, ty, tz, cp.x[k], cp.y[k], cp.z[k]) < T) & (int)(k_off<G>(k) < cp.rem)) {  // (a branch on purpose: a seed pass includes about half of the points, an R-GPF round ~60 %)
            gmask |= 1u << k;
#ifndef PWPP_ABLATE_NO_ACCUM  // (timing experiments only: the pass without its sums)
            const float zc = __builtin_amdgcn_fmed3f(cp.z[k], org.zlo, org.zhi);
            clamp_hit = clamp_hit || zc != cp.z[k];
            m.add_uncounted(cp.x[k], cp.y[k], zc, scale, org);
#endif
        }
    }
    m.n += __popc(gmask);
    return gmask;
}
// the pass's threshold for lane_stage_accum: the R-GPF rounds test the distance to the plane (d) against th_dist, the seed
// stages the height against lpr + th_seeds(_v)
__device__ __forceinline__ float stage_threshold(int kind, double d, double th_dist, double thr_seed) {
    return kind == ST_ITER ? plane_test_threshold(d, th_dist) : plane_test_threshold(0.0, thr_seed);
}

// The points of this lane that the R-VPF plane removes (ref :495-503): |double(s) + d| < th_dist_v, i.e.
// t_lo < s < t_hi with t_hi = plane_test_threshold(d, th_dist_v) and t_lo = -plane_test_threshold(-d, th_dist_v)
// (negation is exact and rounding symmetric).  A point removed before is a NaN and is not hit again.
struct StripBand {
    float lo, hi;
};
__device__ __forceinline__ StripBand strip_band(double d, double th_dist_v) {
    StripBand b;
    b.hi = plane_test_threshold(d, th_dist_v);
    b.lo = -plane_test_threshold(-d, th_dist_v);
    return b;
}
template <int G>
__device__ __forceinline__ unsigned lane_strip(const ChunkPts &cp, bool on, float nx, float ny, float nz, const StripBand &b) {
    unsigned hit = 0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const float sv = plane_s(nx, ny, nz, cp.x[k], cp.y[k], cp.z[k]);
        if (on & (sv < b.hi) & (sv > b.lo) & (k_off<G>(k) < cp.rem)) hit |= 1u << k;  // ref :499
    }
    return hit;
}

// Did a height of the FINAL ground set of a patch lie outside z0 +- ZR (its quantised value was clamped, Moments)?  Noticed where the
// clamped height is formed (lane_stage_accum's clamp_hit) in the passes that leave a set in the membership plane, and counted only if that
// set is final; the frame is flagged (PwppFrameResult.overflow bit 2, pwpp_get_clamped_frames): the plane of such a patch -- more than 32 m
// tall with the default CZM -- is the plane of the clamped heights (include/pwpp.h).
__device__ __forceinline__ void flag_clamped(const PwppBatch &Bt, int f) { atomicOr((unsigned *)&Bt.results[f].overflow, 4u); }

// Contract v3 (pwpp_common.hpp, mean_cov_tiny): a fit set of one, two or three points follows the reference's own float
// arithmetic, which needs the POINTS, not their moments.  The pass that found so few members is repeated here for the
// row's patch with the same test on the same data (both parts: skipping the high part is only ever an optimisation),
// every member is handed to all lanes of the row, which keep them sorted by (z, cloud index) -- the order of the
// reference's z-sorted bin (ref :199; equal heights, which std::sort leaves to libstdc++, in cloud order) -- and
// evaluate mean and covariance side by side.  Rare (num_min_pts < 4, or a patch whose seeds / ground set dwindle to
// a few points), so nothing here is tuned; `on` is row-uniform, the loops are wave-uniform.
template <int G>
__device__ void tiny_fit_row(const PatchRef &pts, bool on, int kind, float T, float nx, float ny, float nz, float mean[3], float c6[6]) {
    const unsigned j = (unsigned)lane_id() & (G - 1);
    const bool iter = kind == ST_ITER;
    const float tx = iter ? nx : 0.0f, ty = iter ? ny : 0.0f, tz = iter ? nz : 1.0f;  // (the one test of lane_stage_accum)
    unsigned long long key[3] = {~0ull, ~0ull, ~0ull};
    float qx[3] = {0.0f, 0.0f, 0.0f}, qy[3] = {0.0f, 0.0f, 0.0f}, qz[3] = {0.0f, 0.0f, 0.0f};
    int cnt = 0;
    const unsigned n = on ? pts.n_lo + pts.n_hi : 0u;
    const unsigned nmax = wave_max_u32(n);
    for (unsigned i0 = 0; i0 < nmax; i0 += G) {  // one point per lane and step: few registers, so that the kernels around it keep theirs
        const unsigned i = i0 + j;
        const bool in = i < n;
        const unsigned sl = in ? patch_slot(pts, i) : pts.off_lo;
        const float pz = pts.z[sl];
        const float2 pxy = pts.xy[sl];
        const int pidx = pts.idx[sl];
        const bool inc = in && (plane_s(tx, ty, tz, pxy.x, pxy.y, pz) < T);  // (a point R-VPF removed is a NaN: it fails)
        unsigned long long mm = Row<G>::ballot(inc);
        while (__any(mm != 0ull)) {  // (wave-uniform: the shuffles below need every lane)
            const int src = Row<G>::first_lane() + (mm ? __ffsll((long long)mm) - 1 : (int)j);
            float ex = __shfl(pxy.x, src, 64), ey = __shfl(pxy.y, src, 64), ez = __shfl(pz, src, 64);
            const int ei = __shfl(pidx, src, 64);
            if (mm) {
                unsigned long long nk = ((unsigned long long)z_key(ez == 0.0f ? 0.0f : ez) << 32) | (unsigned)ei;  // (-0 and +0 compare equal: cloud order)
#pragma unroll
                for (int q = 0; q < 3; ++q) {  // sorted insertion; a fourth member (there is none) would fall off the end
                    const bool sw = nk < key[q];
                    const unsigned long long tk = key[q];
                    const float t0 = qx[q], t1 = qy[q], t2 = qz[q];
                    key[q] = sw ? nk : tk;
                    qx[q] = sw ? ex : t0;
                    qy[q] = sw ? ey : t1;
                    qz[q] = sw ? ez : t2;
                    nk = sw ? tk : nk;
                    ex = sw ? t0 : ex;
                    ey = sw ? t1 : ey;
                    ez = sw ? t2 : ez;
                }
                ++cnt;
                mm &= mm - 1ull;
            }
        }
    }
    mean_cov_tiny(cnt < 3 ? cnt : 3, qx, qy, qz, mean, c6);
}

// the sixteen values Row<64>::reduce16_scatter adds up for a patch: n, S1[3], lower and upper halves of S2[6]
// (a lane's second moment: up to 2^63 on the narrow grid, 2^83 on the wide one -- the upper half fits int64 either way, and stays
// there when 256 lanes are added up)
template <class M>
__device__ __forceinline__ void moments_to_16(const M &mm, long long (&v)[16]) {
    v[0] = mm.n;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[1 + k] = mm.first(k);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const __int128 t = mm.second(k);
        v[4 + k] = (long long)((unsigned long long)t & 0xffffffffull);
        v[10 + k] = (long long)(t >> 32);
    }
}
__device__ __forceinline__ __int128 join_halves(long long lo, long long hi) { return ((__int128)hi << 32) + (__int128)lo; }


// The 64 keys a 16-lane row kept (four per lane, ascending in every lane) in ascending order over element 4 * lane + register: a
// bitonic merge network, lane pairs -> quads -> eights -> the row.  A merge of two ascending halves is one compare-exchange with the
// MIRRORED element (i <-> 2m - 1 - i: the partner lane's registers in reverse) followed by half-cleaners at distances m / 2 ... 1.
// Lane distances 1, 2 (quad_perm), 4 (row_shl / row_shr under bank masks), 8 (row_ror:8), mirrors of 4, 8, 16 lanes (quad_perm [3,2,1,0],
// row_half_mirror, row_mirror) are DPP moves inside the row -- 13 cross-lane steps of four registers and 8 in-lane ones, ~250 dependent-
// chain-short VALU instructions where the extraction of the 20 lowest took 20 rounds of ~30 with a row reduction each (round 6).
#define PWPP_DPPU(x, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ unsigned dpp_lane_xor4(unsigned x) {
    int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104 /* row_shl:4 */, 0xF, 0x5, false);  // lanes 0-3, 8-11 <- lane + 4
    t = __builtin_amdgcn_update_dpp(t, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xA, false);           // lanes 4-7, 12-15 <- lane - 4
    return (unsigned)t;
}
__device__ __forceinline__ unsigned cex_keep(unsigned x, unsigned p, bool keep_min) { return keep_min ? (p < x ? p : x) : (p > x ? p : x); }
__device__ __forceinline__ void row16_sort64(unsigned &k0, unsigned &k1, unsigned &k2, unsigned &k3) {
    const int l = lane_id() & 15;
    auto in_lane = [&]() {  // distances 2 and 1 inside the lane's four elements
        ce(k0, k2);
        ce(k1, k3);
        ce(k0, k1);
        ce(k2, k3);
    };
#define PWPP_MIRROR_STEP(ctrl, keep)                                                                          \
    {                                                                                                         \
        const unsigned p0 = PWPP_DPPU(k3, ctrl), p1 = PWPP_DPPU(k2, ctrl), p2 = PWPP_DPPU(k1, ctrl), p3 = PWPP_DPPU(k0, ctrl); \
        k0 = cex_keep(k0, p0, keep);                                                                          \
        k1 = cex_keep(k1, p1, keep);                                                                          \
        k2 = cex_keep(k2, p2, keep);                                                                          \
        k3 = cex_keep(k3, p3, keep);                                                                          \
    }
#define PWPP_XOR_STEP(MOVE, keep)                                                                             \
    {                                                                                                         \
        const unsigned p0 = MOVE(k0), p1 = MOVE(k1), p2 = MOVE(k2), p3 = MOVE(k3);                            \
        k0 = cex_keep(k0, p0, keep);                                                                          \
        k1 = cex_keep(k1, p1, keep);                                                                          \
        k2 = cex_keep(k2, p2, keep);                                                                          \
        k3 = cex_keep(k3, p3, keep);                                                                          \
    }
#define PWPP_MV_X1(x) PWPP_DPPU(x, PWPP_DPP_XOR1)
#define PWPP_MV_X2(x) PWPP_DPPU(x, PWPP_DPP_XOR2)
#define PWPP_MV_X4(x) dpp_lane_xor4(x)
#define PWPP_MV_X8(x) PWPP_DPPU(x, 0x128 /* row_ror:8 */)
    const bool b0 = (l & 1) == 0, b1 = (l & 2) == 0, b2 = (l & 4) == 0, b3 = (l & 8) == 0;
    // pairs of lanes (8 elements)
    PWPP_MIRROR_STEP(PWPP_DPP_XOR1, b0)
    in_lane();
    // quads (16)
    PWPP_MIRROR_STEP(0x1B /* quad_perm [3,2,1,0] */, b1)
    PWPP_XOR_STEP(PWPP_MV_X1, b0)
    in_lane();
    // eights (32)
    PWPP_MIRROR_STEP(PWPP_DPP_HMIR, b2)
    PWPP_XOR_STEP(PWPP_MV_X2, b1)
    PWPP_XOR_STEP(PWPP_MV_X1, b0)
    in_lane();
    // the row (64)
    PWPP_MIRROR_STEP(PWPP_DPP_MIR, b3)
    PWPP_XOR_STEP(PWPP_MV_X4, b2)
    PWPP_XOR_STEP(PWPP_MV_X2, b1)
    PWPP_XOR_STEP(PWPP_MV_X1, b0)
    in_lane();
    // (a 16-lane row that holds a BITONIC sequence of 64 keys -- 32 ascending in lanes 0-7, 32 descending in lanes 8-15 -- is sorted by
    // the half-cleaners alone: row16_clean64 below, used to merge the lowest keys of two rows of a 64-lane row)
#undef PWPP_MIRROR_STEP
#undef PWPP_XOR_STEP
#undef PWPP_MV_X1
#undef PWPP_MV_X2
#undef PWPP_MV_X4
#undef PWPP_MV_X8
}
__device__ __forceinline__ void row16_clean64(unsigned &k0, unsigned &k1, unsigned &k2, unsigned &k3) {
    const int l = lane_id() & 15;
    const bool b0 = (l & 1) == 0, b1 = (l & 2) == 0, b2 = (l & 4) == 0, b3 = (l & 8) == 0;
    auto xstep = [&](auto mover, bool keep) {
        const unsigned p0 = mover(k0), p1 = mover(k1), p2 = mover(k2), p3 = mover(k3);
        k0 = cex_keep(k0, p0, keep);
        k1 = cex_keep(k1, p1, keep);
        k2 = cex_keep(k2, p2, keep);
        k3 = cex_keep(k3, p3, keep);
    };
    xstep([](unsigned x) { return PWPP_DPPU(x, 0x128 /* row_ror:8 */); }, b3);
    xstep([](unsigned x) { return dpp_lane_xor4(x); }, b2);
    xstep([](unsigned x) { return PWPP_DPPU(x, PWPP_DPP_XOR2); }, b1);
    xstep([](unsigned x) { return PWPP_DPPU(x, PWPP_DPP_XOR1); }, b0);
    ce(k0, k2);
    ce(k1, k3);
    ce(k0, k1);
    ce(k2, k3);
}
// The 32 lowest of the 256 keys a 64-lane row kept, ascending over element 4 * lane + register of lanes 0-7: every 16-lane quarter sorts
// its 64 keys, then the quarters are merged two at a time -- lanes 8-15 of the receiving quarter take the 32 lowest of the other one in
// REVERSE (ds_bpermute), which makes the quarter's 64 keys a bitonic sequence whose lower half is the 32 lowest of both.
__device__ __forceinline__ void row64_lowest32(unsigned &k0, unsigned &k1, unsigned &k2, unsigned &k3) {
    row16_sort64(k0, k1, k2, k3);
    const int ln = lane_id(), l = ln & 15, q = ln >> 4;
    for (int round = 0; round < 2; ++round) {  // quarters 1 -> 0 and 3 -> 2, then 2 -> 0
        const int step = round == 0 ? 1 : 2;
        const bool recv = l >= 8 && (q & (2 * step - 1)) == 0;
        const int src = recv ? ((q + step) * 16 + (15 - l)) : ln;
        const unsigned p0 = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)k3), p1 = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)k2);
        const unsigned p2 = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)k1), p3 = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)k0);
        k0 = recv ? p0 : k0;
        k1 = recv ? p1 : k1;
        k2 = recv ? p2 : k2;
        k3 = recv ? p3 : k3;
        row16_clean64(k0, k1, k2, k3);
    }
}
// sum of a double over a 16-lane row (every lane gets it): exact whatever the order only where the caller has made sure of that
__device__ __forceinline__ double row16_sum_f64(double v) {
    auto step = [&](auto mover) {
        const long long b = __double_as_longlong(v);
        const int lo = (int)(unsigned)(unsigned long long)b, hi = (int)((unsigned long long)b >> 32);
        const unsigned long long o = ((unsigned long long)(unsigned)mover(hi) << 32) | (unsigned)mover(lo);
        v += __longlong_as_double((long long)o);
    };
    step([](int x) { return PWPP_DPP(x, PWPP_DPP_XOR1); });
    step([](int x) { return PWPP_DPP(x, PWPP_DPP_XOR2); });
    step([](int x) { return PWPP_DPP(x, PWPP_DPP_HMIR); });
    step([](int x) { return PWPP_DPP(x, PWPP_DPP_MIR); });
    return v;
}

// LPR (ref :84-103) of streamed rows, normally in ONE pass over the points: every lane keeps its
// four smallest eligible keys and the smallest key it had to drop.  The keff smallest of the
// 4G kept keys are extracted in ascending order; they are the keff smallest of the row unless
// some lane dropped a key below the largest one taken (T).  Only then (a lane held five or more
// of the row's lowest points: ~1e-3 of the rows for G = 64, a few % for G = 16) a second pass
// gathers every key below T (<= 8 per lane) and sums those, topped up with copies of T; if even
// that overflows, an exact but slow extraction by distinct values runs.
// The two parts of the bin: every z of the low part is below every z of the high part (and a NaN, the largest
// key, lives in the high part), so if the low part alone holds num_lpr eligible points the lowest num_lpr of
// the patch are all there and the high part is not read; otherwise the pass goes on through the high part.
template <int G>
__device__ double srow_lpr(const PatchRef &pts, bool need, bool use_cutoff, double cutoff,
                           int num_lpr, int force = 0 /* tests: 1 = take the second pass, 2 = and the exact extraction (PWPP_DEBUG_FLAGS 16384 / 32768) */,
                           const ChunkZ *first = nullptr /* the first chunk of the low part, already requested by the caller (load_chunk_z with need ? n_lo : 0) */) {
#ifdef PWPP_ABLATE_NO_LPR  // (timing experiments only: no lowest-point pass)
    return -1.75;
#endif
    const int j = lane_id() & (G - 1);
    const unsigned INF = 0xFFFFFFFFu;
    unsigned k0 = INF, k1 = INF, k2 = INF, k3 = INF, dropped = INF;
    int elig = 0;
    // (the pass is a pure latency chain -- eight loads, ~100 instructions -- so the next chunk's z values are requested
    // before this chunk's are ranked: eight registers, in a phase that is far from the kernels' register peak)
    auto rank_part = [&](unsigned off, unsigned n, unsigned nchunks, const ChunkZ *pre) {
        PartSel sel;
        sel.off = off;
        sel.n = n;
        sel.c = 0u;
        sel.moff = 0u;
        ChunkZ cp;
        if (pre) cp = *pre;
        else if (nchunks > 0u) load_chunk_z<G>(cp, pts, sel);
        for (unsigned c = 0; c < nchunks; ++c) {
            ChunkZ nx;
            sel.c = c + 1u;
            if (c + 1u < nchunks) load_chunk_z<G>(nx, pts, sel);
            const unsigned act = chunk_act_z<G>(cp);
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const bool e = (act >> k & 1u) && !(use_cutoff && (double)cp.z[k] < cutoff);
                unsigned x = e ? z_key(cp.z[k]) : INF;
                ce(k0, x);
                ce(k1, x);
                ce(k2, x);
                ce(k3, x);
                dropped = x < dropped ? x : dropped;
                elig += e ? 1 : 0;
            }
            if (c + 1u < nchunks) cp = nx;
        }
    };
    rank_part(pts.off_lo, need ? pts.n_lo : 0u, wave_max_u32(need ? part_chunks<G>(pts.n_lo) : 0u), first);
    int total = Row<G>::sum_i32(elig);
    const bool use_hi = need && pts.n_hi > 0u && total < num_lpr;  // row-uniform
    if (__any(use_hi)) {
        rank_part(pts.off_hi, use_hi ? pts.n_hi : 0u, wave_max_u32(use_hi ? part_chunks<G>(pts.n_hi) : 0u), nullptr);
        total = Row<G>::sum_i32(elig);
    }
    const int keff = total < num_lpr ? total : num_lpr;  // row-uniform
    double sum = 0.0;
    unsigned T = 0;
    bool quick = false;  // (row-uniform) the sorted-row shortcut below has delivered sum and T
    if constexpr (G == 16) {
        // Round 6: SORT the row's 64 kept keys (row16_sort64) instead of extracting the lowest one by one.  The keff lowest are then
        // elements 0 .. keff - 1, T is element keff - 1, and their sum is order-free -- hence equal to the reference's ascending sum,
        // ref :99-101 -- whenever every one of them is 0 or has 2^-12 <= |z| < 2^8: multiples of 2^-35 whose partial sums stay below
        // 2^14, exact in a double in any order.  Anything else among the lowest (a denormal, a height of a kilometre, inf, the INF of
        // a row that kept fewer than keff keys) takes the extraction loop below, which works on the sorted registers just as well.
        if (__any(need && keff > 0 && keff <= 64)) {
            row16_sort64(k0, k1, k2, k3);
            const unsigned kk[4] = {k0, k1, k2, k3};
            double part = 0.0;
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool sel = 4 * j + r < keff;
                const float zv = key_z(kk[r]);
                const unsigned e = (__float_as_uint(zv) >> 23) & 0xffu;
                ok = ok && (!sel || zv == 0.0f || (e >= 115u && e < 135u));
                part += sel ? (double)zv : 0.0;
            }
            quick = need && keff > 0 && keff <= 64 && Row<G>::ballot(!ok) == 0ull;
            const double s = row16_sum_f64(part);
            const int last = keff > 0 ? keff - 1 : 0;
            const unsigned pick = (last & 3) == 0 ? k0 : ((last & 3) == 1 ? k1 : ((last & 3) == 2 ? k2 : k3));
            const unsigned t = (unsigned)__shfl((int)pick, last >> 2, 16);
            if (quick) {
                sum = s;
                T = t;
            }
        }
    }
    if constexpr (G == 64) {
        // the same for a 64-lane row (one patch per wave here: everything below is wave-uniform): the 32 lowest of its 256 kept keys,
        // sorted, in lanes 0-7 (row64_lowest32); the registers are put back if the shortcut cannot be taken
        if (need && keff > 0 && keff <= 32) {
            const unsigned o0 = k0, o1 = k1, o2 = k2, o3 = k3;
            row64_lowest32(k0, k1, k2, k3);
            const unsigned kk[4] = {k0, k1, k2, k3};
            double part = 0.0;
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool sel = j < 8 && 4 * j + r < keff;
                const float zv = key_z(kk[r]);
                const unsigned e = (__float_as_uint(zv) >> 23) & 0xffu;
                ok = ok && (!sel || zv == 0.0f || (e >= 115u && e < 135u));
                part += sel ? (double)zv : 0.0;
            }
            quick = __ballot(!ok) == 0ull;
            const double s = row16_sum_f64(part);  // (lanes 0-15: the sum of lanes 0-7's parts)
            const int last = keff - 1;
            const unsigned pick = (last & 3) == 0 ? k0 : ((last & 3) == 1 ? k1 : ((last & 3) == 2 ? k2 : k3));
            if (quick) {
                const long long sb = __double_as_longlong(s);
                const unsigned slo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)sb, 0);
                const unsigned shi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)sb >> 32), 0);
                sum = __longlong_as_double((long long)(((unsigned long long)shi << 32) | slo));
                T = (unsigned)__builtin_amdgcn_readlane((int)pick, last >> 2);
            }
            k0 = o0;
            k1 = o1;
            k2 = o2;
            k3 = o3;
        }
    }
    for (int r = 0; r < num_lpr; ++r) {  // the keff smallest kept keys, ascending
        const bool take = need && !quick && r < keff;
        if (!__any(take)) break;
        const unsigned m = Row<G>::min_u32(k0);
        if (take) {
            sum += (double)key_z(m);
            T = m;
        }
        const int lowest = __ffsll((long long)Row<G>::ballot(take && k0 == m)) - 1;
        if (take && j == lowest) {
            k0 = k1;
            k1 = k2;
            k2 = k3;
            k3 = INF;
        }
    }
    const bool fast = need && keff > 0 && (Row<G>::min_u32(dropped) < T || force != 0);  // second pass needed (row-uniform)
    bool exact_path = false;
    if (__any(fast)) {
        const unsigned U = T;  // an upper bound of the keff-th smallest key of the row
        unsigned key[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) key[q] = INF;
        bool overflow = false;
        const unsigned nchunk_max = wave_max_u32(fast ? patch_chunks<G>(pts, use_hi) : 0u);
        for (unsigned c = 0; c < nchunk_max; ++c) {
            ChunkZ cp;
            load_chunk_z<G>(cp, pts, chunk_sel<G>(pts, c, use_hi, fast));
            const unsigned act = chunk_act_z<G>(cp);
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const bool e = (act >> k & 1u) && !(use_cutoff && (double)cp.z[k] < cutoff);
                const unsigned kk = z_key(cp.z[k]);
                const bool cand = fast && e && kk < U;
                if (__any(cand)) {  // sorted insertion; whatever falls off the end must be "none"
                    unsigned x = cand ? kk : INF;
#pragma unroll
                    for (int q = 0; q < 8; ++q) ce(key[q], x);
                    overflow = overflow || (x != INF);
                }
            }
        }
        const bool row_over = Row<G>::ballot(overflow) != 0ull;
        exact_path = fast && (row_over || force == 2);
        const bool ok = fast && !row_over && force != 2;
        int nless = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) nless += key[q] != INF ? 1 : 0;
        const int c_less = Row<G>::sum_i32(nless);
        const int take_n = c_less < keff ? c_less : keff;
        double sum2 = 0.0;
        for (int r = 0; r < num_lpr; ++r) {  // the take_n smallest gathered keys, ascending
            const bool take = ok && r < take_n;
            if (!__any(take)) break;
            const unsigned head = key[0];
            const unsigned m = Row<G>::min_u32(head);
            if (take) sum2 += (double)key_z(m);
            const int lowest = __ffsll((long long)Row<G>::ballot(take && head == m)) - 1;
            if (take && j == lowest) {
#pragma unroll
                for (int q = 0; q < 7; ++q) key[q] = key[q + 1];
                key[7] = INF;
            }
        }
        const double zu = (double)key_z(U);
        for (int r = 0; r < num_lpr; ++r) {  // copies of U (at least keff keys are <= U)
            const bool take = ok && r >= take_n && r < keff;
            if (take) sum2 += zu;
        }
        if (ok) sum = sum2;
    }
    if (__any(exact_path)) {
        // extraction by distinct values, ascending: one pass per distinct value among the keff smallest
        double xsum = 0.0;
        int remaining = exact_path ? keff : 0;
        bool first = true;
        unsigned prev = 0;
        for (int guard = 0; guard <= num_lpr; ++guard) {
            if (!__any(remaining > 0)) break;
            unsigned vmin = INF;
            int vcnt = 0;
            const unsigned nchunk_max = wave_max_u32(remaining > 0 ? patch_chunks<G>(pts, use_hi) : 0u);
            for (unsigned c = 0; c < nchunk_max; ++c) {
                ChunkZ cp;
                load_chunk_z<G>(cp, pts, chunk_sel<G>(pts, c, use_hi, remaining > 0));
                const unsigned act = chunk_act_z<G>(cp);
#pragma unroll
                for (int k = 0; k < kPPT; ++k) {
                    const bool e = (act >> k & 1u) && !(use_cutoff && (double)cp.z[k] < cutoff);
                    const unsigned kk = z_key(cp.z[k]);
                    if (e && (first || kk > prev)) {
                        if (kk < vmin) {
                            vmin = kk;
                            vcnt = 1;
                        } else if (kk == vmin) {
                            ++vcnt;
                        }
                    }
                }
            }
            const unsigned v = Row<G>::min_u32(vmin);
            const int mult = Row<G>::sum_i32(vmin == v ? vcnt : 0);
            if (remaining > 0) {
                if (v == INF) {
                    remaining = 0;  // (only NaN-keyed leftovers)
                } else {
                    const int take = mult < remaining ? mult : remaining;
                    const double zv = (double)key_z(v);
                    for (int r = 0; r < take; ++r) xsum += zv;
                    remaining -= take;
                    prev = v;
                }
            }
            first = false;
        }
        if (exact_path) sum = xsum;
    }
    return keff ? sum / (double)keff : 0.0;  // ref :103
}


// what every fit kernel needs of a patch before its chain starts
struct PatchCtx {
    int bin, zone;
    unsigned n;                           // points of the bin = n_lo + n_hi
    unsigned off_lo, n_lo, off_hi, n_hi;  // its two parts (slots relative to the frame's first)
    float ox, oy;                         // origin of the bin's fixed-point sums
};
__device__ __forceinline__ PatchCtx patch_ctx(const PwppBatch &Bt, int f, unsigned slot, bool alive) {
    const PwppDevParams &P = Bt.P;
    const int NP = PWPP_NUM_PARTS(P.num_bins);
    PatchCtx c;
    c.bin = alive ? (int)Bt.cls_list[(size_t)f * P.num_bins + slot] : 0;
    const uint2 cnt = *reinterpret_cast<const uint2 *>(Bt.part_count + (size_t)f * NP + PWPP_PART_LO(c.bin));  // (low, high: neighbours)
    const uint2 off = *reinterpret_cast<const uint2 *>(Bt.part_off + (size_t)f * NP + PWPP_PART_LO(c.bin));
    c.n_lo = alive ? cnt.x : 0u;
    c.n_hi = alive ? cnt.y : 0u;
    c.off_lo = alive ? off.x : 0u;
    c.off_hi = alive ? off.y : 0u;
    c.n = c.n_lo + c.n_hi;
    c.zone = c.bin < P.bin_base[1] ? 0 : (c.bin < P.bin_base[2] ? 1 : (c.bin < P.bin_base[3] ? 2 : 3));
    const float2 o = Bt.bin_origin[c.bin];
    c.ox = o.x;
    c.oy = o.y;
    return c;
}
__device__ __forceinline__ PatchRef patch_ref(const PwppBatch &Bt, const PwppFrameDesc &fd, unsigned off_lo, unsigned n_lo, unsigned off_hi, unsigned n_hi,
                                              int bin = 0) {
    PatchRef r;
    r.bin = bin;
    r.arena_base = Bt.arena_base;
    r.num_parts = PWPP_NUM_PARTS(Bt.P.num_bins);
    r.z = Bt.sorted_z + fd.sbase;
    r.xy = Bt.sorted_xy + fd.sbase;
    r.idx = Bt.sorted_idx + fd.sbase;
    r.off_lo = off_lo;
    r.n_lo = n_lo;
    r.off_hi = off_hi;
    r.n_hi = n_hi;
    return r;
}
__device__ __forceinline__ PatchRef patch_ref(const PwppBatch &Bt, const PwppFrameDesc &fd, const PatchCtx &pc) {
    return patch_ref(Bt, fd, pc.off_lo, pc.n_lo, pc.off_hi, pc.n_hi, pc.bin);
}

// Does a pass of this stage have to read the high part of the patch (n_hi points with z >= zs or NaN, x and y inside bb)?
//   seed stages (ref :108,145: z < lpr + th)        no, if the largest threshold of the pass is not above zs;
//   R-GPF round (ref :525: n . p + d < th_dist)     no, if plane_dist is not below th_dist at the corner of the box
//       bb x [zs, inf) where the plane is lowest: plane_dist is a chain of correctly rounded -- hence monotone --
//       operations in every coordinate (no FMA contraction: -ffp-contract=off is part of the contract), so every
//       point of the box gets at least the corner's value, bit for bit.  (nz >= 0 since ref :68; a NaN anywhere
//       fails the comparison and the part is read.)
// An R-VPF strip (|dist| < th_dist_v around a near-vertical plane) always reads it.
__device__ __forceinline__ bool stage_needs_hi(int kind, unsigned n_hi, double thr_seed_max, double th_dist, const PlaneFit &pl,
                                               const float4 &bb, float zs) {
    if (n_hi == 0u) return false;
    if (kind == ST_ITER) {
        if (!(pl.nz >= 0.0f)) return true;
        const float cx = pl.nx >= 0.0f ? bb.x : bb.y, cy = pl.ny >= 0.0f ? bb.z : bb.w;
        return !(plane_dist(pl.nx, pl.ny, pl.nz, pl.d, cx, cy, zs) >= th_dist);
    }
    return !(thr_seed_max <= (double)zs);
}
__device__ __forceinline__ void plane_clear(PlaneFit &pl) {
    pl.nx = pl.ny = pl.nz = 0.0f;
    pl.mean[0] = pl.mean[1] = pl.mean[2] = 0.0f;
    pl.sv[0] = pl.sv[1] = pl.sv[2] = 0.0f;
    pl.d = 0.0;
}

// ------------------------------------------------------------------------------------------
// Streaming rows: G lanes per patch (8, 16 or 64), the points are re-read from L2 at every
// stage in chunks of 8 per lane.  Fewer lanes per patch = more patches per wave = the serial
// eigen-solve (the dominant instruction count) is shared by more patches; the price is more
// points per lane.  Patches of a frame are sorted by size (k_czm_scan), so the rows of one
// wave have similar trip counts.  A workgroup per big patch was tried and rejected: all but
// one wave idle during the solve (2.5 + 7.5 ms per 1024 frames vs 2.6 ms for one wave each).
// ------------------------------------------------------------------------------------------
template <int G, bool WIDE>
__device__ __forceinline__ void fit_srows_body(const PwppBatch &Bt, int b_lo, int b_hi, unsigned by /* block index among the row blocks */) {
    const int f = blockIdx.x;  // frame = fast grid dimension: most blocks of a frame's worst-case grid are empty, and with the
                               // frame in blockIdx.y the working blocks formed a pattern of period 32 = 8 XCDs x 4 SEs (3x slower)
    const PwppDevParams &P = Bt.P;
    const uint32_t *cs = Bt.cls_start + (size_t)f * PWPP_CLS_STRIDE;
    const unsigned cbeg = cs[b_lo], cend = cs[b_hi];
    const unsigned tid = by * kBlock + threadIdx.x;
    if (cbeg + (tid & ~63u) / G >= cend) return;  // this wave has no patch
    const bool alive = cbeg + tid / G < cend;  // row-uniform
    // largest patches first: the list is sorted by size, a wave's run time grows with its patch,
    // and the last waves to start should be the short ones
    const unsigned slot = cend - 1u - tid / G;
    const int j = lane_id() & (G - 1);
    const PatchCtx pc = patch_ctx(Bt, f, slot, alive);
    const int bin = pc.bin, zone = pc.zone;
    const unsigned n = pc.n;
    const PwppFrameDesc fd = Bt.frames[f];
    const PatchRef pts = patch_ref(Bt, fd, pc);
    uint8_t *frame_member = Bt.member + fd.mbase;
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const double cutoff = P.margin * sensor_height;
    const float zs = hi_split_z(P, sensor_height);
    const float4 bb = Bt.bin_bbox[bin];
    const bool use_cutoff = zone == 0;
    const double scale = (double)(1 << P.fxp_shift);
    const bool wide = WIDE || __any(n > 2047u);  // wave-uniform: some row's second moments may leave int64 in the cross-lane sum
    // the totals of the row's last R-GPF round (early termination, see k_fit_w64): n, S1[3], S2[6] as two 64-bit halves each
    __shared__ long long s_prev[kBlock / G][16];
    long long(&prev)[16] = s_prev[threadIdx.x / G];

    PlaneFit pl;
    plane_clear(pl);
    double lpr = 0.0;
    bool lpr_valid = false, z0_set = false;
    float z0 = 0.0f;
    FxpOrg org = fxp_org(pc.ox, pc.oy, 0.0f, scale, P.fxp_zr);
    int kind = !alive ? ST_DONE : ((P.enable_RVPF != 0 && zone == 0) ? ST_VPF : ST_SEED);  // row-uniform
    int it = 0;
    bool fitted = false;  // a plane of this patch's own exists

    for (int guard = 0; guard < 4 * P.num_iter + 8; ++guard) {
        if (!__any(kind != ST_DONE)) break;
        const bool need_lpr = (kind == ST_VPF || kind == ST_SEED) && !lpr_valid;
        if (__any(need_lpr)) {
            const double l = srow_lpr<G>(pts, need_lpr, use_cutoff, cutoff, P.num_lpr, (Bt.debug >> 14) & 3);
            if (need_lpr) {
                lpr = l;
                lpr_valid = true;
                if (!z0_set) {  // the z origin of this patch's sums: its first lowest-point representative
                    z0 = fxp_z_origin(l);
                    z0_set = true;
                    org = fxp_org(pc.ox, pc.oy, z0, scale, P.fxp_zr);
                }
            }
        }
        const double thr_seed = lpr + ((kind == ST_VPF || kind == ST_LAZY) ? P.th_seeds_v : P.th_seeds);
        const bool last = kind == ST_ITER && it == P.num_iter - 1;
        const bool on = kind != ST_DONE;
        const bool use_hi = on && stage_needs_hi(kind, pc.n_hi, thr_seed, P.th_dist, pl, bb, zs);  // row-uniform
        const unsigned nchunk_max = wave_max_u32(on ? patch_chunks<G>(pts, use_hi) : 0u);
        const float T = stage_threshold(kind, pl.d, P.th_dist, thr_seed);  // the pass's test in float (lane_stage_accum)
        // (early termination and the membership plane: see k_fit_w64 -- every R-GPF round from the second on, and the last one,
        // leaves its set in the plane; a round whose totals repeat the round before's ends the patch)
        const bool wbits = kind == ST_ITER && (last || it >= 1);
        MomentsT<WIDE> m;
        m.clear();
        bool clamped = false;
        ChunkPts cp;
        load_chunk<G>(cp, pts, chunk_sel<G>(pts, 0u, use_hi, on));
        for (unsigned c = 0; c < nchunk_max; ++c) {
            ChunkPts nx;  // the next chunk is in flight while this one is accumulated
            load_chunk<G>(nx, pts, chunk_sel<G>(pts, c + 1u, use_hi, on));
            bool hit = false;
            const unsigned gmask = lane_stage_accum<G>(cp, kind, T, pl.nx, pl.ny, pl.nz, scale, org, m, hit);
            clamped = clamped || (wbits && hit);
            if (__any(wbits)) store_member<G>(frame_member, chunk_sel<G>(pts, c, use_hi, on), gmask, wbits);
            cp = nx;
        }
        const long long cnt = Row<G>::sum_i64(m.n);
        bool conv = false;
        {
            long long s1[3];
            __int128 s2[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) s1[k] = Row<G>::sum_i64(m.first(k));
            if (!wide) {
#pragma unroll
                for (int k = 0; k < 6; ++k) s2[k] = (__int128)Row<G>::sum_i64((long long)m.second(k));  // narrow grid, <= 2047 points: fits int64
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const __int128 t = m.second(k);
                    s2[k] = join_halves(Row<G>::sum_i64((long long)((unsigned long long)t & 0xffffffffull)), Row<G>::sum_i64((long long)(t >> 32)));
                }
            }
            if (__any(kind == ST_ITER)) {  // this round's totals against the last round's, then they take their place (lane 0 of the row keeps them)
                bool same = prev[0] == cnt;
#pragma unroll
                for (int k = 0; k < 3; ++k) same = same && prev[1 + k] == s1[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) same = same && prev[4 + k] == (long long)(unsigned long long)s2[k] && prev[10 + k] == (long long)(s2[k] >> 64);
                conv = kind == ST_ITER && !last && it >= 1 && cnt > 3 && same;  // (1-3 points: the tiny-fit path, never compared)
                wave_lds_sync();
                if (kind == ST_ITER && j == 0) {
                    prev[0] = cnt;
#pragma unroll
                    for (int k = 0; k < 3; ++k) prev[1 + k] = s1[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        prev[4 + k] = (long long)(unsigned long long)s2[k];
                        prev[10 + k] = (long long)(s2[k] >> 64);
                    }
                }
                wave_lds_sync();
            }
            if (__any(clamped && (last || conv)) && lane_id() == 0) flag_clamped(Bt, f);
            const bool fit = kind != ST_DONE && cnt > 0 && !conv;  // empty: ref :49; converged: the solve would return the plane in force
            const bool tiny = fit && cnt <= 3;            // contract v3: the reference's float arithmetic (row-uniform)
            float mt[3], ct[6];
            if (__any(tiny)) tiny_fit_row<G>(pts, tiny, kind, T, pl.nx, pl.ny, pl.nz, mt, ct);
            if (fit) {
                float mean[3], c6[6];
                if (tiny) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) mean[k] = mt[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) c6[k] = ct[k];
                } else {
                    mean_cov_from_totals(cnt, s1, s2, P.fxp_shift, pc.ox, pc.oy, z0, mean, c6);
                }
                plane_from_mean_c6(mean, c6, Bt.debug, pl);
                fitted = true;
            }
        }
        if (kind != ST_DONE && needs_previous_plane(P, kind, zone, fitted)) {
            if (j == 0) mark_needs_previous_plane(Bt, f, Bt.recs + (size_t)f * P.num_bins + bin, n);
            kind = ST_DONE;
        }
        // (the rows of a wave may be at different stages: everything wave-wide -- wave_max_u32 -- stays outside the per-stage branches)
        const bool vertical = kind == ST_VPF && (double)pl.nz < P.uprightness_thr;  // ref :489
        if (__any(vertical)) {
            bool any = false;
            const unsigned nstrip_max = wave_max_u32(vertical ? patch_chunks<G>(pts, true) : 0u);
            const StripBand band = strip_band(pl.d, P.th_dist_v);
            for (unsigned c = 0; c < nstrip_max; ++c) {
                ChunkPts cs2;
                const PartSel sel = chunk_sel<G>(pts, c, true, vertical);
                load_chunk<G>(cs2, pts, sel);
                const unsigned hit = lane_strip<G>(cs2, vertical, pl.nx, pl.ny, pl.nz, band);
#pragma unroll
                for (int k = 0; k < kPPT; ++k) {
                    if (hit >> k & 1u) {
                        strip_point(pts, chunk_slot<G>(sel, k, (unsigned)j), it);
                    }
                }
                any = any || hit != 0;
            }
            if (Row<G>::ballot(any) != 0ull) lpr_valid = false;  // the working set changed
        }
        if (kind == ST_VPF) {
            ++it;
            if (!vertical || it >= P.num_iter) {
                kind = ST_SEED;
                it = 0;
            }
        } else if (kind == ST_SEED) {
            kind = (cnt == 0 && P.enable_RVPF != 0 && zone != 0) ? ST_LAZY : ST_ITER;
        } else if (kind == ST_LAZY) {
            kind = ST_ITER;
        } else if (kind == ST_ITER) {
            if (last || conv) {
                if (j == 0)
                    write_record(Bt.recs + (size_t)f * P.num_bins + bin, pl, n, (unsigned)cnt, !use_hi && pc.n_hi > 0u, G == 64 ? 6 : (G == 32 ? 5 : (G == 16 ? 4 : 3)), it + 1);
                kind = ST_DONE;
            }
            ++it;
        }
    }
}

template <int G, bool WIDE>
__global__ __launch_bounds__(kBlock, 3) void k_fit_srows(PwppBatch Bt, int b_lo, int b_hi) {
    fit_srows_body<G, WIDE>(Bt, b_lo, b_hi, blockIdx.y);
}

// ------------------------------------------------------------------------------------------
// k_fit_w64: one wave = up to 64 patches.  The VALU profile of the row kernels showed the serial
// eigen-solve (~1800 VALU instructions, executed by all 64 lanes of a wave for only 1-4 patches)
// to be the largest single consumer of issue cycles of the fit stage.  Here a wave owns PW patches of a frame:
//   * points phases run 64 / G patches at a time in rows of G lanes (points streamed from L2) and
//     leave the integer moments of every patch in LDS;
//   * the solve phase runs ONCE per stage with lane p solving patch p -- PW different 3x3
//     problems per instruction stream instead of 1-4;
//   * lane p ("owner") carries patch p's state machine (stage, LPR, plane) in registers and
//     publishes what the rows need (plane, thresholds, origin) through LDS.
// Patches are handed to the waves of a frame round-robin over the size-sorted list, so every
// wave gets a similar mix and the rows of a points phase have similar trip counts.
// Waves are independent: no workgroup barrier anywhere.
// ------------------------------------------------------------------------------------------
struct W64Patch {
    unsigned off_lo, n_lo, off_hi, n_hi;  // the two parts of the bin
    int kind;        // stage of the coming points phase; ST_DONE = nothing to do
    int flags;       // bit0: last R-GPF round, bit1: zone-0 cut-off applies, bit2: dual seed pass,
                     // bit3: the pass reads the high part too, bit4: an R-VPF strip removed something; bits 8-15: the R-VPF
                     // round of the strip in progress (reference-order output);
                     // bit5: the pass leaves its set in the membership plane (an R-GPF round whose set may be the final one),
                     // bit6: its totals may be compared with the totals in LDS (those of the round before, which gave the plane in
                     // force), bit7 (set by the ROW): they are equal -- the patch has converged (early termination below)
    int bin;
    float nx, ny, nz;
    float z0;        // z origin of the patch's fixed-point sums
    float ox, oy;    // x, y origin
    // what the rows need of the pass's tests, as float thresholds (plane_test_threshold):
    //   points phase: t = the stage's test, t2 = the upper end of the band of a dual seed pass;
    //   R-VPF strip : (t, t2) = the band (lo, hi) around the plane.
    // The lowest-point phase hands the representative (a double) to the owner lane through the same eight bytes.
    union {
        double lpr;
        struct {
            float t, t2;
        } thr;
    } u;
};
// The state machine of a patch (what its owner lane carries from stage to stage).  In the kernels with 64-lane rows it
// lives in LDS, not in registers: only PW = 2-8 lanes own a patch, but a register is 64 lanes wide, and the ~40 registers
// of this state were what the points phase lacked to keep a second chunk in flight (182 spilled registers with the
// prefetch, 8 without).  The 16-lane kernels (64 owners per wave, LDS full) keep it in registers.
struct W64Owner {
    PlaneFit pl;
    double lpr;
    float z0;
    int kind, it;
    int lpr_valid, z0_set, fitted, hi_skipped, stash_valid;
    int bin, zone;
    unsigned n, n_hi;
    float4 bb;
    long long cnt;
};
template <int PW, bool DUAL, int MW>
struct W64Shared {
    W64Owner o[DUAL ? PW + 1 : 1];  // (slot PW: the lanes that own nothing; they only ever read kind == ST_DONE)
    W64Patch p[PW];
    unsigned char order[64];  // the patches that take part in the coming points phase, packed (sub-batches without idle rows)
    long long mom[PW][MW];
    long long mom2[DUAL ? PW : 1][MW];  // dual seed pass: moments of the band; then the stashed seed totals of the R-GPF stage
};


// G = lanes per patch in the points phases (16: four patches at a time; 64: one at a time, for
// big bins), PW = patches owned by the wave = lanes active in the solve phase.
// Moments per patch in LDS: rows of 16 lanes only see patches below 2048 points, whose ten totals fit
// int64; 64-lane rows leave sixteen values (second moments as 32-bit halves, Row<64>::reduce16_scatter).
template <int G, int PW, bool WIDE>
__global__ __launch_bounds__(64, G == 64 ? PWPP_W64_OCC : (WIDE ? PWPP_W16_WIDE_OCC : PWPP_W16_OCC)) void k_fit_w64(PwppBatch Bt, int b_lo, int b_hi) {
    // ONE WAVE PER WORKGROUP: the waves never talk to each other, and a workgroup of four only starts when a CU has
    // room for all four at once -- with waves of very different lifetimes the slots of the early finishers stood empty
    // (27 % of the wave slots of k_fit_w64<64,2>, profiles/).
    constexpr int MW = (G == 64 || WIDE) ? 16 : 10;  // (the wide grid: second moments beyond int64 -- sixteen values in 16-lane rows too)
    typedef MomentsT<WIDE> Moments;
    __shared__ W64Shared<PW, G == 64, MW> sh;
    constexpr int R = 64 / G;      // patches per points-phase sub-batch
    constexpr int NSB = PW / R;    // sub-batches
    static_assert(PW % R == 0 && PW <= 64, "patches per wave");
    const int f = blockIdx.x;  // frame = fast grid dimension, see fit_srows_body
    const PwppDevParams &P = Bt.P;
    const uint32_t *cs = Bt.cls_start + (size_t)f * PWPP_CLS_STRIDE;
    const unsigned cbeg = cs[b_lo], cend = cs[b_hi];
    const unsigned npatch = cend - cbeg;
    const unsigned nwaves = (npatch + PW - 1u) / PW;
    const unsigned w = blockIdx.y;
    if (w >= nwaves) return;  // wave-uniform
    const int ln = lane_id();
    const int j = ln & (G - 1), row = ln / G;
    const PwppFrameDesc fd = Bt.frames[f];
    uint8_t *frame_member = Bt.member + fd.mbase;
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const double cutoff = P.margin * sensor_height;  // ref :90
    const float zs = hi_split_z(P, sensor_height);
    const double scale = (double)(1 << P.fxp_shift);

    // ---- owner lane: patch `ln` of this wave (lanes >= PW own nothing)
    const unsigned slot = cbeg + w + (unsigned)ln * nwaves;
    const bool alive = ln < PW && slot < cend;
    const PatchCtx pc = patch_ctx(Bt, f, slot, alive);
    constexpr bool OWN_LDS = G == 64;  // the owner state in LDS (W64Owner)
    W64Owner own_regs;
    const int oi = ln < PW ? ln : PW;
#define O(fld) (*(OWN_LDS ? &sh.o[OWN_LDS ? oi : 0].fld : &own_regs.fld))
    if (!OWN_LDS || ln <= PW) {
        plane_clear(O(pl));
        O(lpr) = 0.0;
        O(z0) = 0.0f;
        O(kind) = !alive ? ST_DONE : ((P.enable_RVPF != 0 && pc.zone == 0) ? ST_VPF : ST_SEED);
        O(it) = 0;
        O(lpr_valid) = 0;
        O(z0_set) = 0;
        O(fitted) = 0;      // a plane of this patch's own exists
        O(hi_skipped) = 0;  // the last points phase of this patch did not read the high part
        O(stash_valid) = 0;
        O(bin) = pc.bin;
        O(zone) = pc.zone;
        O(n) = pc.n;
        O(n_hi) = pc.n_hi;
        O(bb) = Bt.bin_bbox[pc.bin];
        O(cnt) = 0;
    }
    // Dual seed pass (big bins, G == 64): the R-VPF round and the R-GPF seed stage of a zone-0 patch
    // select seeds from the same working set with the same lowest-point representative and two
    // thresholds (th_seeds_v / th_seeds, ref :480,:511).  The R-VPF pass therefore accumulates the
    // moments below the smaller threshold and those of the band up to the larger one; the sums are
    // exact integers, so one set is A and the other A + B.  If R-VPF removes nothing, the R-GPF seed
    // stage takes its totals from the stash instead of streaming the patch again.
    constexpr bool DUAL = G == 64;
    const bool v_is_hi = P.th_seeds_v >= P.th_seeds;
    if (ln < PW) {
        sh.p[ln].off_lo = pc.off_lo;
        sh.p[ln].n_lo = pc.n_lo;
        sh.p[ln].off_hi = pc.off_hi;
        sh.p[ln].n_hi = pc.n_hi;
        sh.p[ln].kind = ST_DONE;
        sh.p[ln].flags = pc.zone == 0 ? 2 : 0;
        sh.p[ln].bin = pc.bin;
        sh.p[ln].ox = pc.ox;
        sh.p[ln].oy = pc.oy;
        sh.p[ln].z0 = 0.0f;
    }
    wave_lds_sync();

    // Phase profile (build with -DPWPP_PHASE_PROBE, option debug_flags = 4: tools/fit_phases.py): shader-clock cycles every wave spends in
    // each phase of its loop, added up over all waves in Bt.dbg[16 + 8 * (G == 64) + phase] -- 0 set-up, 1 lowest points, 2 publish,
    // 3 points phase, 4 solve (tiny fits included), 5 R-VPF strip, 6 state step; [32 + ...]: the waves counted.
#ifdef PWPP_PHASE_PROBE
    long long probe_t = (long long)clock64();
#define PWPP_PHASE(ph)                                                                                                   \
    do {                                                                                                                 \
        const long long now_ = (long long)clock64();                                                                     \
        if ((Bt.debug & 4) && ln == 0) atomicAdd(&Bt.dbg[16 + (G == 64 ? 8 : 0) + (ph)], (unsigned long long)(now_ - probe_t)); \
        probe_t = now_;                                                                                                  \
    } while (0)
    if ((Bt.debug & 4) && ln == 0) atomicAdd(&Bt.dbg[32 + (G == 64 ? 8 : 0)], 1ull);
    long long sub_t = 0;
#define PWPP_SUB_BEGIN() do { sub_t = (long long)clock64(); } while (0)
#define PWPP_SUB(k)                                                                                                       \
    do {                                                                                                                 \
        const long long now_ = (long long)clock64();                                                                     \
        if ((Bt.debug & 4) && ln == 0) atomicAdd(&Bt.dbg[(G == 64 ? 48 : 56) + (k)], (unsigned long long)(now_ - sub_t)); \
        sub_t = now_;                                                                                                    \
    } while (0)
#else
#define PWPP_PHASE(ph) do { } while (0)
#define PWPP_SUB_BEGIN() do { } while (0)
#define PWPP_SUB(k) do { } while (0)
#endif
    PWPP_PHASE(0);
    for (int guard = 0; guard < 4 * P.num_iter + 8; ++guard) {
        if (!__any(O(kind) != ST_DONE)) break;

        // ---- A. lowest-point representative (ref :84-103) for the patches whose working set is new: z only
        const bool need_lpr = (O(kind) == ST_VPF || O(kind) == ST_SEED) && !O(lpr_valid);
        const unsigned long long lpr_mask = __ballot(need_lpr);
        if (lpr_mask) {
            // (16-lane rows: the z pass is the FIRST touch of a patch -- every sub-batch waited for its own HBM round trip, sixteen in
            // a row.  The first chunk of the NEXT sub-batch that needs a pass is requested before this one's keys are ranked and
            // sorted: eight registers, in a phase far from the kernel's register peak.  Round 6.)
            constexpr bool kAhead = G == 16;
            auto sb_needs = [&](int sb) { return sb < NSB && ((lpr_mask >> (R * sb)) & ((1ull << R) - 1ull)) != 0ull; };
            auto request = [&](int sb, ChunkZ &cz) {
                const int q = R * sb + row;
                const bool need_row = (lpr_mask >> q) & 1ull;
                const PatchRef qpts = patch_ref(Bt, fd, sh.p[q].off_lo, sh.p[q].n_lo, sh.p[q].off_hi, sh.p[q].n_hi);
                PartSel sel;
                sel.off = qpts.off_lo;
                sel.n = need_row ? qpts.n_lo : 0u;
                sel.c = 0u;
                sel.moff = 0u;
                load_chunk_z<G>(cz, qpts, sel);
            };
            ChunkZ ahead;
            int sb = 0;
            while (sb < NSB && !sb_needs(sb)) ++sb;
            if (kAhead && sb < NSB) request(sb, ahead);
            for (; sb < NSB;) {
                int nxt = sb + 1;
                while (nxt < NSB && !sb_needs(nxt)) ++nxt;
                const ChunkZ cur = ahead;
                if (kAhead && nxt < NSB) request(nxt, ahead);
                const int q = R * sb + row;
                const bool need_row = (lpr_mask >> q) & 1ull;
                const bool use_cutoff = (sh.p[q].flags & 2) != 0;
                const PatchRef qpts = patch_ref(Bt, fd, sh.p[q].off_lo, sh.p[q].n_lo, sh.p[q].off_hi, sh.p[q].n_hi);
                const double l = srow_lpr<G>(qpts, need_row, use_cutoff, cutoff, P.num_lpr, (Bt.debug >> 14) & 3, kAhead ? &cur : nullptr);
                if (need_row && j == 0) sh.p[q].u.lpr = l;
                sb = nxt;
            }
            wave_lds_sync();
            if (need_lpr) {
                const double l = sh.p[ln].u.lpr;
                O(lpr) = l;
                O(lpr_valid) = 1;
                if (!O(z0_set)) {  // the z origin of this patch's sums: its first lowest-point representative
                    O(z0) = fxp_z_origin(l);
                    O(z0_set) = 1;
                }
            }
        }

        PWPP_PHASE(1);
        // ---- B. publish the stage of every patch
        const bool dual_now = DUAL && O(kind) == ST_VPF;                         // this round's pass also fills the stash
        const bool from_stash = DUAL && O(kind) == ST_SEED && O(stash_valid) != 0;  // no pass: totals come from the stash
        const int pub_kind = from_stash ? ST_DONE : O(kind);
        if (ln < PW) {
            const int kind = O(kind);
            const PlaneFit pl = O(pl);
            const double lpr = O(lpr);
            const bool last = kind == ST_ITER && O(it) == P.num_iter - 1;
            // EARLY TERMINATION (exact): the plane is a function of the ten integer totals.  If an R-GPF round's totals equal
            // those of the round before -- which gave the plane this round tested with -- the next plane is this plane bit for
            // bit, hence every later round selects this round's set again and the final plane (ref :537-542) is the one in
            // force: the patch is finished, its split is what this round left in the membership plane.  Every round that can
            // be the last one that way (from the second on) writes its bits; the totals of 1-3 points are not compared (the
            // tiny-fit path replaces them in LDS).
            const bool wbits = kind == ST_ITER && (last || O(it) >= 1);
            const bool cmp = kind == ST_ITER && !last && O(it) >= 1 && O(cnt) > 3;
            const double th = (kind == ST_VPF || kind == ST_LAZY) ? P.th_seeds_v : P.th_seeds;
            const double thr_seed = lpr + (dual_now ? (v_is_hi ? P.th_seeds : P.th_seeds_v) : th);
            const double thr_band = lpr + (v_is_hi ? P.th_seeds_v : P.th_seeds);
            const unsigned n_hi = O(n_hi);
            const bool use_hi = pub_kind != ST_DONE &&
                                stage_needs_hi(kind, n_hi, dual_now && thr_band > thr_seed ? thr_band : thr_seed, P.th_dist, pl, O(bb), zs);
            if (pub_kind != ST_DONE) O(hi_skipped) = !use_hi && n_hi > 0u;
            sh.p[ln].kind = pub_kind;
            sh.p[ln].flags = (O(zone) == 0 ? 2 : 0) | (last ? 1 : 0) | (dual_now ? 4 : 0) | (use_hi ? 8 : 0) | (wbits ? 32 : 0) | (cmp ? 64 : 0);
            sh.p[ln].nx = pl.nx;
            sh.p[ln].ny = pl.ny;
            sh.p[ln].nz = pl.nz;
            sh.p[ln].z0 = O(z0);
            if (pub_kind != ST_DONE) {  // the pass's tests as float thresholds (lane_stage_accum)
                sh.p[ln].u.thr.t = stage_threshold(kind, pl.d, P.th_dist, thr_seed);
                sh.p[ln].u.thr.t2 = dual_now ? plane_test_threshold(0.0, thr_band) : 0.0f;
            }
        }
        // the patches of the coming points phase, packed: patches that are finished (early termination), that take their
        // totals from the stash or that are at another point of their chain leave no idle rows in the sub-batches
        const unsigned long long act_mask = __ballot(pub_kind != ST_DONE);
        if (pub_kind != ST_DONE) sh.order[__popcll(act_mask & ((1ull << ln) - 1ull))] = (unsigned char)ln;
        const int nact = __popcll(act_mask);
        wave_lds_sync();

        PWPP_PHASE(2);
        // ---- C. points phase: R patches at a time, G lanes each
        for (int sb = 0; R * sb < nact; ++sb) {
            const bool row_on = R * sb + row < nact;
            const int q = row_on ? (int)sh.order[R * sb + row] : (int)sh.order[0];
            const W64Patch pp = sh.p[q];
            const bool on = row_on && pp.kind != ST_DONE;
            const bool last = on && (pp.flags & 1);
            const bool wbits = on && (pp.flags & 32);
            const bool use_hi = (pp.flags & 8) != 0;
            const FxpOrg org = fxp_org(pp.ox, pp.oy, pp.z0, scale, P.fxp_zr);
            const PatchRef pts = patch_ref(Bt, fd, pp.off_lo, pp.n_lo, pp.off_hi, pp.n_hi, pp.bin);
            const unsigned nchunk_max = wave_max_u32(on ? patch_chunks<G>(pts, use_hi) : 0u);
            const bool dual = DUAL && on && (pp.flags & 4);
            Moments m, m2;
            m.clear();
            m2.clear();
            bool clamped = false;  // a height of the round's set lay outside z0 +- ZR (matters if this set is the final one)
            // The loads of chunk c + 1 are issued before chunk c is consumed: a wave is a chain load -> wait -> ~250
            // instructions, and with 3-4 waves per SIMD the waits were not covered (k_fit_w64<64,2> moved its bytes at
            // 4.8 TB/s where a plain read stream reaches 6.3, tools/ubench/read_bw.hip).  The solve phase sets the
            // register allocation of these kernels, so the second chunk in flight costs the points phase nothing.
            constexpr bool kPrefetch = PWPP_FIT_PREFETCH != 0 && G == 64;  // (the 16-lane kernels have no registers to spare: 76 spilled without it)
            const bool any_wbits = __any(wbits);
            ChunkPts cp;
            if (nchunk_max > 0u) load_chunk<G>(cp, pts, chunk_sel<G>(pts, 0u, use_hi, on));
            for (unsigned c = 0; c < nchunk_max; ++c) {
                ChunkPts nx;
                if (kPrefetch && c + 1u < nchunk_max) load_chunk<G>(nx, pts, chunk_sel<G>(pts, c + 1u, use_hi, on));  // (wave-uniform)
                bool hit = false;
                const unsigned gmask = lane_stage_accum<G>(cp, pp.kind, pp.u.thr.t, pp.nx, pp.ny, pp.nz, scale, org, m, hit);
                clamped = clamped || (wbits && hit);
                if constexpr (DUAL) {
                    if (__any(dual)) {  // the band [thr_seed, thr_band) of a dual seed pass: the heights below t2 that are not in the first set
                        const unsigned rest = dual ? ~gmask : 0u;
#pragma unroll
                        for (int k = 0; k < kPPT; ++k)
                            if ((rest >> k & 1u) & (cp.z[k] < pp.u.thr.t2) & (k_off<G>(k) < cp.rem)) m2.add(cp.x[k], cp.y[k], cp.z[k], scale, org);
                    }
                }
                if (any_wbits) {  // the round's set -> membership plane: the split of the patch if this round turns out to be its last
                    store_member<G>(frame_member, chunk_sel<G>(pts, c, use_hi, on), gmask, wbits);
                }
                if (c + 1u < nchunk_max) {
                    if (kPrefetch)
                        cp = nx;
                    else
                        load_chunk<G>(cp, pts, chunk_sel<G>(pts, c + 1u, use_hi, on));
                }
            }
            // the row's totals -> LDS.  64-lane rows: reduce-scatter of sixteen values, each stored by the lane it
            // ends up with; 16-lane rows: ten butterflies (four steps each; the selects of a scatter cost what its
            // fewer exchanges save: 0.868 -> 0.884 ms)
            // `same`: does the total this lane stores equal the one it replaces (the lanes that store nothing say yes)?
            auto row_totals = [&](const Moments &mm, long long (*dst)[MW], bool store, bool &same) {
                if constexpr (MW == 16) {
                    long long v[16];
                    if constexpr (G == 16) mm.to_row16(v);  // (wide grid, 16-lane rows: the sums as they stand, MomentsT<true>::to_row16)
                    else moments_to_16(mm, v);
                    int slot16;
                    const long long mine = Row<G>::reduce16_scatter(v, slot16);
                    if (store && j < 16) {
                        same = dst[q][slot16] == mine;
                        dst[q][slot16] = mine;
                    }
                } else {
                    long long v[10];
                    v[0] = Row<G>::sum_i64(mm.n);
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[1 + k] = Row<G>::sum_i64(mm.s1[k]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) v[4 + k] = Row<G>::sum_i64(mm.s2[k]);
                    if (store && j < 10) {  // lane j of the row stores moment j
                        long long mine = v[0];
#pragma unroll
                        for (int k = 1; k < 10; ++k) mine = j == k ? v[k] : mine;
                        same = dst[q][j] == mine;
                        dst[q][j] = mine;
                    }
                }
            };
            bool same = true, same2 = true;
            row_totals(m, sh.mom, on, same);
            if (DUAL && __any(dual)) row_totals(m2, sh.mom2, dual, same2);
            // early termination: every total of this round equals the round before's (flags bit 6: comparable) -> bit 7
            const bool conv = on && (pp.flags & 64) && Row<G>::ballot(!same) == 0ull;  // row-uniform
            if (conv && j == 0) sh.p[q].flags = pp.flags | 128;
            if (__any(clamped && (last || conv)) && ln == 0) flag_clamped(Bt, f);  // (only a FINAL ground set counts, pwpp_get_clamped_frames)
        }
        wave_lds_sync();
        PWPP_PHASE(3);

        // ---- D. solve phase: lane p fits patch p (ref :47-75)
        PWPP_SUB_BEGIN();
        long long cnt = 0;
        bool tiny = false;  // contract v3: a fit set of 1-3 points follows the reference's float arithmetic (tiny_fit_row)
        bool conv = false;  // early termination: this round's totals repeat the last round's, the plane in force is the final one
        if (O(kind) != ST_DONE) {
            const long long a0 = sh.mom[ln][0], b0 = DUAL ? sh.mom2[DUAL ? ln : 0][0] : 0;
            cnt = from_stash ? b0 : (dual_now ? (v_is_hi ? a0 + b0 : a0) : a0);
            tiny = cnt >= 1 && cnt <= 3;
            conv = ln < PW && (sh.p[ln < PW ? ln : 0].flags & 128) != 0 && cnt > 3;
        }
        const unsigned long long t_mask = __ballot(tiny);
        if (t_mask) {  // the rows gather the points of those patches again (the stage of phase B is still published)
            for (int sb = 0; sb < NSB; ++sb) {
                if (((t_mask >> (R * sb)) & ((1ull << R) - 1ull)) == 0ull) continue;
                const int q = R * sb + row;
                const bool trow = (t_mask >> q) & 1ull;
                const W64Patch pp = sh.p[q];
                float thr = pp.u.thr.t;
                if constexpr (DUAL) {
                    if ((pp.flags & 4) && v_is_hi) thr = pp.u.thr.t2;  // dual pass: this round's set is the R-VPF one
                }
                const PatchRef pts = patch_ref(Bt, fd, pp.off_lo, pp.n_lo, pp.off_hi, pp.n_hi);
                float mt[3], ct[6];
                tiny_fit_row<G>(pts, trow, pp.kind, thr, pp.nx, pp.ny, pp.nz, mt, ct);
                if (trow && j == 0) {  // mean and covariance take the place of the patch's moments 1..5 (the count stays)
                    float *dst = reinterpret_cast<float *>(&sh.mom[q][1]);
#pragma unroll
                    for (int k = 0; k < 3; ++k) dst[k] = mt[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) dst[3 + k] = ct[k];
                }
            }
            wave_lds_sync();
        }
        // (Round 6, measured and not adopted: the totals -> (mean, covariance) step of the 64-lane kernels spread over the idle lanes,
        // one output per lane -- 64-lane rows leave 56+ lanes idle in this phase.  Bit-exact, and SLOWER (k_fit_w64<64,4> 0.885 -> 0.897 ms):
        // the phase is a chain of dependent 64-bit operations whose LENGTH is the same for one output as for nine interleaved ones, plus
        // two hand-overs through LDS.  tools/fit_phases.py, profiles/r06_fit_phases.txt.)
        PWPP_SUB(1);
        if (O(kind) != ST_DONE) {
            float mean[3], c6[6];
            if (tiny) {
                const float *src = reinterpret_cast<const float *>(&sh.mom[ln][1]);
#pragma unroll
                for (int k = 0; k < 3; ++k) mean[k] = src[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = src[3 + k];
                if (dual_now) O(stash_valid) = 0;  // (its moments are gone: the R-GPF seed stage streams the patch itself)
            } else {
                long long tot[MW];
#pragma unroll
                for (int k = 0; k < MW; ++k) tot[k] = from_stash ? sh.mom2[DUAL ? ln : 0][k] : sh.mom[ln][k];
                if (dual_now) {  // mom = below the smaller threshold (A), mom2 = the band (B)
                    long long stash_cnt = 0;
#pragma unroll
                    for (int k = 0; k < MW; ++k) {
                        const long long a = tot[k], ab = a + sh.mom2[DUAL ? ln : 0][k];
                        tot[k] = v_is_hi ? ab : a;            // this round: the R-VPF seeds (th_seeds_v)
                        sh.mom2[DUAL ? ln : 0][k] = v_is_hi ? a : ab;    // stash: the R-GPF seeds (th_seeds)
                        if (k == 0) stash_cnt = v_is_hi ? a : ab;
                    }
                    O(stash_valid) = !(stash_cnt >= 1 && stash_cnt <= 3);  // (1-3 seeds: that stage gathers the points, it needs its own pass)
                }
                if (cnt > 0 && !conv) {
                    constexpr bool kRaw = MW == 16 && G == 16;  // (to_row16's values)
                    const long long s1[3] = {kRaw ? wide_first(tot[1], cnt) : tot[1], kRaw ? wide_first(tot[2], cnt) : tot[2], kRaw ? wide_first(tot[3], cnt) : tot[3]};
                    __int128 s2[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        s2[k] = kRaw ? wide_second(tot[4 + k], tot[MW == 16 ? 10 + k : 4 + k])
                                     : (MW == 16 ? join_halves(tot[4 + k], tot[MW == 16 ? 10 + k : 4 + k]) : (__int128)tot[4 + k]);
                    mean_cov_from_totals(cnt, s1, s2, P.fxp_shift, sh.p[ln].ox, sh.p[ln].oy, O(z0), mean, c6);
                }
            }
            PWPP_SUB(2);
            if (cnt > 0 && !conv) {  // empty set: the previous plane stays (ref :49); converged: the solve would return the plane in force
                PlaneFit npl;
                plane_from_mean_c6(mean, c6, Bt.debug, npl);
                PWPP_SUB(3);
                O(pl) = npl;
                O(fitted) = 1;
            }
            if (needs_previous_plane(P, O(kind), O(zone), O(fitted) != 0)) {
                mark_needs_previous_plane(Bt, f, Bt.recs + (size_t)f * P.num_bins + O(bin), O(n));
                O(kind) = ST_DONE;
            }
            O(cnt) = cnt;
        }
        PWPP_SUB(4);

        PWPP_PHASE(4);
        // ---- E. R-VPF strip (ref :489-505) for the zone-0 patches whose plane came out vertical
        const bool vertical = O(kind) == ST_VPF && (double)O(pl).nz < P.uprightness_thr;
        const unsigned long long v_mask = __ballot(vertical);
        if (v_mask) {
            if (ln < PW) {
                const PlaneFit pl = O(pl);
                sh.p[ln].nx = pl.nx;
                sh.p[ln].ny = pl.ny;
                sh.p[ln].nz = pl.nz;
                if (vertical) {
                    const StripBand band = strip_band(pl.d, P.th_dist_v);
                    sh.p[ln].u.thr.t = band.lo;
                    sh.p[ln].u.thr.t2 = band.hi;
                }
                sh.p[ln].flags = (sh.p[ln].flags & 0xef) | (O(it) << 8);  // nothing removed yet; the round
            }
            wave_lds_sync();
            for (int sb = 0; sb < NSB; ++sb) {
                if (((v_mask >> (R * sb)) & ((1ull << R) - 1ull)) == 0ull) continue;
                const int q = R * sb + row;
                const bool vrow = (v_mask >> q) & 1ull;
                const W64Patch pp = sh.p[q];
                StripBand band;
                band.lo = pp.u.thr.t;
                band.hi = pp.u.thr.t2;
                const PatchRef pts = patch_ref(Bt, fd, pp.off_lo, pp.n_lo, pp.off_hi, pp.n_hi);
                const int vpf_round = (pp.flags >> 8) & 0xff;
                const unsigned nchunk_max = wave_max_u32(vrow ? patch_chunks<G>(pts, true) : 0u);
                bool any = false;
                for (unsigned c = 0; c < nchunk_max; ++c) {
                    ChunkPts cp;
                    const PartSel sel = chunk_sel<G>(pts, c, true, vrow);
                    load_chunk<G>(cp, pts, sel);
                    const unsigned hit = lane_strip<G>(cp, vrow, pp.nx, pp.ny, pp.nz, band);
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) {
                        if (hit >> k & 1u) {
                            strip_point(pts, chunk_slot<G>(sel, k, (unsigned)j), vpf_round);
                        }
                    }
                    any = any || hit != 0;
                }
                if (Row<G>::ballot(any) != 0ull && j == 0) sh.p[q].flags = pp.flags | 16;
            }
            wave_lds_sync();
            if (vertical && (sh.p[ln].flags & 16)) {  // the working set changed
                O(lpr_valid) = 0;
                O(stash_valid) = 0;
            }
        }

        PWPP_PHASE(5);
        // ---- what comes next for the patch of this lane
        const int kind = O(kind);
        if (kind == ST_VPF) {
            const int it = O(it) + 1;
            O(it) = it;
            if (!vertical || it >= P.num_iter) {  // ref :506 / loop end
                O(kind) = ST_SEED;
                O(it) = 0;
            }
        } else if (kind == ST_SEED) {
            O(kind) = (O(cnt) == 0 && P.enable_RVPF != 0 && O(zone) != 0) ? ST_LAZY : ST_ITER;
        } else if (kind == ST_LAZY) {
            O(kind) = ST_ITER;
        } else if (kind == ST_ITER) {
            const int it = O(it);
            if (it == P.num_iter - 1 || conv) {
                write_record(Bt.recs + (size_t)f * P.num_bins + O(bin), O(pl), O(n), (unsigned)O(cnt), O(hi_skipped) != 0, G == 64 ? 6 : (G == 32 ? 5 : (G == 16 ? 4 : 3)), it + 1);
                O(kind) = ST_DONE;
            }
            O(it) = it + 1;
        }
        if (OWN_LDS) wave_lds_sync();  // (the owner state is LDS: ordered like every other hand-over between the phases)
        PWPP_PHASE(6);
    }
#undef PWPP_PHASE
#undef PWPP_SUB
#undef PWPP_SUB_BEGIN
#undef O
}

struct FitShared {
    long long part[kWaves][22];
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
    long long last_n;  // points of the set reduce_and_fit saw last
};

// per-lane sums of the workgroup-per-patch kernel: it takes patches of any size, so the second moments are
// kept in 128 bits (a lane of the other kernels never sees more than 2047 points, see Moments)
template <bool WIDE>
struct MomentsWide {
    long long n, s1[3];
    __int128 s2[6];
    __device__ __forceinline__ void clear() {
        n = 0;
        s1[0] = s1[1] = s1[2] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s2[k] = 0;
    }
    __device__ __forceinline__ void add(float x, float y, float z, double scale, const FxpOrg &o) {
        const float zc = __builtin_amdgcn_fmed3f(z, o.zlo, o.zhi);
        const long long qx = WIDE ? fxp_q_wide(x, scale, o.cx) : (long long)fxp_q(x, scale, o.cx);
        const long long qy = WIDE ? fxp_q_wide(y, scale, o.cy) : (long long)fxp_q(y, scale, o.cy);
        const long long qz = WIDE ? fxp_q_wide(zc, scale, o.cz) : (long long)fxp_q(zc, scale, o.cz);
        n += 1;
        s1[0] += qx;
        s1[1] += qy;
        s1[2] += qz;
        // (36-bit values on the wide grid: the products need 128 bits; this kernel is the rare path, nothing here is tuned)
        s2[0] += (__int128)qx * (__int128)qx;
        s2[1] += (__int128)qx * (__int128)qy;
        s2[2] += (__int128)qx * (__int128)qz;
        s2[3] += (__int128)qy * (__int128)qy;
        s2[4] += (__int128)qy * (__int128)qz;
        s2[5] += (__int128)qz * (__int128)qz;
    }
};

// Block-wide sum of the moments and, if the set is non-empty, the plane of ref :47-75.
// An empty set leaves the previous plane in force, as ref :49 does.  The 128-bit second moments are
// added up as three limbs (32 + 32 + 64 bits) and recombined: exact at any size.
// `pts`, `iter`, `thr`: the set's membership test once more (R-GPF round: distance to the plane still in `sh` below th_dist = thr;
// seed stages: z < thr) for the sets of 1-3 points, which follow the reference's float arithmetic (tiny_fit_row).
template <class M>
__device__ void reduce_and_fit(FitShared &sh, const M &m, int shift, float ox, float oy, float z0, const PatchRef &pts, bool iter,
                               double thr, int debug = 0) {
    long long v[22];
    v[0] = m.n;
    v[1] = m.s1[0];
    v[2] = m.s1[1];
    v[3] = m.s1[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned long long lo = (unsigned long long)m.s2[k];
        v[4 + k] = (long long)(lo & 0xffffffffull);
        v[10 + k] = (long long)(lo >> 32);
        v[16 + k] = (long long)(m.s2[k] >> 64);
    }
    const int wv = wave_id(), ln = lane_id();
#pragma unroll
    for (int k = 0; k < 22; ++k) {
        const long long t = wave_sum_i64(v[k]);
        if (ln == 0) sh.part[wv][k] = t;
    }
    __syncthreads();
    if (wv == 0) {
        long long t[22];
#pragma unroll
        for (int k = 0; k < 22; ++k) {
            t[k] = 0;
#pragma unroll
            for (int q = 0; q < kWaves; ++q) t[k] += sh.part[q][k];
        }
        const long long n = t[0];
        if (ln == 0) sh.last_n = n;
        if (n > 0) {
            __int128 s2[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) s2[k] = ((__int128)t[16 + k] << 64) + ((__int128)t[10 + k] << 32) + (__int128)t[4 + k];
            const long long s1[3] = {t[1], t[2], t[3]};
            PlaneFit pf;
            float mean[3], c6[6];
            if (n <= 3)
                tiny_fit_row<64>(pts, true, iter ? ST_ITER : ST_SEED, stage_threshold(iter ? ST_ITER : ST_SEED, sh.d, thr, thr), sh.normal[0],
                                 sh.normal[1], sh.normal[2], mean, c6);
            else
                mean_cov_from_totals(n, s1, s2, shift, ox, oy, z0, mean, c6);
            plane_from_mean_c6(mean, c6, debug, pf);
            if (ln == 0) {
                sh.normal[0] = pf.nx;
                sh.normal[1] = pf.ny;
                sh.normal[2] = pf.nz;
                sh.mean[0] = pf.mean[0];
                sh.mean[1] = pf.mean[1];
                sh.mean[2] = pf.mean[2];
                sh.sv[0] = pf.sv[0];
                sh.sv[1] = pf.sv[1];
                sh.sv[2] = pf.sv[2];
                sh.d = pf.d;
            }
        }
    }
    __syncthreads();
}

// Lowest-point representative height, ref :84-103, without sorting the bin: the reference
// needs (a) how many points lie below the adaptive cut-off (zone 0 only, :88-96), (b) the
// num_lpr smallest z among the others, summed in ascending order in double (:99-102).
// A 4-pass 8-bit radix select finds the k-th smallest key; the elements below its 24-bit
// prefix are gathered in the last pass, the rest is implied by the last histogram.
// (reads both parts of the bin: this is the rare exact path, and k_fit_stream's)
__device__ double block_lpr(FitShared &sh, const PatchRef &pts, bool use_cutoff, double cutoff, int num_lpr) {
    const int ln = lane_id(), wv = wave_id();
    unsigned prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int bits = 24 - 8 * pass;
        sh.hist[threadIdx.x] = 0;  // kBlock == 256 counters
        if (threadIdx.x == 0 && pass == 3) sh.sel_count = 0;
        __syncthreads();
        for (unsigned i = threadIdx.x; i < pts.n_lo + pts.n_hi; i += kBlock) {
            const float pz = pts.z[patch_slot(pts, i)];
            if (z_stripped(pz)) continue;
            if (use_cutoff && (double)pz < cutoff) continue;  // init_idx prefix, ref :88-96
            const unsigned key = z_key(pz);
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
        for (unsigned me = (unsigned)ln; me < c; me += 64u) {  // (a key per lane and round: its rank among the gathered keys)
            const unsigned mine = sh.sel_keys[me];
            unsigned rank = 0;
            for (unsigned j = 0; j < c; ++j) {
                const unsigned o = sh.sel_keys[j];
                rank += (o < mine || (o == mine && j < me)) ? 1u : 0u;
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

// ------------------------------------------------------------------------------------------
// k_fit_brows: FOUR waves per patch -- the latency flavour of k_fit_srows<64>.  With a handful of
// frames in flight the run time of the fit stage is the chain of its largest patch (~5000 points:
// 10 chunks per pass, 6 passes, 5 solves on one wave = 118 us per KITTI frame).  Here the four
// waves of a workgroup stream every fourth chunk of the same patch (next chunk prefetched), add
// their integer moments through LDS (exact, so the split over waves changes nothing) and all
// solve the same 3x3 problem; the lowest points are selected per wave and merged.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ long long readlane_i64(long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), lane);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

struct BRowShared {
    long long mom[kWaves][16];   // per wave: n, S1[3], lower and upper halves of S2[6] (Row<64>::reduce16_scatter)
    unsigned cand[kWaves][5][64];  // every lane's four smallest keys + the smallest it dropped, pooled lane by lane
    unsigned dropped[kWaves];
    int elig[kWaves];
    double single_sum;   // a patch of one chunk: wave 0's result
    unsigned single_T;
    long long prev_tot[2][16];   // the totals of the last R-GPF round (early termination), double-buffered by round parity
    long long mom2[kWaves][16];  // dual seed pass: the band between the two seed thresholds
    PlaneFit plane[2];           // R-VPF fit | R-GPF seed fit, solved side by side by different waves
    FitShared fs;  // for block_lpr, the exact fall-back of the lowest-point selection
};

// LPR (ref :84-103) with the points of the patch dealt out to the four waves chunk by chunk.  `use_hi`: read the high
// part too; `total` = eligible points seen (the caller reads the low part alone first and comes back with the high part
// if it held fewer than num_lpr: see srow_lpr).
__device__ double brow_lpr(BRowShared &sh, const PatchRef &pts, bool use_hi, bool use_cutoff, double cutoff, int num_lpr, int &total_out,
                           int force = 0) {
    const unsigned INF = 0xFFFFFFFFu;
    const int wv = wave_id(), ln = lane_id();
    const unsigned nchunk = patch_chunks<64>(pts, use_hi);
    unsigned k0 = INF, k1 = INF, k2 = INF, k3 = INF, dropped = INF;
    int elig = 0;
    ChunkZ cp;
    load_chunk_z<64>(cp, pts, chunk_sel<64>(pts, (unsigned)wv, use_hi));
    for (unsigned c = (unsigned)wv; c < nchunk; c += kWaves) {
        ChunkZ nx;  // the next chunk is in flight while this one is ranked (this is the first, cold touch of the patch)
        load_chunk_z<64>(nx, pts, chunk_sel<64>(pts, c + kWaves, use_hi, c + kWaves < nchunk));
        const unsigned act = chunk_act_z<64>(cp);
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const bool e = (act >> k & 1u) && !(use_cutoff && (double)cp.z[k] < cutoff);
            unsigned x = e ? z_key(cp.z[k]) : INF;
            ce(k0, x);
            ce(k1, x);
            ce(k2, x);
            ce(k3, x);
            dropped = x < dropped ? x : dropped;
            elig += e ? 1 : 0;
        }
        cp = nx;
    }
    // Every lane kept its four smallest keys (and the smallest one it dropped).  The waves pool them lane by
    // lane: lane j of EVERY wave merges the sixteen keys the four lanes j hold into its four smallest (what
    // falls out joins the dropped ones), then all waves extract the keff smallest of the pool the same way,
    // adding them up in ascending order as the reference does (ref :99-101) -- no cross-wave list merge, and
    // the result is in every thread.  A patch of one chunk lives in wave 0 alone and skips the pooling.
    const int total_w = Row<64>::sum_i32(elig);
    const bool single = nchunk <= 1u;
    int total = total_w;
    if (!single) {
        sh.cand[wv][0][ln] = k0;
        sh.cand[wv][1][ln] = k1;
        sh.cand[wv][2][ln] = k2;
        sh.cand[wv][3][ln] = k3;
        sh.cand[wv][4][ln] = dropped;
        if (ln == 0) sh.elig[wv] = total_w;
        __syncthreads();
        k0 = k1 = k2 = k3 = dropped = INF;
        total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            total += sh.elig[w];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned x = sh.cand[w][q][ln];
                ce(k0, x);
                ce(k1, x);
                ce(k2, x);
                ce(k3, x);
                dropped = x < dropped ? x : dropped;
            }
            const unsigned d = sh.cand[w][4][ln];
            dropped = d < dropped ? d : dropped;
        }
        __syncthreads();  // the pool is free again
    } else {
        total = __builtin_amdgcn_readfirstlane(total_w);
        if (wv == 0 && ln == 0) sh.elig[0] = total_w;
        __syncthreads();
        total = sh.elig[0];  // (waves 1-3 hold nothing: they follow wave 0's count and extract INF keys; only wave 0's result is used)
    }
    total_out = total;
    const int keff = total < num_lpr ? total : num_lpr;
    double sum = 0.0;
    unsigned T = 0;
    // Round 6 (as in srow_lpr): the 32 lowest of the wave's 256 kept keys by a merge network instead of keff extraction rounds with a
    // wave reduction each (20 x ~250 cycles = 2 us of the single-frame chain per lowest-point selection), their sum order-free where
    // that is exact; otherwise the extraction loop on the untouched registers.  (A patch of one chunk lives in wave 0: the other waves
    // hold no keys, extract nothing and take wave 0's result below.)
    bool quick = false;
    const int kx = (single && wv != 0) ? 0 : keff;
    if (kx > 0 && kx <= 32) {
        unsigned s0 = k0, s1 = k1, s2 = k2, s3 = k3;
        row64_lowest32(s0, s1, s2, s3);
        const unsigned kk[4] = {s0, s1, s2, s3};
        double part = 0.0;
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool sel = ln < 8 && 4 * ln + r < kx;
            const float zv = key_z(kk[r]);
            const unsigned e = (__float_as_uint(zv) >> 23) & 0xffu;
            ok = ok && (!sel || zv == 0.0f || (e >= 115u && e < 135u));
            part += sel ? (double)zv : 0.0;
        }
        quick = __ballot(!ok) == 0ull;
        if (quick) {
            const double s = row16_sum_f64(part);
            sum = __longlong_as_double(readlane_i64(__double_as_longlong(s), 0));
            const int last = kx - 1;
            const unsigned pick = (last & 3) == 0 ? s0 : ((last & 3) == 1 ? s1 : ((last & 3) == 2 ? s2 : s3));
            T = (unsigned)__builtin_amdgcn_readlane((int)pick, last >> 2);
        }
    }
    for (int r = 0; r < (quick ? 0 : kx); ++r) {  // (wave-uniform trip count; no barrier inside)
        const unsigned m = Row<64>::min_u32(k0);
        sum += (double)key_z(m);
        T = m;
        const int lowest = __ffsll((long long)__ballot(k0 == m)) - 1;
        if (ln == lowest) {
            k0 = k1;
            k1 = k2;
            k2 = k3;
            k3 = INF;
        }
    }
    unsigned dall = Row<64>::min_u32(dropped);
    if (single) {  // wave 0's result for everybody
        if (wv == 0 && ln == 0) {
            sh.single_sum = sum;
            sh.single_T = T;
            sh.dropped[0] = dall;
        }
        __syncthreads();
        sum = sh.single_sum;
        T = sh.single_T;
        dall = sh.dropped[0];
        __syncthreads();
    }
    if (keff > 0 && (dall < T || force != 0)) return block_lpr(sh.fs, pts, use_cutoff, cutoff, num_lpr);  // a lane (pool) held > 4 of the lowest: exact path
    return keff ? sum / (double)keff : 0.0;  // ref :103
}

template <bool WIDE>
__device__ __forceinline__ void fit_brows_body(BRowShared &sh, const PwppBatch &Bt, int b_lo, int b_hi, unsigned by /* block index among the patch blocks */) {
    typedef MomentsT<WIDE> Moments;
    const int f = blockIdx.x;
    const PwppDevParams &P = Bt.P;
    const uint32_t *cs = Bt.cls_start + (size_t)f * PWPP_CLS_STRIDE;
    const unsigned cbeg = cs[b_lo], cend = cs[b_hi];
    if (cbeg + by >= cend) return;                    // workgroup-uniform
    const unsigned slot = cend - 1u - by;             // largest patches first
    const int wv = wave_id(), ln = lane_id();
    const PatchCtx pc = patch_ctx(Bt, f, slot, true);
    const int bin = pc.bin, zone = pc.zone;
    const unsigned n = pc.n;  // (at most 2047 points per lane, see Moments; pwpp_launch_fit sends larger patches to k_fit_stream)
    const PwppFrameDesc fd = Bt.frames[f];
    const PatchRef pts = patch_ref(Bt, fd, pc);
    uint8_t *frame_member = Bt.member + fd.mbase;
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const double cutoff = P.margin * sensor_height;
    const float zs = hi_split_z(P, sensor_height);
    const float4 bb = Bt.bin_bbox[bin];
    const bool use_cutoff = zone == 0;
    const double scale = (double)(1 << P.fxp_shift);

    PlaneFit pl;
    plane_clear(pl);
    double lpr = 0.0;
    bool lpr_valid = false, z0_set = false;
    float z0 = 0.0f;
    FxpOrg org = fxp_org(pc.ox, pc.oy, 0.0f, scale, P.fxp_zr);
    int kind = (P.enable_RVPF != 0 && zone == 0) ? ST_VPF : ST_SEED;  // everything below is workgroup-uniform
    int it = 0;
    // Dual seed pass as in k_fit_w64 (an R-VPF round and the R-GPF seed stage pick their seeds from the
    // same working set with two thresholds: the pass sums the points below the smaller one and the band
    // up to the larger one), and because this kernel is a latency chain with four SIMDs to itself, the two
    // planes are also SOLVED side by side: waves 0-1 fit the R-VPF seeds, waves 2-3 the R-GPF seeds.  If
    // the R-VPF round removes nothing, the R-GPF seed stage is already done: no pass, no solve.
    const bool v_is_hi = P.th_seeds_v >= P.th_seeds;
    bool stash_valid = false;
    long long stash_cnt = 0;
    bool fitted = false;  // a plane of this patch's own exists
    PlaneFit pl_seed = pl;
    // timing probes (PWPP_DEBUG_FLAGS & 4): the chain of the largest patch of frame 0, (code << 56) | 100 MHz ticks
    // (PWPP_DEBUG_FLAGS = 4 | size << 16 follows the patch of that size instead of the largest one)
    const bool probing = (Bt.debug & 4) && blockIdx.x == 0 && threadIdx.x == 0 && ((Bt.debug >> 16) ? n == (unsigned)(Bt.debug >> 16) : by == 0u);
    if ((Bt.debug & 4) && threadIdx.x == 0) atomicMin(&Bt.dbg[60], wall_clock64());
    int probe_i = 0;
    auto probe = [&](unsigned long long code) {
        if (probing && probe_i < 62) Bt.dbg[probe_i++] = (code << 56) | (wall_clock64() & 0x00FFFFFFFFFFFFFFull);
    };
    probe(1);

    for (int guard = 0; guard < 4 * P.num_iter + 8 && kind != ST_DONE; ++guard) {
        if (kind == ST_SEED && stash_valid) {  // ref :513-517 on the unchanged working set: solved above
            if (stash_cnt > 0) pl = pl_seed;   // an empty seed set leaves the R-VPF plane in place (ref :49)
            fitted = fitted || stash_cnt > 0;  // (an R-VPF round came first: had it been empty the patch would have stopped there)
            stash_valid = false;
            kind = ST_ITER;                    // (zone 0: never ST_LAZY)
            continue;
        }
        if ((kind == ST_VPF || kind == ST_SEED) && !lpr_valid) {
            int eligible = 0;
            for (int both = 0; both < 2; ++both) {  // the low part alone, unless it holds fewer than num_lpr eligible points (srow_lpr)
                lpr = brow_lpr(sh, pts, both != 0, use_cutoff, cutoff, P.num_lpr, eligible, (Bt.debug >> 14) & 3);
                if (eligible >= P.num_lpr || pc.n_hi == 0u) break;
            }
            lpr_valid = true;
            if (!z0_set) {  // the z origin of this patch's sums: its first lowest-point representative
                z0 = fxp_z_origin(lpr);
                z0_set = true;
                org = fxp_org(pc.ox, pc.oy, z0, scale, P.fxp_zr);
            }
            probe(2);
        }
        const bool dual_now = kind == ST_VPF;
        const double thr_seed = lpr + (dual_now ? (v_is_hi ? P.th_seeds : P.th_seeds_v) : (kind == ST_LAZY ? P.th_seeds_v : P.th_seeds));
        const double thr_band = lpr + (v_is_hi ? P.th_seeds_v : P.th_seeds);
        const bool last = kind == ST_ITER && it == P.num_iter - 1;
        const bool use_hi = stage_needs_hi(kind, pc.n_hi, dual_now && thr_band > thr_seed ? thr_band : thr_seed, P.th_dist, pl, bb, zs);
        const unsigned nchunk = patch_chunks<64>(pts, use_hi);
        const float T = stage_threshold(kind, pl.d, P.th_dist, thr_seed);  // the pass's tests in float (lane_stage_accum)
        const float T_band = dual_now ? plane_test_threshold(0.0, thr_band) : 0.0f;
        Moments m, m2;
        m.clear();
        m2.clear();
        // (early termination and the membership plane: see k_fit_w64)
        const bool wbits = kind == ST_ITER && (last || it >= 1);
        bool clamped = false;
        ChunkPts cp;
        load_chunk<64>(cp, pts, chunk_sel<64>(pts, (unsigned)wv, use_hi));
        for (unsigned c = (unsigned)wv; c < nchunk; c += kWaves) {
            ChunkPts nx;  // this wave's next chunk is in flight while this one is accumulated
            load_chunk<64>(nx, pts, chunk_sel<64>(pts, c + kWaves, use_hi));
            bool hit = false;
            const unsigned gmask = lane_stage_accum<64>(cp, kind, T, pl.nx, pl.ny, pl.nz, scale, org, m, hit);
            clamped = clamped || (wbits && hit);
            if (dual_now) {  // the band [thr_seed, thr_band)
                const unsigned rest = ~gmask;
#pragma unroll
                for (int k = 0; k < kPPT; ++k)
                    if ((rest >> k & 1u) & (cp.z[k] < T_band) & (k_off<64>(k) < cp.rem)) m2.add(cp.x[k], cp.y[k], cp.z[k], scale, org);
            }
            if (wbits) {  // the round's set -> membership plane (every wave its own chunks)
                store_member<64>(frame_member, chunk_sel<64>(pts, c, use_hi), gmask, true);
            }
            cp = nx;
        }
        probe(3);
        {   // the wave's sums -> LDS (reduce-scatter: each of the sixteen values is stored by the lane it ends up with)
            auto wave_sums = [&](const Moments &mm, long long (*dst)[16]) {
                long long v[16];
                moments_to_16(mm, v);
                int slot16;
                const long long mine = Row<64>::reduce16_scatter(v, slot16);
                if (ln < 16) dst[wv][slot16] = mine;
            };
            wave_sums(m, sh.mom);
            if (dual_now) wave_sums(m2, sh.mom2);
        }
        __syncthreads();
        const bool spec = dual_now && wv >= kWaves / 2;  // this wave solves the R-GPF seed fit
        long long tot[4];
        __int128 s2[6];
        long long cnt = 0;
        bool conv = false;  // this R-GPF round's totals repeat the last round's: the plane in force is the final one
        {   // lane k (mod 16) adds up slot k of the waves' sums; the sixteen totals then go to every lane through v_readlane
            // (scalar registers: the solve's integer part below is scalar work) -- 4 LDS reads per lane instead of 64
            const int k16 = ln & 15;
            long long a = 0, b = 0;  // a: below the smaller threshold, a + b: below the larger one
#pragma unroll
            for (int w2 = 0; w2 < kWaves; ++w2) {
                a += sh.mom[w2][k16];
                if (dual_now) b += sh.mom2[w2][k16];
            }
            const long long t_vpf = dual_now ? (v_is_hi ? a + b : a) : a;
            const long long t_seed = v_is_hi ? a : a + b;
            const long long mine = spec ? t_seed : t_vpf;
            cnt = readlane_i64(t_vpf, 0);
            if (dual_now) stash_cnt = readlane_i64(t_seed, 0);
            long long t16[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t16[k] = readlane_i64(mine, k);
#pragma unroll
            for (int k = 0; k < 4; ++k) tot[k] = t16[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) s2[k] = join_halves(t16[4 + k], t16[10 + k]);
            if (kind == ST_ITER) {  // (prev_tot is double-buffered by round: no barrier between its read and its write)
                const bool differs = sh.prev_tot[(it & 1) ^ 1][k16] != mine;
                conv = !last && it >= 1 && cnt > 3 && __ballot(differs) == 0ull;
                if (wv == 0 && ln < 16) sh.prev_tot[it & 1][ln] = mine;
            }
        }
        if (clamped && (last || conv)) flag_clamped(Bt, f);  // (only a FINAL ground set counts)
        __syncthreads();
        probe(4);
        PlaneFit fitted_pl = pl;
        if (tot[0] > 0 && !conv) {  // empty: ref :49; converged: the solve would return the plane in force
            float mean[3], c6[6];
            if (tot[0] <= 3) {  // contract v3 (wave-uniform): this wave gathers the 1-3 points of its set itself
                const float thr_t = dual_now ? ((spec == v_is_hi) ? T : T_band) : T;
                tiny_fit_row<64>(pts, true, kind, thr_t, pl.nx, pl.ny, pl.nz, mean, c6);
            } else {
                const long long s1[3] = {tot[1], tot[2], tot[3]};
                mean_cov_from_totals_uniform(tot[0], s1, s2, P.fxp_shift, pc.ox, pc.oy, z0, mean, c6);  // (totals are the same in every lane of this wave)
            }
            plane_from_mean_c6(mean, c6, Bt.debug, fitted_pl);
        }
        if (dual_now) {
            if (ln == 0 && (wv == 0 || wv == kWaves / 2)) sh.plane[wv ? 1 : 0] = fitted_pl;
            __syncthreads();
            pl = sh.plane[0];
            pl_seed = sh.plane[1];
            stash_valid = true;
        } else {
            pl = fitted_pl;
        }
        fitted = fitted || cnt > 0;
        if (needs_previous_plane(P, kind, zone, fitted)) {  // (workgroup-uniform)
            if (threadIdx.x == 0) mark_needs_previous_plane(Bt, f, Bt.recs + (size_t)f * P.num_bins + bin, n);
            kind = ST_DONE;
            continue;
        }
        probe(5);
        if (kind == ST_VPF) {
            const bool vertical = (double)pl.nz < P.uprightness_thr;  // ref :489
            if (vertical) {
                int any = 0;
                const unsigned nstrip = patch_chunks<64>(pts, true);
                const StripBand band = strip_band(pl.d, P.th_dist_v);
                for (unsigned c = (unsigned)wv; c < nstrip; c += kWaves) {
                    ChunkPts cs2;
                    const PartSel sel = chunk_sel<64>(pts, c, true);
                    load_chunk<64>(cs2, pts, sel);
                    const unsigned hit = lane_strip<64>(cs2, true, pl.nx, pl.ny, pl.nz, band);
#pragma unroll
                    for (int k = 0; k < kPPT; ++k)
                        if (hit >> k & 1u) strip_point(pts, chunk_slot<64>(sel, k, (unsigned)ln), it);
                    any |= hit != 0u;
                }
                if (__syncthreads_or(any)) {  // the working set changed (and the marks are visible)
                    lpr_valid = false;
                    stash_valid = false;
                }
            }
            ++it;
            if (!vertical || it >= P.num_iter) {
                kind = ST_SEED;
                it = 0;
            }
        } else if (kind == ST_SEED) {
            kind = (cnt == 0 && P.enable_RVPF != 0 && zone != 0) ? ST_LAZY : ST_ITER;
        } else if (kind == ST_LAZY) {
            kind = ST_ITER;
        } else if (kind == ST_ITER) {
            if (last || conv) {
                if (threadIdx.x == 0) write_record(Bt.recs + (size_t)f * P.num_bins + bin, pl, n, (unsigned)cnt, !use_hi && pc.n_hi > 0u, 6, it + 1);
                kind = ST_DONE;
            }
            ++it;
        }
        probe(6);
    }
    if (probing) Bt.dbg[62] = n;
    if ((Bt.debug & 4) && threadIdx.x == 0) atomicMax(&Bt.dbg[61], (wall_clock64() << 24) | (unsigned long long)n);
}

template <bool WIDE>
__global__ __launch_bounds__(kBlock, 2) void k_fit_brows(PwppBatch Bt, int b_lo, int b_hi) {
    __shared__ BRowShared sh;
    fit_brows_body<WIDE>(sh, Bt, b_lo, b_hi, blockIdx.y);
}

// k_fit_hybrid: the single-frame kernel.  A SIMD retires one instruction every ~5 cycles whoever it
// belongs to, so two workgroups sharing a CU stretch each other's solve chains (measured: 5.2 -> 9-12 us
// per solve for the ~45 patches that had to double up when every patch had its own workgroup: a frame has
// ~300 patches, the chip 256 CUs).  Here only the patches above `b_mid` get four waves (k_fit_brows'
// body); the small ones, which fit one chunk of one wave anyway, go four to a workgroup, one wave each
// (k_fit_srows<64>'s body) -- ~110 workgroups per frame, every one alone on its CU, in ONE launch.
template <bool WIDE>
__global__ __launch_bounds__(kBlock, 1) void k_fit_hybrid(PwppBatch Bt, int b_mid, int b_hi, unsigned nb_big) {
    __shared__ BRowShared sh;
    if (blockIdx.y < nb_big)
        fit_brows_body<WIDE>(sh, Bt, b_mid, b_hi, blockIdx.y);
    else
        fit_srows_body<64, WIDE>(Bt, 0, b_mid, blockIdx.y - nb_big);
}

// the whole fit chain of one patch of any size by one workgroup (k_fit_stream: what exceeds the plan's classes;
// k_fit_fixup: a patch that starts from the plane fitted before it, `init`)
template <bool WIDE>
__device__ __forceinline__ void fit_stream_patch(FitShared &sh, const PwppBatch &Bt, int f, const PatchCtx &pc, const PwppPlaneState *init) {
    const PwppDevParams &P = Bt.P;
    const int bin = pc.bin, zone = pc.zone;
    const unsigned n = pc.n;
    PwppPatchRec *rec = Bt.recs + (size_t)f * P.num_bins + bin;
    const PwppFrameDesc fd = Bt.frames[f];
    const PatchRef pts = patch_ref(Bt, fd, pc);  // (always both parts: a patch this large is rare)
    uint8_t *frame_member = Bt.member + fd.mbase;
    const double sensor_height = fd.state_in >= 0 ? Bt.st_scalar[fd.state_in].sensor_height : P.sensor_height;
    const double cutoff = P.margin * sensor_height;  // ref :90
    const bool use_cutoff = zone == 0;
    const double scale = (double)(1 << P.fxp_shift);

    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) {
            sh.normal[k] = init ? init->normal[k] : 0.0f;
            sh.mean[k] = init ? init->mean[k] : 0.0f;
            sh.sv[k] = init ? init->sv[k] : 0.0f;
        }
        sh.d = init ? init->d : 0.0;
    }
    __syncthreads();
    bool fitted = init != nullptr;  // (with the plane fitted before this patch in hand nothing is missing)

    double lpr = 0.0;
    bool lpr_valid = false, z0_set = false;
    float z0 = 0.0f;
    FxpOrg org = fxp_org(pc.ox, pc.oy, 0.0f, scale, P.fxp_zr);
    auto new_lpr = [&]() {
        lpr = block_lpr(sh, pts, use_cutoff, cutoff, P.num_lpr);
        lpr_valid = true;
        if (!z0_set) {  // the z origin of this patch's sums: its first lowest-point representative
            z0 = fxp_z_origin(lpr);
            z0_set = true;
            org = fxp_org(pc.ox, pc.oy, z0, scale, P.fxp_zr);
        }
    };
    MomentsWide<WIDE> m;

    // ---- R-VPF, ref :482-508
    if (P.enable_RVPF) {
        for (int it = 0; it < P.num_iter; ++it) {
            if (!lpr_valid) new_lpr();
            const double thr = lpr + P.th_seeds_v;  // ref :108
            m.clear();
            for (unsigned i = threadIdx.x; i < n; i += kBlock) {
                const unsigned sl = patch_slot(pts, i);
                const float z = pts.z[sl];
                if (!z_stripped(z) && (double)z < thr) {
                    const float2 xy = pts.xy[sl];
                    m.add(xy.x, xy.y, z, scale, org);
                }
            }
            reduce_and_fit(sh, m, P.fxp_shift, pc.ox, pc.oy, z0, pts, false, thr, Bt.debug);
            fitted = fitted || sh.last_n > 0;
            if (!fitted && zone == 0) {  // the verticality test below would consult the plane fitted before this patch
                if (threadIdx.x == 0) mark_needs_previous_plane(Bt, f, rec, n);
                return;
            }
            const float nx = sh.normal[0], ny = sh.normal[1], nz = sh.normal[2];
            const double d = sh.d;
            if (zone == 0 && (double)nz < P.uprightness_thr) {  // ref :489
                int any = 0;
                for (unsigned i = threadIdx.x; i < n; i += kBlock) {
                    const unsigned sl = patch_slot(pts, i);
                    const float z = pts.z[sl];
                    if (z_stripped(z)) continue;
                    const float2 xy = pts.xy[sl];
                    const double dist = plane_dist(nx, ny, nz, d, xy.x, xy.y, z);
                    if (fabs(dist) < P.th_dist_v) {  // ref :499 -> non_ground_dst
                        strip_point(pts, sl, it);
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
    if (!lpr_valid) new_lpr();
    {
        const double thr = lpr + P.th_seeds;  // ref :145
        m.clear();
        for (unsigned i = threadIdx.x; i < n; i += kBlock) {
            const unsigned sl = patch_slot(pts, i);
            const float z = pts.z[sl];
            if (!z_stripped(z) && (double)z < thr) {
                const float2 xy = pts.xy[sl];
                m.add(xy.x, xy.y, z, scale, org);
            }
        }
        reduce_and_fit(sh, m, P.fxp_shift, pc.ox, pc.oy, z0, pts, false, thr, Bt.debug);
        fitted = fitted || sh.last_n > 0;
    }
    if (!fitted) {  // the first R-GPF round would measure distances to the plane fitted before this patch
        if (threadIdx.x == 0) mark_needs_previous_plane(Bt, f, rec, n);
        return;
    }
    const int ln = lane_id();
    // The split goes to the membership plane in the layout of a 64-lane fit row (pwpp_dev.h: byte c * 64 + j, bit k = point
    // c * 512 + (k / 4) * 256 + 4 j + k % 4 of the part).  This kernel walks a patch point by point, so the last round sets single
    // bits (atomicOr on the word around the byte) in areas zeroed first -- slow, and irrelevant: the path is rare.
    const unsigned mo[2] = {member_offset(pts, pc.off_lo, PWPP_PART_LO(bin)), member_offset(pts, pc.off_hi, PWPP_PART_HI(bin))};
    const unsigned mwords[2] = {((pc.n_lo + 511u) >> 9) * 16u, ((pc.n_hi + 511u) >> 9) * 16u};
    for (int h = 0; h < 2; ++h)
        for (unsigned wd = threadIdx.x; wd < mwords[h]; wd += kBlock) reinterpret_cast<uint32_t *>(frame_member + mo[h])[wd] = 0u;
    __threadfence_block();
    __syncthreads();
    for (int it = 0; it < P.num_iter; ++it) {
        const bool last = it == P.num_iter - 1;
        const float nx = sh.normal[0], ny = sh.normal[1], nz = sh.normal[2];
        const double d = sh.d;
        m.clear();
        for (unsigned i0 = 0; i0 < n; i0 += kBlock) {
            const unsigned i = i0 + threadIdx.x;
            const bool in = i < n;
            float z = 0.0f;
            float2 xy = make_float2(0.0f, 0.0f);
            const unsigned sl = in ? patch_slot(pts, i) : pts.off_lo;
            if (in) {
                z = pts.z[sl];
                xy = pts.xy[sl];
            }
            const bool active = in && !z_stripped(z);
            bool g = false;
            if (active) {
                const double dist = plane_dist(nx, ny, nz, d, xy.x, xy.y, z);
                g = dist < P.th_dist;  // ref :525,529 (one-sided)
            }
            if (g) m.add(xy.x, xy.y, z, scale, org);
            if (last) {
                if (__any(g && !(z >= org.zlo && z <= org.zhi)) && ln == 0) flag_clamped(Bt, f);
                if (g) {  // regionwise_ground_ (ref :529): the point's bit
                    const bool hi = i >= pc.n_lo;
                    const unsigned ip = hi ? i - pc.n_lo : i, r = ip & 511u;
                    const unsigned byte = mo[hi ? 1 : 0] + (ip >> 9) * 64u + ((r >> 2) & 63u), k = ((r >> 8) << 2) | (r & 3u);
                    atomicOr(reinterpret_cast<unsigned *>(frame_member + (byte & ~3u)), 1u << (8u * (byte & 3u) + k));
                }
            }
        }
        reduce_and_fit(sh, m, P.fxp_shift, pc.ox, pc.oy, z0, pts, true, P.th_dist, Bt.debug);  // ref :537-542
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
        rec->n_ground = (int)sh.last_n;  // (the set the final plane was fitted on: the last round's ground set)
        rec->n_nonground = (int)(n - (unsigned)sh.last_n);
        rec->decision = 0;
        rec->valid = 1 | (6 << 3) | ((P.num_iter & 0xff) << 8);  // the split is in the membership plane, 64-lane layout; every round ran
    }
}

template <bool WIDE>
__global__ __launch_bounds__(kBlock) void k_fit_stream(PwppBatch Bt, int b_lo) {
    __shared__ FitShared sh;
    const int f = blockIdx.x;  // frame = fast grid dimension, see fit_srows_body
    const uint32_t *cs = Bt.cls_start + (size_t)f * PWPP_CLS_STRIDE;
    const unsigned slot = cs[b_lo] + blockIdx.y;
    if (slot >= cs[PWPP_NUM_BUCKETS]) return;
    fit_stream_patch<WIDE>(sh, Bt, f, patch_ctx(Bt, f, slot, true), nullptr);
}

// k_fit_fixup: the patches of a frame that need the plane fitted before them (needs_previous_plane), in the reference's
// order: one workgroup walks the frame's bins in traversal order with the object's plane members in hand -- the plane
// of the stream's last frame at first (PwppPlaneState), then the final plane of every fitted patch it passes -- and
// fits the marked ones from there.  Launched by the host for the frames whose flag is set when a batch lands (never,
// for real scans); K5 and K6 follow for those frames.
template <bool WIDE>
__global__ __launch_bounds__(kBlock) void k_fit_fixup(PwppBatch Bt) {
    __shared__ FitShared sh;
    __shared__ PwppPlaneState cur;
    const int f = blockIdx.x;
    const PwppDevParams &P = Bt.P;
    const PwppFrameDesc fd = Bt.frames[f];
    const int B = P.num_bins, NP = PWPP_NUM_PARTS(B);
    if (threadIdx.x == 0) {
        if (fd.state_in >= 0) {
            cur = Bt.st_plane[fd.state_in];
        } else {
            for (int k = 0; k < 3; ++k) cur.mean[k] = cur.normal[k] = cur.sv[k] = 0.0f;
            cur.d = 0.0;
        }
    }
    __syncthreads();
    for (int bin = 0; bin < B; ++bin) {  // (workgroup-uniform)
        const uint2 cnt = *reinterpret_cast<const uint2 *>(Bt.part_count + (size_t)f * NP + PWPP_PART_LO(bin));
        const unsigned n = cnt.x + cnt.y;
        if ((uint64_t)n < P.min_pts || n == 0u) continue;  // not a patch / nothing to fit (the members stay as they are)
        PwppPatchRec *rec = Bt.recs + (size_t)f * B + bin;
        if (rec->valid == 4) {
            PatchCtx pc;
            pc.bin = bin;
            pc.zone = bin < P.bin_base[1] ? 0 : (bin < P.bin_base[2] ? 1 : (bin < P.bin_base[3] ? 2 : 3));
            const uint2 off = *reinterpret_cast<const uint2 *>(Bt.part_off + (size_t)f * NP + PWPP_PART_LO(bin));
            pc.n_lo = cnt.x;
            pc.n_hi = cnt.y;
            pc.off_lo = off.x;
            pc.off_hi = off.y;
            pc.n = n;
            const float2 o = Bt.bin_origin[bin];
            pc.ox = o.x;
            pc.oy = o.y;
            fit_stream_patch<WIDE>(sh, Bt, f, pc, &cur);
            __threadfence_block();
            __syncthreads();
        }
        if (threadIdx.x == 0) {  // the members after this patch
            for (int k = 0; k < 3; ++k) {
                cur.mean[k] = rec->mean[k];
                cur.normal[k] = rec->normal[k];
                cur.sv[k] = rec->sv[k];
            }
            cur.d = rec->d;
        }
        __syncthreads();
    }
}


}  // namespace

extern "C" int pwpp_launch_fixup(const PwppBatch *batch, hipStream_t stream) {
    if (batch->num_frames <= 0) return 0;
    if (batch->P.fxp_wide) hipLaunchKernelGGL(k_fit_fixup<true>, dim3(batch->num_frames), dim3(kBlock), 0, stream, *batch);
    else hipLaunchKernelGGL(k_fit_fixup<false>, dim3(batch->num_frames), dim3(kBlock), 0, stream, *batch);
    return (int)hipGetLastError();
}

// launches of K4; ev (optional) = 7 events recorded around up to six launches
#define PWPP_DEFAULT_FIT_PLAN "W16:1023,W64.4:65535"
#define PWPP_DEFAULT_FIT_PLAN_WIDE "W16:1023,W64.8:65535"
#define PWPP_DENSE_FIT_PLAN "W16:1023,W64.2:65535"
// the plan of a big batch: four big bins per wave share a solve on scans of KITTI density; with several times more points
// per bin (dense 128-beam frames) a wave's chain gets too long and two per wave win (profiles/r03_bench_dense_1024.json).
// On the wide grid (contract v4) the 64-lane kernel is bound by its instruction issue (93 % of the SIMDs' issue cycles,
// profiles/r06_pmc_summary.json), and the phases that run on one lane per patch -- solve, thresholds, lowest points -- are 46 % of
// its wave time (profiles/r06_fit_phases.txt): EIGHT big bins per wave halve those instructions per patch.  Alone the kernel is no
// faster (0.90 vs 0.88 ms), two batches in flight are: 2.49-2.50 instead of 2.56 ms per step (profiles/r06_plan_sweep_wide.txt);
// dense frames: no difference between 2, 4 and 8 (r06_plan_sweep_dense.txt).
extern "C" const char *pwpp_big_batch_plan(int max_n, int num_bins, int wide) {
    if (!((double)max_n / (double)(num_bins > 0 ? num_bins : 1) < 500.0)) return PWPP_DENSE_FIT_PLAN;
    return wide ? PWPP_DEFAULT_FIT_PLAN_WIDE : PWPP_DEFAULT_FIT_PLAN;
}
#define PWPP_LATENCY_FIT_PLAN "H64:511"  // (round 6, both grids: four waves from 512 points up -- frames 4 and 5 of the KITTI samples 102 -> 94 and 144 -> 123 us,
                                         // the others unchanged: profiles/r06_latency_plans.txt; rounds 3-5: from 1024 up)
// `aux` (optional): a second stream + two events, for the fit_concurrent option (classes of a plan side by side).
extern "C" int pwpp_launch_fit(const PwppBatch *batch, hipStream_t stream, hipEvent_t *ev, hipStream_t aux,
                               hipEvent_t aux_fork, hipEvent_t aux_join) {
    const PwppBatch &B = *batch;
    const int F = B.num_frames, nb = B.P.num_bins;
    const unsigned min_pts = B.P.min_pts < 1 ? 1u : (unsigned)(B.P.min_pts > 0xffffffffull ? 0xffffffffu : B.P.min_pts);
    // a class whose patches have more than `lo` points holds at most max_n / lo of them per frame
    auto cap = [&](unsigned lo) -> unsigned {
        unsigned c = (unsigned)B.max_n / (lo > min_pts ? lo : min_pts);
        if (c > (unsigned)nb) c = (unsigned)nb;
        return c < 1 ? 1u : c;
    };
    // The plan: which kernel handles which size range (ranges = runs of the quarter-octave size
    // buckets k_czm_scan sorts the patches into).  "S16:1023" = streaming rows of 16 lanes up to
    // 1023 points, "W64.2:65535" = two big bins per wave, ...; whatever is larger than the last entry
    // goes to the workgroup-per-patch kernel.  pwpp_set_option(h, "fit_plan", ...) overrides the default
    // for tuning experiments (the environment variable PWPP_FIT_PLAN sets that option at pwpp_create).
    const char *plan = B.fit_plan;
    // The right granularity depends on how much work there is to spread over 1024 SIMDs (measured with
    // tools/plan_by_frames.sh on KITTI frames; "frames" below = frames x sqrt(points per frame / 125 000)):
    //   <= 9    k_fit_hybrid: four waves per patch above 1023 points, one wave per smaller patch, one launch,
    //           one workgroup per CU                                         (chain latency is what counts;
    //           tools/small_batches.py: 1 / 2 / 4 / 8 frames 128 / 133 / 201 / 218 us per call, against
    //           174 / 184 / 229 / 229 us with the next plan and 134 / 167 / 225 / 323 us with four waves for
    //           every patch, "B64:65535"; from 12 frames on the next plan wins, 236 vs 307 us)
    //   <= 15   the same with four waves only above 2047 points
    //   <= 76   one prefetching wave per patch, every patch at once
    //   <= 320  16 small patches per wave; big bins one wave each         (8 -> 32 waves per frame)
    //   <= 576  16 small patches per wave; big bins two per wave
    //   <= 704  32 small patches per wave; big bins two per wave
    //   <= 896  32 small patches per wave; big bins four per wave
    //   more    64 small patches per wave; big bins four per wave         (fewest solve instances)
    // e.g. 1 frame: 0.128 ms instead of 0.174 with the second plan; 32 frames: 93 k frames/s instead of
    // 44 k with the last plan; 256 frames: 237 k instead of 208 k.  Round 4 (tools/plans_by_frames.py, profiles/
    // r04_plans_by_frames.txt): the thresholds re-measured with early termination in every kernel, and a frame range of
    // the overlap schedule takes the plan of the WHOLE call (B.plan_frames) -- the ranges share the machine, and choosing
    // by their own size gave 512 / 640 frames a plan 10-12 % slower than the best.
    if (!plan || !*plan) {
        // (frames of KITTI size; denser frames count by the square root of their size -- their patches are larger, not more:
        // 256 / 384 / 512 dense 486 k-point frames are best served by the plans of 505 / 757 / 1010 KITTI frames, bench.py
        // --workload dense with every plan, profiles/r04_plans_by_frames.txt)
        const double eff = (double)(B.plan_frames > 0 ? B.plan_frames : F) * std::sqrt((double)B.max_n / 125000.0);
        plan = eff <= 9.0 ? PWPP_LATENCY_FIT_PLAN
             : eff <= 15.0 ? "H64:2047"  // (10 / 12 frames: 169 / 184 us per call against 215 / 217 with the next plan; 16: equal)
             : eff <= 76.0 ? "S64:65535"
             : eff <= 320.0 ? "W16.16:1023,S64:65535"
             : eff <= 576.0 ? "W16.16:1023,W64.2:65535"
             : eff <= 704.0 ? "W16.32:1023,W64.2:65535"
             : eff <= 896.0 ? (B.max_n / (nb > 0 ? nb : 1) >= 500 ? "W16.32:1023,W64.2:65535" : "W16.32:1023,W64.4:65535")
                            : pwpp_big_batch_plan(B.max_n, nb, B.P.fxp_wide);
    }
    int k_lo = 0, slot = 0;
    unsigned n_lo = 1;
    // The classes of a plan are independent of each other: the fit_concurrent option runs the later ones on
    // the aux stream beside the first (fork after K3, join before K5; +3.5 % on the 1024-frame batch;
    // off by default: the per-kernel times bench.py reports would lose their meaning).
    const bool concurrent = aux != nullptr && !ev && B.fit_concurrent != 0;
    const bool fork = concurrent;
    if (fork) {  // every class only depends on K3
        hipError_t e = hipEventRecord(aux_fork, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(aux, aux_fork, 0);
        if (e != hipSuccess) return (int)e;  // (unordered classes would race with K3: nothing has been launched yet)
    }
    const char *p = plan;
    while (*p && slot < 5) {
        char mode = p[0];
        int g = 0, pw = 0;
        unsigned upper = 0;
        if (sscanf(p + 1, "%d.%d:%u", &g, &pw, &upper) != 3) {
            pw = 0;
            if (sscanf(p + 1, "%d:%u", &g, &upper) != 2) return (int)hipErrorInvalidValue;
        }
        if (g != 8 && g != 16 && g != 32 && g != 64) return (int)hipErrorInvalidValue;
        // A lane adds up at most 2047 points (Moments).  A patch is dealt out in chunks of 8 points per lane, its two
        // parts separately, the chunks of the four-waves kernel four at a time: a lane sees up to n / G + 16 (n / 256 + 24)
        // points.  Rows of 16 lanes in k_fit_w64 keep their ten totals in int64, which ends at 2047 points per PATCH.
        const unsigned lane_cap = mode == 'B' || mode == 'H' ? 2023u * 256u : (mode == 'W' && g == 16 ? 2047u : 2031u * (unsigned)g);
        if (upper > lane_cap) upper = lane_cap;
        if (upper > 65535u && mode != 'B' && mode != 'H') upper = 65535u;  // one wave per patch: keep the classes short
        const int k_hi = pwpp_size_bucket(upper + 1u);
        if (k_hi > k_lo) {
            if (ev) (void)hipEventRecord(ev[slot], stream);
            const hipStream_t ls = (concurrent && slot >= 1) ? aux : stream;  // later classes beside the first one
            const unsigned patches = cap(n_lo);
            const dim3 grid(F, (patches * (unsigned)g + kBlock - 1) / kBlock);
            // (every kernel exists for both widths of the arithmetic contract, PwppDevParams.fxp_wide)
#define PWPP_LAUNCH2(kern, ...) do { if (B.P.fxp_wide) hipLaunchKernelGGL((kern<true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<false>), __VA_ARGS__); } while (0)
#define PWPP_LAUNCH_T(kern, a, ...) do { if (B.P.fxp_wide) hipLaunchKernelGGL((kern<a, true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<a, false>), __VA_ARGS__); } while (0)
#define PWPP_LAUNCH_W(a, b, ...) do { if (B.P.fxp_wide) hipLaunchKernelGGL((k_fit_w64<a, b, true>), __VA_ARGS__); else hipLaunchKernelGGL((k_fit_w64<a, b, false>), __VA_ARGS__); } while (0)
            if (mode == 'S' && g == 8) PWPP_LAUNCH_T(k_fit_srows, 8, grid, dim3(kBlock), 0, ls, B, k_lo, k_hi);
            else if (mode == 'S' && g == 16) PWPP_LAUNCH_T(k_fit_srows, 16, grid, dim3(kBlock), 0, ls, B, k_lo, k_hi);
            else if (mode == 'S' && g == 32) PWPP_LAUNCH_T(k_fit_srows, 32, grid, dim3(kBlock), 0, ls, B, k_lo, k_hi);
            else if (mode == 'S' && g == 64) PWPP_LAUNCH_T(k_fit_srows, 64, grid, dim3(kBlock), 0, ls, B, k_lo, k_hi);
            else if (mode == 'H') {  // "H64:<n>": up to n points a wave per patch, four waves above (up to 2^19 - 1 points), everything in one launch
                const int k_top = pwpp_size_bucket(2023u * 256u + 1u);
                const unsigned nb_big = cap(pwpp_bucket_floor(k_hi));
                PWPP_LAUNCH2(k_fit_hybrid, dim3(F, nb_big + (patches + kWaves - 1) / kWaves), dim3(kBlock), 0, ls, B, k_hi, k_top, nb_big);
                k_lo = k_top;
                n_lo = pwpp_bucket_floor(k_top);
                ++slot;
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
                continue;
            }
            else if (mode == 'B') PWPP_LAUNCH2(k_fit_brows, dim3(F, patches), dim3(kBlock), 0, ls, B, k_lo, k_hi);
            else if (mode == 'W') {  // "W<lanes per patch>.<patches per wave>"
                if (pw == 0) pw = 64;
                const dim3 wgrid(F, (patches + (unsigned)pw - 1) / (unsigned)pw), wblock(64);  // one wave per workgroup
                if (g == 16 && pw == 64) PWPP_LAUNCH_W(16, 64, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else if (g == 16 && pw == 32) PWPP_LAUNCH_W(16, 32, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else if (g == 16 && pw == 16) PWPP_LAUNCH_W(16, 16, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else if (g == 64 && pw == 16) PWPP_LAUNCH_W(64, 16, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else if (g == 64 && pw == 8) PWPP_LAUNCH_W(64, 8, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else if (g == 64 && pw == 4) PWPP_LAUNCH_W(64, 4, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else if (g == 64 && pw == 2) PWPP_LAUNCH_W(64, 2, wgrid, wblock, 0, ls, B, k_lo, k_hi);
                else return (int)hipErrorInvalidValue;
            }
            else return (int)hipErrorInvalidValue;
            ++slot;
            k_lo = k_hi;
            n_lo = pwpp_bucket_floor(k_hi);
        }
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
    }
    for (; slot < 5; ++slot)
        if (ev) (void)hipEventRecord(ev[slot], stream);
    if (ev) (void)hipEventRecord(ev[5], stream);
    // whatever is larger than the plan's last class: a workgroup per patch (none can exist when the largest
    // frame of the batch is smaller than that class's upper bound -- one launch less on the latency path)
    const bool rest_possible = (unsigned)B.max_n >= n_lo;
    if (!rest_possible) {
        // nothing left
    } else if (fork) {
        PWPP_LAUNCH2(k_fit_stream, dim3(F, cap(n_lo)), dim3(kBlock), 0, aux, B, k_lo);
    } else {
        PWPP_LAUNCH2(k_fit_stream, dim3(F, cap(n_lo)), dim3(kBlock), 0, stream, B, k_lo);
    }
    if (fork) {
        hipError_t e = hipEventRecord(aux_join, aux);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, aux_join, 0);
        if (e != hipSuccess) return (int)e;  // (the caller synchronises every stream on an error: pwpp_capi.cpp)
    }
    if (ev) (void)hipEventRecord(ev[6], stream);
    return (int)hipGetLastError();
}
