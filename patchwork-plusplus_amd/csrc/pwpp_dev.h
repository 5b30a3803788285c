// pwpp_dev.h -- structures shared by the host side (pwpp_capi.cpp) and the gfx950 kernels
// (pwpp_kernels.hip).  Internal; the public boundary is include/pwpp.h.
#ifndef PWPP_DEV_H
#define PWPP_DEV_H

#include <stdint.h>
#include <hip/hip_runtime.h>

#define PWPP_MAX_BINS 2048        // CZM bins per frame (default model: 504)
#define PWPP_MAX_NEAR_BINS 1024   // bins inside the rings of interest (default: 96)
#define PWPP_MAX_LPR 256          // num_lpr upper bound (default 20; the reference has none, patchworkpp.cpp:99-103: beyond ~4 x the lanes of
                                  // a fit row the lowest-point selection takes its exact, slow path)
#define PWPP_MAX_ROI 4            // rings of interest (reference keeps update_*_[4])
#define PWPP_NUM_BUCKETS 96        // patch size buckets (quarter octaves up to 2^24 points)
#define PWPP_CLS_STRIDE 104       // uint32 per frame in cls_start (PWPP_NUM_BUCKETS + 1, padded)

// per-point CZM codes written by k_czm_bin (uint16): 0..B-1 real bins, then
#define PWPP_CODE_RNR(B) ((B))        // reflected-noise hit      (patchworkpp.cpp:391-396)
#define PWPP_CODE_OOR(B) ((B) + 1)    // outside (min,max] range  (patchworkpp.cpp:595,618)
#define PWPP_CODE_DROP 0xFFFFu        // z == FLT_MIN in the input: the reference silently drops it (:591)
// A bin is stored in two PARTS: the points below the split height (z < -sensor_height + hi_split: the "low" part)
// and the others (the "high" part: z at or above it, or NaN).  The fit kernels skip the high part of a patch in every
// pass that provably cannot use it (pwpp_fit.hip).  Parts are numbered in memory order -- the two parts of a bin are
// neighbours, so a bin's slot range [first slot of its low part, +points of the bin) exists in every layout:
//   2 b, 2 b + 1   low / high part of bin b;   2 B, 2 B + 1   the two pseudo-bins (RNR hits, out of range)
// Every part starts at a multiple of PWPP_SLOT_ALIGN slots (16-byte loads of four points per lane; whole bytes of the membership
// plane).  MEMBERSHIP PLANE (PwppBatch.member): one bit per slot, written by the R-GPF rounds whose set may be the patch's
// final ground set, read by k_emit -- the split of a patch costs 1/8 byte per point instead of an index list.  The bits of part p
// of a frame start at byte  fd.mbase + (first slot of the part) / 8 + PWPP_MEMBER_PAD * p  and are stored chunk by chunk the way a
// fit row of G lanes sees its points (pwpp_fit.hip, chunk_point): byte  c * G + j  = the 8 points of lane j in chunk c, bit k = its
// k-th.  A part's bits therefore end at most G <= PWPP_MEMBER_PAD bytes beyond (its points) / 8: hence the pad per part.
#define PWPP_SLOT_ALIGN 32
#define PWPP_MEMBER_PAD 64
#define PWPP_PART_LO(bin) (2 * (bin))
#define PWPP_PART_HI(bin) (2 * (bin) + 1)
#define PWPP_NUM_PARTS(B) (2 * (B) + 2)
// One-pass binning with an OVERFLOW ARENA (round 6): a part's fixed segment holds ~1.06 x the largest count the part has had, and
// every frame owns PwppBatch.arena_slots more slots behind its segments.  A point whose part is full is written into the arena
// instead (k_czm_bin_scatter reserves a run per workgroup with one atomic on the frame's cursor -- bits 8.. of PwppFrameResult.overflow
// -- and tags the record with its part and its rank inside the part); k_czm_scan then MOVES every part that outgrew its segment into a
// contiguous region of the arena behind the spilled records (the first `capacity` points from the segment, the others by their tags)
// and points part_off at it: the fit kernels and k_emit see a contiguous part as always.  Only a frame whose arena runs out (or with
// more than PWPP_MAX_RELOC overflowing parts) raises the overflow flag and is binned again by the host.  The bits of a moved part
// live PWPP_MEMBER_PAD * parts bytes further on in the membership plane (the moved parts are placed in part order, so the plane's
// "slot / 8 + pad * part" addressing stays collision-free): pwpp_member_offset.
#define PWPP_MAX_RELOC 64
#define PWPP_EMIT_LONG_BLOCKS 8   // blocks of 512 list entries the main wave of a "long" bin copies in k_emit; the rest goes to the extra waves
#define PWPP_EMIT_LONG_MIN 8192   // a bin whose count has exceeded this in some frame of the handle is "long"

struct PwppDevParams {
    int32_t enable_RNR, enable_RVPF, enable_TGR;
    int32_t num_iter, num_lpr, num_rings_of_interest;
    uint64_t min_pts;  // (size_t)num_min_pts, the reference compares size_t < int (patchworkpp.cpp:191)
    double RNR_ver_angle_thr, RNR_intensity_thr;
    double sensor_height;  // initial value (fresh state)
    double th_seeds, th_dist, th_seeds_v, th_dist_v;
    double max_range, min_range, uprightness_thr, margin;
    double min_ranges[4], ring_sizes[4], sector_sizes[4];  // patchworkpp.h:122-134, computed on the host in double
    int32_t rings[4], sectors[4];
    int32_t bin_base[5];  // first bin of zone k; [4] = B
    int32_t num_bins;     // B
    int32_t fxp_shift;    // s of the plane-fit arithmetic contract: the sums run on a 2^-s m grid
    int32_t fxp_wide;     // 1 (default, contract v4): |Q| <= 2^35, s <= 30; 0 (option "exact_moments" = 0, rounds 3-5's v3): |Q| <= 2^26, s <= 21
    float fxp_zr;         // 2^(35 - s) / 2^(26 - s) metres: a fit's z coordinates are clamped to z0 +- fxp_zr before they are quantised
    float hi_split;       // metres above the ground level (-sensor_height) where the high part of a bin begins; huge = no high parts
    int32_t split_end;    // bins [0, split_end) are stored in two parts, the others keep all their points in the low part
    int32_t max_elev_storage, max_flat_storage;
    int32_t hist_cap;     // doubles per (state, which, ring) history slab
    int32_t near_bins;    // bins with concentric_idx < num_rings_of_interest
    double elevation_thr0[4], flatness_thr0[4];  // initial thresholds (fresh state)
    // float mirrors for the fast path of k_czm_bin (never decide a point near a boundary)
    float f_min_range, f_max_range, f_margin_r, f_margin_t;
    float f_zone[4], f_inv_ring[4], f_inv_sector[4];
};

struct PwppFrameDesc {
    const float *pts;
    int32_t n;
    int32_t cols;      // 3 or 4
    int32_t layout;    // PWPP_LAYOUT_*
    int32_t state_in;  // index into the state arrays, -1 = fresh (defaults from params)
    int64_t base;      // first slot of this frame in the compact per-point workspaces (codes, out_idx)
    int32_t state_out; // index the updated state is written to
    int32_t pad_;
    int32_t step;      // PWPP_LAYOUT_FIELDS: bytes from one point to the next (sensor_msgs/PointCloud2 point_step) ...
    int32_t off[4];    // ... and the byte offsets of x, y, z, intensity inside a point (intensity < 0: none)
    int32_t pad2_;
    int64_t mbase;     // first byte of this frame in the membership plane (= sbase / 8 + 2 * PWPP_MEMBER_PAD * parts * frame index)
    int64_t sbase;     // first slot of this frame in the part-ordered buffers (sorted_*): compact on the two-pass
                       // path, frame * slots_per_frame on the one-pass path (see cap_off)
};

struct PwppStateScalar {  // = pwpp_state
    double sensor_height;
    double elevation_thr[4];
    double flatness_thr[4];
    int32_t elev_len[4];
    int32_t flat_len[4];
};

// The reference object's plane members (pc_mean_, normal_, singular_values_, d_: patchworkpp.h:177-182) as they stand
// after a frame: they survive into the next estimateGround() call, where a bin that is processed without any fit -- an
// empty bin let through by num_min_pts <= 0, the ROS launch file's setting -- reports them (patchworkpp.cpp:49).
struct PwppPlaneState {
    float mean[3], normal[3], sv[3];
    float pad_;
    double d;
};

struct PwppPatchRec {  // one per (frame, bin); written by k_patch_fit, finished by k_gle_tgr
    float mean[3];
    float normal[3];
    float sv[3];
    int32_t n_ground;
    double d;
    int32_t n_points;
    int32_t n_nonground;
    int32_t decision;
    int32_t valid;  // 0: no fit ran in this bin (empty bin let through by num_min_pts <= 0); bit 1: the last pass skipped the
                    // high part (its points are non-ground, its bits in the membership plane were not written);
                    // bits 3-5: log2(G) of the fit rows that left the patch's split in the membership plane (PWPP_SLOT_ALIGN
                    // above), i.e. the layout of its bits: k_emit compacts the two lists from them; bits 8-15: the R-GPF rounds
                    // the fit ran (num_iter, or fewer: early termination -- a diagnostic, pwpp_patch_record.rounds);
                    // bit 2 (alone): the patch's first fit set was empty, so it works with the plane the reference object
                    // fitted LAST (the patch before it, or the frame before): nothing was fitted yet, k_fit_fixup does it
};

// first byte of a part's bits in its frame's share of the membership plane (see PWPP_SLOT_ALIGN / the overflow arena above)
__host__ __device__ inline uint32_t pwpp_member_offset(uint32_t off, int part, uint32_t arena_base, int num_parts) {
    return (off >> 3) + (uint32_t)(PWPP_MEMBER_PAD * (part + (off >= arena_base ? num_parts : 0)));
}

struct PwppFrameResult {
    int32_t n_ground, n_nonground, n_patches, n_rnr, n_oor, n_dropped;
    int32_t hist_state;  // (entries of the fullest A-GLE history after this frame << 1) | a push found its slab full
    int32_t overflow;  // (bits 8..: scratch of the binning kernels -- the fused scan's tickets, or the cursor of the frame's overflow arena;
                       // zero again when k_czm_scan is done)
                       // bit 0: one-pass binning: some bin of this frame outgrew its segment AND the arena (the batch is redone on the two-pass
                       // path); bit 1: some patch of the frame needs the plane fitted before it (PwppPatchRec.valid bit 2): K5 and K6
                       // leave the frame alone and the host runs k_fit_fixup + K5 + K6 for it when the batch lands; bit 2: the final
                       // ground set of some patch held a height outside z0 +- ZR (clamped before it was quantised: pwpp_get_clamped_frames);
                       // bit 3: k_czm_scan moved parts of this frame into the overflow arena (statistics only: pwpp_get_arena_stats)
};

// everything a launch needs, by value in the kernarg segment
// The fit passes re-read a patch 5-6 times and are bound by that traffic, so the bin-ordered records are
// planes: z (all a lowest-point pass needs, 4 B), {x, y} (8 B) and the cloud index (4 B, read by k_emit and the tiny-fit
// gather only; the fit passes leave the split of a patch in the membership plane, a bit per slot).

struct PwppBatch {
    PwppDevParams P;
    const PwppFrameDesc *frames;
    int32_t num_frames;
    int32_t max_n;               // largest frame of the batch
    int32_t debug;               // option "debug_flags": 4 = timing probes of the fit chain, 16 = exact binning only, 16384 / 32768 =
                                 // force the fall-back paths of the lowest-point selection (tests); results never depend on it
    int32_t no_clear;            // the caller already launched k_clear for these frames (overlap mode: two frame ranges, two streams)
    int32_t fixup_run;           // this launch finishes frames whose patches needed the plane fitted before them (k_fit_fixup ran): K5 / K6 do not skip them
    const uint32_t *cap_off;     // one-pass binning: [2B+3] first slot of every PART's fixed segment inside a frame
                                 // (cap_off[2B+2] = end of the segments = first slot of the frame's overflow arena); null on the two-pass path
    uint32_t arena_base;         // = cap_off[2B+2] on the one-pass path, 0xffffffff otherwise (no part lies at or beyond it)
    uint32_t arena_slots;        // slots of a frame's overflow arena (0: none -- a full segment raises the overflow flag as in rounds 1-5)
    uint32_t arena_spill;        // ... of which the first arena_spill may hold spilled records (the tags' capacity); moved parts follow them
    uint2 *arena_tag;            // [frames][arena_spill] {part, rank inside the part} of every spilled record
    PwppStateScalar *st_scalar;  // [num_states]
    double *st_hist;             // [num_states][2][4][hist_cap]
    PwppPlaneState *st_plane;    // [num_states] the plane members after the state's last frame (zero for a new object)
    // One-pass binning of stateful streams: what a redo on the exact path must start from (a segment overflow is only known when
    // the batch has landed, and K5 has updated the streams by then).  Written by one extra workgroup per frame of the binning
    // kernel -- the CUs are there, a copy command in front of the pipeline costs the host and the stream more than the binning
    // saves (round 4).  Null: no snapshot (fresh frames, two-pass binning).
    PwppStateScalar *snap_scalar;  // [num_states]
    double *snap_hist;             // [num_states][2][4][hist_cap]
    PwppPlaneState *snap_plane;    // [num_states]
    uint16_t *codes;             // [total points]
    uint32_t *part_count;        // [frames][2B+2] points per part (K1 / K1' histogram)
    uint32_t *part_off;          // [frames][2B+2] first slot of every part in the sorted_* planes (relative to sbase)
    uint32_t *part_cursor;       // [frames][2B+2] two-pass scatter cursors
    uint32_t *bin_count;         // [frames][B+2] points per bin = low + high part (K2); what K5 / K6 size the lists with
    uint32_t *bin_off;           // [frames][B+2] first slot of the bin (= of its low part), relative to sbase
    uint32_t *cls_start;         // [frames][PWPP_CLS_STRIDE] first entry of each size bucket in cls_list
    uint16_t *cls_list;          // [frames][B] patch bins sorted by size bucket
    float *sorted_z;             // [total points | frames x slots per frame] z of the points grouped by part.  A NaN z of the cloud is
                                 // stored as 0x7fc00000; 0x7fc00000 | (round + 1) marks a point an R-VPF round removed
    float2 *sorted_xy;           // same slots: {x, y}
    int *sorted_idx;             // same slots: cloud index of the point (read by K6, and by the tiny-fit gather of K4)
    uint32_t *bin_max;           // [2B+2] largest count every PART has had in any frame so far (k_czm_scan): sizes the one-pass segments
    const float4 *bin_bbox;      // [B] {xmin, xmax, ymin, ymax} of every bin (a little generous): the skip test of the high parts
    const float2 *bin_origin;    // [B] origin of every bin's fixed-point plane-fit sums (its polar centre rounded to 1/8 m)
    uint32_t *order_work;        // reference-order mode: per frame [1 + 2 (B + 2)] words -- the number of sub-lists above 256 entries, then
                                 // bin | which << 16 of each (k_order_worklist -> k_order_sublists); null otherwise
    uint8_t *member;             // membership plane (see PWPP_SLOT_ALIGN): the final ground set of a patch, one bit per slot
    PwppPatchRec *recs;          // [frames][B]
    uint32_t *dst_a;             // [frames][B+2] output offset of sub-list A (candidates / whole bin)
    uint32_t *dst_b;             // [frames][B+2] output offset of sub-list B (regionwise non-ground)
    int32_t *out_idx;            // [total points] per frame: ground list then non-ground list
    float *centers;              // [frames][B][3] compacted to n_patches rows
    float *normals;              // [frames][B][3]
    PwppFrameResult *results;    // [frames]
    PwppFrameResult *results_host;  // [frames] pinned host mirror, written by K6 (no D2H copy command behind the pipeline)
    int fuse_scan;               // a few frames: K2 inside K1' (the last workgroup of a frame to take a ticket scans) instead of a kernel of its own
    int k5_split;                // K5 in two launches (k_gle_tgr PART 1 / 2): the second on the handle's other stream, joined by the host before the next call
    unsigned long long *dbg;     // [64] timing probes, only written when debug & 4
    // host side only (the kernels never read these)
    const char *fit_plan;        // option "fit_plan": overrides the plan pwpp_launch_fit would choose; null or empty = automatic
    int32_t plan_frames;         // frames the automatic fit plan is chosen for: the WHOLE call's when this batch is one of its frame ranges (0: num_frames)
    int32_t fit_concurrent;      // option "fit_concurrent": the classes of a plan side by side on two streams
    int32_t emit_parts;          // waves per bin in k_emit (1..8, from the largest bin seen so far)
    // Big batches (round 5): one wave per bin, and the FEW bins that have held long lists so far (pseudo-bins of a sensor that sees
    // beyond max_range or its own vehicle, the near bins of a dense cloud) get extra waves from a second, small launch -- not every
    // bin of every frame (a 55 k-point pseudo-bin used to put seven waves on each of 500 k bins: k_emit 0.29 -> 0.92 ms).
    const uint8_t *emit_long;        // [B+2] 1 = the bin is in emit_long_list: its main wave stops after PWPP_EMIT_LONG_BLOCKS blocks; null = none
    const uint16_t *emit_long_list;  // [emit_long_n] those bins
    int32_t emit_long_n;
    int32_t emit_long_parts;     // waves per listed bin of the second launch
    int32_t emit_long_pass;      // set in the copy of the batch the second launch gets
    int32_t bin_block;           // option "bin_block": threads per workgroup of k_czm_bin_scatter (256, 512, 1024; four points each)
    // The counters a call starts from (part_count [+ part_off, part_cursor], results) exist TWICE; a call works on one copy
    // and its K5 zeroes the frame's share of the OTHER copy, so that the next call of the same shape needs no clearing
    // kernel in front of its binning (a single frame: k_clear and the dispatch gap behind it were 5-7 of 105 us).
    uint32_t *next_part_count;   // the other copy's part_count slab, this batch's frames (null: nothing to prepare)
    PwppFrameResult *next_results;
    int64_t next_slab_stride;    // words from part_count to part_off / part_cursor in the other copy (= frames of the CALL x parts)
    int32_t next_slabs;          // 1 (one-pass binning: counts only) or 3
};

#endif
