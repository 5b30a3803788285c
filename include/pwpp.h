/* include/pwpp.h -- C-ABI of the MI355X-native Patchwork++ hot path (libpwpp_hip.so).
 *
 * This is the drop-in boundary for the reference's per-frame path
 *     patchwork::PatchWorkpp::estimateGround()      /root/reference/cpp/patchworkpp/src/patchworkpp.cpp:151-336
 * and its result getters                            /root/reference/cpp/patchworkpp/include/patchwork/patchworkpp.h:152-163
 * Plain C: POD structs, raw pointers and sizes only; no C++ / torch / Eigen types cross it.
 * The reference has no FFI of its own for this path (its Python module binds the C++ class
 * directly, python/patchworkpp/pybinding.cpp:45-55); the C++ class and pybind11 module that
 * sit on top of this header (patchwork-plusplus_amd/include/patchwork/patchworkpp.h,
 * patchwork-plusplus_amd/python/pybinding.cpp) are what a maintainer would swap in --
 * see INTEGRATION.md.
 *
 * Every entry point returns PWPP_OK (0) or a negative pwpp_status; pwpp_last_error() gives
 * the text for the calling thread.  One handle = one device + one HIP stream; a handle is
 * not re-entrant (like the reference object, patchworkpp.h:177-195), distinct handles are
 * independent and may be driven from different threads.
 */
#ifndef PWPP_H
#define PWPP_H

#include <stdint.h>

/* Only the entry points below are exported: libpwpp_hip.so is built with -fvisibility=hidden. */
#ifndef PWPP_API
#define PWPP_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define PWPP_VERSION_MAJOR 0
#define PWPP_VERSION_MINOR 2 /* round 6: pwpp_pipe_submit takes a mode; pwpp_pipe_set_num_streams, pwpp_get_arena_stats; options exact_moments, split_k5 */

typedef enum pwpp_status {
    PWPP_OK = 0,
    PWPP_E_ARG = -1,         /* bad argument / parameter combination              */
    PWPP_E_HIP = -2,         /* a HIP runtime call failed                          */
    PWPP_E_NOMEM = -3,       /* device or host allocation failed                   */
    PWPP_E_STATE = -4,       /* call sequence error (e.g. getter before any frame) */
    PWPP_E_UNSUPPORTED = -5, /* legal for the reference, not implemented here      */
    PWPP_E_NODEVICE = -6     /* no usable GPU: the product path never falls back to CPU */
} pwpp_status;

/* Mirror of patchwork::Params, field for field (reference patchworkpp.h:42-112).
 * bools are int32 (0/1); the four std::vector members are fixed arrays of 4 because the
 * reference hard-codes four zones (patchworkpp.h:122-134, patchworkpp.cpp:580-615). */
typedef struct pwpp_params {
    int32_t verbose;      /* patchworkpp.h:44 */
    int32_t enable_RNR;   /* :45 */
    int32_t enable_RVPF;  /* :46 */
    int32_t enable_TGR;   /* :47 */
    int32_t num_iter;     /* :49 */
    int32_t num_lpr;      /* :50 */
    int32_t num_min_pts;  /* :51 */
    int32_t num_zones;    /* :52  (must be 4, see above) */
    int32_t num_rings_of_interest; /* :53 (<= 4: the reference keeps update_*_[4], patchworkpp.h:174-175) */
    int32_t pad0_;
    double RNR_ver_angle_thr; /* :55 */
    double RNR_intensity_thr; /* :56 */
    double sensor_height;     /* :58 */
    double th_seeds;          /* :59 */
    double th_dist;           /* :60 */
    double th_seeds_v;        /* :61 */
    double th_dist_v;         /* :62 */
    double max_range;         /* :63 */
    double min_range;         /* :64 */
    double uprightness_thr;   /* :65 */
    double adaptive_seed_selection_margin; /* :66 */
    double intensity_thr;     /* :67 (never read by the hot path; kept for API parity) */
    int32_t num_sectors_each_zone[4]; /* :69 */
    int32_t num_rings_each_zone[4];   /* :70 */
    int32_t max_flatness_storage;     /* :72 */
    int32_t max_elevation_storage;    /* :73 */
    double elevation_thr[4];          /* :75 */
    double flatness_thr[4];           /* :76 */
} pwpp_params;

/* Adaptive per-stream state the reference keeps inside the object and mutates every frame
 * (params_.sensor_height / elevation_thr / flatness_thr, patchworkpp.cpp:347-350,368). */
typedef struct pwpp_state {
    double sensor_height;
    double elevation_thr[4];
    double flatness_thr[4];
    int32_t elevation_len[4]; /* entries in update_elevation_[i] */
    int32_t flatness_len[4];  /* entries in update_flatness_[i]  */
} pwpp_state;

/* One processed patch (CZM bin with >= num_min_pts points), in bin traversal order. */
typedef struct pwpp_patch_record {
    int32_t bin;            /* flattened zone->ring->sector index (patchworkpp.cpp:184-189) */
    int32_t concentric_idx; /* patchworkpp.cpp:174,309 */
    int32_t n_points;
    int32_t n_ground;       /* |regionwise_ground_|     (patchworkpp.cpp:529-531) */
    int32_t n_nonground;    /* |regionwise_nonground_|  (patchworkpp.cpp:500,532) */
    int32_t decision;       /* pwpp_decision */
    float mean[3];          /* pc_mean_            (patchworkpp.cpp:59-60) */
    float normal[3];        /* normal_             (patchworkpp.cpp:66-68) */
    float sv[3];            /* singular_values_    (patchworkpp.cpp:63)    */
    int32_t rounds;         /* diagnostic, no reference counterpart: R-GPF rounds the fit ran -- num_iter, or fewer when a round's
                             * integer totals repeated the round before's (early termination: the remaining rounds and the final fit
                             * of patchworkpp.cpp:516-543 provably reproduce this round's set and plane) */
    double d;               /* d_                  (patchworkpp.cpp:74)    */
} pwpp_patch_record;

typedef enum pwpp_decision {
    PWPP_DEC_NOT_UPRIGHT = 1, /* patchworkpp.cpp:262-265 */
    PWPP_DEC_FAR_GROUND = 2,  /* :266-269 */
    PWPP_DEC_HEADING = 3,     /* :270-273 */
    PWPP_DEC_GROUND = 4,      /* :274-277 */
    PWPP_DEC_TGR_REJECT = 5,  /* :278-282 then :452-459 / :296-300 */
    PWPP_DEC_TGR_REVERT = 6   /* :444-451 */
} pwpp_decision;

enum { PWPP_LAYOUT_ROW_MAJOR = 0, /* (n, cols) C-order: np.fromfile(..).reshape(-1,4), python/examples/demo_visualize.py:10-14 */
       PWPP_LAYOUT_COL_MAJOR = 1, /* Eigen::MatrixXf default storage: cols planes of n floats (patchworkpp.h:152) */
       PWPP_LAYOUT_FIELDS = 2     /* (internal to pwpp_estimate_ground_fields*: float32 fields at byte offsets of a record) */ };
enum { PWPP_MEM_HOST = 0,         /* pageable or pinned host memory; the call returns when the results are ready */
       PWPP_MEM_DEVICE = 1,       /* device memory; asynchronous */
       PWPP_MEM_HOST_PINNED = 2   /* page-locked host memory (pwpp_host_alloc / hipHostMalloc) that stays valid and
                                     unchanged until pwpp_synchronize(): the copies and the launches are only
                                     enqueued, so a second handle can overlap its own transfers and kernels
                                     (double-buffered ingest) */ };
enum { PWPP_MODE_FRESH = 0,   /* every frame starts from the handle's Params (= a fresh PatchWorkpp object per frame) */
       PWPP_MODE_STREAMS = 1  /* frame i belongs to stream i and reads+updates that stream's adaptive state
                                 (= one long-lived PatchWorkpp object per stream, demo_sequential.cpp:54-67) */ };

typedef struct pwpp_handle pwpp_handle;

/* ---- lifetime ------------------------------------------------------------------------- */
/* Params() defaults, reference patchworkpp.h:79-111 */
PWPP_API int pwpp_params_default(pwpp_params *p);
/* PatchWorkpp::PatchWorkpp(Params), reference patchworkpp.h:120-150: validates, computes the
 * CZM geometry in double exactly as the reference constructor, creates stream + workspace. */
PWPP_API int pwpp_create(const pwpp_params *p, int device, pwpp_handle **out);
PWPP_API int pwpp_destroy(pwpp_handle *h);
PWPP_API const char *pwpp_last_error(void);
PWPP_API int pwpp_device_count(void);

/* ---- the hot path ---------------------------------------------------------------------- */
/* void PatchWorkpp::estimateGround(Eigen::MatrixXf cloud_in), reference patchworkpp.cpp:151.
 * One frame, host memory, stream 0 of the handle, stateful like the reference object.
 * cols is 3 or 4 (3 only legal with enable_RNR == 0 semantics of patchworkpp.cpp:379-382:
 * RNR is skipped).  Synchronous: results are ready on return. */
PWPP_API int pwpp_estimate_ground(pwpp_handle *h, const float *points, int n, int cols, int layout);

/* Many independent frames in one set of launches.  points[i] is frame i (host or device
 * memory according to `mem`), n[i] its point count.  PWPP_MODE_FRESH: each frame is
 * processed with fresh state.  PWPP_MODE_STREAMS: frames <= streams configured with
 * pwpp_set_num_streams(); frame i continues stream i.  Asynchronous when mem is
 * PWPP_MEM_DEVICE or PWPP_MEM_HOST_PINNED: call pwpp_synchronize() before reading results. */
PWPP_API int pwpp_estimate_ground_batch(pwpp_handle *h, const float *const *points, const int32_t *n, int frames,
                               int cols, int layout, int mem, int mode);
/* The ROS 2 wrapper's input (reference ros/src/GroundSegmentationServer.cpp:72-75, ros/src/Utils.hpp:158-172
 * PointCloud2ToEigenMat: x, y, z read through one float32 iterator per field): `data` = msg->data, n = height * width,
 * point_step and the byte offsets of the fields as the message declares them (4-byte aligned; off_intensity < 0 when
 * there is none -- RNR is then skipped as for an N x 3 matrix, patchworkpp.cpp:379-382).  The fields are read where
 * they lie: no repacked copy on the host.  pwpp_estimate_ground_fields = one frame on stream 0 from host memory, like
 * pwpp_estimate_ground; the _batch form takes `mem` and `mode` like pwpp_estimate_ground_batch. */
PWPP_API int pwpp_estimate_ground_fields(pwpp_handle *h, const void *data, int n, int point_step, int off_x, int off_y, int off_z, int off_intensity);
PWPP_API int pwpp_estimate_ground_fields_batch(pwpp_handle *h, const void *const *data, const int32_t *n, int frames, int point_step,
                                      int off_x, int off_y, int off_z, int off_intensity, int mem, int mode);
PWPP_API int pwpp_synchronize(pwpp_handle *h);
PWPP_API int pwpp_set_num_streams(pwpp_handle *h, int streams); /* (re)creates `streams` fresh stream states */

/* ---- results of the last call ---------------------------------------------------------- */
/* sizes of getGround()/getNonground()/getCenters(), reference patchworkpp.h:157-163 */
PWPP_API int pwpp_get_counts(pwpp_handle *h, int frame, int32_t *n_ground, int32_t *n_nonground, int32_t *n_patches);
/* getGroundIndices()/getNongroundIndices(), reference patchworkpp.h:159-160, patchworkpp.cpp:18-26.
 * The index SETS are those of the reference's control flow with the plane-fit sums of patchworkpp.cpp:56-60
 * evaluated (contract v4)
 *   - for a fit set of 1, 2 or 3 points: in the reference's own float arithmetic, which is determinate there (Eigen
 *     reduces fewer elements than one SIMD packet sequentially; two terms commute) -- points in the order of the
 *     reference's z-sorted bin, equal heights in cloud order (the reference's std::sort is stable only for bins of up
 *     to 16 points -- libstdc++'s insertion sort; beyond that members of EQUAL height may reach it in another order, and
 *     the last bits of a 2-3 point float mean / covariance with them: the bit-for-bit statement holds for distinct heights,
 *     tests/test_tiny_fits.py records the behaviour with duplicated heights).  All three builds of the reference under oracle/_ref agree
 *     on such sets and this library agrees with them: identical ground sets under the ROS launch file's parameters,
 *     num_min_pts 0-3, num_lpr 1-3 on the KITTI samples (tests/test_tiny_fits.py, CPU and GPU);
 *   - for 4 points and more: in EXACT arithmetic on the reference's own floats -- integer moments on a 2^-30 m grid around
 *     per-bin / per-patch origins, on which every float of magnitude >= 2^-7 m lies (smaller ones are rounded to it: an error of
 *     at most 2^-31 m); z clamped to z0 +- 2^(35-s) m = 32 m with the default CZM (pwpp_get_fxp_geometry).  Eigen's float
 *     summation order there depends on the vector width it was built for; exact sums do not depend on any order, and they are what
 *     every order approximates.  Bit-identical to the CPU restatement of the contract (oracle/).
 *     Measured on 10 400 frames against all three builds of the reference (float sums in two orders, exact-f64 sums; tools/parity_statistics.py,
 *     profiles/r06_parity_statistics_10k.json -- CPU restatement of the contract, which the HIP path equals bit for bit): 6 000 varied 64-beam
 *     frames with fresh state, 20 stateful sequences of 200 frames, 400 dense 128-beam frames with the 36-sector CZM.  The builds are
 *     unanimous on 5 829 / 3 869 / 334 of them, and this library returns exactly their ground set on EVERY one of those 10 032 frames.
 *     Where the builds differ among themselves (2.9 % / 3.3 % / 16.5 % of the frames: there is no single reference result) the library
 *     equals one of the builds on all 368: the exact-f64 build on 303, both float builds on 63 (frames where a fit set of 1-3 points
 *     decides -- the reference sums those in float, determinately; the exact-f64 build is the odd one out there), one float build on 2.
 *     Adaptive sensor height over the 200-frame sequences: within 1.2e-3 m of the exact build, as the float build of the reference is
 *     (a split frame's differing patch decision enters the elevation history).  On the reference's own KITTI samples: identical index
 *     sets with every build, plane normals within 3.1e-5 of the float build's and 6e-8 of the exact build's.
 *     Option "exact_moments" = 0 selects rounds 3-5's coarser 2^-21 m grid (contract v3: |Q| <= 2^26, nine multiply-adds per point instead of
 *     twenty-one): 7 % faster on 1024-frame batches (2.36 vs 2.52 ms per batch on one MI355X), 5-6 us on a single frame -- and off the unanimous
 *     reference by 1-31 indices of ~120 000 on 0.2 % of varied frames (18 of the 10 032 above), because that grid is coarser than the float
 *     ulp of heights around -1.7 m and of |x|, |y| < 4 m.  Both widths are tested bit for bit against their restatements.
 * (NaN heights are undefined in the reference itself: it sorts bins with `a.z < b.z`.)
 * The order inside a list is not the reference's unless pwpp_set_output_order asks for it. */
PWPP_API int pwpp_get_ground_indices(pwpp_handle *h, int frame, int32_t *out);
PWPP_API int pwpp_get_nonground_indices(pwpp_handle *h, int frame, int32_t *out);
/* getGround()/getNonground(), reference patchworkpp.h:157-158: row-major (count,3) float32,
 * rows aligned with the index getters above. */
PWPP_API int pwpp_get_ground_xyz(pwpp_handle *h, int frame, float *out);
PWPP_API int pwpp_get_nonground_xyz(pwpp_handle *h, int frame, float *out);
/* getCenters()/getNormals(), reference patchworkpp.h:162-163: row-major (n_patches,3), bin traversal order */
PWPP_API int pwpp_get_centers(pwpp_handle *h, int frame, float *out);
PWPP_API int pwpp_get_normals(pwpp_handle *h, int frame, float *out);
/* per-patch detail for parity checks (no reference getter; fields are the reference's scratch members) */
PWPP_API int pwpp_get_patch_records(pwpp_handle *h, int frame, pwpp_patch_record *out, int capacity);
/* getHeight(), reference patchworkpp.h:154 (stream 0) */
PWPP_API double pwpp_get_height(pwpp_handle *h);
/* getTimeTaken(), reference patchworkpp.h:155: microseconds of the last estimate call
 * (GPU time between HIP events on the handle's stream, batch calls: whole batch: first kernel -> index lists written.  With up to 64
 * stateful streams the update of the streams' adaptive thresholds -- K5's second launch, option "split_k5" -- runs on the handle's second
 * stream and ends ~8 us after the lists; the next estimate call and every call that touches a stream's state wait for it, pwpp_synchronize
 * and the getters of a call's results -- counts, lists, patch rows -- do not) */
PWPP_API double pwpp_get_time_us(pwpp_handle *h);

/* ---- adaptive state --------------------------------------------------------------------- */
/* state after the last call; PWPP_MODE_FRESH: `index` is the frame, PWPP_MODE_STREAMS: the stream */
PWPP_API int pwpp_get_state(pwpp_handle *h, int index, pwpp_state *out);
PWPP_API int pwpp_get_history(pwpp_handle *h, int index, int which /*0 elevation, 1 flatness*/, int ring, double *out, int capacity);
/* overwrite the scalars of a stream state; its histories are cleared (elevation_len / flatness_len of `in` are ignored),
 * its plane members (pwpp_set_plane_state) are left as they are */
PWPP_API int pwpp_set_state(pwpp_handle *h, int stream, const pwpp_state *in);
/* ... and put a history back: after pwpp_set_state + eight pwpp_set_history calls with what pwpp_get_state /
 * pwpp_get_history returned, a stream continues exactly where the checkpointed one stood. */
PWPP_API int pwpp_set_history(pwpp_handle *h, int stream, int which /*0 elevation, 1 flatness*/, int ring, const double *values, int count);
/* The plane members of the reference object (pc_mean_, normal_, singular_values_, d_: patchworkpp.h:177-182) as they
 * stand after the state's last frame -- {mean[3], normal[3], singular values[3], d}.  They are part of what a stream
 * carries from frame to frame: a bin that is processed without a fit (an empty bin let through by num_min_pts <= 0, the
 * ROS launch file's setting) reports whatever plane was fitted last, also across frames (patchworkpp.cpp:49).  Zero for a
 * new stream; a checkpoint is pwpp_get_state + the histories + this. */
PWPP_API int pwpp_get_plane_state(pwpp_handle *h, int index, float out[10]);
/* Frames this handle had to finish with the serial fix-up kernel: a patch whose first fit set is empty consults the
 * plane the reference object fitted last (the patch before it, or the frame before), which the parallel fit kernels
 * only recognise; the host then runs k_fit_fixup + the GLE and list kernels for that frame.  It takes a lowest height
 * of -inf, one beyond 1e15 m, or num_lpr = 0 -- no real scan; the count exists for tests. */
PWPP_API int64_t pwpp_get_fixed_up_frames(pwpp_handle *h);
/* Frames this handle has seen in which the FINAL ground set of some patch held a height outside z0 +- 2^(26-s) m (32 m with
 * the default CZM), i.e. a fit of 4+ points whose z coordinates were clamped before they were quantised (see
 * pwpp_get_fxp_origins): that patch's plane is the plane of the clamped heights, not the reference's.  No ground patch is
 * that tall; a steep facade or cliff filling a bin can be (it is rejected as "not upright" either way with the default
 * parameters).  0 on every scan of the test suite.  (2^(35-s) with the default "exact_moments" = 1, 2^(26-s) with 0: 32 m either way
 * for the default CZM, s = 30 / 21.) */
PWPP_API int64_t pwpp_get_clamped_frames(pwpp_handle *h);
/* Host only (no device needed): shift and per-bin origins {ox, oy} of the fixed-point plane-fit sums a handle created with
 * these parameters uses by default (= pwpp_get_fxp_shift / pwpp_get_fxp_origins of that handle; contract v4: s <= 30, |Q| <= 2^35.
 * The origins do not depend on the option "exact_moments"; the shift does -- pwpp_get_fxp_shift reports the handle's current one).  The CPU tests compare them with
 * the restatement's for many CZM shapes.  Returns the number of bins; shift / out_xy may be NULL. */
PWPP_API int pwpp_get_fxp_geometry(const pwpp_params *p, int *shift, float *out_xy, int capacity_bins);
/* Host only (no device needed): the axis-aligned box {xmin, xmax, ymin, ymax} the library assumes around every CZM bin of
 * a parameter set, bins in traversal order (zone, ring, sector).  The fit kernels prove with it that no point of a bin's
 * high part can lie below a plane, so every point the reference bins into b must lie inside box b:
 * tests/test_capi_cpu.py checks exactly that.  Returns the number of bins (with out_boxes == NULL: just that). */
PWPP_API int pwpp_get_bin_boxes(const pwpp_params *p, float *out_boxes, int capacity_bins);
PWPP_API int pwpp_set_plane_state(pwpp_handle *h, int stream, const float in[10]);

/* ---- ingest ----------------------------------------------------------------- */
/* Page-locked host memory: frames handed over in such buffers are DMA'd straight to the device
 * (pageable memory is staged by the runtime at a fraction of the PCIe rate), and result copies
 * into them do not block. */
PWPP_API int pwpp_host_alloc(void **out, uint64_t bytes);
PWPP_API int pwpp_host_free(void *p);
/* All index lists of the last batch in ONE device-to-host copy: out[frame_base[f] .. +n_ground)
 * is frame f's ground list, followed by its non-ground list; frame_base (frames+1 entries) and
 * counts (frames x 8 int32, see pwpp_device_view) are filled if not NULL.  `out` must hold the
 * total number of points of the batch. */
PWPP_API int pwpp_get_all_indices(pwpp_handle *h, int32_t *out, int64_t *frame_base, int32_t *counts);

/* ---- device-side views and measurement --------------------------------------------------- */
typedef struct pwpp_device_view {
    const int32_t *indices;      /* all frames: [frame_base[f] .. +n_ground) ground, then nonground */
    const int64_t *frame_base;   /* frames+1 prefix sums of n (host pointer, pinned)                */
    const int32_t *counts;       /* frames x 8 int32 (host pointer, pinned): n_ground, n_nonground, n_patches, n_rnr, n_out_of_range, n_dropped,
                                    history fill (library bookkeeping), 0 */
    int32_t frames;
    int32_t pad_;
} pwpp_device_view;
PWPP_API int pwpp_get_device_view(pwpp_handle *h, pwpp_device_view *out);

/* per-kernel GPU time (HIP events on the handle's stream around every launch) */
#define PWPP_NUM_KERNELS 11
PWPP_API int pwpp_set_profiling(pwpp_handle *h, int enable);
PWPP_API int pwpp_get_kernel_profile(pwpp_handle *h, double *sum_ms /*[PWPP_NUM_KERNELS]*/, int64_t *launches /*[PWPP_NUM_KERNELS]*/);
PWPP_API int pwpp_reset_kernel_profile(pwpp_handle *h);
PWPP_API const char *pwpp_kernel_name(int k);
/* the fixed-point contract of the plane-fit sums for this handle: the shift s (grid 2^-s m) ... */
PWPP_API int pwpp_get_fxp_shift(pwpp_handle *h);
/* ... and every bin's origin (its polar centre rounded to 1/8 m): out_xy = B x {x, y}; returns B (out_xy = NULL: only B).
 * The z coordinates of a fit of 4+ points are clamped to z0 +- 2^(35-s) m (32 m with the default CZM; z0 = the patch's first
 * lowest-point representative rounded to 1/8 m) before they are quantised: a fit set that spans more than that vertically --
 * no ground patch does; a facade that R-VPF did not strip could -- gets the plane of the clamped heights. */
PWPP_API int pwpp_get_fxp_origins(pwpp_handle *h, float *out_xy, int capacity_bins);
/* Order of the points INSIDE a patch's part of the index lists (the parts themselves always follow the
 * reference: bin traversal order, TGR candidates at the end of their ring).
 *   PWPP_ORDER_SCATTER   (default) whatever the binning atomics produced -- same sets, fastest;
 *   PWPP_ORDER_REFERENCE the reference's order (patchworkpp.cpp:199 sorts a bin by z; :500,:532): ground
 *                        candidates ascending in z; non-ground: the points each R-VPF round removed, then
 *                        the rest, each ascending in z; small bins / RNR / out-of-range in cloud order.
 *                        Equal z: cloud order (the reference's std::sort leaves ties unspecified).
 * Applies to the batches launched after the call. */
enum { PWPP_ORDER_SCATTER = 0, PWPP_ORDER_REFERENCE = 1 };
PWPP_API int pwpp_set_output_order(pwpp_handle *h, int order);

/* Overlap mode (ON by default): batches of 128 frames or more are processed as two frame ranges -- binning
 * and index lists of both on the handle's main stream, each range's plane fits on a stream of its own -- so
 * that the stages of one range fill the wave slots the other leaves empty (binning and index lists are bound
 * by memory, the plane fits by their dependent chains).  Same results; per-kernel profiling (pwpp_set_profiling) and PWPP_ORDER_REFERENCE use the single-stream
 * schedule.  pwpp_set_overlap(h, 0) / PWPP_OVERLAP=0 select that schedule for everything. */
PWPP_API int pwpp_set_overlap(pwpp_handle *h, int on);
/* one-pass binning (fixed bin segments): batches launched that way and how many of
 * them had at least one frame redone on the exact two-pass path because a bin outgrew its segment.  Finishes the
 * batch in flight first.  No reference counterpart (its bins are unbounded vectors, patchworkpp.cpp:578-622). */
PWPP_API int pwpp_get_one_pass_stats(pwpp_handle *h, int64_t *batches, int64_t *redone);
/* ... and the same per FRAME: frames that went through one-pass binning, and how many of them were redone.  An
 * overflow costs the frames it happened in, not their batch (round 5): such a frame is binned again, exactly and
 * in place, while the other frames' results stand; only a frame too large for its own slots of the one-pass layout
 * (or the option "redo_whole_batch") sends the whole batch through the two-pass path, and then every frame counts. */
PWPP_API int pwpp_get_redo_stats(pwpp_handle *h, int64_t *frames_one_pass, int64_t *frames_redone);
/* ... and what keeps that rare (round 6): a part's segment holds ~1.06 x the largest count the part has had, and every frame owns an
 * OVERFLOW ARENA behind its segments -- a part that outgrows its segment is moved there as a whole by the scan kernel, on the device
 * (the cost of copying that part), and only a frame whose arena runs out goes back to the host.  frames_with_moved_parts counts the
 * frames that took that path; slots_per_frame / arena_slots describe the current table (0 before the first one-pass batch). */
PWPP_API int pwpp_get_arena_stats(pwpp_handle *h, int64_t *frames_with_moved_parts, int64_t *slots_per_frame, int64_t *arena_slots);


/* ---- batches in flight (no reference counterpart; round 5) --------------------------------------------------------------------
 * A pipe keeps `depth` batches of independent frames enqueued: it owns `depth` handles (each with its own workspace and HIP streams,
 * each on the single-stream schedule) and deals the submitted batches to them in turn.  pwpp_pipe_submit waits for the batch the
 * next handle launched `depth` submits ago (its results are complete then, and are replaced by the new batch), launches the new one
 * and returns that handle: the caller reads the results through the usual getters after pwpp_synchronize(handle), any time before
 * the handle comes round again.  The ramp-up of one batch (binning, nothing to overlap with) then runs under the ramp-down of
 * the one before (last plane fits, index lists): 2.32-2.46 instead of 2.49-2.63 ms per 1024-frame batch with depth 2
 * (profiles/r05_pipelined_batches.txt; depth 3: +0.5 %).  `mem` as in pwpp_estimate_ground_batch (PWPP_MEM_DEVICE or
 * PWPP_MEM_HOST_PINNED to stay asynchronous), `mode` likewise: PWPP_MODE_FRESH, or PWPP_MODE_STREAMS (round 6) for stateful streams
 * in disjoint GROUPS -- a stream's frames must stay in order on ONE handle, so handle g owns the streams of group g
 * (pwpp_pipe_set_num_streams sizes every handle's group) and the caller submits the groups round robin: submit k carries the next
 * frames of group k mod depth (reference use: demo_sequential.cpp:54-67, one long-lived object per stream).  depth 1..4.
 * The handles belong to the pipe: a pointer from pwpp_pipe_handle / pwpp_pipe_submit is INVALID after pwpp_pipe_destroy. */
typedef struct pwpp_pipe pwpp_pipe;
PWPP_API int pwpp_pipe_create(const pwpp_params *p, int device, int depth, pwpp_pipe **out);
PWPP_API int pwpp_pipe_submit(pwpp_pipe *pipe, const float *const *points, const int32_t *n, int frames, int cols, int layout, int mem,
                              int mode, pwpp_handle **holder);
/* pwpp_set_num_streams(streams_per_handle) on every handle of the pipe: `depth` groups of that many fresh streams */
PWPP_API int pwpp_pipe_set_num_streams(pwpp_pipe *pipe, int streams_per_handle);
/* waits for every batch in flight */
PWPP_API int pwpp_pipe_drain(pwpp_pipe *pipe);
/* the pipe's handles (options, statistics, getters): index 0 .. depth - 1; NULL beyond */
PWPP_API pwpp_handle *pwpp_pipe_handle(pwpp_pipe *pipe, int index);
PWPP_API int pwpp_pipe_destroy(pwpp_pipe *pipe);

/* Tuning and test switches (no reference counterpart).  The environment variables PWPP_DEBUG_FLAGS,
 * PWPP_FIT_PLAN, PWPP_FIT_CONCURRENT, PWPP_NO_ONE_PASS, PWPP_ONE_PASS_MIN_FRAMES, PWPP_ONE_PASS_SCALE,
 * PWPP_OVERLAP, PWPP_OVERLAP_MODE, PWPP_OVERLAP_RANGES, PWPP_FIT_STREAMS, PWPP_BIN_BLOCK, PWPP_HI_SPLIT,
 * PWPP_HI_SPLIT_ZONES and PWPP_EXACT_MOMENTS set the same options ONCE, in pwpp_create (which says so on stderr); nothing reads the
 * environment afterwards.  None of them changes a result, except the one that says so:
 *   "exact_moments"       "1" (default): the plane-fit sums of 4+ points exact on the reference's floats (2^-30 m grid, contract v4);
 *                         "0": rounds 3-5's 2^-21 m grid -- 7 % faster, off the reference by a few indices on 0.2 % of varied frames
 *                         (see pwpp_get_ground_indices above).  May be changed between calls.
 *   "split_k5"            "1" (default): calls with up to 64 stateful streams run K5 (GLE / TGR / thresholds) in two launches -- the index lists
 *                         wait for the first only; the statistics over the streams' A-GLE histories (two chains of ~1000 dependent f64 adds
 *                         in the reference's order) run on the handle's second stream, under K6 and the host's turn-around: one stream in steady
 *                         state 108 -> 100 us per frame.  "2": the second launch starts only when the lists are written (not beside K6): the
 *                         lists another ~4-5 us earlier, the state ~15 us later (a caller that steps the stream again at once waits for it there).  "0": one kernel
 *   "fuse_scan"           "1": fewer than eight frames run the part scan (K2) inside the binning kernel -- the workgroup that takes a frame's
 *                         last ticket scans (rounds 4-5's default).  "0" (default): a kernel of its own -- 1-2 us faster per frame since the
 *                         ticket is an agent-scope acquire-release
 *   "fit_plan"            which fit kernel handles which patch sizes, e.g. "W16:1023,W64.2:65535"; "" = automatic
 *   "fit_concurrent"      "1": the classes of a plan side by side on two streams
 *   "one_pass"            "0": always the two-pass binning
 *   "redo_whole_batch"    "1": a segment overflow of the one-pass binning redoes every frame of the batch (rounds 1-4) instead of
 *                         the frames that overflowed
 *   "one_pass_min_frames" smallest batch that takes the one-pass binning (default 1; rounds 1-3: 5 for stream batches, whose state
 *                         must be copied aside for a redo -- the binning pipeline now copies it itself, off the chain)
 *   "one_pass_scale"      scales the head-room of the one-pass segments (default 4 = 1.0625 x the largest count seen + 2 sqrt + 16 slots; tests
 *                         use small values to force overflows: first the arena, then the host's redo)
 *   "overlap_ranges"      frame ranges of the overlap mode (default 2; more were slower: 3.38 ms vs 2.86 ms with 4)
 *   "overlap_mode"        "1" (default): binning and lists on the main stream, the ranges' fits on "fit_streams" more;
 *                         "0": every range as a whole pipeline, alternating between two streams
 *   "fit_streams"         streams the ranges' fit stages are dealt to (1..8, default 2)
 *   "cu_split"            "N" or "N:mode" (experiment, default "0" = off): the overlap schedule's memory stream on N of the 256 CUs
 *                         and its fit streams on the others (hipExtStreamCreateWithCUMask).  Slower in every configuration
 *                         measured (profiles/r05_cu_mask_sweep.txt)
 *   "bin_block"           threads per workgroup of the one-pass binning kernel (128 / 256 / 512 / 1024, default 256)
 *   "hi_split"            metres above the ground level (-sensor_height) where the "high" part of a bin begins
 *                         (default 0.6; 1e30 = no high parts): the fit passes skip a high part whenever they can
 *                         prove that none of its points can enter the pass
 *   "hi_split_zones"      how many zones' bins are stored in two parts (0..4, default 1: the near zone)
 *   "debug_flags"         4: timing probes of the fit chain; 8: timing probes of the binning, scan and GLE kernels;
 *                         16: exact binning arithmetic only;
 *                         128: the first pass of the history statistics always as the reference's sequential sum (no exact shortcut);
 *                         256: fewer than eight frames: the scan as a kernel of its own (by default the last workgroup of the
 *                         binning kernel to finish a frame runs it in place);
 *                         64: before a call that skips the clearing kernel (the last call's K5 zeroed this call's counters),
 *                         read the counters back and fail with PWPP_E_STATE unless every word is zero;
 *                         2048: no overflow arena (a full segment sends its frame back to the host, as in rounds 2-5);
 *                         16384 / 32768: force the fall-back paths of the lowest-point selection
 * Returns PWPP_E_ARG for an unknown name or a value out of range. */
PWPP_API int pwpp_set_option(pwpp_handle *h, const char *name, const char *value);
/* The 64 timing probes of the last call when "debug_flags" has bit 2 set (device-side timestamps along the fit chain of the
 * largest patch, (code << 56) | 100 MHz ticks; slots 60-62: first start, last end, the patch's size): what tools/brows_chain.py
 * prints.  PWPP_E_STATE if no call has run.  A measurement aid; results never depend on it. */
PWPP_API int pwpp_debug_read(pwpp_handle *h, unsigned long long *out64);
/* Frees everything whose size follows the batch (a handle that processed one large batch otherwise keeps it, e.g. 9.7 GB
 * after 1024 KITTI frames with one-pass binning): inputs staged from the host, the bin-ordered planes, the index lists,
 * every per-frame table and patch record, the state of PWPP_MODE_FRESH frames and the one-pass snapshots.  Kept: the
 * streams' state (thresholds, histories, plane members) and the per-handle tables (a few KB).  The results of the last
 * call are gone: fetch them first.  The next call allocates again. */
PWPP_API int pwpp_trim_workspace(pwpp_handle *h);
/* device memory the handle holds right now, in bytes: EVERY device allocation of the handle (inputs handed over as device
 * buffers are the caller's); after pwpp_trim_workspace what is left is the streams' state and the per-handle tables */
PWPP_API int64_t pwpp_get_workspace_bytes(pwpp_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* PWPP_H */
