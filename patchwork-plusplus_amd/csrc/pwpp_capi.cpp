// pwpp_capi.cpp -- host side of libpwpp_hip.so: the C-ABI of include/pwpp.h.
//
// Owns the device workspace, the per-stream adaptive state and the HIP stream of a handle,
// validates parameters exactly where the reference constructor would read them
// (/root/reference/cpp/patchworkpp/include/patchwork/patchworkpp.h:120-150) and drives the six
// kernels of pwpp_kernels.hip.  There is no CPU fallback: without a GPU every compute entry
// point fails with PWPP_E_NODEVICE / PWPP_E_HIP.
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <new>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/pwpp.h"
#include "pwpp_dev.h"

extern "C" int pwpp_launch_pipeline(const PwppBatch *batch, hipStream_t stream, hipEvent_t *ev, hipStream_t aux,
                                    hipEvent_t aux_fork, hipEvent_t aux_join, unsigned long long *order_a, unsigned long long *order_b,
                                    int stages);
extern "C" int pwpp_launch_clear(const PwppBatch *batch, hipStream_t stream);
extern "C" int pwpp_launch_k5_tail(const PwppBatch *batch, hipStream_t aux, hipEvent_t aux_fork, hipEvent_t aux_join);
extern "C" int pwpp_launch_histogram(const PwppBatch *batch, hipStream_t stream);
extern "C" int pwpp_launch_gather_xyz(const PwppFrameDesc *fd, const int *idx, int count, float *out, hipStream_t stream);

static_assert(sizeof(pwpp_state) == sizeof(PwppStateScalar), "pwpp_state must mirror PwppStateScalar");

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(PWPP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n) {
        if (n <= cap) return PWPP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        // (head-room so that sizes creeping up do not reallocate every call: an eighth, at most 4 M elements -- an eighth of the
        // bin-ordered planes of a 1024-frame batch was 0.7 GB of nothing)
        const size_t want = n + (n / 8 < ((size_t)4 << 20) ? n / 8 : ((size_t)4 << 20)) + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) return fail(PWPP_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return PWPP_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <class T>
struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return PWPP_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 8 + 64;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return fail(PWPP_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return PWPP_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

const char *kKernelNames[PWPP_NUM_KERNELS] = {"k_czm_bin", "k_czm_scan", "k_czm_scatter", "k_fit_w64<16,64>", "k_fit_w64<64,p>",
                                              "k_fit[2]", "k_fit[3]", "k_fit[4]", "k_fit_stream", "k_gle_tgr", "k_emit"};  // fit slots: the classes of the big-batch plan (p = 4 big bins per wave on scans of KITTI density, 2 on denser ones)

std::atomic<bool> g_slot0_one_pass{false};  // diagnostic only (pwpp_kernel_name); handles on different threads may race to set it

}  // namespace

struct pwpp_handle {
    pwpp_params params;
    PwppDevParams dp;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;  // second stream for the latency plan (few frames)
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    bool k5_tail_unjoined = false;  // the last call left K5's second part on aux_stream (aux_join recorded behind it): the main stream has not waited for it yet
    bool k5_tail_unsynced = false;  // ... and neither has the host
    bool hist_check_due = false;    // the fill of the streams' histories after the last call has not been looked at yet (settle_k5_tail)
    int fuse_scan = 0;              // option "fuse_scan": K2 inside K1' for fewer than eight frames (rounds 4-5's default)
    int split_k5 = 1;               // option "split_k5": a few stateful streams run K5 in two launches (k_gle_tgr PART 1 / 2)
    bool overlap = true;   // pwpp_set_overlap: big batches as a pipeline of frame ranges over the two streams (default on)
    int overlap_ranges = 2;  // (more ranges were slower at every setting tried: each range's fit kernels end with the tail of their
                             // longest waves -- docs/history/design_through_round4.md section 9)
    int overlap_mode = 1;  // option "overlap_mode": 1 = memory stream / fit stream pipeline, 0 = whole ranges alternating between the streams
    std::vector<hipEvent_t> ev_ranges;  // two per frame range: binned, fitted
    int bin_block = 256;                // option "bin_block": threads per workgroup of the one-pass binning kernel
    int num_fit_streams = 2;            // option "fit_streams": streams the ranges' fit stages are dealt to (aux_stream + extra_streams)
    std::vector<hipStream_t> extra_streams;
    // option "cu_split" (round 5 experiment): the overlap schedule's memory stream and fit streams on disjoint
    // sets of CUs (hipExtStreamCreateWithCUMask).  cu_split_mem = CUs of the memory stream (0 = off), cu_split_mode: which bits.
    int cu_split_mem = 0, cu_split_mode = 0;
    hipStream_t masked_mem = nullptr;
    std::vector<hipStream_t> masked_fit;
    // tuning / test options (pwpp_set_option; the PWPP_* environment variables are read ONCE, in pwpp_create)
    int debug_flags = 0;
    std::string fit_plan;
    bool fit_concurrent = false;
    bool no_one_pass = false;
    int one_pass_min_frames = 1;     // stream batches (round 4: the state a redo starts from is copied by extra workgroups of k_czm_scan, no
                                     // copy commands in front of the pipeline any more: one stream 112.8 -> 106.9 us of GPU time, 139 -> 131 us per call)
    int one_pass_min_fresh = 1;      // ... and batches of FRESH frames (a single frame: 118 -> 107 us in round 3)
    double one_pass_scale = 4.0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_k[PWPP_NUM_KERNELS + 1] = {};
    bool profiling = false;
    bool profile_pending = false;
    double prof_ms[PWPP_NUM_KERNELS] = {};
    int64_t prof_launches[PWPP_NUM_KERNELS] = {};

    // last call
    int frames = 0;
    int mode = PWPP_MODE_FRESH;
    bool have_results = false;
    bool pending = false;  // launches in flight, results not yet fetched
    double time_us = 0.0;
    std::vector<PwppFrameDesc> descs;  // host copy of the last batch (device pointers inside)
    std::vector<PwppFrameDesc> descs_on_device;  // what d_frames holds
    const PwppFrameDesc *descs_dev_ptr = nullptr;
    // one-pass binning (fixed bin segments, DESIGN.md section 3 K1'): state of the batch in flight
    bool one_pass = false;           // the batch in flight uses fixed segments; overflow flags are checked when it lands
    int one_pass_holdoff = 0;        // batches to run on the two-pass path after an overflow
    int64_t slots_per_frame = 0;     // of the current capacity table: the parts' segments + the frame's overflow arena
    uint32_t arena_base = 0, arena_slots = 0, arena_spill = 0;  // the arena of the current table (pwpp_dev.h): first slot, slots (0: none), room for spilled records
    DevBuf<uint2> d_arena_tag;       // [frames][arena_spill]
    int cap_max_n = -1;              // largest frame the capacity table on the device was built for
    bool cap_few = false;            // ... and whether it was built with the head-room of calls of a few frames
    int max_n = 0;
    int64_t total_points = 0;
    int cols = 4, layout = 0;
    long long one_pass_batches = 0, one_pass_redone = 0;  // batches on the one-pass path; of those, batches with at least one frame redone
    long long one_pass_frames = 0, one_pass_redone_frames = 0;  // the same per frame (pwpp_get_redo_stats)
    std::vector<uint8_t> frame_two_pass;  // frames of the last one-pass batch that were redone in place (compact layout at their own first slot)
    bool redo_whole_batch = false;        // option "redo_whole_batch": an overflow redoes every frame of the batch (rounds 1-4; tests, A/B)
    long long fixed_up_frames = 0;   // frames finished by k_fit_fixup (a patch needed the plane fitted before it)
    long long arena_frames = 0;      // frames in which k_czm_scan moved overgrown parts into the overflow arena (PwppFrameResult.overflow bit 3)
    long long clamped_frames = 0;    // frames with a patch whose final ground set spanned more than z0 +- ZR (PwppFrameResult.overflow bit 2)

    // workspace
    DevBuf<PwppFrameDesc> d_frames;
    PinnedBuf<PwppFrameDesc> h_frames;
    PinnedBuf<int64_t> h_base;
    DevBuf<float> d_in;  // staging for host inputs
    DevBuf<uint16_t> d_codes;
    DevBuf<float> d_sorted_z;     // the bin-ordered planes (pwpp_dev.h): z, {x, y}, cloud index
    DevBuf<float2> d_sorted_xy;
    DevBuf<int> d_sorted_idx;
    DevBuf<float2> d_bin_origin;  // [B] origins of the fixed-point plane-fit sums
    DevBuf<float4> d_bin_bbox;    // [B] {xmin, xmax, ymin, ymax} of every bin (the fit kernels' skip test of the high parts)
    DevBuf<uint8_t> d_member;     // membership plane (pwpp_dev.h, PWPP_SLOT_ALIGN): one bit per slot + PWPP_MEMBER_PAD bytes per part
    DevBuf<int32_t> d_out;
    DevBuf<unsigned long long> d_ord_a, d_ord_b;  // scratch of the reference-order mode (long sub-lists)
    DevBuf<uint32_t> d_ord_work;                  // ... and its work lists: the sub-lists above 256 entries of every frame
    int output_order = PWPP_ORDER_SCATTER;
    DevBuf<uint32_t> d_bins;   // 4 slabs of frames*(B+2): bin_count, bin_off, dst_a, dst_b
    DevBuf<uint32_t> d_parts;  // TWO copies of 3 slabs of frames*(2B+2): part_count, part_off, part_cursor (pwpp_dev.h: a bin is stored in two
                               // parts); a call works on one copy while its K5 zeroes the other for the next call (PwppBatch.next_part_count)
    DevBuf<uint32_t> d_cls_start;  // frames * 8
    DevBuf<uint32_t> d_cap_off;    // 2B + 3 segment starts of the one-pass path (one segment per part)
    DevBuf<uint32_t> d_bin_max;    // 2B + 2: largest count of every part so far (k_czm_scan)
    DevBuf<uint8_t> d_emit_long;   // k_emit's table of long bins: B + 2 flags, then (4-byte aligned) the list of those bins as uint16
    std::vector<uint8_t> emit_long_flags;
    int emit_long_n = 0, emit_long_parts = 1;
    std::vector<uint32_t> observed, cap_table, cap_seen;  // host copies: d_bin_max as last read; capacities of the table on the device and the counts it was built from
    bool have_observation = false, table_stale = true;
    DevBuf<PwppFrameDesc> d_frames_probe;
    PinnedBuf<uint32_t> h_bin_max;
    DevBuf<uint16_t> d_cls_list;   // frames * B
    DevBuf<PwppPatchRec> d_recs;
    DevBuf<float> d_centers, d_normals;
    DevBuf<PwppFrameResult> d_results;  // two copies (see d_parts)
    int counters_copy = 0;           // which copy of d_parts / d_results the call in flight (or the last one) works on
    bool next_clean = false;         // the OTHER copy was zeroed by the last call's K5 ...
    int next_clean_frames = 0, next_clean_slabs = 0;   // ... for a call of this many frames and slabs (1: one-pass, 3: two-pass)
    const void *next_clean_parts = nullptr, *next_clean_results = nullptr;  // ... in these allocations
    size_t next_clean_parts_cap = 0, next_clean_results_cap = 0;
    PinnedBuf<PwppFrameResult> h_results;
    DevBuf<float> d_xyz;  // gather scratch
    DevBuf<unsigned long long> d_dbg;  // timing probes (PWPP_DEBUG_FLAGS & 4)

    // adaptive state: streams (long history slabs) and per-frame fresh outputs (short slabs)
    int num_streams = 0;
    int stream_hist_cap = 0, fresh_hist_cap = 0;
    int max_pushes_per_frame = 0;  // bins of the widest ring of interest: what one frame can add to a history
    DevBuf<PwppStateScalar> d_st_stream, d_st_fresh;
    DevBuf<double> d_hist_stream, d_hist_fresh;
    DevBuf<PwppPlaneState> d_pl_stream, d_pl_fresh, d_pl_snap;  // the reference object's plane members after a frame (pwpp_dev.h)
    // one-pass batches of stateful streams: the state of the streams before the batch, so that a redo after an
    // overflow starts from it (the first attempt has already advanced sensor height, thresholds, histories)
    DevBuf<PwppStateScalar> d_st_snap;
    DevBuf<double> d_hist_snap;
};

namespace {

// The fixed-point contract of the plane-fit sums: every bin sums its points around an
// ORIGIN, its polar centre rounded to 1/8 m (a sector wider than a quarter turn keeps the sensor), and the
// shift s is the largest one <= 30 that keeps every point of every bin within 2^35 grid steps of its origin (contract v4, `wide`;
// v3 of rounds 3-5, option "exact_moments" = 0: <= 21 and 2^26).  Returns the z half-range ZR = 2^35 / 2^s (2^26 / 2^s) metres.
double fxp_geometry(const PwppDevParams &d, std::vector<float2> &origin, int &shift, bool wide) {
    double rmax = 0.0;
    origin.clear();
    for (int z = 0; z < 4; ++z)
        for (int r = 0; r < d.rings[z]; ++r)
            for (int k = 0; k < d.sectors[z]; ++k) {
                const double r0 = d.min_ranges[z] + r * d.ring_sizes[z];
                const double r1 = (z == 3 && r == d.rings[z] - 1) ? d.max_range : r0 + d.ring_sizes[z];
                const double t0 = k * d.sector_sizes[z], t1 = (k + 1) * d.sector_sizes[z];
                double cx = 0.0, cy = 0.0, far = r1;
                if (d.sectors[z] >= 4) {
                    const double rc = 0.5 * (r0 + r1), tc = 0.5 * (t0 + t1);
                    cx = std::rint(rc * std::cos(tc) * 8.0) / 8.0;
                    cy = std::rint(rc * std::sin(tc) * 8.0) / 8.0;
                    far = 0.0;
                    const double cr[2] = {r0, r1}, ct[2] = {t0, t1};
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            const double dx = cr[a] * std::cos(ct[b]) - cx, dy = cr[a] * std::sin(ct[b]) - cy;
                            far = std::max(far, std::sqrt(dx * dx + dy * dy));
                        }
                }
                origin.push_back(make_float2((float)cx, (float)cy));
                rmax = std::max(rmax, far);
            }
    const double qmax = wide ? 34359738368.0 : 67108864.0;
    int s = wide ? 30 : 21;
    while (s > 0 && (rmax + 0.01) * (double)(1 << s) > qmax) --s;
    shift = s;
    return qmax / (double)(1 << s);
}

// Bounding box {xmin, xmax, ymin, ymax} of every bin: an annular sector reaches its extremes at its four corners or where
// it crosses an axis.  A little generous (1e-4 of the radius + 1 mm; the binning itself decides in double, and a point on
// the positive x axis belongs to the LAST sector, patchworkpp.cpp:570): the fit kernels only use it to prove that no point
// of a bin's high part can lie below a plane (stage_needs_hi, pwpp_fit.hip), so too large is safe and too small is not.
void bin_boxes(const PwppDevParams &d, std::vector<float4> &box) {
    box.clear();
    for (int z = 0; z < 4; ++z)
        for (int r = 0; r < d.rings[z]; ++r)
            for (int k = 0; k < d.sectors[z]; ++k) {
                const double r0 = d.min_ranges[z] + r * d.ring_sizes[z];
                const double r1 = (r == d.rings[z] - 1) ? (z == 3 ? d.max_range : d.min_ranges[z + 1]) : r0 + d.ring_sizes[z];
                const double t0 = k * d.sector_sizes[z], t1 = (k == d.sectors[z] - 1) ? 2 * M_PI : (k + 1) * d.sector_sizes[z];
                std::vector<double> th = {t0, t1};
                for (int q = 0; q <= 4; ++q)
                    if (q * (M_PI / 2) > t0 && q * (M_PI / 2) < t1) th.push_back(q * (M_PI / 2));
                double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
                for (double t : th)
                    for (double rr : {std::min(r0, r1), std::max(r0, r1)}) {
                        xmin = std::min(xmin, rr * std::cos(t));
                        xmax = std::max(xmax, rr * std::cos(t));
                        ymin = std::min(ymin, rr * std::sin(t));
                        ymax = std::max(ymax, rr * std::sin(t));
                    }
                const double m = 1e-4 * std::max(r0, r1) + 1e-3;
                box.push_back(make_float4((float)(xmin - m), (float)(xmax + m), (float)(ymin - m), (float)(ymax + m)));
            }
}

int build_dev_params(const pwpp_params &p, PwppDevParams &d) {
    if (p.num_zones != 4)
        return fail(PWPP_E_UNSUPPORTED, "num_zones=%d: the reference hard-codes four zones (patchworkpp.h:122-134)", p.num_zones);
    if (p.num_iter < 1) return fail(PWPP_E_UNSUPPORTED, "num_iter=%d: must be >= 1", p.num_iter);
    if (p.num_lpr > PWPP_MAX_LPR) return fail(PWPP_E_UNSUPPORTED, "num_lpr=%d: at most %d supported", p.num_lpr, PWPP_MAX_LPR);
    if (p.num_rings_of_interest < 0 || p.num_rings_of_interest > PWPP_MAX_ROI)
        return fail(PWPP_E_ARG, "num_rings_of_interest=%d: the reference keeps update_*_[4] (patchworkpp.h:174-175)", p.num_rings_of_interest);
    if (!(p.max_range > p.min_range)) return fail(PWPP_E_ARG, "max_range must exceed min_range");
    if (!(p.max_range <= 8388607.0)) return fail(PWPP_E_UNSUPPORTED, "max_range=%g: at most 2^23 - 1 m supported", p.max_range);
    if (p.max_flatness_storage < 0 || p.max_elevation_storage < 0) return fail(PWPP_E_ARG, "negative history storage");
    std::memset(&d, 0, sizeof(d));
    int bins = 0, total_rings = 0, near = 0, max_near_sectors = 0;
    for (int k = 0; k < 4; ++k) {
        if (p.num_rings_each_zone[k] < 1 || p.num_sectors_each_zone[k] < 1)
            return fail(PWPP_E_ARG, "zone %d: rings and sectors must be >= 1", k);
        d.rings[k] = p.num_rings_each_zone[k];
        d.sectors[k] = p.num_sectors_each_zone[k];
        d.bin_base[k] = bins;
        for (int r = 0; r < d.rings[k]; ++r) {
            if (total_rings < p.num_rings_of_interest) {
                near += d.sectors[k];
                if (d.sectors[k] > max_near_sectors) max_near_sectors = d.sectors[k];
            }
            ++total_rings;
        }
        bins += d.rings[k] * d.sectors[k];
    }
    d.bin_base[4] = bins;
    if (bins > PWPP_MAX_BINS) return fail(PWPP_E_UNSUPPORTED, "%d CZM bins: at most %d supported", bins, PWPP_MAX_BINS);
    if (near > PWPP_MAX_NEAR_BINS) return fail(PWPP_E_UNSUPPORTED, "%d bins inside the rings of interest: at most %d", near, PWPP_MAX_NEAR_BINS);
    d.num_bins = bins;
    d.near_bins = near;
    // CZM geometry, the reference constructor's expressions in double (patchworkpp.h:122-134)
    const double mn = p.min_range, mx = p.max_range;
    const double z2 = (7 * mn + mx) / 8.0, z3 = (3 * mn + mx) / 4.0, z4 = (mn + mx) / 2.0;
    d.min_ranges[0] = mn;
    d.min_ranges[1] = z2;
    d.min_ranges[2] = z3;
    d.min_ranges[3] = z4;
    d.ring_sizes[0] = (z2 - mn) / p.num_rings_each_zone[0];
    d.ring_sizes[1] = (z3 - z2) / p.num_rings_each_zone[1];
    d.ring_sizes[2] = (z4 - z3) / p.num_rings_each_zone[2];
    d.ring_sizes[3] = (mx - z4) / p.num_rings_each_zone[3];
    for (int k = 0; k < 4; ++k) d.sector_sizes[k] = 2 * M_PI / p.num_sectors_each_zone[k];
    d.enable_RNR = p.enable_RNR != 0;
    d.enable_RVPF = p.enable_RVPF != 0;
    d.enable_TGR = p.enable_TGR != 0;
    d.num_iter = p.num_iter;
    d.num_lpr = p.num_lpr < 0 ? 0 : p.num_lpr;
    d.num_rings_of_interest = p.num_rings_of_interest;
    d.min_pts = (uint64_t)(size_t)p.num_min_pts;  // int -> size_t as in `size() < params_.num_min_pts` (patchworkpp.cpp:191)
    d.RNR_ver_angle_thr = p.RNR_ver_angle_thr;
    d.RNR_intensity_thr = p.RNR_intensity_thr;
    d.sensor_height = p.sensor_height;
    d.th_seeds = p.th_seeds;
    d.th_dist = p.th_dist;
    d.th_seeds_v = p.th_seeds_v;
    d.th_dist_v = p.th_dist_v;
    d.max_range = p.max_range;
    d.min_range = p.min_range;
    d.uprightness_thr = p.uprightness_thr;
    d.margin = p.adaptive_seed_selection_margin;
    d.fxp_shift = 0;  // (pwpp_create: fxp_geometry)
    d.f_min_range = (float)p.min_range;
    d.f_max_range = (float)p.max_range;
    d.f_margin_r = (float)(p.max_range * 1.5e-6 + 5e-5);  // metres; ~5x the float error budget of the radius
    d.f_margin_t = 8e-6f;                                  // radians; ~13x the error of the float angle
    for (int k = 0; k < 4; ++k) {
        d.f_zone[k] = (float)d.min_ranges[k];
        d.f_inv_ring[k] = (float)(1.0 / d.ring_sizes[k]);
        d.f_inv_sector[k] = (float)(1.0 / d.sector_sizes[k]);
    }
    d.max_elev_storage = p.max_elevation_storage;
    d.max_flat_storage = p.max_flatness_storage;
    for (int k = 0; k < 4; ++k) {
        d.elevation_thr0[k] = p.elevation_thr[k];
        d.flatness_thr0[k] = p.flatness_thr[k];
    }
    return max_near_sectors;  // >= 0
}

int use_device(pwpp_handle *h) {
    HIPCHK(hipSetDevice(h->device));
    return PWPP_OK;
}

void fill_default_state(const pwpp_handle *h, PwppStateScalar &s) {
    std::memset(&s, 0, sizeof(s));
    s.sensor_height = h->params.sensor_height;
    for (int k = 0; k < 4; ++k) {
        s.elevation_thr[k] = h->params.elevation_thr[k];
        s.flatness_thr[k] = h->params.flatness_thr[k];
    }
}

int finish_pending(pwpp_handle *h, bool lists_only = false);
extern "C" const char *pwpp_big_batch_plan(int max_n, int num_bins, int wide);  // pwpp_fit.hip

// every stream a schedule may have put work on (error paths, pwpp_destroy): the main stream alone is not the join of a
// schedule that stopped half way
void sync_all_streams(pwpp_handle *h) {
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    for (hipStream_t st : h->extra_streams) (void)hipStreamSynchronize(st);
    if (h->masked_mem) (void)hipStreamSynchronize(h->masked_mem);
    for (hipStream_t st : h->masked_fit) (void)hipStreamSynchronize(st);
}

// the stream history slabs [stream][2][4][cap] re-laid out for a larger cap (contents kept)
int grow_stream_histories(pwpp_handle *h, int new_cap) {
    new_cap = (new_cap + 1) & ~1;  // (even: k_gle_tgr fetches two entries per load)
    const size_t rows = (size_t)h->num_streams * 8;
    DevBuf<double> bigger;
    int rc = bigger.ensure(rows * (size_t)new_cap);
    if (rc) return rc;
    HIPCHK(hipMemcpy2D(bigger.p, (size_t)new_cap * sizeof(double), h->d_hist_stream.p, (size_t)h->stream_hist_cap * sizeof(double),
                       (size_t)h->stream_hist_cap * sizeof(double), rows, hipMemcpyDeviceToDevice));
    h->d_hist_stream.release();
    h->d_hist_stream = bigger;
    h->stream_hist_cap = new_cap;
    return PWPP_OK;
}

// Segment sizes of the one-pass path.  Round 6: a part's segment holds 1.0625 x the largest count that part has had in any frame
// this handle has seen (d_bin_max, kept by k_czm_scan; before the first batch a histogram of up to 256 sample frames fills it:
// probe_histogram) + 2 sqrt(that) + 16 slots, and every frame has an OVERFLOW ARENA behind its segments (pwpp_dev.h): a part that
// outgrows its segment is moved there by k_czm_scan, on the device, at the cost of a copy of that part -- ~1.7 slots per point of a
// KITTI frame, 27 B per point.  (Rounds 2-5: 1.5 x + 256 slots and no arena -- 2.7 slots per point, every overflow a frame binned
// twice by the host; round 1: 20 slots per point.)  one_pass_scale scales the 1.0625 (tests use a tiny one to force overflows).
// A frame whose arena runs out raises the overflow flag and is binned again on the exact two-pass path, whose counts enter
// d_bin_max, and the table is rebuilt.
int build_capacity_table(pwpp_handle *h, int max_n) {
    const PwppDevParams &P = h->dp;
    const int NB = PWPP_NUM_PARTS(P.num_bins);  // one segment per part
    // A FEW frames per call (a single sensor's stream, the reference's own use): memory is no concern there (2 048 slots more per part
    // are 40 MB per frame), but every overflow is a frame binned twice (the fused binning + scan kernel of fewer than eight frames has
    // no arena), and a handle that has seen a handful of frames knows little about its bins -- 3.75 x the largest count + 2048.
    const bool few = h->frames <= 16 && h->one_pass_scale >= 1.0;
    h->cap_few = few;
    const double scale = (few ? 3.75 : 1.0625) * h->one_pass_scale / 4.0;
    std::vector<uint32_t> off((size_t)NB + 1);
    h->cap_table.assign((size_t)NB, 0u);
    h->cap_seen = h->observed;
    uint64_t run = 0;
    for (int b = 0; b < NB; ++b) {
        off[(size_t)b] = (uint32_t)run;
        const double seen = (double)h->observed[(size_t)b];
        double cap = scale * seen + (h->one_pass_scale >= 1.0 ? (few ? 2048.0 : 2.0 * std::sqrt(seen) + 16.0) : 16.0);
        if (cap > (double)max_n + 16.0) cap = (double)max_n + 16.0;
        uint64_t c = ((uint64_t)cap + (PWPP_SLOT_ALIGN - 1)) & ~(uint64_t)(PWPP_SLOT_ALIGN - 1);
        if (b < 2 * P.num_bins && (b & 1) && b / 2 >= P.split_end) c = 0;  // the high part of a bin that is not split: never used
        h->cap_table[(size_t)b] = (uint32_t)c;
        run += c;
        if (run >= ((uint64_t)1 << 31)) return fail(PWPP_E_NOMEM, "one-pass capacity table overflows 31-bit offsets");
    }
    off[(size_t)NB] = (uint32_t)run;
    // the arena: room for the largest part to move as a whole, plus a sixteenth of the largest frame -- 4 096 slots at least; a
    // quarter of it (2 048 at least) may fill with spilled records
    uint64_t arena = 0;
    if (!few && !(h->debug_flags & 2048)) {  // (debug 2048: no arena -- rounds 2-5's behaviour on today's segments, for tests and A/B runs)
        uint32_t biggest = 0;
        for (int b = 0; b < NB; ++b) biggest = h->observed[(size_t)b] > biggest ? h->observed[(size_t)b] : biggest;
        arena = (uint64_t)biggest + biggest / 4 + (uint64_t)max_n / 16 + 4096;
        if (h->one_pass_scale < 1.0) arena = (uint64_t)((double)arena * h->one_pass_scale);  // (tests: a small arena overflows too)
        arena = (arena + (PWPP_SLOT_ALIGN - 1)) & ~(uint64_t)(PWPP_SLOT_ALIGN - 1);
        if (run + arena >= ((uint64_t)1 << 31)) arena = 0;
    }
    int rc = h->d_cap_off.ensure((size_t)NB + 1);
    if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_cap_off.p, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    h->arena_base = (uint32_t)run;
    h->arena_slots = (uint32_t)arena;
    h->arena_spill = (uint32_t)(arena / 4 > 2048 ? arena / 4 : (arena < 2048 ? arena : 2048));
    h->slots_per_frame = (int64_t)(run + arena);
    h->cap_max_n = max_n;
    h->table_stale = false;
    return PWPP_OK;
}

void fill_batch(pwpp_handle *h, PwppBatch &bt);
// Words of ONE copy of the part counters (part_count, part_off, part_cursor: three slabs of frames x parts words).  A multiple of
// four words with four to spare: k_clear zeroes whole 16-byte words from the copy's first word, so the SECOND copy must start
// on a 16-byte boundary too (3 x frames x parts + 4 is 8 mod 16 bytes for a single frame of the default CZM).
inline size_t counters_copy_words(size_t slab_words) { return (3 * slab_words + 4 + 3) & ~(size_t)3; }

int read_observed(pwpp_handle *h) {
    const size_t NB = (size_t)PWPP_NUM_PARTS(h->dp.num_bins);
    h->observed.resize(NB);
    HIPCHK(hipMemcpy(h->observed.data(), h->d_bin_max.p, NB * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PWPP_OK;
}

// Histogram of up to 256 frames spread over the batch described by h->descs (inputs already on their way to the device):
// K0 + K1 + K2 on a private descriptor array, which leaves the bins' largest counts in d_bin_max.
int probe_histogram(pwpp_handle *h) {
    // (round 5: up to 256 sample frames -- with 32, 353 of the first 512 frames of a varied drive outgrew some segment (bench.py,
    // `distinct` leg); the probe is a histogram pass, ~1 us per frame)
    const int S = h->frames < 256 ? h->frames : 256;
    std::vector<PwppFrameDesc> sample((size_t)S);
    int max_n = 0;
    for (int i = 0; i < S; ++i) {
        // (scattered over the batch, not evenly spaced: replayed or periodic inputs would alias with a fixed stride)
        const size_t pick = S == h->frames ? (size_t)i : (size_t)(((uint64_t)(i + 1) * 2654435761ull) % (uint64_t)h->frames);
        sample[(size_t)i] = h->descs[pick];
        if (h->mode != PWPP_MODE_FRESH) sample[(size_t)i].state_in = sample[(size_t)i].state_out;  // (RNR reads the stream's sensor height)
        sample[(size_t)i].state_out = i;  // never written: the probe stops after K2
        if (sample[(size_t)i].n > max_n) max_n = sample[(size_t)i].n;
    }
    int rc = h->d_frames_probe.ensure((size_t)S);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->d_frames_probe.p, sample.data(), (size_t)S * sizeof(PwppFrameDesc), hipMemcpyHostToDevice, h->stream));
    PwppBatch bt;
    fill_batch(h, bt);
    bt.frames = h->d_frames_probe.p;
    bt.num_frames = S;
    bt.max_n = max_n;
    bt.cap_off = nullptr;
    h->next_clean = false;  // (the probe's own K0-K2 run on sample frames with their own frame count)
    bt.next_part_count = nullptr;
    const int lrc = pwpp_launch_histogram(&bt, h->stream);
    if (lrc != 0) return fail(PWPP_E_HIP, "histogram probe failed: %s", hipGetErrorString((hipError_t)lrc));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_observation = true;
    return read_observed(h);
}

// k_emit's table of "long" bins (pwpp_dev.h: emit_long) from the parts' largest counts so far; uploaded when it changes.
// Called with no launch in flight (estimate_batch, after finish_pending).
int refresh_emit_long(pwpp_handle *h) {
    const int B = h->dp.num_bins, NB = B + 2;
    std::vector<uint8_t> flag((size_t)NB, 0);
    std::vector<uint16_t> list;
    uint32_t biggest = 0;
    for (int b = 0; b < NB && h->observed.size() == (size_t)PWPP_NUM_PARTS(B) && !(h->debug_flags & 1024); ++b) {  // (debug 1024: no long-list table)
        const uint32_t c = b < B ? h->observed[2 * (size_t)b] + h->observed[2 * (size_t)b + 1] : h->observed[(size_t)B + b];  // (pseudo-bin s = part B + s)
        if (c > PWPP_EMIT_LONG_MIN) {
            flag[(size_t)b] = 1;
            list.push_back((uint16_t)b);
            biggest = c > biggest ? c : biggest;
        }
    }
    int parts = (int)((biggest > 512u * PWPP_EMIT_LONG_BLOCKS ? biggest - 512u * PWPP_EMIT_LONG_BLOCKS : 0u) / 4096u) + 1;
    parts = parts > 32 ? 32 : parts;
    if (flag == h->emit_long_flags && parts == h->emit_long_parts) return PWPP_OK;
    const size_t list_at = (size_t)((NB + 3) & ~3);
    std::vector<uint8_t> blob(list_at + 2 * (size_t)NB, 0);
    std::memcpy(blob.data(), flag.data(), flag.size());
    if (!list.empty()) std::memcpy(blob.data() + list_at, list.data(), list.size() * sizeof(uint16_t));
    int rc = h->d_emit_long.ensure(blob.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_emit_long.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    h->emit_long_flags = flag;
    h->emit_long_n = (int)list.size();
    h->emit_long_parts = parts;
    return PWPP_OK;
}

// everything a launch needs except the frame range, the capacity table and the events
void fill_batch(pwpp_handle *h, PwppBatch &bt) {
    const int B = h->dp.num_bins, NB = B + 2;
    std::memset(&bt, 0, sizeof(bt));
    bt.P = h->dp;
    bt.frames = h->d_frames.p;
    bt.num_frames = h->frames;
    bt.max_n = h->max_n;
    bt.debug = h->debug_flags;
    bt.fit_plan = h->fit_plan.empty() ? nullptr : h->fit_plan.c_str();
    bt.plan_frames = 0;
    bt.fit_concurrent = h->fit_concurrent ? 1 : 0;
    bt.bin_block = h->bin_block;
    if (h->mode == PWPP_MODE_FRESH) {
        bt.P.hist_cap = h->fresh_hist_cap;
        bt.st_scalar = h->d_st_fresh.p;
        bt.st_hist = h->d_hist_fresh.p;
        bt.st_plane = h->d_pl_fresh.p;
    } else {
        bt.P.hist_cap = h->stream_hist_cap;
        bt.st_scalar = h->d_st_stream.p;
        bt.st_hist = h->d_hist_stream.p;
        bt.st_plane = h->d_pl_stream.p;
    }
    bt.codes = h->d_codes.p;
    const size_t slab = (size_t)h->frames * NB, pslab = (size_t)h->frames * PWPP_NUM_PARTS(B);
    bt.bin_count = h->d_bins.p;
    bt.bin_off = h->d_bins.p + slab;
    bt.dst_a = h->d_bins.p + 2 * slab;
    bt.dst_b = h->d_bins.p + 3 * slab;
    const size_t pcopy = counters_copy_words(pslab);
    uint32_t *parts = h->d_parts.p + (size_t)h->counters_copy * pcopy;
    bt.part_count = parts;
    bt.part_off = parts + pslab;
    bt.part_cursor = parts + 2 * pslab;
    bt.next_part_count = h->d_parts.p + (size_t)(h->counters_copy ^ 1) * pcopy;
    bt.next_results = h->d_results.p + (size_t)(h->counters_copy ^ 1) * (size_t)h->frames;
    bt.next_slab_stride = (int64_t)pslab;
    bt.next_slabs = 3;  // (launch_prepared narrows it to 1 for a one-pass batch)
    bt.cls_start = h->d_cls_start.p;
    bt.cls_list = h->d_cls_list.p;
    bt.sorted_z = h->d_sorted_z.p;
    bt.sorted_xy = h->d_sorted_xy.p;
    bt.sorted_idx = h->d_sorted_idx.p;
    bt.bin_origin = h->d_bin_origin.p;
    bt.bin_bbox = h->d_bin_bbox.p;
    bt.member = h->d_member.p;
    bt.order_work = h->output_order == PWPP_ORDER_REFERENCE ? h->d_ord_work.p : nullptr;
    bt.recs = h->d_recs.p;
    bt.out_idx = h->d_out.p;
    bt.centers = h->d_centers.p;
    bt.normals = h->d_normals.p;
    bt.results = h->d_results.p + (size_t)h->counters_copy * (size_t)h->frames;
    bt.results_host = h->h_results.p;  // hipHostMalloc'ed: the same address on the device
    bt.dbg = h->d_dbg.p;

    bt.bin_max = h->d_bin_max.p;
    bt.arena_base = 0xffffffffu;  // (launch_prepared sets the arena of a one-pass batch)
    bt.arena_slots = bt.arena_spill = 0u;
    bt.arena_tag = nullptr;
    {   // k_emit: one wave per bin copies a list of a few thousand entries well; the bins of a dense cloud get more
        uint32_t biggest = 0;  // (observed: per part; a bin's two parts are neighbours, the pseudo-bins the last two entries)
        for (size_t b = 0; b + 1 < h->observed.size(); b += 2) {
            const uint32_t both = b < 2 * (size_t)B ? h->observed[b] + h->observed[b + 1] : (h->observed[b] > h->observed[b + 1] ? h->observed[b] : h->observed[b + 1]);
            biggest = both > biggest ? both : biggest;
        }
        int parts = (int)(biggest / 8192u) + 1;
        // a few frames: the kernel is as long as the copy of its longest list by ONE wave (a 4 900-point patch: ten rounds of
        // 512 entries, ~1 us each) -- eight waves per bin cut that chain (single frame: k_emit 13.2 -> see DESIGN.md), where a
        // big batch would only pay for the empty waves (0.24 -> 0.30 ms with two per bin)
        if (h->frames <= 8) parts = 8;
        else if (h->frames <= 64 && parts < 4) parts = 4;
        bt.emit_parts = parts > 8 ? 8 : parts;
        if (h->frames > 64) {
            // Big batches (round 5): ONE wave per bin, and extra waves only for the bins that have held more than PWPP_EMIT_LONG_MIN
            // entries in some frame of this handle (refresh_emit_long: a table on the device, rebuilt when the maxima change).
            // A long list in a bin that is not in the table yet is copied whole by its one wave: slower, never wrong.
            bt.emit_parts = 1;
            if (h->emit_long_n > 0) {
                bt.emit_long = h->d_emit_long.p;
                bt.emit_long_list = reinterpret_cast<const uint16_t *>(h->d_emit_long.p + (size_t)((B + 2 + 3) & ~3));
                bt.emit_long_n = h->emit_long_n;
                bt.emit_long_parts = h->emit_long_parts;
            }
        }
    }
}

// a view of the batch's workspaces that covers the frames [f0, f0 + nf): every per-frame array is indexed by the frame
PwppBatch frame_range(const pwpp_handle *h, const PwppBatch &bt, int f0, int nf) {
    const int B = h->dp.num_bins, NB = B + 2, NP = PWPP_NUM_PARTS(B);
    PwppBatch v = bt;
    v.frames += f0;
    v.num_frames = nf;
    v.no_clear = 1;
    v.bin_count += (size_t)f0 * NB;
    v.bin_off += (size_t)f0 * NB;
    v.part_count += (size_t)f0 * NP;
    v.part_off += (size_t)f0 * NP;
    v.part_cursor += (size_t)f0 * NP;
    v.dst_a += (size_t)f0 * NB;
    v.dst_b += (size_t)f0 * NB;
    v.cls_start += (size_t)f0 * PWPP_CLS_STRIDE;
    v.cls_list += (size_t)f0 * B;
    v.recs += (size_t)f0 * B;
    v.centers += (size_t)f0 * B * 3;
    v.normals += (size_t)f0 * B * 3;
    v.results += f0;
    v.results_host += f0;
    if (v.order_work) v.order_work += (size_t)f0 * (size_t)(1 + 2 * NB);
    if (v.arena_tag) v.arena_tag += (size_t)f0 * v.arena_spill;
    v.next_part_count += (size_t)f0 * NP;  // (the other copy: same frame, same slab stride)
    v.next_results += f0;
    return v;
}

// Launches the pipeline over the batch described by h->descs (buffers sized, inputs on the device).
// one_pass: fixed bin segments (k_czm_bin_scatter); otherwise the exact two-pass binning.
int launch_prepared(pwpp_handle *h, bool one_pass) {
    const int NP = PWPP_NUM_PARTS(h->dp.num_bins);
    const int frames = h->frames;
    if (h->k5_tail_unjoined) {  // the threshold update of the streams' last frames (K5's second part) comes before anything of this call
        HIPCHK(hipStreamWaitEvent(h->stream, h->aux_join, 0));
        h->k5_tail_unjoined = false;
    }
    // where a frame's bins live in the bin-ordered buffers
    int64_t base = 0;
    for (int f = 0; f < frames; ++f) {
        PwppFrameDesc &d = h->descs[(size_t)f];
        d.sbase = one_pass ? (int64_t)f * h->slots_per_frame : base;
        d.mbase = d.sbase / 8 + (int64_t)2 * PWPP_MEMBER_PAD * NP * f;  // (sbase is a multiple of PWPP_SLOT_ALIGN in both layouts; a pad per part, and one more for a part moved into the arena)
        // compact layout: every part starts at a multiple of PWPP_SLOT_ALIGN slots (k_czm_scan)
        base += ((int64_t)d.n + (PWPP_SLOT_ALIGN - 1) * (int64_t)NP + (PWPP_SLOT_ALIGN - 1)) & ~(int64_t)(PWPP_SLOT_ALIGN - 1);
    }
    // the descriptors on the device are reused when nothing changed (a caller cycling through the same
    // device buffers, a replayed batch): one host-to-device copy less in front of the first kernel
    const size_t desc_bytes = (size_t)frames * sizeof(PwppFrameDesc);
    if (h->descs_on_device.size() != (size_t)frames || h->descs_dev_ptr != h->d_frames.p ||
        std::memcmp(h->descs_on_device.data(), h->descs.data(), desc_bytes) != 0) {
        std::memcpy(h->h_frames.p, h->descs.data(), desc_bytes);
        HIPCHK(hipMemcpyAsync(h->d_frames.p, h->h_frames.p, desc_bytes, hipMemcpyHostToDevice, h->stream));
        h->descs_on_device = h->descs;
        h->descs_dev_ptr = h->d_frames.p;
    }

    // this call works on the other copy of the counters; if the last call's K5 zeroed exactly what this one needs, no
    // clearing kernel runs in front of the binning
    h->counters_copy ^= 1;
    const int slabs = one_pass ? 1 : 3;
    const bool pre_cleared = h->next_clean && h->next_clean_frames == frames && h->next_clean_slabs >= slabs &&
                             h->next_clean_parts == h->d_parts.p && h->next_clean_results == h->d_results.p &&
                             h->next_clean_parts_cap == h->d_parts.cap && h->next_clean_results_cap == h->d_results.cap;
    h->next_clean = false;  // (set again below, once this call's K5 is on its way for every frame)
    PwppBatch bt;
    fill_batch(h, bt);
    bt.next_slabs = slabs;
    bt.no_clear = pre_cleared ? 1 : 0;

    // A few stateful streams (the latency plan): the index lists wait for the first part of K5 only; the statistics over the streams'
    // histories -- what only their NEXT frames need -- run on aux_stream under K6 and the host's turn-around.
    const bool split_k5 = h->split_k5 != 0 && h->mode == PWPP_MODE_STREAMS && frames <= 64 && !h->profiling && bt.debug == 0 &&
                          h->output_order != PWPP_ORDER_REFERENCE && h->dp.min_pts != 0;
    bt.k5_split = split_k5 ? h->split_k5 : 0;
    bt.fuse_scan = h->fuse_scan;
    bt.cap_off = one_pass ? h->d_cap_off.p : nullptr;
    if (one_pass) {
        bt.arena_base = h->arena_base;
        bt.arena_slots = h->arena_slots;
        bt.arena_spill = h->arena_spill;
        bt.arena_tag = h->arena_slots ? h->d_arena_tag.p : nullptr;
    }
    if (one_pass && h->mode == PWPP_MODE_STREAMS) {  // what a redo after a segment overflow starts from: copied by the binning kernel
        bt.snap_scalar = h->d_st_snap.p;
        bt.snap_hist = h->d_hist_snap.p;
        bt.snap_plane = h->d_pl_snap.p;
    }
    if (pre_cleared && (bt.debug & 64)) {
        // the pre-cleared path rests on every K5 variant zeroing the OTHER copy of the counters for every frame.
        // Debug option: read this call's copy back before the binning touches it; anything but zeros is a broken invariant.
        HIPCHK(hipStreamSynchronize(h->stream));
        const size_t words = (size_t)slabs * (size_t)frames * (size_t)PWPP_NUM_PARTS(h->dp.num_bins);
        std::vector<uint32_t> hp(words);
        std::vector<PwppFrameResult> hr((size_t)frames);
        HIPCHK(hipMemcpy(hp.data(), bt.part_count, words * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hr.data(), bt.results, hr.size() * sizeof(PwppFrameResult), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (uint32_t v : hp) bad += v != 0u;
        const unsigned char *rb = reinterpret_cast<const unsigned char *>(hr.data());
        for (size_t i = 0; i < hr.size() * sizeof(PwppFrameResult); ++i) bad += rb[i] != 0;
        if (bad) return fail(PWPP_E_STATE, "pre-cleared counters are not zero (%zu words / bytes): a K5 variant skipped clear_next_counters", bad);
    }
    HIPCHK(hipEventRecord(h->ev_begin, h->stream));
    // (the histogram, the scatter cursors and the per-frame result counters are zeroed by the pipeline's
    // first kernel, k_clear)
    if (bt.debug & 4) {
        HIPCHK(hipMemsetAsync(bt.dbg, 0, 64 * sizeof(unsigned long long), h->stream));
        HIPCHK(hipMemsetAsync(bt.dbg + 60, 0xFF, sizeof(unsigned long long), h->stream));  // slot 60 is a minimum
    }
    const bool ordered = h->output_order == PWPP_ORDER_REFERENCE;
    int lrc;
    // Overlap mode: the pipeline alternates memory-bound stages (binning, emit) and VALU-bound ones (the
    // fits).  Two halves of the batch, each with its own launches on its own stream, put one kind under the
    // other (tools/two_handles.py: +7.5 % on 1024 KITTI frames).  Every per-frame array is indexed by the
    // frame, so the halves are two views of the same workspaces with shifted base pointers.
    if (h->overlap && !h->profiling && !ordered && frames >= 128 && bt.debug == 0) {
        auto range = [&](int f0, int nf) { return frame_range(h, bt, f0, nf); };
        // R frame ranges of whole groups of eight frames (K1' deals frames to the 8 XCDs).  Every range is launched
        // with the fit plan of the WHOLE batch (the machine is shared, not split).
        int R = h->overlap_ranges < 2 ? 2 : h->overlap_ranges;
        while (R > 2 && frames / R < 64) --R;
        // big bins per wave: four on scans of KITTI density (+2 % over two in interleaved end-to-end runs), two where the
        // bins are several times larger (dense 128-beam frames, 36-sector CZM: 110.9 k against 106.8 k frames/s)
        bt.plan_frames = frames;  // (pwpp_launch_fit picks the plan; round 4: also below 640 frames -- the ranges used to pick one for their own size)
        std::vector<int> first((size_t)R + 1, 0);
        for (int r = 1; r <= R; ++r) {
            int f1 = r == R ? frames : (int)(((int64_t)frames * r / R + 7) / 8 * 8);
            first[(size_t)r] = f1 > frames ? frames : f1;
        }
        lrc = pre_cleared ? 0 : pwpp_launch_clear(&bt, h->stream);
        if (h->overlap_mode == 1) {
            // A software pipeline over the two streams: the MEMORY stream bins range r + 2 and then writes the lists of
            // range r, the FIT stream runs the plane fits (and K5) of range r + 1 meanwhile -- the streams never run the
            // same kind of stage at the same time (two whole-range pipelines side by side start in lock step: both
            // binning, then both fitting).  Events: range binned -> its fits may start; range fitted -> its lists.
            while (h->ev_ranges.size() < 2 * (size_t)R) {
                hipEvent_t e = nullptr;
                HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                h->ev_ranges.push_back(e);
            }
            auto stage = [&](int r, int stages, hipStream_t st) {
                if (lrc != 0 || first[(size_t)r + 1] <= first[(size_t)r]) return;
                const PwppBatch v = range(first[(size_t)r], first[(size_t)r + 1] - first[(size_t)r]);
                lrc = pwpp_launch_pipeline(&v, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stages);
            };
            // (the fit stages go round a few streams: a fit kernel ends with the tail of its longest waves, and the next
            // range's kernels fill the machine meanwhile)
            while ((int)h->extra_streams.size() + 1 < h->num_fit_streams) {
                hipStream_t st = nullptr;
                HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                h->extra_streams.push_back(st);
            }
            hipStream_t mem = h->stream;
            const bool masked = h->masked_mem != nullptr && (int)h->masked_fit.size() >= h->num_fit_streams;
            if (masked) {  // the CU-partitioned streams: fork from the main stream (descriptor copy, k_clear), join at the end
                mem = h->masked_mem;
                HIPCHK(hipEventRecord(h->aux_fork, h->stream));
                HIPCHK(hipStreamWaitEvent(mem, h->aux_fork, 0));
            }
            // (every record / wait is checked: a failed one would leave the fit streams unordered against binning and
            // k_emit, and the batch would "succeed" with racy results -- ADVICE r02)
            auto ordered_ok = [&](hipError_t e) {
                if (e != hipSuccess && lrc == 0) lrc = (int)e;
            };
            auto bin = [&](int r) {
                const int k = r % h->num_fit_streams;
                hipStream_t fit = masked ? h->masked_fit[(size_t)k] : (k == 0 ? h->aux_stream : h->extra_streams[(size_t)k - 1]);
                stage(r, 1, mem);
                if (lrc == 0) ordered_ok(hipEventRecord(h->ev_ranges[2 * (size_t)r], mem));
                if (lrc == 0) ordered_ok(hipStreamWaitEvent(fit, h->ev_ranges[2 * (size_t)r], 0));
                stage(r, 2, fit);
                if (lrc == 0) ordered_ok(hipEventRecord(h->ev_ranges[2 * (size_t)r + 1], fit));
            };
            auto lists = [&](int r) {
                if (lrc == 0) ordered_ok(hipStreamWaitEvent(mem, h->ev_ranges[2 * (size_t)r + 1], 0));
                stage(r, 4, mem);
            };
            bin(0);
            if (R > 1) bin(1);
            for (int r = 0; r < R; ++r) {
                lists(r);
                if (r + 2 < R) bin(r + 2);
            }
            // (the memory stream ends with the lists of the last range, which wait for its fits: h->stream is the join)
            if (masked && lrc == 0) {
                ordered_ok(hipEventRecord(h->aux_join, mem));
                ordered_ok(hipStreamWaitEvent(h->stream, h->aux_join, 0));
            }
        } else {
            // whole ranges alternating between the two streams
            HIPCHK(hipEventRecord(h->aux_fork, h->stream));
            HIPCHK(hipStreamWaitEvent(h->aux_stream, h->aux_fork, 0));
            for (int r = 0; r < R && lrc == 0; ++r)
                if (first[(size_t)r + 1] > first[(size_t)r]) {
                    const PwppBatch v = range(first[(size_t)r], first[(size_t)r + 1] - first[(size_t)r]);
                    lrc = pwpp_launch_pipeline(&v, (r & 1) ? h->aux_stream : h->stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 7);
                }
            HIPCHK(hipEventRecord(h->aux_join, h->aux_stream));
            HIPCHK(hipStreamWaitEvent(h->stream, h->aux_join, 0));
        }
    } else {
        lrc = pwpp_launch_pipeline(&bt, h->stream, h->profiling ? h->ev_k : nullptr, h->aux_stream, h->aux_fork, h->aux_join,
                                   ordered ? h->d_ord_a.p : nullptr, ordered ? h->d_ord_b.p : nullptr, 7);
    }
    if (lrc != 0) {  // nothing of a half-launched schedule may still be running when the caller sees the error
        sync_all_streams(h);
        return fail(PWPP_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    }
    HIPCHK(hipEventRecord(h->ev_end, h->stream));
    if (bt.k5_split && !(h->overlap && !h->profiling && !ordered && frames >= 128 && bt.debug == 0)) {
        if (bt.k5_split == 2) {  // the second part behind the lists
            HIPCHK(hipEventRecord(h->aux_fork, h->stream));
            const int trc = pwpp_launch_k5_tail(&bt, h->aux_stream, h->aux_fork, h->aux_join);
            if (trc != 0) {
                sync_all_streams(h);
                return fail(PWPP_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)trc));
            }
        }
        h->k5_tail_unjoined = h->k5_tail_unsynced = true;
    }
    h->next_clean = true;  // every frame's K5 is enqueued: the other copy will be zero for a call of this shape
    h->next_clean_frames = frames;
    h->next_clean_slabs = slabs;
    h->next_clean_parts = h->d_parts.p;
    h->next_clean_results = h->d_results.p;
    h->next_clean_parts_cap = h->d_parts.cap;
    h->next_clean_results_cap = h->d_results.cap;
    if (one_pass)  // the bins' largest counts, for the segment sizes of the next batches (finish_pending)
        HIPCHK(hipMemcpyAsync(h->h_bin_max.p, h->d_bin_max.p, (size_t)NP * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    h->profile_pending = h->profiling;
    h->one_pass = one_pass;
    g_slot0_one_pass = one_pass;
    return PWPP_OK;
}

// What K5's second launch (split_k5) leaves behind the index lists: the streams' state, and how full their histories are.
int settle_k5_tail(pwpp_handle *h) {
    if (h->k5_tail_unsynced) {  // (the streams' state and results[].hist_state are complete once K5's second part has run)
        HIPCHK(hipStreamSynchronize(h->aux_stream));
        h->k5_tail_unsynced = false;
    }
    if (h->hist_check_due) {
        h->hist_check_due = false;
        // The reference's history vectors are unbounded (update_flatness_thr stops trimming the higher rings while a
        // lower one holds <= 1 entries, patchworkpp.cpp:363-364).  The slabs grow before a call could fill them.
        int fill = 0;
        bool dropped = false;
        for (int f = 0; f < h->frames; ++f) {
            const int hs = h->h_results.p[f].hist_state;
            fill = (hs >> 1) > fill ? (hs >> 1) : fill;
            dropped = dropped || (hs & 1);
        }
        if (dropped) return fail(PWPP_E_STATE, "an A-GLE history outgrew its slab (%d entries): the adaptive thresholds of this stream are no longer the reference's", h->stream_hist_cap);
        if (fill + h->max_pushes_per_frame + 8 > h->stream_hist_cap) {
            const int rc = grow_stream_histories(h, 2 * h->stream_hist_cap);
            if (rc) return rc;
        }
    }
    return PWPP_OK;
}

// lists_only (pwpp_synchronize, the getters of a call's results, pwpp_get_time_us): the caller wants what the main stream delivers -- counts,
// index lists, patch rows.  K5's second launch may then stay on its stream (the next estimate call and everything that touches a stream's
// state settle it first), unless a frame of the call needs the host: a redo or a fix-up restores / continues the streams' state.
int finish_pending(pwpp_handle *h, bool lists_only) {
    if (!h->pending) return lists_only ? PWPP_OK : settle_k5_tail(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    bool defer_tail = lists_only && h->k5_tail_unsynced;
    for (int f = 0; f < h->frames && defer_tail; ++f)
        if (h->h_results.p[f].overflow & 3) defer_tail = false;
    if (h->k5_tail_unsynced && !defer_tail) {
        HIPCHK(hipStreamSynchronize(h->aux_stream));
        h->k5_tail_unsynced = false;
    }
    h->pending = false;
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
    h->time_us = (double)ms * 1000.0;
    if (h->profile_pending) {
        for (int k = 0; k < PWPP_NUM_KERNELS; ++k) {
            float kms = 0.0f;
            HIPCHK(hipEventElapsedTime(&kms, h->ev_k[k], h->ev_k[k + 1]));
            h->prof_ms[k] += kms;
            h->prof_launches[k] += 1;
        }
        h->profile_pending = false;
    }
    const bool was_one_pass = h->one_pass;
    bool redone_in_place = false;
    if (h->one_pass) {  // did every bin of every frame fit its segment?  the frames that did not are redone on the exact two-pass path
        std::vector<int> redo;
        for (int f = 0; f < h->frames; ++f)
            if (h->h_results.p[f].overflow & 1) redo.push_back(f);
        h->one_pass = false;
        h->frame_two_pass.assign((size_t)h->frames, 0);
        if (!redo.empty()) {
            {   // the exact path needs the per-point codes (not held for one-pass batches)
                const int rc = h->d_codes.ensure((size_t)(h->total_points > 0 ? h->total_points : 1));
                if (rc) return rc;
            }
            ++h->one_pass_redone;
            h->one_pass_holdoff = 0;  // (the redo's exact counts enter d_bin_max and the table is rebuilt: no need to stay away)
            h->table_stale = true;
            // A frame is redone IN PLACE: its compact two-pass layout (points + the parts' alignment pads) starts at the frame's own
            // first slot of the one-pass layout, whose segments are ~2.7 slots per point -- the other frames of the batch are not
            // touched (frames are independent: fresh state, or one stream each).  Only a frame that would not fit its own
            // slots (a table built from much smaller frames) sends the whole batch through the compact layout.
            const int NP = PWPP_NUM_PARTS(h->dp.num_bins);
            bool in_place = !h->redo_whole_batch;
            {   //  every run of neighbouring frames is a pipeline of its own -- four memsets and ten small launches, one after
                // the other on one stream.  A cold or out-of-distribution batch with overflows all over it (353 of 512 frames once) is
                // cheaper as ONE whole-batch redo on the two-pass path (~2.5 ms per 1024 frames) than as hundreds of those.
                size_t runs = 0;
                for (size_t i = 0; i < redo.size(); ++i) runs += i == 0 || redo[i] != redo[i - 1] + 1;
                if (runs > 16 && (redo.size() > (size_t)h->frames / 8 || runs > 64)) in_place = false;
            }
            for (int f : redo) {
                const int64_t need = (int64_t)h->descs[(size_t)f].n + (int64_t)(PWPP_SLOT_ALIGN - 1) * NP + PWPP_SLOT_ALIGN;
                if (need > h->slots_per_frame) in_place = false;
            }
            if (h->mode == PWPP_MODE_STREAMS) {  // back to the state before the first attempt -- of the streams that are redone
                const size_t slab = (size_t)8 * (size_t)h->stream_hist_cap;
                if (!in_place) {
                    HIPCHK(hipMemcpyAsync(h->d_st_stream.p, h->d_st_snap.p, (size_t)h->frames * sizeof(PwppStateScalar), hipMemcpyDeviceToDevice, h->stream));
                    HIPCHK(hipMemcpyAsync(h->d_pl_stream.p, h->d_pl_snap.p, (size_t)h->frames * sizeof(PwppPlaneState), hipMemcpyDeviceToDevice, h->stream));
                    HIPCHK(hipMemcpyAsync(h->d_hist_stream.p, h->d_hist_snap.p, (size_t)h->frames * slab * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
                } else {
                    for (int f : redo) {
                        HIPCHK(hipMemcpyAsync(h->d_st_stream.p + f, h->d_st_snap.p + f, sizeof(PwppStateScalar), hipMemcpyDeviceToDevice, h->stream));
                        HIPCHK(hipMemcpyAsync(h->d_pl_stream.p + f, h->d_pl_snap.p + f, sizeof(PwppPlaneState), hipMemcpyDeviceToDevice, h->stream));
                        HIPCHK(hipMemcpyAsync(h->d_hist_stream.p + (size_t)f * slab, h->d_hist_snap.p + (size_t)f * slab, slab * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
                    }
                }
            }
            if (!in_place) {
                h->one_pass_redone_frames += h->frames;
                int rc = launch_prepared(h, false);
                if (rc) return rc;
                h->pending = true;
                if ((rc = finish_pending(h))) return rc;
                return read_observed(h);  // the exact counts of the redo size the next table
            }
            h->one_pass_redone_frames += (long long)redo.size();
            PwppBatch bt;
            fill_batch(h, bt);
            bt.cap_off = nullptr;  // compact layout, exact two-pass binning (K1, K2, K3)
            bt.next_slabs = 1;     // (the other copy of the counters was zeroed for a one-pass call by the first attempt's K5 already)
            const bool ordered = h->output_order == PWPP_ORDER_REFERENCE;
            for (size_t i = 0; i < redo.size();) {  // runs of neighbouring frames go through the pipeline together
                size_t j = i + 1;
                while (j < redo.size() && redo[j] == redo[j - 1] + 1) ++j;
                const int f0 = redo[i], nf = (int)(j - i);
                PwppBatch v = frame_range(h, bt, f0, nf);  // (no_clear set: the three counter slabs of a frame RANGE are not adjacent)
                int vmax = 0;
                for (int f = f0; f < f0 + nf; ++f) {
                    h->frame_two_pass[(size_t)f] = 1;
                    vmax = h->descs[(size_t)f].n > vmax ? h->descs[(size_t)f].n : vmax;
                }
                v.max_n = vmax;
                const size_t words = (size_t)nf * (size_t)NP * sizeof(uint32_t);
                HIPCHK(hipMemsetAsync(v.part_count, 0, words, h->stream));
                HIPCHK(hipMemsetAsync(v.part_off, 0, words, h->stream));
                HIPCHK(hipMemsetAsync(v.part_cursor, 0, words, h->stream));
                HIPCHK(hipMemsetAsync(v.results, 0, (size_t)nf * sizeof(PwppFrameResult), h->stream));
                const int lrc = pwpp_launch_pipeline(&v, h->stream, nullptr, nullptr, nullptr, nullptr, ordered ? h->d_ord_a.p : nullptr,
                                                     ordered ? h->d_ord_b.p : nullptr, 7);
                if (lrc != 0) return fail(PWPP_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
                i = j;
            }
            HIPCHK(hipStreamSynchronize(h->stream));
            redone_in_place = true;
        }
    }
    // Frames with a patch whose first fit set was empty (pwpp_fit.hip: needs_previous_plane -- a lowest height of -inf or
    // beyond 1e15 m, num_lpr = 0; never a real scan): K5 / K6 left them alone; k_fit_fixup fits those patches in the
    // reference's order from the plane fitted before them, then K5 and K6 finish the frame.
    {
        bool any = false;
        for (int f = 0; f < h->frames; ++f) any = any || (h->h_results.p[f].overflow & 2) != 0;
        if (any) {
            PwppBatch bt;
            fill_batch(h, bt);
            bt.fixup_run = 1;
            const bool ordered = h->output_order == PWPP_ORDER_REFERENCE;
            for (int f = 0; f < h->frames; ++f) {
                if (!(h->h_results.p[f].overflow & 2)) continue;
                PwppBatch v = frame_range(h, bt, f, 1);
                v.cap_off = was_one_pass && !h->frame_two_pass[(size_t)f] ? h->d_cap_off.p : nullptr;  // (a frame redone in place is in the compact layout)
                if (v.cap_off) {  // (where its moved parts' bits live: pwpp_member_offset)
                    v.arena_base = h->arena_base;
                    v.arena_slots = h->arena_slots;
                    v.arena_spill = h->arena_spill;
                    v.arena_tag = h->arena_slots ? h->d_arena_tag.p + (size_t)f * h->arena_spill : nullptr;
                }
                const int lrc = pwpp_launch_pipeline(&v, h->stream, nullptr, nullptr, nullptr, nullptr, ordered ? h->d_ord_a.p : nullptr,
                                                     ordered ? h->d_ord_b.p : nullptr, 2 | 4 | 8);
                if (lrc != 0) return fail(PWPP_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
                ++h->fixed_up_frames;
            }
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    for (int f = 0; f < h->frames; ++f) {
        if (h->h_results.p[f].overflow & 4) ++h->clamped_frames;
        if ((h->h_results.p[f].overflow & 8) && !(h->frame_two_pass.size() > (size_t)f && h->frame_two_pass[(size_t)f])) ++h->arena_frames;
    }
    h->have_results = true;
    if (redone_in_place) {  // the exact counts of the redone frames are in d_bin_max now: they size the next table
        const int rc = read_observed(h);
        if (rc) return rc;
    } else if (was_one_pass) {
        // a part that held more points than the table was built for (it ate into its head-room, or was moved into the arena): size
        // the table anew before the next batch
        h->observed.assign(h->h_bin_max.p, h->h_bin_max.p + h->cap_table.size());
        for (size_t b = 0; b < h->cap_table.size() && !h->table_stale; ++b)
            if (h->observed[b] > (b < h->cap_seen.size() ? h->cap_seen[b] : 0u) && h->cap_table[b] > 0u) h->table_stale = true;
    }
    if (h->mode == PWPP_MODE_STREAMS) h->hist_check_due = true;
    return defer_tail ? PWPP_OK : settle_k5_tail(h);
}

int check_frame(pwpp_handle *h, int frame) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    rc = finish_pending(h, true);  // (a call's results: counts, lists, patch rows)
    if (rc) return rc;
    if (!h->have_results) return fail(PWPP_E_STATE, "no frame has been processed yet");
    if (frame < 0 || frame >= h->frames) return fail(PWPP_E_ARG, "frame %d out of range [0,%d)", frame, h->frames);
    return PWPP_OK;
}

}  // namespace

extern "C" {

const char *pwpp_last_error(void) { return g_err.c_str(); }

int pwpp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *pwpp_kernel_name(int k) {
    if (k == 0 && g_slot0_one_pass) return "k_czm_bin_scatter";  // what the last launch ran in the first slot
    return (k >= 0 && k < PWPP_NUM_KERNELS) ? kKernelNames[k] : "";
}

int pwpp_params_default(pwpp_params *p) {  // reference patchworkpp.h:79-111
    if (!p) return fail(PWPP_E_ARG, "null params");
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
    p->intensity_thr = 0.0;  // uninitialised in the reference (patchworkpp.h:67), never read
    const int sectors[4] = {16, 32, 54, 32}, rings[4] = {2, 4, 4, 4};
    for (int k = 0; k < 4; ++k) {
        p->num_sectors_each_zone[k] = sectors[k];
        p->num_rings_each_zone[k] = rings[k];
    }
    p->max_flatness_storage = 1000;
    p->max_elevation_storage = 1000;
    return PWPP_OK;
}

int pwpp_set_num_streams(pwpp_handle *h, int streams) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    if (streams < 1) return fail(PWPP_E_ARG, "streams must be >= 1");
    int rc = use_device(h);
    if (rc) return rc;
    rc = finish_pending(h);
    if (rc) return rc;
    if ((rc = h->d_st_stream.ensure((size_t)streams))) return rc;
    if ((rc = h->d_hist_stream.ensure((size_t)streams * 8 * (size_t)h->stream_hist_cap))) return rc;
    if ((rc = h->d_pl_stream.ensure((size_t)streams))) return rc;
    HIPCHK(hipMemset(h->d_pl_stream.p, 0, (size_t)streams * sizeof(PwppPlaneState)));  // a new object's members
    std::vector<PwppStateScalar> init((size_t)streams);
    for (auto &s : init) fill_default_state(h, s);
    HIPCHK(hipMemcpy(h->d_st_stream.p, init.data(), init.size() * sizeof(PwppStateScalar), hipMemcpyHostToDevice));
    h->num_streams = streams;
    return PWPP_OK;
}

int pwpp_create(const pwpp_params *p, int device, pwpp_handle **out) {
    if (!p || !out) return fail(PWPP_E_ARG, "null argument");
    *out = nullptr;
    PwppDevParams dp;
    const int max_near_sectors = build_dev_params(*p, dp);
    if (max_near_sectors < 0) return max_near_sectors;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(PWPP_E_NODEVICE, "no HIP device available (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    if (device < 0 || device >= ndev) return fail(PWPP_E_ARG, "device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    std::vector<float2> origin;
    dp.fxp_wide = 1;  // contract v4 (option "exact_moments")
    if (const char *e = std::getenv("PWPP_EXACT_MOMENTS")) dp.fxp_wide = std::atoi(e) != 0;
    dp.fxp_zr = (float)fxp_geometry(dp, origin, dp.fxp_shift, dp.fxp_wide != 0);
    std::vector<float4> boxes;
    bin_boxes(dp, boxes);
    dp.hi_split = 0.6f;  // option "hi_split"
    dp.split_end = dp.bin_base[1];  // option "hi_split_zones": the zones whose bins are stored in two parts (default: the first)
    pwpp_handle *h = new pwpp_handle;
    h->params = *p;
    h->dp = dp;
    h->device = device;
    // The tuning / test switches: read once, here (not in the launch path), and said out loud.
    if (const char *e = std::getenv("PWPP_DEBUG_FLAGS")) h->debug_flags = std::atoi(e);
    if ((h->debug_flags & 4) && (h->debug_flags & 8)) {  // (the two sets of timing probes share one array: pwpp_set_option refuses the pair, the environment path says what it does)
        h->debug_flags &= ~8;
        std::fprintf(stderr, "pwpp: PWPP_DEBUG_FLAGS has both timing-probe sets (4 and 8), which share one array: 8 dropped\n");
    }
    if (const char *e = std::getenv("PWPP_FIT_PLAN")) h->fit_plan = e;
    h->fit_concurrent = std::getenv("PWPP_FIT_CONCURRENT") != nullptr;
    h->no_one_pass = std::getenv("PWPP_NO_ONE_PASS") != nullptr;
    if (const char *e = std::getenv("PWPP_ONE_PASS_MIN_FRAMES")) h->one_pass_min_frames = h->one_pass_min_fresh = std::atoi(e);
    if (const char *e = std::getenv("PWPP_OVERLAP_RANGES")) h->overlap_ranges = std::atoi(e) < 2 ? 2 : std::atoi(e);
    if (const char *e = std::getenv("PWPP_OVERLAP_MODE")) h->overlap_mode = std::atoi(e) != 0;
    if (const char *e = std::getenv("PWPP_BIN_BLOCK")) h->bin_block = std::atoi(e) == 128 ? 128 : (std::atoi(e) == 512 ? 512 : (std::atoi(e) == 1024 ? 1024 : 256));
    if (const char *e = std::getenv("PWPP_FIT_STREAMS")) h->num_fit_streams = std::atoi(e) < 1 ? 1 : (std::atoi(e) > 8 ? 8 : std::atoi(e));
    if (const char *e = std::getenv("PWPP_HI_SPLIT")) h->dp.hi_split = (float)std::atof(e);
    if (const char *e = std::getenv("PWPP_HI_SPLIT_ZONES")) h->dp.split_end = h->dp.bin_base[std::atoi(e) < 0 ? 0 : (std::atoi(e) > 4 ? 4 : std::atoi(e))];
    if (const char *e = std::getenv("PWPP_ONE_PASS_SCALE")) {
        const double v = std::atof(e);
        if (v > 0.0 && v <= 1024.0) h->one_pass_scale = v;
        else std::fprintf(stderr, "pwpp: ignoring PWPP_ONE_PASS_SCALE=%s (a positive number up to 1024 expected)\n", e);
    }
    if (std::getenv("PWPP_EXACT_MOMENTS")) std::fprintf(stderr, "pwpp: exact_moments=%d taken from the environment (PWPP_EXACT_MOMENTS)\n", h->dp.fxp_wide);
    if (h->debug_flags || !h->fit_plan.empty() || h->fit_concurrent || h->no_one_pass || std::getenv("PWPP_ONE_PASS_MIN_FRAMES") ||
        std::getenv("PWPP_ONE_PASS_SCALE") || std::getenv("PWPP_OVERLAP") || std::getenv("PWPP_OVERLAP_RANGES") || std::getenv("PWPP_HI_SPLIT") ||
        std::getenv("PWPP_HI_SPLIT_ZONES"))
        std::fprintf(stderr, "pwpp: tuning options taken from the environment (PWPP_*): debug_flags=%d fit_plan='%s' fit_concurrent=%d "
                             "no_one_pass=%d one_pass_min_frames=%d one_pass_scale=%g overlap=%d hi_split=%g (first %d bins)\n", h->debug_flags,
                     h->fit_plan.c_str(), (int)h->fit_concurrent, (int)h->no_one_pass, h->one_pass_min_frames, h->one_pass_scale,
                     std::getenv("PWPP_OVERLAP") ? std::atoi(std::getenv("PWPP_OVERLAP")) : 1, (double)h->dp.hi_split, h->dp.split_end);
    const int storage = p->max_elevation_storage > p->max_flatness_storage ? p->max_elevation_storage : p->max_flatness_storage;
    h->stream_hist_cap = (storage + max_near_sectors + 1024 + 1) & ~1;  // (even: k_gle_tgr fetches the histories two entries at a time)
    h->max_pushes_per_frame = max_near_sectors;
    h->fresh_hist_cap = (max_near_sectors + 2 + 1) & ~1;
    if (const char *e = std::getenv("PWPP_OVERLAP")) h->overlap = std::atoi(e) != 0;  // (pwpp_set_overlap; PWPP_OVERLAP=0 runs existing programs on one stream)
    hipError_t se = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking);
    // The streams of the overlap schedule are created HERE, together (round 5): the HIP runtime deals streams to its four hardware
    // queues in creation order, and a fit stream created lazily -- after some other handle's streams -- could land on the queue of
    // this handle's own memory stream: the two then run one after the other (3.18 instead of 2.6 ms per 1024-frame step,
    // tools/sync_after_pipelined.py).
    for (int k = 1; se == hipSuccess && k < h->num_fit_streams; ++k) {
        hipStream_t st = nullptr;
        se = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (se == hipSuccess) h->extra_streams.push_back(st);
    }
    if (se == hipSuccess) se = hipEventCreateWithFlags(&h->aux_fork, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&h->aux_join, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreate(&h->ev_begin);
    if (se == hipSuccess) se = hipEventCreate(&h->ev_end);
    for (int k = 0; k <= PWPP_NUM_KERNELS && se == hipSuccess; ++k) se = hipEventCreate(&h->ev_k[k]);
    if (se != hipSuccess) {
        pwpp_destroy(h);
        return fail(PWPP_E_HIP, "stream/event creation failed: %s", hipGetErrorString(se));
    }
    int rc = pwpp_set_num_streams(h, 1);
    const size_t NP = (size_t)PWPP_NUM_PARTS(dp.num_bins);
    if (!rc) rc = h->h_bin_max.ensure(NP);
    if (!rc) rc = h->d_bin_max.ensure(NP);
    if (!rc && hipMemset(h->d_bin_max.p, 0, NP * sizeof(uint32_t)) != hipSuccess) rc = fail(PWPP_E_HIP, "hipMemset failed");
    if (!rc) rc = h->d_bin_bbox.ensure(boxes.size());
    if (!rc && hipMemcpy(h->d_bin_bbox.p, boxes.data(), boxes.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PWPP_E_HIP, "uploading the bin boxes failed");
    if (!rc) rc = h->d_bin_origin.ensure(origin.size());
    if (!rc && hipMemcpy(h->d_bin_origin.p, origin.data(), origin.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PWPP_E_HIP, "uploading the bin origins failed");
    if (rc) {
        pwpp_destroy(h);
        return rc;
    }
    if (p->verbose) std::printf("PatchWorkpp::PatchWorkpp() - INITIALIZATION COMPLETE (MI355X/HIP, %d CZM bins, fxp shift %d)\n", dp.num_bins, dp.fxp_shift);
    *out = h;
    return PWPP_OK;
}

int pwpp_destroy(pwpp_handle *h) {
    if (!h) return PWPP_OK;
    (void)hipSetDevice(h->device);
    sync_all_streams(h);
    h->d_frames.release();
    h->h_frames.release();
    h->h_base.release();
    h->d_in.release();
    h->d_codes.release();
    h->d_sorted_z.release();
    h->d_sorted_xy.release();
    h->d_sorted_idx.release();
    h->d_bin_origin.release();
    h->d_bin_bbox.release();
    h->d_parts.release();
    h->d_bin_max.release();
    h->h_bin_max.release();
    h->d_frames_probe.release();
    h->d_member.release();
    h->d_arena_tag.release();
    h->d_out.release();
    h->d_ord_a.release();
    h->d_ord_b.release();
    h->d_ord_work.release();
    h->d_bins.release();
    h->d_recs.release();
    h->d_cls_start.release();
    h->d_cap_off.release();
    h->d_emit_long.release();
    h->d_cls_list.release();
    h->d_centers.release();
    h->d_normals.release();
    h->d_results.release();
    h->h_results.release();
    h->d_xyz.release();
    h->d_dbg.release();
    h->d_st_stream.release();
    h->d_st_fresh.release();
    h->d_hist_stream.release();
    h->d_hist_fresh.release();
    h->d_st_snap.release();
    h->d_hist_snap.release();
    h->d_pl_stream.release();
    h->d_pl_fresh.release();
    h->d_pl_snap.release();
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    for (int k = 0; k <= PWPP_NUM_KERNELS; ++k)
        if (h->ev_k[k]) (void)hipEventDestroy(h->ev_k[k]);
    for (hipEvent_t e : h->ev_ranges) (void)hipEventDestroy(e);
    for (hipStream_t st : h->extra_streams) (void)hipStreamDestroy(st);
    for (hipStream_t st : h->masked_fit) (void)hipStreamDestroy(st);
    if (h->masked_mem) (void)hipStreamDestroy(h->masked_mem);
    if (h->aux_fork) (void)hipEventDestroy(h->aux_fork);
    if (h->aux_join) (void)hipEventDestroy(h->aux_join);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PWPP_OK;
}

}  // extern "C"

namespace {
struct FieldSpec {  // PWPP_LAYOUT_FIELDS: where the float32 fields of a point lie
    int step, off[4];
};
int estimate_batch(pwpp_handle *h, const float *const *points, const int32_t *n, int frames, int cols, int layout, int mem, int mode,
                   const FieldSpec *fs);
}  // namespace

extern "C" {

int pwpp_estimate_ground_batch(pwpp_handle *h, const float *const *points, const int32_t *n, int frames, int cols,
                               int layout, int mem, int mode) {
    if (layout != PWPP_LAYOUT_ROW_MAJOR && layout != PWPP_LAYOUT_COL_MAJOR) return fail(PWPP_E_ARG, "bad layout %d", layout);
    return estimate_batch(h, points, n, frames, cols, layout, mem, mode, nullptr);
}

int pwpp_estimate_ground_fields_batch(pwpp_handle *h, const void *const *data, const int32_t *n, int frames, int point_step,
                                      int off_x, int off_y, int off_z, int off_intensity, int mem, int mode) {
    if (point_step < 12 || (point_step & 3)) return fail(PWPP_E_ARG, "point_step=%d: a multiple of 4, at least 12, expected", point_step);
    const int off[4] = {off_x, off_y, off_z, off_intensity};
    for (int k = 0; k < 4; ++k) {
        if (k == 3 && off[k] < 0) continue;  // no intensity field: RNR is skipped, as for an N x 3 matrix (patchworkpp.cpp:379-382)
        if (off[k] < 0 || (off[k] & 3) || off[k] + 4 > point_step)
            return fail(PWPP_E_ARG, "field offset %d does not name a 4-byte aligned float32 inside a point of %d bytes", off[k], point_step);
    }
    // the fields are read in place as float32 (host: merged copies computed in units of floats; device: aligned loads): a
    // blob that does not start on a 4-byte boundary cannot be read that way
    if (!data || !n) return fail(PWPP_E_ARG, "null argument");  // (the handle is checked by estimate_batch: the alignment check needs none)
    if (frames < 1 || frames > 65535) return fail(PWPP_E_ARG, "frames=%d: 1 ... 65535 per call expected", frames);  // (before n[] / data[] are walked)
    for (int i = 0; i < frames; ++i)
        if (n[i] > 0 && (reinterpret_cast<uintptr_t>(data[i]) & 3u) != 0)
            return fail(PWPP_E_ARG, "frame %d: data pointer %p is not 4-byte aligned", i, data[i]);
    FieldSpec fs;
    fs.step = point_step;
    for (int k = 0; k < 4; ++k) fs.off[k] = off[k] < 0 ? -1 : off[k];
    return estimate_batch(h, reinterpret_cast<const float *const *>(data), n, frames, off_intensity >= 0 ? 4 : 3, PWPP_LAYOUT_FIELDS, mem, mode, &fs);
}

int pwpp_estimate_ground_fields(pwpp_handle *h, const void *data, int n, int point_step, int off_x, int off_y, int off_z, int off_intensity) {
    const void *ptrs[1] = {data};
    const int32_t ns[1] = {n};
    return pwpp_estimate_ground_fields_batch(h, ptrs, ns, 1, point_step, off_x, off_y, off_z, off_intensity, PWPP_MEM_HOST, PWPP_MODE_STREAMS);
}

}  // extern "C"

namespace {
int estimate_batch(pwpp_handle *h, const float *const *points, const int32_t *n, int frames, int cols, int layout, int mem, int mode,
                   const FieldSpec *fs) {
    if (!h || !points || !n) return fail(PWPP_E_ARG, "null argument");
    if (frames < 1) return fail(PWPP_E_ARG, "frames must be >= 1");
    if (frames > 65535) return fail(PWPP_E_ARG, "%d frames: at most 65535 per call (the frame is a grid dimension of the kernels)", frames);
    if (cols != 3 && cols != 4) return fail(PWPP_E_ARG, "cols=%d: 3 or 4 expected", cols);
    const int64_t floats_per_point = fs ? fs->step / 4 : cols;  // what a frame occupies in the staging buffer
    if (mem != PWPP_MEM_HOST && mem != PWPP_MEM_DEVICE && mem != PWPP_MEM_HOST_PINNED) return fail(PWPP_E_ARG, "bad mem %d", mem);
    const bool from_host = mem != PWPP_MEM_DEVICE;
    if (mode != PWPP_MODE_FRESH && mode != PWPP_MODE_STREAMS) return fail(PWPP_E_ARG, "bad mode %d", mode);
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (mode == PWPP_MODE_STREAMS && frames > h->num_streams)
        return fail(PWPP_E_ARG, "%d frames but only %d streams (pwpp_set_num_streams)", frames, h->num_streams);

    const int B = h->dp.num_bins, NB = B + 2, NP = PWPP_NUM_PARTS(B);
    int64_t total = 0, total_in = 0;
    int max_n = 0;
    for (int f = 0; f < frames; ++f) {
        if (n[f] < 0 || n[f] > (1 << 22)) return fail(PWPP_E_ARG, "frame %d: %d points (0..4194304 supported)", f, n[f]);
        if (n[f] > 0 && !points[f]) return fail(PWPP_E_ARG, "frame %d: null points", f);
        if (mem == PWPP_MEM_DEVICE && layout == PWPP_LAYOUT_ROW_MAJOR && cols == 4 && ((uintptr_t)points[f] & 15u))
            return fail(PWPP_E_ARG, "frame %d: device buffer must be 16-byte aligned", f);
        total += n[f];
        total_in += ((int64_t)n[f] * floats_per_point + 3) & ~(int64_t)3;
        if (n[f] > max_n) max_n = n[f];
    }
    if (total >= ((int64_t)1 << 31)) return fail(PWPP_E_ARG, "batch of %lld points exceeds 2^31", (long long)total);
    const size_t tp = (size_t)(total > 0 ? total : 1);

    // ---- 1. everything but the bin-ordered buffers
    if ((rc = h->d_frames.ensure((size_t)frames))) return rc;
    if ((rc = h->h_frames.ensure((size_t)frames))) return rc;
    if ((rc = h->h_base.ensure((size_t)frames + 1))) return rc;
    // (the per-point codes are the two-pass path's: a one-pass batch only needs them for its histogram probe and for a frame that is
    // binned again -- allocated there, 2 bytes per point less to hold otherwise)
    if ((rc = h->d_out.ensure(tp))) return rc;
    if (h->output_order == PWPP_ORDER_REFERENCE) {
        if ((rc = h->d_ord_a.ensure(tp))) return rc;
        if ((rc = h->d_ord_b.ensure(tp))) return rc;
        if ((rc = h->d_ord_work.ensure((size_t)frames * (size_t)(1 + 2 * NB)))) return rc;
    }
    if ((rc = h->d_bins.ensure((size_t)frames * NB * 4))) return rc;
    if ((rc = h->d_parts.ensure(2 * counters_copy_words((size_t)frames * NP)))) return rc;  // two copies
    if ((rc = h->d_recs.ensure((size_t)frames * B))) return rc;
    if ((rc = h->d_cls_start.ensure((size_t)frames * PWPP_CLS_STRIDE))) return rc;
    if ((rc = h->d_cls_list.ensure((size_t)frames * B))) return rc;
    if ((rc = h->d_centers.ensure((size_t)frames * B * 3))) return rc;
    if ((rc = h->d_normals.ensure((size_t)frames * B * 3))) return rc;
    if ((rc = h->d_results.ensure(2 * (size_t)frames))) return rc;  // two copies
    if ((rc = h->d_dbg.ensure(64))) return rc;
    if ((rc = h->h_results.ensure((size_t)frames))) return rc;
    if (mode == PWPP_MODE_FRESH) {
        if ((rc = h->d_st_fresh.ensure((size_t)frames))) return rc;
        if ((rc = h->d_hist_fresh.ensure((size_t)frames * 8 * (size_t)h->fresh_hist_cap))) return rc;
        if ((rc = h->d_pl_fresh.ensure((size_t)frames))) return rc;
    }
    if (from_host && (rc = h->d_in.ensure((size_t)(total_in > 0 ? total_in : 4)))) return rc;

    // ---- 2. frame descriptors; host inputs start their way to the device
    h->descs.resize((size_t)frames);
    int64_t base = 0, in_off = 0;
    const float *run_src = nullptr;  // pending host-to-device copy (merged run of adjacent frames)
    float *run_dst = nullptr;
    int64_t run_len = 0;
    for (int f = 0; f < frames; ++f) {
        PwppFrameDesc &d = h->descs[(size_t)f];
        std::memset(&d, 0, sizeof(d));
        d.n = n[f];
        d.cols = cols;
        d.layout = layout;
        if (fs) {
            d.step = fs->step;
            for (int k = 0; k < 4; ++k) d.off[k] = fs->off[k];
        }
        d.base = base;
        if (mode == PWPP_MODE_FRESH) {
            d.state_in = -1;
            d.state_out = f;
        } else {
            d.state_in = f;
            d.state_out = f;
        }
        if (from_host) {
            d.pts = h->d_in.p + in_off;
            // Frames that lie back to back in host memory (one slab per chunk) go over in copies of up to
            // 32 MB: 2 MB copies reach 34 GB/s on this PCIe Gen5 x16 link, 8-32 MB ones 49-52 GB/s.
            const int64_t fl = (int64_t)n[f] * floats_per_point;
            if (fl > 0) {
                if (run_len > 0 && points[f] == run_src + run_len && h->d_in.p + in_off == run_dst + run_len &&
                    (run_len + fl) * (int64_t)sizeof(float) <= ((int64_t)32 << 20)) {
                    run_len += fl;
                } else {
                    if (run_len > 0) HIPCHK(hipMemcpyAsync(run_dst, run_src, (size_t)run_len * sizeof(float), hipMemcpyHostToDevice, h->stream));
                    run_src = points[f];
                    run_dst = h->d_in.p + in_off;
                    run_len = fl;
                }
            }
            in_off += (fl + 3) & ~(int64_t)3;
        } else {
            d.pts = points[f];
        }
        h->h_base.p[f] = base;
        base += n[f];
    }
    h->h_base.p[frames] = base;
    if (run_len > 0) HIPCHK(hipMemcpyAsync(run_dst, run_src, (size_t)run_len * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->frames = frames;
    h->mode = mode;

    // ---- 3. One-pass binning (fixed bin segments, k_czm_bin_scatter) for batches of independent frames: the
    // bin-ordered buffers hold frames x slots_per_frame records instead of one per point.  Used when the memory
    // is there; any overflow is caught when the batch lands and the batch is redone exactly.
    bool one_pass = false;
    const size_t compact_slots = tp + (size_t)frames * (size_t)((PWPP_SLOT_ALIGN - 1) * NP + PWPP_SLOT_ALIGN);  // parts padded to multiples of PWPP_SLOT_ALIGN slots
    size_t bin_slots = compact_slots;
    if (h->one_pass_holdoff > 0) {
        --h->one_pass_holdoff;
    } else if (!h->no_one_pass && frames >= (mode == PWPP_MODE_FRESH ? h->one_pass_min_fresh : h->one_pass_min_frames) && max_n > 0) {
        if (2 * (int64_t)max_n < h->cap_max_n) {  // a much smaller sensor than the table was built for: start over
            HIPCHK(hipMemsetAsync(h->d_bin_max.p, 0, (size_t)NP * sizeof(uint32_t), h->stream));
            h->have_observation = false;
        }
        if (!h->have_observation) {
            if ((rc = h->d_codes.ensure(tp))) return rc;
            if ((rc = probe_histogram(h))) return rc;
            h->table_stale = true;
        }
        if (h->table_stale || max_n > h->cap_max_n || 2 * (int64_t)max_n < h->cap_max_n || h->cap_few != (frames <= 16 && h->one_pass_scale >= 1.0))
            if ((rc = build_capacity_table(h, max_n + max_n / 8))) return rc;
        const size_t want = (size_t)frames * (size_t)h->slots_per_frame;
        size_t free_b = 0, total_b = 0;
        const size_t per_slot = 3 * sizeof(float) + sizeof(int32_t) + 1;  // z, {x, y}, cloud index (+ the slot's bit and its share of the pads in the membership plane)
        const size_t held = (h->d_sorted_z.cap + h->d_sorted_idx.cap) * sizeof(int32_t) + h->d_sorted_xy.cap * sizeof(float2) + h->d_member.cap;
        const bool have = h->d_sorted_z.cap >= want + 1024 && h->d_sorted_xy.cap >= want + 1024 && h->d_sorted_idx.cap >= want;  // already allocated
        if (have || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (want + want / 8 + 64) * per_slot * 21 / 20 <= free_b + held &&
                     want < ((size_t)1 << 40))) {
            one_pass = true;
            bin_slots = want > compact_slots ? want : compact_slots;  // (a redo after an overflow uses the compact layout)
        }
    }

    // ---- 4. the bin-ordered buffers (slack: the fit kernels fetch whole chunks, up to 512 points beyond a patch's end)
    const size_t member_pads = (size_t)frames * (size_t)NP * 2 * PWPP_MEMBER_PAD + 4096;
    if (one_pass && (h->d_sorted_z.ensure(bin_slots + 1024) || h->d_sorted_xy.ensure(bin_slots + 1024) || h->d_sorted_idx.ensure(bin_slots) ||
                     h->d_member.ensure(bin_slots / 8 + member_pads))) {  // (the membership plane is part of the trial, not a hard error after it)
        one_pass = false;  // the big allocation failed after all (fragmentation): compact layout, two-pass binning
        bin_slots = compact_slots;
        (void)hipGetLastError();
    }
    if ((rc = h->d_sorted_z.ensure(bin_slots + 1024))) return rc;
    if ((rc = h->d_sorted_xy.ensure(bin_slots + 1024))) return rc;
    if ((rc = h->d_sorted_idx.ensure(bin_slots))) return rc;
    if ((rc = h->d_member.ensure(bin_slots / 8 + member_pads))) return rc;
    if (one_pass) {
        h->d_codes.release();  // (a cold handle's probe needed them; a redo gets them back: finish_pending)
    } else if ((rc = h->d_codes.ensure(tp))) {
        return rc;
    }
    if (one_pass && h->arena_slots && (rc = h->d_arena_tag.ensure((size_t)frames * h->arena_spill))) return rc;

    h->frames = frames;
    h->mode = mode;
    h->max_n = max_n;
    h->total_points = total;
    h->cols = cols;
    h->layout = layout;
    if (one_pass) {
        ++h->one_pass_batches;
        h->one_pass_frames += frames;
    }
    h->frame_two_pass.clear();
    if (one_pass && mode == PWPP_MODE_STREAMS) {  // stream i = frame i: keep what a redo must start from
        const size_t slab = (size_t)8 * (size_t)h->stream_hist_cap;
        if ((rc = h->d_st_snap.ensure((size_t)frames))) return rc;
        if ((rc = h->d_hist_snap.ensure((size_t)frames * slab))) return rc;
        if ((rc = h->d_pl_snap.ensure((size_t)frames))) return rc;
        // (the copies themselves are made by the binning kernel: PwppBatch.snap_*, set in launch_prepared)
    }
    if (frames > 64 && (rc = refresh_emit_long(h))) return rc;
    if ((rc = launch_prepared(h, one_pass))) return rc;
    h->pending = true;
    h->have_results = false;
    if (mem == PWPP_MEM_HOST) return finish_pending(h);  // the caller's buffers may go away
    return PWPP_OK;
}
}  // namespace

extern "C" {

int pwpp_estimate_ground(pwpp_handle *h, const float *points, int n, int cols, int layout) {
    const float *ptrs[1] = {points};
    const int32_t ns[1] = {n};
    return pwpp_estimate_ground_batch(h, ptrs, ns, 1, cols, layout, PWPP_MEM_HOST, PWPP_MODE_STREAMS);
}

int pwpp_synchronize(pwpp_handle *h) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    return finish_pending(h, true);  // (the call's results; a stream's state is settled by whoever touches it: settle_k5_tail)
}

int pwpp_get_counts(pwpp_handle *h, int frame, int32_t *n_ground, int32_t *n_nonground, int32_t *n_patches) {
    int rc = check_frame(h, frame);
    if (rc) return rc;
    const PwppFrameResult &r = h->h_results.p[frame];
    if (n_ground) *n_ground = r.n_ground;
    if (n_nonground) *n_nonground = r.n_nonground;
    if (n_patches) *n_patches = r.n_patches;
    return PWPP_OK;
}

static int copy_indices(pwpp_handle *h, int frame, int32_t *out, bool ground) {
    int rc = check_frame(h, frame);
    if (rc) return rc;
    const PwppFrameResult &r = h->h_results.p[frame];
    const int64_t base = h->h_base.p[frame];
    const int count = ground ? r.n_ground : r.n_nonground;
    if (count > 0) {
        if (!out) return fail(PWPP_E_ARG, "null output");
        HIPCHK(hipMemcpy(out, h->d_out.p + base + (ground ? 0 : r.n_ground), (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return PWPP_OK;
}
int pwpp_get_ground_indices(pwpp_handle *h, int frame, int32_t *out) { return copy_indices(h, frame, out, true); }
int pwpp_get_nonground_indices(pwpp_handle *h, int frame, int32_t *out) { return copy_indices(h, frame, out, false); }

static int copy_xyz(pwpp_handle *h, int frame, float *out, bool ground) {
    int rc = check_frame(h, frame);
    if (rc) return rc;
    const PwppFrameResult &r = h->h_results.p[frame];
    const int64_t base = h->h_base.p[frame];
    const int count = ground ? r.n_ground : r.n_nonground;
    if (count <= 0) return PWPP_OK;
    if (!out) return fail(PWPP_E_ARG, "null output");
    if ((rc = h->d_xyz.ensure((size_t)count * 3))) return rc;
    const int lrc = pwpp_launch_gather_xyz(&h->descs[(size_t)frame], h->d_out.p + base + (ground ? 0 : r.n_ground), count, h->d_xyz.p, h->stream);
    if (lrc != 0) return fail(PWPP_E_HIP, "gather launch failed: %s", hipGetErrorString((hipError_t)lrc));
    HIPCHK(hipMemcpyAsync(out, h->d_xyz.p, (size_t)count * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PWPP_OK;
}
int pwpp_get_ground_xyz(pwpp_handle *h, int frame, float *out) { return copy_xyz(h, frame, out, true); }
int pwpp_get_nonground_xyz(pwpp_handle *h, int frame, float *out) { return copy_xyz(h, frame, out, false); }

static int copy_patch_rows(pwpp_handle *h, int frame, float *out, const float *src_all) {
    int rc = check_frame(h, frame);
    if (rc) return rc;
    const int np = h->h_results.p[frame].n_patches;
    if (np <= 0) return PWPP_OK;
    if (!out) return fail(PWPP_E_ARG, "null output");
    HIPCHK(hipMemcpy(out, src_all + (size_t)frame * h->dp.num_bins * 3, (size_t)np * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return PWPP_OK;
}
int pwpp_get_centers(pwpp_handle *h, int frame, float *out) { return copy_patch_rows(h, frame, out, h ? h->d_centers.p : nullptr); }
int pwpp_get_normals(pwpp_handle *h, int frame, float *out) { return copy_patch_rows(h, frame, out, h ? h->d_normals.p : nullptr); }

int pwpp_get_patch_records(pwpp_handle *h, int frame, pwpp_patch_record *out, int capacity) {
    int rc = check_frame(h, frame);
    if (rc) return rc;
    const int B = h->dp.num_bins, NB = B + 2;
    std::vector<PwppPatchRec> recs((size_t)B);
    std::vector<uint32_t> cnt((size_t)NB);
    HIPCHK(hipMemcpy(recs.data(), h->d_recs.p + (size_t)frame * B, (size_t)B * sizeof(PwppPatchRec), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cnt.data(), h->d_bins.p + (size_t)frame * NB, (size_t)NB * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int k = 0, concentric = 0;
    for (int zone = 0; zone < 4; ++zone)
        for (int ring = 0; ring < h->dp.rings[zone]; ++ring, ++concentric)
            for (int sector = 0; sector < h->dp.sectors[zone]; ++sector) {
                const int bin = h->dp.bin_base[zone] + ring * h->dp.sectors[zone] + sector;
                if ((uint64_t)cnt[(size_t)bin] < h->dp.min_pts) continue;
                if (k < capacity && out) {
                    const PwppPatchRec &r = recs[(size_t)bin];
                    pwpp_patch_record &o = out[k];
                    std::memset(&o, 0, sizeof(o));
                    o.bin = bin;
                    o.concentric_idx = concentric;
                    o.n_points = (int)cnt[(size_t)bin];
                    o.n_ground = r.n_ground;
                    o.n_nonground = r.n_nonground;
                    o.decision = r.decision;
                    o.rounds = (r.valid >> 8) & 0xff;
                    for (int i = 0; i < 3; ++i) {
                        o.mean[i] = r.mean[i];
                        o.normal[i] = r.normal[i];
                        o.sv[i] = r.sv[i];
                    }
                    o.d = r.d;
                }
                ++k;
            }
    return k;  // number of patches (>= 0)
}

double pwpp_get_height(pwpp_handle *h) {
    pwpp_state s;
    if (!h) return 0.0;
    if (use_device(h) || finish_pending(h)) return 0.0;
    if (hipMemcpy(&s, h->d_st_stream.p, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
    return s.sensor_height;
}

double pwpp_get_time_us(pwpp_handle *h) {
    if (!h) return 0.0;
    if (use_device(h) || finish_pending(h, true)) return 0.0;
    return h->time_us;
}

int pwpp_get_state(pwpp_handle *h, int index, pwpp_state *out) {
    if (!h || !out) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    const bool fresh = h->have_results && h->mode == PWPP_MODE_FRESH;
    const int limit = fresh ? h->frames : h->num_streams;
    if (index < 0 || index >= limit) return fail(PWPP_E_ARG, "state index %d out of range [0,%d)", index, limit);
    const PwppStateScalar *src = (fresh ? h->d_st_fresh.p : h->d_st_stream.p) + index;
    HIPCHK(hipMemcpy(out, src, sizeof(pwpp_state), hipMemcpyDeviceToHost));
    return PWPP_OK;
}

int pwpp_get_history(pwpp_handle *h, int index, int which, int ring, double *out, int capacity) {
    pwpp_state s;
    int rc = pwpp_get_state(h, index, &s);
    if (rc) return rc;
    if (which < 0 || which > 1 || ring < 0 || ring > 3) return fail(PWPP_E_ARG, "bad history selector");
    const bool fresh = h->mode == PWPP_MODE_FRESH && h->have_results;
    const int cap = fresh ? h->fresh_hist_cap : h->stream_hist_cap;
    const double *base = (fresh ? h->d_hist_fresh.p : h->d_hist_stream.p) + ((size_t)index * 8 + (size_t)which * 4 + ring) * cap;
    const int len = which == 0 ? s.elevation_len[ring] : s.flatness_len[ring];
    const int ncopy = len < capacity ? len : capacity;
    if (ncopy > 0) {
        if (!out) return fail(PWPP_E_ARG, "null output");
        HIPCHK(hipMemcpy(out, base, (size_t)ncopy * sizeof(double), hipMemcpyDeviceToHost));
    }
    return len;
}

int pwpp_set_state(pwpp_handle *h, int stream, const pwpp_state *in) {
    if (!h || !in) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (stream < 0 || stream >= h->num_streams) return fail(PWPP_E_ARG, "stream %d out of range", stream);
    PwppStateScalar s;
    std::memcpy(&s, in, sizeof(s));
    for (int k = 0; k < 4; ++k) s.elev_len[k] = s.flat_len[k] = 0;
    HIPCHK(hipMemcpy(h->d_st_stream.p + stream, &s, sizeof(s), hipMemcpyHostToDevice));
    return PWPP_OK;
}

int pwpp_get_fxp_geometry(const pwpp_params *p, int *shift, float *out_xy, int capacity_bins) {
    if (!p) return fail(PWPP_E_ARG, "null params");
    PwppDevParams dp;
    const int rc = build_dev_params(*p, dp);
    if (rc < 0) return rc;
    std::vector<float2> origin;
    int s = 0;
    (void)fxp_geometry(dp, origin, s, true);
    if (shift) *shift = s;
    if (out_xy) {
        if (capacity_bins < dp.num_bins) return fail(PWPP_E_ARG, "room for %d bins, %d needed", capacity_bins, dp.num_bins);
        for (int b = 0; b < dp.num_bins; ++b) {
            out_xy[2 * b] = origin[(size_t)b].x;
            out_xy[2 * b + 1] = origin[(size_t)b].y;
        }
    }
    return dp.num_bins;
}

int pwpp_get_bin_boxes(const pwpp_params *p, float *out_boxes, int capacity_bins) {
    if (!p) return fail(PWPP_E_ARG, "null params");
    PwppDevParams dp;
    const int rc = build_dev_params(*p, dp);
    if (rc < 0) return rc;
    if (!out_boxes) return dp.num_bins;
    if (capacity_bins < dp.num_bins) return fail(PWPP_E_ARG, "room for %d bins, %d needed", capacity_bins, dp.num_bins);
    std::vector<float4> boxes;
    bin_boxes(dp, boxes);
    for (int b = 0; b < dp.num_bins; ++b) {
        out_boxes[4 * b] = boxes[(size_t)b].x;
        out_boxes[4 * b + 1] = boxes[(size_t)b].y;
        out_boxes[4 * b + 2] = boxes[(size_t)b].z;
        out_boxes[4 * b + 3] = boxes[(size_t)b].w;
    }
    return dp.num_bins;
}

int64_t pwpp_get_clamped_frames(pwpp_handle *h) {
    if (!h) return PWPP_E_ARG;
    if (use_device(h) || finish_pending(h)) return PWPP_E_HIP;
    return (int64_t)h->clamped_frames;
}

int64_t pwpp_get_fixed_up_frames(pwpp_handle *h) {
    if (!h) return PWPP_E_ARG;
    if (use_device(h) || finish_pending(h)) return PWPP_E_HIP;
    return (int64_t)h->fixed_up_frames;
}

int pwpp_get_plane_state(pwpp_handle *h, int index, float out[10]) {
    if (!h || !out) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    const bool fresh = h->have_results && h->mode == PWPP_MODE_FRESH;
    const int limit = fresh ? h->frames : h->num_streams;
    if (index < 0 || index >= limit) return fail(PWPP_E_ARG, "state index %d out of range [0,%d)", index, limit);
    PwppPlaneState ps;
    HIPCHK(hipMemcpy(&ps, (fresh ? h->d_pl_fresh.p : h->d_pl_stream.p) + index, sizeof(ps), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; ++i) {
        out[i] = ps.mean[i];
        out[3 + i] = ps.normal[i];
        out[6 + i] = ps.sv[i];
    }
    out[9] = (float)ps.d;  // (d is a float dot product widened to double in the reference, patchworkpp.cpp:74)
    return PWPP_OK;
}

int pwpp_set_plane_state(pwpp_handle *h, int stream, const float in[10]) {
    if (!h || !in) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (stream < 0 || stream >= h->num_streams) return fail(PWPP_E_ARG, "stream %d out of range", stream);
    PwppPlaneState ps;
    for (int i = 0; i < 3; ++i) {
        ps.mean[i] = in[i];
        ps.normal[i] = in[3 + i];
        ps.sv[i] = in[6 + i];
    }
    ps.pad_ = 0.0f;
    ps.d = (double)in[9];
    HIPCHK(hipMemcpy(h->d_pl_stream.p + stream, &ps, sizeof(ps), hipMemcpyHostToDevice));
    return PWPP_OK;
}

int pwpp_set_history(pwpp_handle *h, int stream, int which, int ring, const double *values, int count) {
    if (!h || (count > 0 && !values)) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (stream < 0 || stream >= h->num_streams) return fail(PWPP_E_ARG, "stream %d out of range", stream);
    if (which < 0 || which > 1 || ring < 0 || ring > 3) return fail(PWPP_E_ARG, "bad history selector");
    if (count < 0 || count > (1 << 24)) return fail(PWPP_E_ARG, "count %d out of range", count);
    // (count is NOT limited to max_*_storage: the reference only trims a history in update_elevation_thr / update_flatness_thr,
    // which stop at the first ring without data, so a sensor that sees no ground in ring 0 grows the others without bound --
    // test_histories_the_reference_never_trims restores exactly such a state)
    for (int i = 0; i < count; ++i)
        if (!(std::fabs(values[i]) <= DBL_MAX)) return fail(PWPP_E_ARG, "history value %d is not finite: the reference only ever pushes finite heights and flatness values", i);
    while (count + h->max_pushes_per_frame + 8 > h->stream_hist_cap)
        if ((rc = grow_stream_histories(h, 2 * h->stream_hist_cap))) return rc;
    double *dst = h->d_hist_stream.p + ((size_t)stream * 8 + (size_t)which * 4 + ring) * h->stream_hist_cap;
    if (count > 0) HIPCHK(hipMemcpy(dst, values, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    const size_t field = which == 0 ? offsetof(PwppStateScalar, elev_len) : offsetof(PwppStateScalar, flat_len);
    char *dlen = reinterpret_cast<char *>(h->d_st_stream.p + stream) + field + (size_t)ring * sizeof(int32_t);
    const int32_t c32 = count;
    HIPCHK(hipMemcpy(dlen, &c32, sizeof(int32_t), hipMemcpyHostToDevice));
    return PWPP_OK;
}

int pwpp_get_device_view(pwpp_handle *h, pwpp_device_view *out) {
    if (!h || !out) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (!h->have_results) return fail(PWPP_E_STATE, "no frame has been processed yet");
    out->indices = h->d_out.p;
    out->frame_base = h->h_base.p;
    out->counts = reinterpret_cast<const int32_t *>(h->h_results.p);
    out->frames = h->frames;
    out->pad_ = 0;
    return PWPP_OK;
}

int pwpp_host_alloc(void **out, uint64_t bytes) {
    if (!out) return fail(PWPP_E_ARG, "null argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, (size_t)(bytes ? bytes : 1), hipHostMallocDefault);
    if (e != hipSuccess) return fail(PWPP_E_NOMEM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return PWPP_OK;
}
int pwpp_host_free(void *p) {
    if (p) HIPCHK(hipHostFree(p));
    return PWPP_OK;
}

int pwpp_get_all_indices(pwpp_handle *h, int32_t *out, int64_t *frame_base, int32_t *counts) {
    int rc = check_frame(h, 0);
    if (rc) return rc;
    if (!out) return fail(PWPP_E_ARG, "null output");
    const int64_t total = h->h_base.p[h->frames];
    if (total > 0) {
        HIPCHK(hipMemcpyAsync(out, h->d_out.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (frame_base) std::memcpy(frame_base, h->h_base.p, ((size_t)h->frames + 1) * sizeof(int64_t));
    if (counts) std::memcpy(counts, h->h_results.p, (size_t)h->frames * sizeof(PwppFrameResult));
    return PWPP_OK;
}

int pwpp_set_profiling(pwpp_handle *h, int enable) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    h->profiling = enable != 0;
    return PWPP_OK;
}
int pwpp_get_kernel_profile(pwpp_handle *h, double *sum_ms, int64_t *launches) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    for (int k = 0; k < PWPP_NUM_KERNELS; ++k) {
        if (sum_ms) sum_ms[k] = h->prof_ms[k];
        if (launches) launches[k] = h->prof_launches[k];
    }
    return PWPP_OK;
}
int pwpp_reset_kernel_profile(pwpp_handle *h) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    for (int k = 0; k < PWPP_NUM_KERNELS; ++k) {
        h->prof_ms[k] = 0.0;
        h->prof_launches[k] = 0;
    }
    return PWPP_OK;
}
int pwpp_get_fxp_shift(pwpp_handle *h) { return h ? h->dp.fxp_shift : PWPP_E_ARG; }

int pwpp_get_fxp_origins(pwpp_handle *h, float *out_xy, int capacity_bins) {
    if (!h) return fail(PWPP_E_ARG, "null argument");
    const int B = h->dp.num_bins;
    if (!out_xy) return B;  // size query
    if (capacity_bins < B) return fail(PWPP_E_ARG, "room for %d bins, %d needed", capacity_bins, B);
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(hipMemcpy(out_xy, h->d_bin_origin.p, (size_t)B * sizeof(float2), hipMemcpyDeviceToHost));
    return B;
}

int pwpp_set_option(pwpp_handle *h, const char *name, const char *value) {
    if (!h || !name || !value) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    const std::string k = name;
    if (k == "fit_plan") {
        for (const char *c = value; *c; ++c)
            if (!std::strchr("SWBH0123456789.:,", *c)) return fail(PWPP_E_ARG, "fit_plan '%s': unexpected character '%c'", value, *c);
        h->fit_plan = value;
    } else if (k == "fit_concurrent") {
        h->fit_concurrent = std::atoi(value) != 0;
    } else if (k == "one_pass") {
        h->no_one_pass = std::atoi(value) == 0;
    } else if (k == "cu_split") {
        // "N" or "N:mode": the overlap schedule's memory stream (binning, index lists) on N of the 256 CUs, the fit streams on the
        // others.  mode 0: the N lowest bits of the mask; mode 1: the bits i with (i mod 8) < N / 32 (whichever of the two is "whole
        // XCDs" depends on how the driver numbers the CUs of the eight XCDs in a queue's mask).  "0": off.
        int n = 0, mode = 0;
        if (std::sscanf(value, "%d:%d", &n, &mode) < 1 || n < 0 || n >= 256 || (mode != 0 && mode != 1))
            return fail(PWPP_E_ARG, "cu_split=%s: N or N:mode with 0 <= N < 256, mode 0 or 1 expected", value);
        //  mode 1 deals whole groups of 32 CUs: (i mod 8) < N / 32 is empty below 32 and everything at 256 -- an all-zero
        // CU mask is not a stream the runtime can create
        if (mode == 1 && n != 0 && (n % 32 != 0 || n < 32 || n > 224))
            return fail(PWPP_E_ARG, "cu_split=%s: mode 1 takes N = 32, 64, ... 224 (whole groups of 32 CUs)", value);
        sync_all_streams(h);
        for (hipStream_t st : h->masked_fit) (void)hipStreamDestroy(st);
        h->masked_fit.clear();
        if (h->masked_mem) (void)hipStreamDestroy(h->masked_mem);
        h->masked_mem = nullptr;
        h->cu_split_mem = n;
        h->cu_split_mode = mode;
        if (n > 0) {
            uint32_t mm[8] = {}, mf[8] = {};
            for (int i = 0; i < 256; ++i) {
                const bool to_mem = mode == 0 ? i < n : (i % 8) < n / 32;
                (to_mem ? mm : mf)[i / 32] |= 1u << (i % 32);
            }
            bool mem_empty = true, fit_empty = true;
            for (int w = 0; w < 8; ++w) {
                mem_empty = mem_empty && mm[w] == 0u;
                fit_empty = fit_empty && mf[w] == 0u;
            }
            if (mem_empty || fit_empty) {
                h->cu_split_mem = 0;
                return fail(PWPP_E_ARG, "cu_split=%s: one of the two CU masks comes out empty", value);
            }
            HIPCHK(hipExtStreamCreateWithCUMask(&h->masked_mem, 8, mm));
            for (int k2 = 0; k2 < 8; ++k2) {
                hipStream_t st = nullptr;
                HIPCHK(hipExtStreamCreateWithCUMask(&st, 8, mf));
                h->masked_fit.push_back(st);
            }
        }
    } else if (k == "exact_moments") {
        // the width of the plane-fit sums (include/pwpp.h): "1" = contract v4, a 2^-30 m grid; "0" = rounds 3-5's 2^-21 m grid.
        // The per-bin origins do not depend on it; shift and z half-range do.
        const int v = std::atoi(value);
        if ((v != 0 && v != 1) || (value[0] != '0' && value[0] != '1') || value[1] != 0) return fail(PWPP_E_ARG, "exact_moments=%s: 0 or 1 expected", value);
        std::vector<float2> origin;
        h->dp.fxp_wide = v;
        h->dp.fxp_zr = (float)fxp_geometry(h->dp, origin, h->dp.fxp_shift, v != 0);
    } else if (k == "fuse_scan") {
        h->fuse_scan = std::atoi(value) != 0;
    } else if (k == "split_k5") {
        const int v = std::atoi(value);
        if (v < 0 || v > 2) return fail(PWPP_E_ARG, "split_k5=%s: 0, 1 or 2 expected", value);
        h->split_k5 = v;
    } else if (k == "redo_whole_batch") {
        h->redo_whole_batch = std::atoi(value) != 0;
    } else if (k == "one_pass_min_frames") {
        const int v = std::atoi(value);
        if (v < 1) return fail(PWPP_E_ARG, "one_pass_min_frames=%s: >= 1 expected", value);
        h->one_pass_min_frames = h->one_pass_min_fresh = v;
    } else if (k == "one_pass_scale") {
        const double v = std::atof(value);
        if (!(v > 0.0 && v <= 1024.0)) return fail(PWPP_E_ARG, "one_pass_scale=%s: a positive number up to 1024 expected", value);
        h->one_pass_scale = v;
        h->table_stale = true;  // rebuild the capacity table
    } else if (k == "bin_block") {
        const int v = std::atoi(value);
        if (v != 128 && v != 256 && v != 512 && v != 1024) return fail(PWPP_E_ARG, "bin_block=%s: 128, 256, 512 or 1024 expected", value);
        h->bin_block = v;
    } else if (k == "fit_streams") {
        const int v = std::atoi(value);
        if (v < 1 || v > 8) return fail(PWPP_E_ARG, "fit_streams=%s: 1..8 expected", value);
        h->num_fit_streams = v;
    } else if (k == "overlap_mode") {
        const int v = std::atoi(value);
        if (v != 0 && v != 1) return fail(PWPP_E_ARG, "overlap_mode=%s: 0 (whole ranges side by side) or 1 (memory / fit pipeline) expected", value);
        h->overlap_mode = v;
    } else if (k == "overlap_ranges") {
        const int v = std::atoi(value);
        if (v < 2 || v > 64) return fail(PWPP_E_ARG, "overlap_ranges=%s: 2..64 expected", value);
        h->overlap_ranges = v;
    } else if (k == "hi_split") {
        // height over the ground level where the high part of a bin begins (pwpp_dev.h); results never depend on it
        const double v = std::atof(value);
        if (!(v >= -1e30 && v <= 1e30)) return fail(PWPP_E_ARG, "hi_split=%s: a number of metres expected (1e30: no high parts)", value);
        h->dp.hi_split = (float)v;
        HIPCHK(hipMemsetAsync(h->d_bin_max.p, 0, (size_t)PWPP_NUM_PARTS(h->dp.num_bins) * sizeof(uint32_t), h->stream));
        h->have_observation = false;  // the parts change: size the one-pass segments anew
        h->table_stale = true;
    } else if (k == "hi_split_zones") {
        const int v = std::atoi(value);
        if (v < 0 || v > 4) return fail(PWPP_E_ARG, "hi_split_zones=%s: 0..4 expected", value);
        h->dp.split_end = h->dp.bin_base[v];
        HIPCHK(hipMemsetAsync(h->d_bin_max.p, 0, (size_t)PWPP_NUM_PARTS(h->dp.num_bins) * sizeof(uint32_t), h->stream));
        h->have_observation = false;
        h->table_stale = true;
    } else if (k == "debug_flags") {
        const int v = std::atoi(value);
        //  the fit-chain probes (4) and the binning / scan / GLE probes (8) write the same 64-entry array
        if ((v & 4) && (v & 8)) return fail(PWPP_E_ARG, "debug_flags=%s: the timing probes 4 and 8 share one probe array; set one of them", value);
        h->debug_flags = v;
    } else {
        return fail(PWPP_E_ARG, "unknown option '%s'", name);
    }
    return PWPP_OK;
}

int64_t pwpp_get_workspace_bytes(pwpp_handle *h) {
    if (!h) return PWPP_E_ARG;
    // EVERY device allocation of the handle (the sum used to leave out the descriptors, the stream state and
    // the tables, and reported the per-frame buffers pwpp_trim_workspace did not free)
    auto b = [](size_t cap, size_t elt) { return (int64_t)(cap * elt); };
    return b(h->d_frames.cap, sizeof(PwppFrameDesc)) + b(h->d_frames_probe.cap, sizeof(PwppFrameDesc)) + b(h->d_in.cap, 4) + b(h->d_codes.cap, 2) +
           b(h->d_sorted_z.cap, 4) + b(h->d_sorted_xy.cap, 8) + b(h->d_sorted_idx.cap, 4) + b(h->d_bin_origin.cap, 8) + b(h->d_bin_bbox.cap, 16) +
           b(h->d_member.cap, 1) + b(h->d_arena_tag.cap, 8) + b(h->d_out.cap, 4) + b(h->d_ord_a.cap, 8) + b(h->d_ord_b.cap, 8) + b(h->d_ord_work.cap, 4) + b(h->d_bins.cap, 4) + b(h->d_parts.cap, 4) +
           b(h->d_cls_start.cap, 4) + b(h->d_cap_off.cap, 4) + b(h->d_emit_long.cap, 1) + b(h->d_bin_max.cap, 4) + b(h->d_cls_list.cap, 2) +
           b(h->d_recs.cap, sizeof(PwppPatchRec)) + b(h->d_centers.cap, 4) + b(h->d_normals.cap, 4) + b(h->d_results.cap, sizeof(PwppFrameResult)) +
           b(h->d_xyz.cap, 4) + b(h->d_dbg.cap, 8) + b(h->d_st_stream.cap, sizeof(PwppStateScalar)) + b(h->d_st_fresh.cap, sizeof(PwppStateScalar)) +
           b(h->d_st_snap.cap, sizeof(PwppStateScalar)) + b(h->d_hist_stream.cap, 8) + b(h->d_hist_fresh.cap, 8) + b(h->d_hist_snap.cap, 8) +
           b(h->d_pl_stream.cap, sizeof(PwppPlaneState)) + b(h->d_pl_fresh.cap, sizeof(PwppPlaneState)) + b(h->d_pl_snap.cap, sizeof(PwppPlaneState));
}

int pwpp_trim_workspace(pwpp_handle *h) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    // everything whose size follows the batch: inputs, bin-ordered planes, lists, per-frame tables and records, the state of
    // FRESH frames and the snapshots of a one-pass batch.  What stays: the streams' state (thresholds, histories, plane
    // members), the per-handle tables (origins, boxes, segment table, largest counts) and 512 bytes of timing probes.
    h->d_in.release();
    h->d_codes.release();
    h->d_sorted_z.release();
    h->d_sorted_xy.release();
    h->d_sorted_idx.release();
    h->d_member.release();
    h->d_arena_tag.release();
    h->d_out.release();
    h->d_ord_a.release();
    h->d_ord_b.release();
    h->d_ord_work.release();
    h->d_xyz.release();
    h->d_frames.release();
    h->d_frames_probe.release();
    h->d_bins.release();
    h->d_parts.release();
    h->d_recs.release();
    h->d_cls_start.release();
    h->d_cls_list.release();
    h->d_centers.release();
    h->d_normals.release();
    h->d_results.release();
    h->d_st_fresh.release();
    h->d_hist_fresh.release();
    h->d_pl_fresh.release();
    h->d_st_snap.release();
    h->d_hist_snap.release();
    h->d_pl_snap.release();
    h->have_results = false;  // the index lists lived in d_out, the records in d_recs
    h->next_clean = false;    // (the counters' copies are gone with d_parts / d_results)
    h->descs_on_device.clear();
    return PWPP_OK;
}

int pwpp_set_output_order(pwpp_handle *h, int order) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    if (order != PWPP_ORDER_SCATTER && order != PWPP_ORDER_REFERENCE) return fail(PWPP_E_ARG, "bad order %d", order);
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    h->output_order = order;
    return PWPP_OK;
}

int pwpp_set_overlap(pwpp_handle *h, int on) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    h->overlap = on != 0;
    return PWPP_OK;
}

int pwpp_get_one_pass_stats(pwpp_handle *h, int64_t *batches, int64_t *redone) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (batches) *batches = h->one_pass_batches;
    if (redone) *redone = h->one_pass_redone;
    return PWPP_OK;
}

int pwpp_get_arena_stats(pwpp_handle *h, int64_t *frames_with_moved_parts, int64_t *slots_per_frame, int64_t *arena_slots) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (frames_with_moved_parts) *frames_with_moved_parts = h->arena_frames;
    if (slots_per_frame) *slots_per_frame = h->slots_per_frame;
    if (arena_slots) *arena_slots = h->arena_slots;
    return PWPP_OK;
}

int pwpp_get_redo_stats(pwpp_handle *h, int64_t *frames_one_pass, int64_t *frames_redone) {
    if (!h) return fail(PWPP_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (frames_one_pass) *frames_one_pass = h->one_pass_frames;
    if (frames_redone) *frames_redone = h->one_pass_redone_frames;
    return PWPP_OK;
}

/* timing probes of the last call (PWPP_DEBUG_FLAGS & 4) */
int pwpp_debug_read(pwpp_handle *h, unsigned long long *out64) {
    if (!h || !out64) return fail(PWPP_E_ARG, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = finish_pending(h))) return rc;
    if (!h->d_dbg.p) return fail(PWPP_E_STATE, "no probes");
    HIPCHK(hipMemcpy(out64, h->d_dbg.p, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PWPP_OK;
}


/* ---- batches in flight (include/pwpp.h) ---- */
struct pwpp_pipe {
    std::vector<pwpp_handle *> handles;
    unsigned long long submits = 0;
};

int pwpp_pipe_create(const pwpp_params *p, int device, int depth, pwpp_pipe **out) {
    if (!p || !out) return fail(PWPP_E_ARG, "null argument");
    if (depth < 1 || depth > 4) return fail(PWPP_E_ARG, "depth %d: 1..4 expected", depth);
    pwpp_pipe *pipe = new (std::nothrow) pwpp_pipe;
    if (!pipe) return fail(PWPP_E_NOMEM, "out of host memory");
    for (int k = 0; k < depth; ++k) {
        pwpp_handle *h = nullptr;
        const int rc = pwpp_create(p, device, &h);
        if (rc) {
            (void)pwpp_pipe_destroy(pipe);
            return rc;
        }
        if (depth > 1) h->overlap = false;  // (the in-handle overlap schedule on top of batches in flight is slower: 2.83 against 2.46 ms)
        pipe->handles.push_back(h);
    }
    *out = pipe;
    return PWPP_OK;
}

int pwpp_pipe_set_num_streams(pwpp_pipe *pipe, int streams_per_handle) {
    if (!pipe || pipe->handles.empty()) return fail(PWPP_E_ARG, "null pipe");
    for (pwpp_handle *h : pipe->handles) {
        const int rc = pwpp_set_num_streams(h, streams_per_handle);
        if (rc) return rc;
    }
    return PWPP_OK;
}

int pwpp_pipe_submit(pwpp_pipe *pipe, const float *const *points, const int32_t *n, int frames, int cols, int layout, int mem, int mode,
                     pwpp_handle **holder) {
    if (!pipe || pipe->handles.empty()) return fail(PWPP_E_ARG, "null pipe");
    pwpp_handle *h = pipe->handles[(size_t)(pipe->submits % pipe->handles.size())];
    // (pwpp_estimate_ground_batch first waits for this handle's own batch in flight: the one submitted `depth` submits ago.
    // PWPP_MODE_STREAMS: the handle owns the streams of ITS group -- submit k carries group k mod depth, so a stream's frames
    // stay in order on one handle; the mode is checked there, together with the stream count)
    const int rc = pwpp_estimate_ground_batch(h, points, n, frames, cols, layout, mem, mode);
    if (rc) return rc;
    ++pipe->submits;
    if (holder) *holder = h;
    return PWPP_OK;
}

int pwpp_pipe_drain(pwpp_pipe *pipe) {
    if (!pipe) return fail(PWPP_E_ARG, "null pipe");
    for (pwpp_handle *h : pipe->handles) {
        const int rc = pwpp_synchronize(h);
        if (rc) return rc;
    }
    return PWPP_OK;
}

pwpp_handle *pwpp_pipe_handle(pwpp_pipe *pipe, int index) {
    return pipe && index >= 0 && (size_t)index < pipe->handles.size() ? pipe->handles[(size_t)index] : nullptr;
}

int pwpp_pipe_destroy(pwpp_pipe *pipe) {
    if (!pipe) return PWPP_OK;
    int rc = PWPP_OK;
    for (pwpp_handle *h : pipe->handles) {
        const int r = pwpp_destroy(h);
        rc = rc ? rc : r;
    }
    delete pipe;
    return rc;
}

}  // extern "C"
