// oracle/ref_capi.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C wrapper (oracle/oracle_api.h) around the reference's own class
// patchwork::PatchWorkpp.  Built by oracle/Makefile together with the
// UNMODIFIED /root/reference/cpp/patchworkpp/src/patchworkpp.cpp and the
// Eigen stand-in oracle/eigen_shim into oracle/_ref/libpwpp_ref.so (eigen-f32
// flavour), libpwpp_ref_exact.so (exact-f64 arbiter) and libpwpp_ref_pk4.so
// (float, 4-lane summation order).
// No reference source is copied: the compiler reads it where it lies.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <numeric>
#include <thread>
#include <vector>

#include <Eigen/Dense>  // the shim; also pulls in every std header the reference needs

// The adaptive thresholds and histories are private members with no getter
// (reference patchworkpp.h:169-175).  Tests must compare them, so this one
// translation unit sees the class with its private section opened.  Access
// specifiers do not change the object layout; patchworkpp.cpp itself is
// compiled without this.
#define private public
#include "patchwork/patchworkpp.h"
#undef private

#include "oracle_api.h"

namespace {

patchwork::Params to_ref_params(const pwo_params &p) {
    patchwork::Params r;
    r.verbose = p.verbose != 0;
    r.enable_RNR = p.enable_RNR != 0;
    r.enable_RVPF = p.enable_RVPF != 0;
    r.enable_TGR = p.enable_TGR != 0;
    r.num_iter = p.num_iter;
    r.num_lpr = p.num_lpr;
    r.num_min_pts = p.num_min_pts;
    r.num_zones = p.num_zones;
    r.num_rings_of_interest = p.num_rings_of_interest;
    r.RNR_ver_angle_thr = p.RNR_ver_angle_thr;
    r.RNR_intensity_thr = p.RNR_intensity_thr;
    r.sensor_height = p.sensor_height;
    r.th_seeds = p.th_seeds;
    r.th_dist = p.th_dist;
    r.th_seeds_v = p.th_seeds_v;
    r.th_dist_v = p.th_dist_v;
    r.max_range = p.max_range;
    r.min_range = p.min_range;
    r.uprightness_thr = p.uprightness_thr;
    r.adaptive_seed_selection_margin = p.adaptive_seed_selection_margin;
    r.num_sectors_each_zone.assign(p.num_sectors_each_zone, p.num_sectors_each_zone + 4);
    r.num_rings_each_zone.assign(p.num_rings_each_zone, p.num_rings_each_zone + 4);
    r.max_flatness_storage = p.max_flatness_storage;
    r.max_elevation_storage = p.max_elevation_storage;
    r.elevation_thr.assign(p.elevation_thr, p.elevation_thr + 4);
    r.flatness_thr.assign(p.flatness_thr, p.flatness_thr + 4);
    return r;
}

Eigen::MatrixXf to_matrix(const float *pts, int n, int cols) {
    Eigen::MatrixXf m(n, cols);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < cols; ++j) m(i, j) = pts[(size_t)i * cols + j];
    return m;
}

void copy_rows(const Eigen::MatrixX3f &m, float *out) {
    for (int i = 0; i < m.rows(); ++i)
        for (int j = 0; j < 3; ++j) out[(size_t)i * 3 + j] = m(i, j);
}

// The reference prints a line from its constructor (patchworkpp.h:149).
struct MuteCout {
    std::streambuf *old;
    MuteCout() : old(std::cout.rdbuf(nullptr)) {}
    ~MuteCout() { std::cout.rdbuf(old); }
};

struct Handle {
    patchwork::PatchWorkpp *pw;
};

}  // namespace

extern "C" {

void pwo_default_params(pwo_params *p) {
    patchwork::Params d;
    std::memset(p, 0, sizeof(*p));
    p->verbose = d.verbose;
    p->enable_RNR = d.enable_RNR;
    p->enable_RVPF = d.enable_RVPF;
    p->enable_TGR = d.enable_TGR;
    p->num_iter = d.num_iter;
    p->num_lpr = d.num_lpr;
    p->num_min_pts = d.num_min_pts;
    p->num_zones = d.num_zones;
    p->num_rings_of_interest = d.num_rings_of_interest;
    p->RNR_ver_angle_thr = d.RNR_ver_angle_thr;
    p->RNR_intensity_thr = d.RNR_intensity_thr;
    p->sensor_height = d.sensor_height;
    p->th_seeds = d.th_seeds;
    p->th_dist = d.th_dist;
    p->th_seeds_v = d.th_seeds_v;
    p->th_dist_v = d.th_dist_v;
    p->max_range = d.max_range;
    p->min_range = d.min_range;
    p->uprightness_thr = d.uprightness_thr;
    p->adaptive_seed_selection_margin = d.adaptive_seed_selection_margin;
    for (int k = 0; k < 4; ++k) {
        p->num_sectors_each_zone[k] = d.num_sectors_each_zone[k];
        p->num_rings_each_zone[k] = d.num_rings_each_zone[k];
        p->elevation_thr[k] = d.elevation_thr[k];
        p->flatness_thr[k] = d.flatness_thr[k];
    }
    p->max_flatness_storage = d.max_flatness_storage;
    p->max_elevation_storage = d.max_elevation_storage;
}

int pwo_arith_supported(int arith) { return arith == PWPP_SHIM_ARITH; }  // flavours are compile-time here

void *pwo_create(const pwo_params *p, int arith) {
    if (!pwo_arith_supported(arith)) return nullptr;
    MuteCout mute;
    Handle *h = new Handle;
    h->pw = new patchwork::PatchWorkpp(to_ref_params(*p));
    return h;
}

void pwo_destroy(void *hv) {
    Handle *h = (Handle *)hv;
    if (!h) return;
    delete h->pw;
    delete h;
}

int pwo_estimate_ground(void *hv, const float *pts, int n, int cols) {
    Handle *h = (Handle *)hv;
    h->pw->estimateGround(to_matrix(pts, n, cols));
    return 0;
}

int pwo_num_ground(void *hv) { return (int)((Handle *)hv)->pw->cloud_ground_.size(); }
int pwo_num_nonground(void *hv) { return (int)((Handle *)hv)->pw->cloud_nonground_.size(); }
int pwo_num_patches(void *hv) { return (int)((Handle *)hv)->pw->centers_.size(); }

void pwo_get_ground_indices(void *hv, int32_t *out) {
    Eigen::VectorXi v = ((Handle *)hv)->pw->getGroundIndices();
    for (int i = 0; i < v.rows(); ++i) out[i] = v(i);
}
void pwo_get_nonground_indices(void *hv, int32_t *out) {
    Eigen::VectorXi v = ((Handle *)hv)->pw->getNongroundIndices();
    for (int i = 0; i < v.rows(); ++i) out[i] = v(i);
}
void pwo_get_ground(void *hv, float *out) { copy_rows(((Handle *)hv)->pw->getGround(), out); }
void pwo_get_nonground(void *hv, float *out) { copy_rows(((Handle *)hv)->pw->getNonground(), out); }
void pwo_get_centers(void *hv, float *out) { copy_rows(((Handle *)hv)->pw->getCenters(), out); }
void pwo_get_normals(void *hv, float *out) { copy_rows(((Handle *)hv)->pw->getNormals(), out); }
double pwo_get_height(void *hv) { return ((Handle *)hv)->pw->getHeight(); }
double pwo_get_time_taken(void *hv) { return ((Handle *)hv)->pw->getTimeTaken(); }

void pwo_get_thresholds(void *hv, double *sensor_height, double *elev, double *flat) {
    patchwork::PatchWorkpp *pw = ((Handle *)hv)->pw;
    *sensor_height = pw->params_.sensor_height;
    for (int k = 0; k < 4; ++k) {
        elev[k] = pw->params_.elevation_thr[k];
        flat[k] = pw->params_.flatness_thr[k];
    }
}
int pwo_get_history_len(void *hv, int which, int ring) {
    patchwork::PatchWorkpp *pw = ((Handle *)hv)->pw;
    return (int)(which == 0 ? pw->update_elevation_[ring].size() : pw->update_flatness_[ring].size());
}
void pwo_get_history(void *hv, int which, int ring, double *out) {
    patchwork::PatchWorkpp *pw = ((Handle *)hv)->pw;
    const std::vector<double> &v = which == 0 ? pw->update_elevation_[ring] : pw->update_flatness_[ring];
    std::copy(v.begin(), v.end(), out);
}

void pwo_get_counters(long *plane_fits, long *jacobi_sweeps) {
    *plane_fits = Eigen::shim::counters().plane_fits;
    *jacobi_sweeps = Eigen::shim::counters().jacobi_sweeps;
}

double pwo_bench(const pwo_params *p, int arith, const float *const *frames, const int *n_points, int cols,
                 int num_distinct, int total, int threads, double *sum_call_seconds) {
    if (!pwo_arith_supported(arith) || threads < 1) return -1.0;
    std::vector<Eigen::MatrixXf> mats;
    for (int k = 0; k < num_distinct; ++k) mats.push_back(to_matrix(frames[k], n_points[k], cols));
    const patchwork::Params rp = to_ref_params(*p);
    MuteCout mute;
    std::vector<double> call_s((size_t)threads, 0.0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            for (int i = t; i < total; i += threads) {
                patchwork::PatchWorkpp pw(rp);  // fresh state per frame
                auto a = std::chrono::steady_clock::now();
                pw.estimateGround(mats[(size_t)(i % num_distinct)]);
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
