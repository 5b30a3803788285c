// patchwork/patchworkpp.h -- host-side C++ mirror of the reference's public interface for the
// estimateGround() path, backed by the MI355X library (include/pwpp.h, libpwpp_hip.so).
//
// Mirrors /root/reference/cpp/patchworkpp/include/patchwork/patchworkpp.h:
//   patchwork::Params        (:42-112)  same field names, types and defaults
//   patchwork::PatchWorkpp   (:114-163) same constructor, estimateGround(), and getters
// so that code written against the reference header compiles against this one (see
// INTEGRATION.md).  Eigen is optional here (the container / no-network build has none):
//  * with <Eigen/Dense> on the include path the class has the reference's EXACT signatures --
//    estimateGround(Eigen::MatrixXf), getGround() etc. return real Eigen::MatrixX3f / Eigen::VectorXi
//    objects, so `pw.getGround().row(i)`, `.transpose()`, `auto g = pw.getNormals(); g.col(2)` compile
//    as against the reference;
//  * without it (or with PWPP_NO_EIGEN) the getters return the small row-major containers below and
//    estimateGround takes a raw pointer.
//
// Differences a caller can observe, all documented in DESIGN.md section 7 and INTEGRATION.md section 5:
//  * index and point lists hold the same SETS as the reference, ordered by the device
//    pipeline, not by the reference's bin traversal / z order;
//  * getTimeTaken() is GPU time in microseconds (the reference reports CPU clock ticks = us);
//  * errors throw std::runtime_error instead of printing.
#ifndef PATCHWORKPP_AMD_H
#define PATCHWORKPP_AMD_H

#include <cstdint>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pwpp.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(PWPP_NO_EIGEN)
#include <Eigen/Dense>
#define PWPP_HAVE_EIGEN 1
#endif
#endif

namespace patchwork {

// reference patchworkpp.h:42-112
struct Params {
    bool verbose;
    bool enable_RNR;
    bool enable_RVPF;
    bool enable_TGR;

    int num_iter;
    int num_lpr;
    int num_min_pts;
    int num_zones;
    int num_rings_of_interest;

    double RNR_ver_angle_thr;
    double RNR_intensity_thr;

    double sensor_height;
    double th_seeds;
    double th_dist;
    double th_seeds_v;
    double th_dist_v;
    double max_range;
    double min_range;
    double uprightness_thr;
    double adaptive_seed_selection_margin;
    double intensity_thr;

    std::vector<int> num_sectors_each_zone;
    std::vector<int> num_rings_each_zone;

    int max_flatness_storage;
    int max_elevation_storage;

    std::vector<double> elevation_thr;
    std::vector<double> flatness_thr;

    Params() {
        verbose = false;
        enable_RNR = true;
        enable_RVPF = true;
        enable_TGR = true;

        num_iter = 3;
        num_lpr = 20;
        num_min_pts = 10;
        num_zones = 4;
        num_rings_of_interest = 4;

        RNR_ver_angle_thr = -15.0;
        RNR_intensity_thr = 0.2;

        sensor_height = 1.723;
        th_seeds = 0.125;
        th_dist = 0.125;
        th_seeds_v = 0.25;
        th_dist_v = 0.1;
        max_range = 80.0;
        min_range = 2.7;
        uprightness_thr = 0.707;
        adaptive_seed_selection_margin = -1.2;
        intensity_thr = 0.0;  // left uninitialised by the reference (:67); never read by the path

        num_sectors_each_zone = {16, 32, 54, 32};
        num_rings_each_zone = {2, 4, 4, 4};

        max_flatness_storage = 1000;
        max_elevation_storage = 1000;
        elevation_thr = {0, 0, 0, 0};
        flatness_thr = {0, 0, 0, 0};
    }
};

// Row-major (rows, 3) float container returned by getGround()/getNonground()/getCenters()/getNormals().
class Cloud {
public:
    Cloud() : rows_(0) {}
    explicit Cloud(int rows) : rows_(rows), v_((size_t)rows * 3) {}
    int rows() const { return rows_; }
    int cols() const { return 3; }
    float operator()(int i, int j) const { return v_[(size_t)i * 3 + j]; }
    float *data() { return v_.data(); }
    const float *data() const { return v_.data(); }
#ifdef PWPP_HAVE_EIGEN
    operator Eigen::MatrixX3f() const {
        Eigen::MatrixX3f m(rows_, 3);
        for (int i = 0; i < rows_; ++i)
            for (int j = 0; j < 3; ++j) m(i, j) = (*this)(i, j);
        return m;
    }
#endif
private:
    int rows_;
    std::vector<float> v_;
};

class Indices {
public:
    Indices() {}
    explicit Indices(int rows) : v_((size_t)rows) {}
    int rows() const { return (int)v_.size(); }
    int operator()(int i) const { return v_[(size_t)i]; }
    int32_t *data() { return v_.data(); }
    const int32_t *data() const { return v_.data(); }
#ifdef PWPP_HAVE_EIGEN
    operator Eigen::VectorXi() const {
        Eigen::VectorXi m(rows());
        for (int i = 0; i < rows(); ++i) m(i) = v_[(size_t)i];
        return m;
    }
#endif
private:
    std::vector<int32_t> v_;
};

// reference patchworkpp.h:114-163
class PatchWorkpp {
public:
    PatchWorkpp(patchwork::Params _params, int device = 0) : params_(_params), h_(nullptr) {
        pwpp_params p;
        pwpp_params_default(&p);
        p.verbose = params_.verbose;
        p.enable_RNR = params_.enable_RNR;
        p.enable_RVPF = params_.enable_RVPF;
        p.enable_TGR = params_.enable_TGR;
        p.num_iter = params_.num_iter;
        p.num_lpr = params_.num_lpr;
        p.num_min_pts = params_.num_min_pts;
        p.num_zones = params_.num_zones;
        p.num_rings_of_interest = params_.num_rings_of_interest;
        p.RNR_ver_angle_thr = params_.RNR_ver_angle_thr;
        p.RNR_intensity_thr = params_.RNR_intensity_thr;
        p.sensor_height = params_.sensor_height;
        p.th_seeds = params_.th_seeds;
        p.th_dist = params_.th_dist;
        p.th_seeds_v = params_.th_seeds_v;
        p.th_dist_v = params_.th_dist_v;
        p.max_range = params_.max_range;
        p.min_range = params_.min_range;
        p.uprightness_thr = params_.uprightness_thr;
        p.adaptive_seed_selection_margin = params_.adaptive_seed_selection_margin;
        p.intensity_thr = params_.intensity_thr;
        // the reference constructor reads .at(0..3) of both vectors (:127-134) and indexes
        // elevation_thr / flatness_thr up to num_rings_of_interest (patchworkpp.cpp:244-245)
        for (int k = 0; k < 4; ++k) {
            p.num_sectors_each_zone[k] = params_.num_sectors_each_zone.at((size_t)k);
            p.num_rings_each_zone[k] = params_.num_rings_each_zone.at((size_t)k);
            p.elevation_thr[k] = (size_t)k < params_.elevation_thr.size() ? params_.elevation_thr[(size_t)k] : 0.0;
            p.flatness_thr[k] = (size_t)k < params_.flatness_thr.size() ? params_.flatness_thr[(size_t)k] : 0.0;
        }
        p.max_flatness_storage = params_.max_flatness_storage;
        p.max_elevation_storage = params_.max_elevation_storage;
        p.verbose = 0;  // the line below is this class' job
        check(pwpp_create(&p, device, &h_));
        std::cout << "PatchWorkpp::PatchWorkpp() - INITIALIZATION COMPLETE" << std::endl;  // reference :149
    }
    ~PatchWorkpp() { pwpp_destroy(h_); }
    PatchWorkpp(const PatchWorkpp &) = delete;
    PatchWorkpp &operator=(const PatchWorkpp &) = delete;

    // estimateGround, reference patchworkpp.cpp:151.  `data` is rows x cols float32, cols = 3 or 4.
    void estimateGround(const float *data, int rows, int cols, bool row_major = true) {
        if (params_.verbose) check(pwpp_set_profiling(h_, 1));
        check(pwpp_estimate_ground(h_, data, rows, cols, row_major ? PWPP_LAYOUT_ROW_MAJOR : PWPP_LAYOUT_COL_MAJOR));
        if (params_.verbose) report_times();
    }
#ifdef PWPP_HAVE_EIGEN
    void estimateGround(Eigen::MatrixXf cloud_in) {  // the reference's exact signature
        estimateGround(cloud_in.data(), (int)cloud_in.rows(), (int)cloud_in.cols(), false);
    }
#endif

    double getHeight() { return pwpp_get_height(h_); }      // reference :154
    double getTimeTaken() { return pwpp_get_time_us(h_); }  // reference :155 (microseconds)
    // extension: true = the points of a patch come out in the reference's own order (bins sorted by z,
    // patchworkpp.cpp:199) instead of the scatter order; same sets either way (pwpp.h, pwpp_set_output_order)
    void setReferenceOrder(bool on) { check(pwpp_set_output_order(h_, on ? PWPP_ORDER_REFERENCE : PWPP_ORDER_SCATTER)); }

#ifdef PWPP_HAVE_EIGEN
    // the reference's return types (fresh objects on every call, as the reference's toEigenCloud / toIndices, :8-26)
    Eigen::MatrixX3f getGround() { return to_eigen(xyz(true)); }               // reference :157
    Eigen::MatrixX3f getNonground() { return to_eigen(xyz(false)); }           // reference :158
    Eigen::VectorXi getGroundIndices() { return to_eigen(idx(true)); }         // reference :159
    Eigen::VectorXi getNongroundIndices() { return to_eigen(idx(false)); }     // reference :160
    Eigen::MatrixX3f getCenters() { return to_eigen(rows(true)); }             // reference :162
    Eigen::MatrixX3f getNormals() { return to_eigen(rows(false)); }            // reference :163
#else
    Cloud getGround() { return xyz(true); }          // reference :157
    Cloud getNonground() { return xyz(false); }      // reference :158
    Indices getGroundIndices() { return idx(true); }      // reference :159
    Indices getNongroundIndices() { return idx(false); }  // reference :160
    Cloud getCenters() { return rows(true); }        // reference :162
    Cloud getNormals() { return rows(false); }       // reference :163
#endif
    // the same results as plain containers, whatever the build (row-major (rows, 3) floats / int32)
    Cloud groundCloud() { return xyz(true); }
    Cloud nongroundCloud() { return xyz(false); }
    Indices groundIndexList() { return idx(true); }
    Indices nongroundIndexList() { return idx(false); }

    pwpp_handle *handle() { return h_; }  // escape hatch to the batch API of include/pwpp.h

private:
    patchwork::Params params_;
    pwpp_handle *h_;

#ifdef PWPP_HAVE_EIGEN
    static Eigen::MatrixX3f to_eigen(const Cloud &c) {
        Eigen::MatrixX3f m(c.rows(), 3);
        for (int i = 0; i < c.rows(); ++i)
            for (int j = 0; j < 3; ++j) m(i, j) = c(i, j);
        return m;
    }
    static Eigen::VectorXi to_eigen(const Indices &v) {
        Eigen::VectorXi m(v.rows());
        for (int i = 0; i < v.rows(); ++i) m(i) = v(i);
        return m;
    }
#endif
    static void check(int rc) {
        if (rc < 0) throw std::runtime_error(std::string("patchworkpp (HIP): ") + pwpp_last_error());
    }
    // the reference's verbose lines (patchworkpp.cpp:323-335), with GPU times: czm = binning kernels, sort = 0 (this
    // design has no sort), pca = the fit kernels, estimate = GLE / TGR + the index lists
    void report_times() {
        double ms[PWPP_NUM_KERNELS];
        int64_t launches[PWPP_NUM_KERNELS];
        check(pwpp_get_kernel_profile(h_, ms, launches));
        check(pwpp_reset_kernel_profile(h_));
        const double czm = ms[0] + ms[1] + ms[2], pca = ms[3] + ms[4] + ms[5] + ms[6] + ms[7] + ms[8], est = ms[9] + ms[10];
        std::cout << "Time taken : " << pwpp_get_time_us(h_) / 1e6 << "(sec) ~ " << czm / 1e3 << "(czm) + " << 0.0 << "(sort) + "
                  << pca / 1e3 << "(pca) + " << est / 1e3 << "(estimate)" << std::endl;
        std::cout << "\033[1;32m" << "PatchWorkpp::estimateGround() - Estimation is finished !" << "\033[0m" << std::endl;
    }
    void counts(int32_t &g, int32_t &n, int32_t &p) { check(pwpp_get_counts(h_, 0, &g, &n, &p)); }
    Cloud xyz(bool ground) {
        int32_t g, n, p;
        counts(g, n, p);
        Cloud c(ground ? g : n);
        check(ground ? pwpp_get_ground_xyz(h_, 0, c.data()) : pwpp_get_nonground_xyz(h_, 0, c.data()));
        return c;
    }
    Indices idx(bool ground) {
        int32_t g, n, p;
        counts(g, n, p);
        Indices v(ground ? g : n);
        check(ground ? pwpp_get_ground_indices(h_, 0, v.data()) : pwpp_get_nonground_indices(h_, 0, v.data()));
        return v;
    }
    Cloud rows(bool centers) {
        int32_t g, n, p;
        counts(g, n, p);
        Cloud c(p);
        check(centers ? pwpp_get_centers(h_, 0, c.data()) : pwpp_get_normals(h_, 0, c.data()));
        return c;
    }
};

}  // namespace patchwork

#endif
