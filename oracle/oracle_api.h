/* oracle/oracle_api.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * One C interface, exported under the same names by two different CPU builds:
 *
 *   oracle/liboracle.so          the restatement (oracle/pwpp_oracle.cpp)
 *   oracle/_ref/libpwpp_ref*.so  the reference's own patchworkpp.cpp, compiled
 *                                unmodified from /root/reference against
 *                                oracle/eigen_shim (oracle/ref_capi.cpp wraps it)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * these libraries.  The product (libpwpp_hip.so) never does.
 */
#ifndef PWPP_ORACLE_API_H
#define PWPP_ORACLE_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of patchwork::Params (reference patchworkpp.h:42-112), POD form. */
typedef struct pwo_params {
    int32_t verbose, enable_RNR, enable_RVPF, enable_TGR;
    int32_t num_iter, num_lpr, num_min_pts, num_zones, num_rings_of_interest;
    double RNR_ver_angle_thr, RNR_intensity_thr;
    double sensor_height, th_seeds, th_dist, th_seeds_v, th_dist_v;
    double max_range, min_range, uprightness_thr, adaptive_seed_selection_margin;
    int32_t num_sectors_each_zone[4];
    int32_t num_rings_each_zone[4];
    int32_t max_flatness_storage, max_elevation_storage;
    double elevation_thr[4];
    double flatness_thr[4];
} pwo_params;

/* Arithmetic of the plane fit's sums (reference patchworkpp.cpp:56-60), the one place the
 * reference defers to Eigen:
 *   EIGEN_F32    float accumulators, rows in storage order (plainest reading of Eigen)
 *   FXP          the product's order-independent fixed-point contract: exact integer
 *                moments on a 2^-30 m grid (contract v4); the restatement only (it needs the bin and its
 *                first LPR, which the shim cannot see)
 *   EXACT_F64    reference-neutral arbiter: double accumulation of the unquantised floats,
 *                one rounding to float per output
 *   F32_PACKET4  float again, four partial sums (a 4-wide SIMD reduction order)
 *   FXP21        rounds 3-5's contract v3: FXP on a 2^-21 m grid (restatement only; a witness, not the product) */
enum { PWO_ARITH_EIGEN_F32 = 0, PWO_ARITH_FXP = 1, PWO_ARITH_EXACT_F64 = 2, PWO_ARITH_F32_PACKET4 = 3, PWO_ARITH_FXP21 = 4 };

void pwo_default_params(pwo_params *p);                 /* patchworkpp.h:79-111 */
int pwo_arith_supported(int arith);                     /* 1 if this build can do it */
void *pwo_create(const pwo_params *p, int arith);       /* NULL on bad arith */
void pwo_destroy(void *h);

/* points: row-major n x cols float32 (cols = 3 or 4), as np.fromfile(..).reshape(-1,4) */
int pwo_estimate_ground(void *h, const float *pts, int n, int cols);

int pwo_num_ground(void *h);
int pwo_num_nonground(void *h);
int pwo_num_patches(void *h);
void pwo_get_ground_indices(void *h, int32_t *out);     /* reference order */
void pwo_get_nonground_indices(void *h, int32_t *out);
void pwo_get_ground(void *h, float *out);               /* row-major (n,3) */
void pwo_get_nonground(void *h, float *out);
void pwo_get_centers(void *h, float *out);
void pwo_get_normals(void *h, float *out);
double pwo_get_height(void *h);
double pwo_get_time_taken(void *h);

/* adaptive state after the last frame (patchworkpp.cpp:338-375) */
void pwo_get_thresholds(void *h, double *sensor_height, double *elevation_thr4, double *flatness_thr4);
int pwo_get_history_len(void *h, int which /*0 elevation, 1 flatness*/, int ring);
void pwo_get_history(void *h, int which, int ring, double *out);

/* work counters since creation of the process: plane fits, Jacobi sweeps */
void pwo_get_counters(long *plane_fits, long *jacobi_sweeps);

/* CPU baseline: `total` frames (frame i = distinct[i % num_distinct]), each through a
 * fresh-state object, spread over `threads` host threads.  Returns wall seconds of the
 * parallel region; *sum_call_seconds gets the sum of the estimateGround() call times. */
double pwo_bench(const pwo_params *p, int arith, const float *const *frames, const int *n_points,
                 int cols, int num_distinct, int total, int threads, double *sum_call_seconds);

#ifdef __cplusplus
}
#endif
#endif
