/* oracle/oracle_ext.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Extras only the restatement (liboracle.so) exports: per-patch records for
 * fine-grained parity checks, state injection, and the arithmetic primitives. */
#ifndef PWPP_ORACLE_EXT_H
#define PWPP_ORACLE_EXT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    PWO_DEC_NOT_UPRIGHT = 1, /* reference patchworkpp.cpp:262-265 */
    PWO_DEC_FAR_GROUND = 2,  /* :266-269 */
    PWO_DEC_HEADING = 3,     /* :270-273 */
    PWO_DEC_GROUND = 4,      /* :274-277 */
    PWO_DEC_TGR_REJECT = 5,  /* :278-282 then :452-459 / :296-300 */
    PWO_DEC_TGR_REVERT = 6   /* :444-451 */
};

typedef struct pwo_patch_record {
    int32_t bin;            /* flattened zone->ring->sector index */
    int32_t concentric_idx; /* :174,309 */
    int32_t n_points;
    int32_t n_ground;       /* |regionwise_ground_| */
    int32_t n_nonground;    /* |regionwise_nonground_| */
    int32_t decision;
    float mean[3];          /* pc_mean_ after the last fit */
    float normal[3];
    float sv[3];            /* singular_values_ */
    float pad_;
    double d;               /* d_ */
} pwo_patch_record;

int pwo_ext_num_records(void *h);
void pwo_ext_get_records(void *h, pwo_patch_record *out);
void pwo_ext_set_state(void *h, double sensor_height, const double *elevation_thr4, const double *flatness_thr4);
void pwo_ext_jacobi(const float *cov9_rowmajor, float *u9_rowmajor, float *sv3);
/* the fixed-point contract: shift, z half-range, per-bin origins (ox/oy may be NULL) */
void pwo_ext_fxp_geometry(void *h, int *shift, double *zr, float *ox, float *oy);
long long pwo_ext_quantise(float v, double origin, int shift);
long long pwo_ext_quantise_z(float v, double z0, int shift, double zr);
double pwo_ext_z_origin(double lpr);
long pwo_ext_max_sweeps(int reset); /* largest Jacobi sweep count of one fit since the last reset (this thread) */

#ifdef __cplusplus
}
#endif
#endif
