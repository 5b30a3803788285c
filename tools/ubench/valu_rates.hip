// valu_rates.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU operations the fit
// kernels lean on, measured on the device: N dependent-free copies of one instruction per loop trip,
// 4 waves per SIMD, s_memtime around the loop of one wave.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    int i0 = (int)a0, i1 = (int)a1, i2 = (int)a2, i3 = (int)a3, i4 = (int)a4, i5 = (int)a5, i6 = (int)a6, i7 = (int)a7;
    long long l0 = i0, l1 = i1, l2 = i2, l3 = i3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {  // v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                         "v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
        } else if (OP == 1) {  // v_cvt_f64_i32
            asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n"
                         "v_cvt_f64_i32 %4, %12\n v_cvt_f64_i32 %5, %13\n v_cvt_f64_i32 %6, %14\n v_cvt_f64_i32 %7, %15\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                         : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));
        } else if (OP == 2) {  // v_cvt_f64_f32
            asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                         "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (OP == 3) {  // v_add_f64
            asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n"
                         "v_add_f64 %4, %4, %4\n v_add_f64 %5, %5, %5\n v_add_f64 %6, %6, %6\n v_add_f64 %7, %7, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
        } else if (OP == 4) {  // v_cmp_lt_f64 (to vcc)
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %4\n"
                         "v_cmp_lt_f64 vcc, %4, %5\n v_cmp_lt_f64 vcc, %5, %6\n v_cmp_lt_f64 vcc, %6, %7\n v_cmp_lt_f64 vcc, %7, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)::"vcc");
        } else if (OP == 5) {  // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 6) {  // v_cvt_i32_f32
            asm volatile("v_cvt_i32_f32 %0, %8\n v_cvt_i32_f32 %1, %9\n v_cvt_i32_f32 %2, %10\n v_cvt_i32_f32 %3, %11\n"
                         "v_cvt_i32_f32 %4, %12\n v_cvt_i32_f32 %5, %13\n v_cvt_i32_f32 %6, %14\n v_cvt_i32_f32 %7, %15\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (OP == 7) {  // v_rndne_f32
            asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n"
                         "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 8) {  // v_mad_i64_i32
            asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %5, %6, %1\n v_mad_i64_i32 %2, vcc, %6, %7, %2\n v_mad_i64_i32 %3, vcc, %7, %4, %3\n"
                         "v_mad_i64_i32 %0, vcc, %4, %6, %0\n v_mad_i64_i32 %1, vcc, %5, %7, %1\n v_mad_i64_i32 %2, vcc, %6, %4, %2\n v_mad_i64_i32 %3, vcc, %7, %5, %3\n"
                         : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");
        } else if (OP == 9) {  // v_med3_i32
            asm volatile("v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %1, %1, %2, %3\n v_med3_i32 %2, %2, %3, %4\n v_med3_i32 %3, %3, %4, %5\n"
                         "v_med3_i32 %4, %4, %5, %6\n v_med3_i32 %5, %5, %6, %7\n v_med3_i32 %6, %6, %7, %0\n v_med3_i32 %7, %7, %0, %1\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (OP == 10) {  // v_mov_b32 dpp row_shr:1
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (OP == 11) {  // v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %4\n"
                         "v_mul_f64 %4, %4, %5\n v_mul_f64 %5, %5, %6\n v_mul_f64 %6, %6, %7\n v_mul_f64 %7, %7, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
        } else if (OP == 12) {  // v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
        } else if (OP == 13) {  // v_cmp_lt_f32 + s_and_saveexec + s_or exec (an if without a skip branch)
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n"
                         "v_cmp_lt_f32 vcc, %2, %3\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %2, %2, %3\n s_or_b64 exec, exec, s[20:21]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc", "s20", "s21");
        } else if (OP == 14) {  // v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 15) {  // v_rndne_f64
            asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n"
                         "v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
        } else if (OP == 16) {  // v_cvt_i32_f64
            asm volatile("v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %9\n v_cvt_i32_f64 %2, %10\n v_cvt_i32_f64 %3, %11\n"
                         "v_cvt_i32_f64 %4, %12\n v_cvt_i32_f64 %5, %13\n v_cvt_i32_f64 %6, %14\n v_cvt_i32_f64 %7, %15\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
        } else if (OP == 17) {  // v_max_f64
            asm volatile("v_max_f64 %0, %0, %1\n v_max_f64 %1, %1, %2\n v_max_f64 %2, %2, %3\n v_max_f64 %3, %3, %4\n"
                         "v_max_f64 %4, %4, %5\n v_max_f64 %5, %5, %6\n v_max_f64 %6, %6, %7\n v_max_f64 %7, %7, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
        } else if (OP == 18) {  // v_lshl_add_u64 (64-bit integer add)
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                         "v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %3\n v_lshl_add_u64 %2, %2, 0, %0\n v_lshl_add_u64 %3, %3, 0, %1\n"
                         : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
        } else if (OP == 19) {  // v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n"
                         "v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + l0 + l1 + l2 + l3;
    if (s == 1.2345) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_trip, unsigned long long *d_out) {
    const int iters = 20000;
    // 256 CUs x 4 blocks of 256 threads = 4 waves per SIMD everywhere
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d_out, 100, 1.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d_out, iters, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = 4 waves * iters * per_trip ; cycles at 2.4 GHz nominal
    const double inst_per_simd = 4.0 * iters * per_trip;
    printf("%-28s %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / inst_per_simd,
           ms * 1e6 / inst_per_simd * 2.4);
}

int main() {
    unsigned long long *d_out;
    hipMalloc(&d_out, 64);
    run<5>("v_fma_f32", 8, d_out);
    run<0>("v_fma_f64", 8, d_out);
    run<11>("v_mul_f64", 8, d_out);
    run<3>("v_add_f64", 8, d_out);
    run<4>("v_cmp_lt_f64", 8, d_out);
    run<1>("v_cvt_f64_i32", 8, d_out);
    run<2>("v_cvt_f64_f32", 8, d_out);
    run<6>("v_cvt_i32_f32", 8, d_out);
    run<7>("v_rndne_f32", 8, d_out);
    run<9>("v_med3_i32", 8, d_out);
    run<8>("v_mad_i64_i32", 8, d_out);
    run<10>("v_mov_b32 dpp", 8, d_out);
    run<12>("v_pk_mul_f32", 8, d_out);
    run<14>("v_rcp_f32", 8, d_out);
    run<15>("v_rndne_f64", 8, d_out);
    run<16>("v_cvt_i32_f64", 8, d_out);
    run<17>("v_max_f64", 8, d_out);
    run<18>("v_lshl_add_u64", 8, d_out);
    run<19>("v_mul_lo_u32", 8, d_out);
    run<13>("cmp+saveexec+add+or (x2)", 2, d_out);
    return 0;
}
