// Issue cost of the VALU instructions the fit kernels are made of, per wave-instruction, relative to v_fma_f32 (2 cycles on a SIMD-32).
// One workgroup of 256 threads per CU (one wave per SIMD) and four per CU (four waves per SIMD); eight independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/ubench/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP8(x) x x x x x x x x
#define KERNEL(name, decl, body, sink)                                                          \
    __global__ __launch_bounds__(256) void name(long long *out, int iters) {                    \
        decl;                                                                                   \
        for (int it = 0; it < iters; ++it) { REP8(body) }                                       \
        sink;                                                                                   \
    }
typedef int v4i __attribute__((ext_vector_type(4)));
KERNEL(k_fma_f32, float a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; float b = out[0] * 1e-30f + 1.0f,
       asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                    "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));,
       float s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345.f) out[1] = 1)
KERNEL(k_fma_f64, double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; double b = out[0] * 1e-30 + 1.0,
       asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                    "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));,
       double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345.) out[1] = 1)
KERNEL(k_add_f64, double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; double b = out[0] * 1e-30 + 1.0,
       asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                    "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));,
       double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345.) out[1] = 1)
KERNEL(k_mad_i64_i32, long long a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; int b = (int)out[0] + 3; int c = (int)out[0] + 5,
       asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n"
                    "v_mad_i64_i32 %4, vcc, %8, %9, %4\n v_mad_i64_i32 %5, vcc, %8, %9, %5\n v_mad_i64_i32 %6, vcc, %8, %9, %6\n v_mad_i64_i32 %7, vcc, %8, %9, %7\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc");,
       long long s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_mad_u32_u24, unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; unsigned b = (int)out[0] + 3; unsigned c = (int)out[0] + 5,
       asm volatile("v_mad_u32_u24 %0, %8, %9, %0\n v_mad_u32_u24 %1, %8, %9, %1\n v_mad_u32_u24 %2, %8, %9, %2\n v_mad_u32_u24 %3, %8, %9, %3\n"
                    "v_mad_u32_u24 %4, %8, %9, %4\n v_mad_u32_u24 %5, %8, %9, %5\n v_mad_u32_u24 %6, %8, %9, %6\n v_mad_u32_u24 %7, %8, %9, %7\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));,
       unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_mul_lo_u32, unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; unsigned b = (int)out[0] + 3,
       asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                    "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));,
       unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_lshl_add_u64, unsigned long long a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; unsigned long long b = out[0] + 3,
       asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                    "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));,
       unsigned long long s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_alignbit, unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; unsigned b = (int)out[0] + 3,
       asm volatile("v_alignbit_b32 %0, %0, %8, 9\n v_alignbit_b32 %1, %1, %8, 9\n v_alignbit_b32 %2, %2, %8, 9\n v_alignbit_b32 %3, %3, %8, 9\n"
                    "v_alignbit_b32 %4, %4, %8, 9\n v_alignbit_b32 %5, %5, %8, 9\n v_alignbit_b32 %6, %6, %8, 9\n v_alignbit_b32 %7, %7, %8, 9\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));,
       unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_mad_i32_i24, int a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; int b = (int)out[0] + 3; int c = (int)out[0] + 5,
       asm volatile("v_mad_i32_i24 %0, %8, %9, %0\n v_mad_i32_i24 %1, %8, %9, %1\n v_mad_i32_i24 %2, %8, %9, %2\n v_mad_i32_i24 %3, %8, %9, %3\n"
                    "v_mad_i32_i24 %4, %8, %9, %4\n v_mad_i32_i24 %5, %8, %9, %5\n v_mad_i32_i24 %6, %8, %9, %6\n v_mad_i32_i24 %7, %8, %9, %7\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));,
       int s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_dot2_i32_i16, int a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; int b = (int)out[0] + 3; int c = (int)out[0] + 5,
       asm volatile("v_dot2_i32_i16 %0, %8, %9, %0\n v_dot2_i32_i16 %1, %8, %9, %1\n v_dot2_i32_i16 %2, %8, %9, %2\n v_dot2_i32_i16 %3, %8, %9, %3\n"
                    "v_dot2_i32_i16 %4, %8, %9, %4\n v_dot2_i32_i16 %5, %8, %9, %5\n v_dot2_i32_i16 %6, %8, %9, %6\n v_dot2_i32_i16 %7, %8, %9, %7\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));,
       int s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_dot4_i32_i8, int a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; int b = (int)out[0] + 3; int c = (int)out[0] + 5,
       asm volatile("v_dot4_i32_i8 %0, %8, %9, %0\n v_dot4_i32_i8 %1, %8, %9, %1\n v_dot4_i32_i8 %2, %8, %9, %2\n v_dot4_i32_i8 %3, %8, %9, %3\n"
                    "v_dot4_i32_i8 %4, %8, %9, %4\n v_dot4_i32_i8 %5, %8, %9, %5\n v_dot4_i32_i8 %6, %8, %9, %6\n v_dot4_i32_i8 %7, %8, %9, %7\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));,
       int s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_perm_b32, unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i; unsigned b = (int)out[0] + 3; unsigned c = 0x05010400u + (unsigned)out[0],
       asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                    "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));,
       unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
KERNEL(k_cvt_f64_f32, double a[8]; float f[8]; for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i + (float)out[0],
       asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                    "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
                    : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
                    : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]));,
       double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345.) out[1] = 1)
KERNEL(k_mov_dpp, unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i,
       asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));,
       unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345) out[1] = 1)
// the int8 matrix instruction: four independent 16 x 16 accumulators, K = 64 per instruction (counts as 4 "instructions" per body: see main)
__global__ __launch_bounds__(256) void k_mfma_i8(long long *out, int iters) {
    v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v4i a = {(int)threadIdx.x, 2, 3, (int)out[0]}, b = {5, (int)threadIdx.x, 7, 8};
    for (int it = 0; it < iters; ++it) {
        REP8(acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[1], 0, 0, 0);
             acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[3], 0, 0, 0);)
    }
    int s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345) out[1] = 1;
}
struct Case { const char *name; void (*fn)(long long *, int); int per_body; };
int main() {
    long long *d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate * 1e-6;
    std::vector<Case> cases = {{"v_fma_f32", k_fma_f32, 64}, {"v_fma_f64", k_fma_f64, 64}, {"v_add_f64", k_add_f64, 64}, {"v_mad_i64_i32", k_mad_i64_i32, 64},
                               {"v_mad_u32_u24", k_mad_u32_u24, 64}, {"v_mad_i32_i24", k_mad_i32_i24, 64}, {"v_mul_lo_u32", k_mul_lo_u32, 64},
                               {"v_lshl_add_u64", k_lshl_add_u64, 64}, {"v_alignbit_b32", k_alignbit, 64}, {"v_dot2_i32_i16", k_dot2_i32_i16, 64},
                               {"v_dot4_i32_i8", k_dot4_i32_i8, 64}, {"v_perm_b32", k_perm_b32, 64}, {"v_cvt_f64_f32", k_cvt_f64_f32, 64},
                               {"v_mov_b32_dpp", k_mov_dpp, 64}, {"v_mfma_i32_16x16x64_i8", k_mfma_i8, 32}};
    const int iters = 4000;
    printf("%d CUs, %.2f GHz nominal; cycles per wave-instruction on one SIMD (nominal clock; compare with v_fma_f32)\n", cus, ghz);
    printf("%-26s %14s %14s\n", "instruction", "1 wave/SIMD", "4 waves/SIMD");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &c : cases) {
        double cyc[2];
        for (int w = 0; w < 2; ++w) {
            const int wgs = cus * (w ? 4 : 1);
            hipLaunchKernelGGL(c.fn, dim3(wgs), dim3(256), 0, 0, d, 10);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(c.fn, dim3(wgs), dim3(256), 0, 0, d, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double instr_per_simd = (double)iters * c.per_body * (w ? 4 : 1);
            cyc[w] = best * 1e-3 * ghz * 1e9 / instr_per_simd;
        }
        printf("%-26s %14.2f %14.2f\n", c.name, cyc[0], cyc[1]);
    }
    return 0;
}
