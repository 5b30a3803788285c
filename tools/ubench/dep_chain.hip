// dep_chain.hip -- latency of a DEPENDENT chain of f64 adds in a single wave (what bounds the history
// statistics of k_gle_tgr: the reference's sums are sequential), with and without LDS reads feeding it.
//   hipcc --offload-arch=gfx950 -O3 dep_chain.hip -o dep_chain && ./dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int n, double seed) {
    __shared__ __attribute__((aligned(16))) double tile[8 * 498];
    for (int i = threadIdx.x; i < 8 * 498; i += 256) tile[i] = seed * i;
    __syncthreads();
    if (threadIdx.x >= 8) return;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    double acc = seed;
    if (MODE == 0) {  // pure register chain
        const double a = seed * 3;
        for (int i = 0; i < n; i += 16) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(a));
        }
    } else if (MODE == 1) {  // batches of 8 ds_read_b128, then 16 adds
        const double *row = tile + threadIdx.x * 498;
        for (int i = 0; i < n; i += 16) {
            const int o = i % 496 - (i % 496) % 16;
            double2 t[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) t[k2] = *reinterpret_cast<const double2 *>(row + o + 2 * k2);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                acc += t[k2].x;
                acc += t[k2].y;
            }
        }
    } else {  // the same, next batch's reads issued before this batch's adds
        const double *row = tile + threadIdx.x * 498;
        double2 cur[8], nxt[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) cur[k2] = *reinterpret_cast<const double2 *>(row + 2 * k2);
        for (int i = 0; i < n; i += 16) {
            const int o = (i + 16) % 496 - ((i + 16) % 496) % 16;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) nxt[k2] = *reinterpret_cast<const double2 *>(row + o + 2 * k2);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(cur[k2].x));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(cur[k2].y));
            }
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) cur[k2] = nxt[k2];
        }
    }
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) {  // shader clock ticks per 100 MHz tick over the loop
        out[8] = (double)(__builtin_readcyclecounter() - c0);
        out[9] = (double)(wall_clock64() - r0);
    }
}

template <int MODE>
static void run(const char *what) {
    double *out;
    hipMalloc(&out, 128);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int n = 1 << 20;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, 1024, 1e-9);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, n, 1e-9);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double host[16];
    hipMemcpy(host, out, 128, hipMemcpyDeviceToHost);
    printf("%-60s %.2f ns per add; s_memtime / s_memrealtime = %.2f (x 100 MHz)\n", what, 1e6 * ms / n, host[8] / host[9]);
    hipFree(out);
}

int main() {
    run<0>("dependent v_add_f64, registers only");
    run<1>("8 x ds_read_b128 then 16 dependent adds");
    run<2>("the same, software-pipelined reads");
    return 0;
}
