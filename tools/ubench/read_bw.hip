// read_bw.hip -- what a read-only stream reaches on this GPU, as a function of the bytes a wave keeps in flight
// (16-byte loads per lane before the first use) and of the waves per SIMD.  The fit kernels are read streams of
// 4-12 B per point; this is their practical roof.   build: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int U, int OCC>
__global__ __launch_bounds__(256, OCC) void k_read(const float4 *__restrict__ src, size_t n4, float *out) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    float acc = 0.0f;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base + (U - 1) * 256 < n4; base += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[base + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int U, int OCC>
void run(const float4 *src, size_t n4, float *out, int wg_per_cu) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = 256 * wg_per_cu;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<U, OCC>), dim3(grid), dim3(256), 0, 0, src, n4, out);
    hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_read<U, OCC>), dim3(grid), dim3(256), 0, 0, src, n4, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("loads in flight %2d x 16 B, launch_bounds occupancy %d, %2d workgroups per CU: %.2f TB/s\n", U, OCC, wg_per_cu,
           (double)n4 * 16 * reps / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
    float4 *src;
    float *out;
    hipMalloc(&src, bytes);
    hipMalloc(&out, 4);
    hipMemset(src, 0, bytes);
    for (int wg : {2, 4, 8, 16, 32}) {
        run<1, 8>(src, n4, out, wg);
        run<2, 8>(src, n4, out, wg);
        run<4, 8>(src, n4, out, wg);
        run<8, 4>(src, n4, out, wg);
        run<16, 2>(src, n4, out, wg);
    }
    return 0;
}
