// copy_bw.hip -- what a device-to-device copy reaches on this GPU: the roof of a kernel that reads 16 bytes and writes 16 bytes per point
// (K1', k_czm_bin_scatter).  (a) float4 -> float4, fully coalesced; (b) float4 -> three planes (4 + 8 + 4 bytes per point: the layout K1'
// writes), coalesced; (c) as (b) with the points of a 1024-point tile dealt to 64 segments in runs of 16 (short contiguous runs, the
// shape of K1's stores).  2 GiB in, 2 GiB out per launch; best of 5.     hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_planes(const float4 *__restrict__ src, float *__restrict__ z, float2 *__restrict__ xy, int *__restrict__ idx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        z[i] = v.z;
        xy[i] = make_float2(v.x, v.y);
        idx[i] = (int)i;
    }
}
__global__ __launch_bounds__(256) void k_runs(const float4 *__restrict__ src, float *__restrict__ z, float2 *__restrict__ xy, int *__restrict__ idx, size_t n) {
    // tile of 1024 points -> 64 segments of the frame-sized region it belongs to (stride: 1/64 of a 128 K-point frame), runs of 16 points
    for (size_t t = blockIdx.x; t * 1024 < n; t += gridDim.x) {
        const size_t frame = t / 128, tile = t % 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned p = q * 256 + threadIdx.x;  // point of the tile
            const size_t i = t * 1024 + p;
            const unsigned seg = p / 16, r = p % 16;
            const size_t o = frame * 131072 + (size_t)seg * 2048 + tile * 16 + r;
            const float4 v = src[i];
            z[o] = v.z;
            xy[o] = make_float2(v.x, v.y);
            idx[o] = (int)i;
        }
    }
}
int main() {
    const size_t n = (size_t)1 << 27;  // 128 Mi points: 2 GiB in
    float4 *src, *dst;
    (void)hipMalloc(&src, n * 16);
    (void)hipMalloc(&dst, n * 16);
    (void)hipMemset(src, 1, n * 16);
    float *z = reinterpret_cast<float *>(dst);
    float2 *xy = reinterpret_cast<float2 *>(z + n);
    int *idx = reinterpret_cast<int *>(xy + n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char *names[3] = {"float4 -> float4", "float4 -> z | xy | idx planes", "float4 -> planes, runs of 16 into 64 segments"};
    for (int v = 0; v < 3; ++v)
        for (int grid : {2048, 8192, 65536}) {
            float best = 1e30f;
            for (int r = 0; r < 6; ++r) {
                (void)hipEventRecord(e0, 0);
                if (v == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, src, dst, n);
                else if (v == 1) hipLaunchKernelGGL(k_planes, dim3(grid), dim3(256), 0, 0, src, z, xy, idx, n);
                else hipLaunchKernelGGL(k_runs, dim3(grid), dim3(256), 0, 0, src, z, xy, idx, n);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r) best = ms < best ? ms : best;
            }
            printf("%-48s grid %6d: %7.3f ms  %6.2f TB/s (read + written)\n", names[v], grid, best, 2.0 * n * 16 / (best * 1e-3) / 1e12);
        }
    return 0;
}
