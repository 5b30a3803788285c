// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// kernels of this library use: a 1 GiB buffer streamed once with 4, 8, 12 (4 + 8, two planes) and 16 bytes per
// lane and load instruction, and a 256 MiB buffer written with 4 and 8 bytes per lane.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and compare the counter with the known byte count (tools/profile_r02.sh does).
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T>
__global__ __launch_bounds__(256) void k_read(const T *p, size_t n, T *sink) {
    T acc = p[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = p[i];
        const unsigned char *a = reinterpret_cast<const unsigned char *>(&v);
        unsigned char *b = reinterpret_cast<unsigned char *>(&acc);
        for (unsigned k = 0; k < sizeof(T); ++k) b[k] ^= a[k];
    }
    if (reinterpret_cast<unsigned char *>(&acc)[0] == 0x5a) sink[blockIdx.x & 1] = acc;  // (data dependent: the loads stay)
}
__global__ __launch_bounds__(256) void k_read_4_8(const float *z, const float2 *xy, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2 v = xy[i];
        acc += z[i] + v.x + v.y;
    }
    if (acc == 1.2345f) sink[blockIdx.x & 1] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void k_write(T *p, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    const dim3 g(256 * 16), b(256);
    hipLaunchKernelGGL(k_read<float>, g, b, 0, 0, (const float *)buf, bytes / 4, (float *)sink);
    hipLaunchKernelGGL(k_read<float2>, g, b, 0, 0, (const float2 *)buf, bytes / 8, (float2 *)sink);
    hipLaunchKernelGGL(k_read<float4>, g, b, 0, 0, (const float4 *)buf, bytes / 16, (float4 *)sink);
    // two planes: z = first third, xy = the rest (12 bytes per point, 1 GiB in total)
    const size_t npts = bytes / 12;
    hipLaunchKernelGGL(k_read_4_8, g, b, 0, 0, (const float *)buf, (const float2 *)((const char *)buf + npts * 4), npts, (float *)sink);
    hipLaunchKernelGGL(k_write<float>, g, b, 0, 0, (float *)buf, bytes / 16, 1.0f);        // 256 MiB
    hipLaunchKernelGGL(k_write<float2>, g, b, 0, 0, (float2 *)buf, bytes / 32, make_float2(1.f, 2.f));  // 256 MiB
    (void)hipDeviceSynchronize();
    printf("bytes read per k_read launch: %zu; per k_read_4_8: %zu; written per k_write launch: %zu\n", bytes, npts * 12, bytes / 4);
    return 0;
}
