// block_sort.hip -- what a workgroup sort of a list of k_order_sublists costs: the register-blocked bitonic network of
// pwpp_kernels.hip against rocprim::block_radix_sort over all 56 key bits and over the bits that differ in a list.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o block_sort block_sort.hip && ./block_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ int ord_at(int i) { return i + (i >> 4); }
__device__ __forceinline__ void ord_ce(unsigned long long &a, unsigned long long &b, bool up) {
    const bool sw = (a > b) == up;
    const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}
template <int BLOCK, int E>
__device__ __forceinline__ void ord_sort_blocked(unsigned long long *s_key) {
    constexpr int np = BLOCK * E;
    const int t = threadIdx.x, base = t * E;
    unsigned long long r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = s_key[ord_at(base + e)];
#pragma unroll
    for (int k = 2; k <= E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & j) == 0) ord_ce(r[e], r[e | j], ((base + e) & k) == 0);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s_key[ord_at(base + e)] = r[e];
    __syncthreads();
    for (int k = 2 * E; k <= np; k <<= 1) {
        for (int j = k >> 1; j >= E; j >>= 1) {
#pragma unroll
            for (int q = 0; q < E / 2; ++q) {
                const int p = t + q * BLOCK;
                const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
                unsigned long long a = s_key[ord_at(lo)], b = s_key[ord_at(hi)];
                ord_ce(a, b, (lo & k) == 0);
                s_key[ord_at(lo)] = a;
                s_key[ord_at(hi)] = b;
            }
            __syncthreads();
        }
        const bool up = (base & k) == 0;
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = s_key[ord_at(base + e)];
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & j) == 0) ord_ce(r[e], r[e | j], up);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) s_key[ord_at(base + e)] = r[e];
        __syncthreads();
    }
}

template <int E>
__global__ __launch_bounds__(256) void k_bitonic(const unsigned long long *in, unsigned long long *out) {
    __shared__ unsigned long long s_key[256 * E + 16 * E];
    const size_t base = (size_t)blockIdx.x * 256 * E;
    for (int i = threadIdx.x; i < 256 * E; i += 256) s_key[ord_at(i)] = in[base + i];
    __syncthreads();
    ord_sort_blocked<256, E>(s_key);
    for (int i = threadIdx.x; i < 256 * E; i += 256) out[base + i] = s_key[ord_at(i)];
}

template <int E, int BITS_PER_PASS>
__global__ __launch_bounds__(256) void k_radix(const unsigned long long *in, unsigned long long *out, int begin_bit, int end_bit) {
    using sorter = rocprim::block_radix_sort<unsigned long long, 256, E, rocprim::empty_type, 1, 1, BITS_PER_PASS>;
    __shared__ typename sorter::storage_type storage;
    const size_t base = (size_t)blockIdx.x * 256 * E;
    unsigned long long k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) k[e] = in[base + threadIdx.x * E + e];
    sorter().sort(k, storage, begin_bit, end_bit);
#pragma unroll
    for (int e = 0; e < E; ++e) out[base + threadIdx.x * E + e] = k[e];
}

static unsigned zkey(float z) {
    unsigned b;
    memcpy(&b, &z, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <class F>
static float time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

template <int E>
static void run(int blocks) {
    const size_t n = (size_t)blocks * 256 * E;
    std::vector<unsigned long long> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) {
        const float z = -1.9f + 0.3f * (float)rand() / (float)RAND_MAX;  // the heights of one patch: one exponent, 23 bits differ
        h[i] = ((unsigned long long)zkey(z) << 24) | (unsigned long long)(rand() & 0xffffff);
    }
    unsigned long long *d_in, *d_out;
    hipMalloc(&d_in, n * 8);
    hipMalloc(&d_out, n * 8);
    hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
    const float tb = time_ms([&] { hipLaunchKernelGGL(k_bitonic<E>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); });
    std::vector<unsigned long long> o1(n), o2(n);
    hipMemcpy(o1.data(), d_out, n * 8, hipMemcpyDeviceToHost);
    const float tr56 = time_ms([&] { hipLaunchKernelGGL((k_radix<E, 0>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 0, 56); });
    hipMemcpy(o2.data(), d_out, n * 8, hipMemcpyDeviceToHost);
    const bool same = memcmp(o1.data(), o2.data(), n * 8) == 0;
    const float tr47 = time_ms([&] { hipLaunchKernelGGL((k_radix<E, 0>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 0, 47); });
    const float tr23 = time_ms([&] { hipLaunchKernelGGL((k_radix<E, 0>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 24, 47); });
    const float tr23_8 = time_ms([&] { hipLaunchKernelGGL((k_radix<E, 8>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 24, 47); });
    const float tr23_6 = time_ms([&] { hipLaunchKernelGGL((k_radix<E, 6>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 24, 47); });
    printf("%5d keys per list, %6d lists: bitonic %.3f ms | radix 56 bits %.3f (%s) | 47 bits %.3f | bits 24..46: %.3f (default digit) %.3f (8-bit) %.3f (6-bit)\n",
           256 * E, blocks, tb, tr56, same ? "same order" : "DIFFERENT", tr47, tr23, tr23_8, tr23_6);
    hipFree(d_in);
    hipFree(d_out);
}

int main() {
    run<16>(16384);
    run<8>(32768);
    run<4>(65536);
    run<2>(131072);
    return 0;
}
