// Fill-rate probe: how fast can all CUs pull an L2/MALL-resident buffer into LDS (global_load_lds, 16 B/lane) or into
// registers (global_load_dwordx4)?  Sets the ceiling for LDS-staged GEMM tiles (bytes/flop x PFLOP/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>   // 0: global_load_lds, 1: global_load to registers
__global__ void __launch_bounds__(512) fill_kernel(const uint4* __restrict__ src, size_t n16, int iters, int inflight, uint4* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t idx = ((size_t)blockIdx.x * nw + wave) * 64 * 8 + lane;
    const size_t stride = (size_t)gridDim.x * nw * 64 * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t i = (idx + (size_t)j * 64) % n16;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                                 (__attribute__((address_space(3))) void*)(smem + (wave * 8 + j) * 1024), 16, 0, 0);
            } else {
                const uint4 v = src[i];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        idx += stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 && acc.x == 0x12345678) sink[0] = acc;
}

int main() {
    const size_t bytes_list[] = {16u << 20, 64u << 20, 512u << 20};
    for (size_t bytes : bytes_list) {
        uint4* src; uint4* sink; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(src, 1, bytes));
        const size_t n16 = bytes / 16;
        for (int mode = 0; mode < 2; ++mode) for (int blocks_per_cu : {1, 2}) for (int threads : {256, 512}) {
            const int grid = 256 * blocks_per_cu, iters = 400;
            hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(grid), dim3(threads), 64 * 1024, 0, src, n16, iters, 0, sink);
                else hipLaunchKernelGGL(fill_kernel<1>, dim3(grid), dim3(threads), 0, 0, src, n16, iters, 0, sink);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0)); launch(); CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
            float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
            const double moved = (double)grid * (threads / 64) * iters * 8 * 1024.0;
            printf("PROBE fill buf=%4zuMB mode=%s blocks/CU=%d threads=%d : %.2f TB/s\n", bytes >> 20, mode == 0 ? "lds-dma" : "regs   ",
                   blocks_per_cu, threads, moved / ms / 1e9);
        }
        CK(hipFree(src)); CK(hipFree(sink));
    }
    return 0;
}
