// tools/probe_dma.hip measured ONE workgroup per CU: 39-42 B/clk/CU global -> LDS with two 64-KiB k-steps in flight.  Does the CU move
// more with TWO workgroups (eight waves) issuing?  gemm_nt2_kernel (128 x 256 tiles, two workgroups per CU) needs 2 x 48 KiB per 2048
// matrix clocks = 48 B/clk/CU.  Every workgroup (256 threads) moves a 384-row x 128-byte k-slab per step (12 buffer_load ... lds
// instructions of 8 rows x 128 B per wave: A 128 rows + B 256 rows), DEPTH steps in flight, 80 KiB of LDS (slots alias: nothing reads).
//   mode 0: all workgroups read the same rows; mode 1: the GEMM's sharing pattern (A rows by b % 94, B rows by b / 94 % 5);
//   mode 2: every workgroup its own rows.     wgs = 256 (one per CU) or 512 (two per CU)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma2.hip -o tools/probe_dma2 && tools/probe_dma2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int DEPTH>
__global__ void __launch_bounds__(256, 2) k2(const char* base, long long ld_bytes, int nsteps, int mode, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    long long rowA, rowB;
    if (mode == 0) { rowA = 0; rowB = 128; }
    else if (mode == 1) { rowA = (long long)(b % 94) * 128; rowB = 94 * 128 + (long long)((b / 94) % 5) * 256; }
    else { rowA = (long long)b * 384; rowB = rowA + 128; }
    unsigned off[12];
#pragma unroll
    for (int d = 0; d < 12; ++d) {
        const int q = d < 4 ? wave * 4 + d : wave * 8 + (d - 4), row = q * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        const long long g = (d < 4 ? rowA : rowB) + row;
        off[d] = (unsigned)(g * ld_bytes + c * 16);
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffffu, 0x00020000);
    const long long t0 = clock64();
    for (int t = 0; t < nsteps + DEPTH; ++t) {
        if (t < nsteps) {
            char* s = smem + (t % 5) * 16384;
#pragma unroll
            for (int d = 0; d < 12; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(s + ((d < 4 ? wave * 4 + d : wave * 8 + d - 4) & 15) * 1024), 16, off[d], t * 128, 0, 0);
        }
        if (t >= DEPTH - 1) {
            if (DEPTH == 1 || t >= nsteps - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (tid == 0) cyc[b] = clock64() - t0;
}

int main() {
    const long long ld = 10240;
    const long long rows = 512LL * 384 + 4096;
    char* buf; long long* cyc;
    hipMalloc(&buf, rows * ld); hipMemset(buf, 1, rows * ld); hipMalloc(&cyc, 512 * 8);
    hipFuncSetAttribute((const void*)k2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k2<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nsteps = 80;
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs = 256; wgs <= 512; wgs *= 2)
            for (int depth = 1; depth <= 3; ++depth) {
                hipEventRecord(e0);
                for (int i = 0; i < 10; ++i) {
                    if (depth == 1) hipLaunchKernelGGL(k2<1>, dim3(wgs), dim3(256), 81920, 0, buf, ld, nsteps, mode, cyc);
                    else if (depth == 2) hipLaunchKernelGGL(k2<2>, dim3(wgs), dim3(256), 81920, 0, buf, ld, nsteps, mode, cyc);
                    else hipLaunchKernelGGL(k2<3>, dim3(wgs), dim3(256), 81920, 0, buf, ld, nsteps, mode, cyc);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<long long> h(wgs); hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
                double mean = 0; for (auto v : h) mean += v; mean /= wgs;
                const double per_cu = (wgs / 256.0) * 48.0 * 1024 * nsteps / mean;
                printf("mode %d  %d workgroups  depth %d: %7.1f us per launch, %6.0f clocks per 48-KiB step per workgroup = %5.1f B/clk/CU, %6.2f TB/s chip\n",
                       mode, wgs, depth, ms / 10 * 1e3, mean / nsteps, per_cu, (double)wgs * 48 * 1024 * nsteps / (ms / 10 * 1e-3) / 1e12);
            }
    return 0;
}
