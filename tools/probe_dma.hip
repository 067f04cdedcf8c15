// How fast can one workgroup per CU stream operand tiles global -> LDS with the GEMM's DMA pattern (no MFMA, no LDS reads)?
// Every workgroup (256 threads, 4 waves) moves a 512-row x 128-byte k-slab per step (16 buffer_load ... lds instructions of
// 8 rows x 128 B per wave), double buffered with the same vmcnt(0) + barrier per step as gemm_ntw_kernel.
//   mode 0: all workgroups read the same 512 rows                (L2 hits, maximal sharing)
//   mode 1: workgroup b reads A rows of tile (b % 64) and B rows of tile (b / 64): the GEMM's sharing pattern
//   mode 2: every workgroup reads its own rows                   (streams from HBM / MALL)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o tools/probe_dma && tools/probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int DEPTH, int AUX = 0>     // DEPTH k-steps in flight before the oldest is waited for (1 = the GEMM's scheme); AUX = cache-policy bits
__global__ void __launch_bounds__(256) k(const char* base, long long ld_bytes, int nsteps, int mode, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    long long rowA, rowB;
    if (mode == 0) { rowA = 0; rowB = 256; }
    else if (mode == 1) { rowA = (long long)(b % 64) * 256; rowB = 64 * 256 + (long long)(b / 64) * 256; }
    else { rowA = (long long)b * 512; rowB = rowA + 256; }
    unsigned off[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int q = wave * 8 + (d & 7), row = q * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        const long long g = (d < 8 ? rowA : rowB) + row;
        off[d] = (unsigned)(g * ld_bytes + c * 16);
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffffu, 0x00020000);
    const long long t0 = clock64();
    for (int t = 0; t < nsteps + DEPTH; ++t) {
        if (t < nsteps) {
            char* s = smem + (t & 1) * 65536;       // (deeper rings alias stages: nothing reads LDS here)
#pragma unroll
            for (int d = 0; d < 16; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(s + (d < 8 ? 0 : 32768) + (wave * 8 + (d & 7)) * 1024), 16,
                                                         off[d], t * 128, 0, AUX);
        }
        if (t >= DEPTH - 1) {
            if (DEPTH == 1 || t >= nsteps - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (tid == 0) cyc[b] = clock64() - t0;
}

// Same, with 32-KiB half stages: a step moves 512 rows x 64 B (8 instructions of 16 rows x 64 B per wave); DEPTH half steps in flight.
template <int DEPTH>
__global__ void __launch_bounds__(256) kh(const char* base, long long ld_bytes, int nsteps, int mode, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    long long rowA, rowB;
    if (mode == 0) { rowA = 0; rowB = 256; }
    else if (mode == 1) { rowA = (long long)(b % 64) * 256; rowB = 64 * 256 + (long long)(b / 64) * 256; }
    else { rowA = (long long)b * 512; rowB = rowA + 256; }
    unsigned off[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int q = wave * 4 + (d & 3), row = q * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        const long long g = (d < 4 ? rowA : rowB) + row;
        off[d] = (unsigned)(g * ld_bytes + c * 16);
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffffu, 0x00020000);
    const long long t0 = clock64();
    for (int t = 0; t < nsteps + DEPTH; ++t) {
        if (t < nsteps) {
            char* s = smem + (t & 3) * 32768;
#pragma unroll
            for (int d = 0; d < 8; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(s + (d < 4 ? 0 : 16384) + (wave * 4 + (d & 3)) * 1024), 16,
                                                         off[d], t * 64, 0, 0);
        }
        if (t >= DEPTH - 1) {
            if (DEPTH == 1 || t >= nsteps - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (tid == 0) cyc[b] = clock64() - t0;
}

// Same k-slabs through the VGPR path: 16 buffer_load_dwordx4 per wave and step into registers (no LDS at all).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
template <int DEPTH>
__global__ void __launch_bounds__(256) kv(const char* base, long long ld_bytes, int nsteps, int mode, long long* cyc, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    long long rowA, rowB;
    if (mode == 0) { rowA = 0; rowB = 256; }
    else if (mode == 1) { rowA = (long long)(b % 64) * 256; rowB = 64 * 256 + (long long)(b / 64) * 256; }
    else { rowA = (long long)b * 512; rowB = rowA + 256; }
    unsigned off[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int q = wave * 8 + (d & 7), row = q * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        const long long g = (d < 8 ? rowA : rowB) + row;
        off[d] = (unsigned)(g * ld_bytes + c * 16);
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffffu, 0x00020000);
    u32x4_t r[DEPTH][16];
    unsigned acc = 0;
    const long long t0 = clock64();
#pragma unroll 1
    for (int t = 0; t < nsteps; t += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int d = 0; d < 16; ++d) r[u][d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[d], (t + u) * 128, 0);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (u == 0 && DEPTH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int d = 0; d < 16; ++d) { asm volatile("" : "+v"(r[u][d])); acc ^= r[u][d][0]; }
            __builtin_amdgcn_s_barrier();
        }
    }
    if (tid == 0) cyc[b] = clock64() - t0;
    if (acc == 0x12345678u) sink[tid] = acc;
}

int main() {
    const long long ld = 10240;                          // K = 5120 bf16
    const long long rows = 256LL * 512 + 1024;
    char* buf; long long* cyc;
    hipMalloc(&buf, rows * ld); hipMemset(buf, 1, rows * ld); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nsteps = 80;
    for (int mode = 0; mode < 3; ++mode) {
        for (int depth = 1; depth <= 4; depth *= 2) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (depth == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 131072, 0, buf, ld, nsteps, mode, cyc);
                else if (depth == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 131072, 0, buf, ld, nsteps, mode, cyc);
                else hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 131072, 0, buf, ld, nsteps, mode, cyc);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += v; mean /= 256;
            const double us = ms * 100.0;               // per launch
            printf("mode %d depth %d: %.1f us per launch, %.0f ns per k-step, %.0f shader clocks per k-step (64 KiB) = %.1f B/clk/CU, %.2f TB/s chip\n",
                   mode, depth, us, us * 1e3 / nsteps, mean / nsteps, 65536.0 / (mean / nsteps), 256 * 65536.0 * nsteps / us / 1e6);
        }
    }
    hipFuncSetAttribute((const void*)kh<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)kh<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)kh<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)kh<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int mode = 0; mode < 3; ++mode) {
        for (int depth = 1; depth <= 4; ++depth) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (depth == 1) hipLaunchKernelGGL(kh<1>, dim3(256), dim3(256), 131072, 0, buf, ld, 2 * nsteps, mode, cyc);
                else if (depth == 2) hipLaunchKernelGGL(kh<2>, dim3(256), dim3(256), 131072, 0, buf, ld, 2 * nsteps, mode, cyc);
                else if (depth == 3) hipLaunchKernelGGL(kh<3>, dim3(256), dim3(256), 131072, 0, buf, ld, 2 * nsteps, mode, cyc);
                else hipLaunchKernelGGL(kh<4>, dim3(256), dim3(256), 131072, 0, buf, ld, 2 * nsteps, mode, cyc);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += v; mean /= 256;
            const double us = ms * 100.0;
            printf("half-row mode %d depth %d: %.1f us per launch, %.0f shader clocks per 64 KiB = %.1f B/clk/CU, %.2f TB/s chip\n",
                   mode, depth, us, mean / nsteps, 65536.0 / (mean / nsteps), 256 * 65536.0 * nsteps / us / 1e6);
        }
    }
    // cache-policy bits of the DMA loads (aux: 1 = sc0, 2 = nt, 16 = sc1), GEMM sharing pattern, two k-steps in flight
#define AUXRUN(AX) { hipFuncSetAttribute((const void*)k<2, AX>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);            \
        hipEventRecord(e0);                                                                                                      \
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<2, AX>), dim3(256), dim3(256), 131072, 0, buf, ld, nsteps, 1, cyc); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                                             \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                                              \
        std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);                                 \
        double mean = 0; for (auto v : h) mean += v; mean /= 256;                                                                \
        printf("aux %2d: %.1f us per launch, %.0f clocks per 64 KiB = %.1f B/clk/CU\n", AX, ms * 100.0, mean / nsteps, 65536.0 / (mean / nsteps)); }
    AUXRUN(0) AUXRUN(1) AUXRUN(2) AUXRUN(3) AUXRUN(16) AUXRUN(17) AUXRUN(18) AUXRUN(19)
    unsigned* sink; hipMalloc(&sink, 4096);
    for (int mode = 0; mode < 2; ++mode) {
        for (int depth = 1; depth <= 2; ++depth) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (depth == 1) hipLaunchKernelGGL(kv<1>, dim3(256), dim3(256), 0, 0, buf, ld, nsteps, mode, cyc, sink);
                else hipLaunchKernelGGL(kv<2>, dim3(256), dim3(256), 0, 0, buf, ld, nsteps, mode, cyc, sink);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += v; mean /= 256;
            const double us = ms * 100.0;
            printf("to-VGPR mode %d depth %d: %.1f us per launch, %.0f shader clocks per 64 KiB = %.1f B/clk/CU, %.2f TB/s chip\n",
                   mode, depth, us, mean / nsteps, 65536.0 / (mean / nsteps), 256 * 65536.0 * nsteps / us / 1e6);
        }
    }
    return 0;
}
