// Measures the shader clock the chip actually sustains under an all-CU MFMA load (power management lowers it well below
// the 2.4 GHz peak), so that MFMA-bound kernels can be judged against the attainable rate, not the datasheet one.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_clock.hip -o tools/probe_clock && tools/probe_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NACC, bool RANDOM = false>
__global__ void __launch_bounds__(256) mfma_loop(float* out, long long* clk, int iters) {
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (short)(0x3f80 + threadIdx.x); b[r] = (short)(0x3f00 + r); }
    if (RANDOM) {     // operands with random sign / mantissa / small exponent spread per lane (N(0,1)-like bit toggling)
        unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
        for (int r = 0; r < 8; ++r) {
            x = x * 1664525u + 1013904223u; a[r] = (short)(((x >> 16) & 0x807f) | (0x3e00 + ((x >> 9) & 0x180)));
            x = x * 1664525u + 1013904223u; b[r] = (short)(((x >> 16) & 0x807f) | (0x3e00 + ((x >> 9) & 0x180)));
        }
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// the 16x16x32 shape (same flops per instruction / 2, twice the instructions): does the board sustain a different rate with it?
__global__ void __launch_bounds__(256) mfma16_loop(float* out, int iters) {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < 8; ++r) {
        x = x * 1664525u + 1013904223u; a[r] = (short)(((x >> 16) & 0x807f) | (0x3e00 + ((x >> 9) & 0x180)));
        x = x * 1664525u + 1013904223u; b[r] = (short)(((x >> 16) & 0x807f) | (0x3e00 + ((x >> 9) & 0x180)));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int blocks = 256, iters = 20000;
    if (argc > 1) {   // sustained mode: tools/probe_clock <seconds> [random]   (sample rocm-smi from another shell meanwhile)
        const double secs = atof(argv[1]); const bool rnd = argc > 2;
        if (argc > 2 && !strcmp(argv[2], "m16")) {        // 16x16x32, random operands: 16 x 16384 flops x 2 per inner trip = the 32x32x16 loop's 8 x 32768 x 2
            float* o16; hipMalloc(&o16, blocks * 256 * 4);
            hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
            double tot = 0; int n16 = 0;
            while (tot < secs * 1e3) {
                hipEventRecord(a0);
                for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma16_loop, dim3(blocks), dim3(256), 0, 0, o16, iters);
                hipEventRecord(a1); hipEventSynchronize(a1);
                float ms; hipEventElapsedTime(&ms, a0, a1); tot += ms; n16 += 10;
                if (n16 % 200 == 0) printf("16x16x32 random operands, after %.1f s: %.1f TF sustained\n", tot * 1e-3,
                                           10.0 * blocks * 4 * iters * 16 * 16384.0 / (ms * 1e-3) * 1e-12);
            }
            return 0;
        }
        float* out; long long* clk; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double total_ms = 0; int n = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) {
                if (rnd) hipLaunchKernelGGL((mfma_loop<8, true>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
                else hipLaunchKernelGGL((mfma_loop<8, false>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; n += 10;
            if (n % 200 == 0) printf("%s operands, after %.1f s: %.1f TF sustained over the last 10 launches\n", rnd ? "random" : "constant", total_ms * 1e-3,
                                     10.0 * blocks * 4 * iters * 8 * 32768.0 / (ms * 1e-3) * 1e-12);
        }
        return 0;
    }
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    std::vector<long long> h(2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz\n", wall_khz);
    for (int rep = 0; rep < 3; ++rep) {
        for (int nb : {256, 32, 1}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(nb), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), clk, nb * 16, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0; for (int i = 0; i < nb; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
            cyc /= nb; wall /= nb;
            const double secs = wall / (wall_khz * 1e3);
            const double flops = (double)nb * 4 * iters * 8 * 32768.0;
            printf("blocks %3d: %.3f ms  shader clk %.3f GHz (s_memtime) | cycles/MFMA %.2f | %.1f TF (%.0f%% of 2500)\n", nb, ms,
                   cyc / secs * 1e-9, cyc / (iters * 8.0), flops / (ms * 1e-3) * 1e-12, flops / (ms * 1e-3) * 1e-12 / 25.0);
        }
    }
    return 0;
}
