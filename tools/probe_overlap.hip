// Do MFMA and plain VALU instructions of DIFFERENT waves on one SIMD overlap?  512-thread workgroups (2 waves per SIMD):
// waves 0-3 run an MFMA loop, waves 4-7 a v_fma / v_exp loop.  Compare both-together against each alone.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_overlap.hip -o tools/probe_overlap && tools/probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int USE_EXP>
__global__ void __launch_bounds__(512) k(float* out, int mfma_iters, int valu_iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        f32x16_t acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8_t a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3f00 + e); }
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    } else {
        float v[16];                                    // 16 independent chains: throughput-, not latency-bound
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (USE_EXP) v[i] = __builtin_amdgcn_exp2f(v[i]);
                else v[i] = fmaf(v[i], 1.0001f, 0.5f);
            }
        }
        for (int i = 0; i < 16; ++i) r += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int mi, int vi, int ex) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (ex) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, mi, vi);
            else hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, mi, vi);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        return best;
    };
    const int MI = 20000;                       // 80000 MFMAs x 32 cycles = 2.56 M cycles per MFMA wave
    for (int ex = 0; ex < 2; ++ex) {
        const int VI = ex ? 10000 : 40000;      // 16 VALU per iteration
        const float m = run(MI, 0, ex), v = run(0, VI, ex), both = run(MI, VI, ex);
        printf("%s: MFMA alone %.3f ms, VALU alone %.3f ms (%.1f cycles/instr @2 GHz), together %.3f ms  (sum %.3f, max %.3f)\n",
               ex ? "v_exp_f32" : "v_fma_f32", m, v, v * 2e6 / (VI * 16.0), both, m + v, m > v ? m : v);
    }
    return 0;
}
