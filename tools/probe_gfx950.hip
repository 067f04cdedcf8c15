// Hardware-semantics probe for gfx950 (run once on the GPU box; output kept in profiles/).
// Pins the assumptions the hand-written kernels rely on:
//   1. ds_read_b64_tr_b16 lane/element mapping
//   2. MFMA 32x32x16 bf16 operand/result layout (and that the k-slot order only has to be
//      consistent between A and B)
//   3. global_load_lds (16 B) lane-linear destination
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_tr(const uint16_t* in, uint16_t* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    // canonical addressing: 16-lane group G reads a [4 rows][16 cols] block; input lane u -> row u>>2, cols 4*(u&3)
    int l = threadIdx.x, G = l >> 4, u = l & 15;
    uint32_t addr = (uint32_t)(uintptr_t)lds + (G * 4 + (u >> 2)) * stride_bytes + (u & 3) * 8;
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}

__device__ inline uint16_t f2bf(float f) { uint32_t u = __float_as_uint(f); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

__global__ void k_mfma(const uint16_t* A, const uint16_t* B, float* D) {
    // A [32][16], B [32 cols][16] (both k-contiguous rows), D [32][32] = A * B^T
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e]; b[e] = B[(l & 31) * 16 + 8 * (l >> 5) + e]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = c[r];
    }
}

__global__ void k_glds(const uint32_t* in, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256];
    int l = threadIdx.x;
    // each lane fetches a DIFFERENT 16-B source chunk (reversed order); destination must be base + lane*16
    const uint32_t* src = in + (63 - l) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = lds[l * 4 + j];
}

int main() {
    // ---- 1. tr_b16
    for (int stride : {32, 128}) {
        std::vector<uint16_t> h(4096);
        for (int i = 0; i < 4096; ++i) h[i] = (uint16_t)i;
        uint16_t *din, *dout; CK(hipMalloc(&din, 8192)); CK(hipMalloc(&dout, 512));
        CK(hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice));
        k_tr<<<1, 64>>>(din, dout, stride);
        std::vector<uint16_t> o(256); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            int G = l >> 4, i = l & 15;
            int expect = ((G * 4 + j) * stride) / 2 + i;       // row (4G+j), col i
            if (o[l * 4 + j] != expect) ok = 0;
        }
        printf("PROBE tr_b16 stride=%d assumed_mapping_ok=%d\n", stride, ok);
        if (!ok) for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    }
    // ---- 2. MFMA 32x32x16
    {
        std::vector<uint16_t> A(512), B(512); std::vector<float> Af(512), Bf(512);
        srand(1);
        for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 7 - 3); Bf[i] = (float)(rand() % 5 - 2);
            uint32_t ua, ub; memcpy(&ua, &Af[i], 4); memcpy(&ub, &Bf[i], 4); A[i] = ua >> 16; B[i] = ub >> 16; }
        uint16_t *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dD);
        std::vector<float> D(1024); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float s = 0; for (int k = 0; k < 16; ++k) s += Af[i * 16 + k] * Bf[j * 16 + k];
            if (s != D[i * 32 + j]) ok = 0;
        }
        printf("PROBE mfma_32x32x16 layout_ok=%d\n", ok);
    }
    // ---- 3. global_load_lds
    {
        std::vector<uint32_t> h(256); for (int i = 0; i < 256; ++i) h[i] = i;
        uint32_t *din, *dout; CK(hipMalloc(&din, 1024)); CK(hipMalloc(&dout, 1024));
        CK(hipMemcpy(din, h.data(), 1024, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(din, dout);
        std::vector<uint32_t> o(256); CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (o[l * 4 + j] != (uint32_t)((63 - l) * 4 + j)) ok = 0;
        printf("PROBE global_load_lds lane_linear_ok=%d\n", ok);
    }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("PROBE device=%s CUs=%d clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    return 0;
}
