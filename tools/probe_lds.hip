// LDS read throughput per CU for the access patterns of the attention / GEMM kernels, at 4 and 8 waves per CU:
//   mode 0: ds_read_b128 row fragments of a swizzled [64 rows][128 B] tile (lane -> row lane&31, chunk (2kk+hh) ^ swz)
//   mode 1: ds_read_b64_tr_b16 transposing reads of the same tile
//   mode 2: ds_read_b128 broadcast (32 lanes share an address: the seed reads of attn_bwd_dkv_kernel)
//   mode 3: ds_read_b64 (plain, lane-contiguous)
// Each wave issues NI independent reads per loop iteration and waits for all of them; cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds.hip -o tools/probe_lds && tools/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ int rev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const int hh = lane >> 5;
    unsigned a[4];
    if (MODE == 0) {
        for (int kk = 0; kk < 4; ++kk) { const int row = lane & 31, c = (kk * 2 + hh) ^ rev3((row >> 1) & 7); a[kk] = (unsigned)(uintptr_t)(smem + row * 128 + c * 16); }
    } else if (MODE == 1) {
        const int G = lane >> 4, u = lane & 15;
        for (int kk = 0; kk < 4; ++kk) {
            const int dblk = kk >> 1, sec = kk & 1;
            const int row = 8 * sec + 4 * (G >> 1) + (u >> 2);
            const int c = (dblk * 4 + 2 * (G & 1) + ((u & 3) >> 1)) ^ rev3((row >> 1) & 7);
            a[kk] = (unsigned)(uintptr_t)(smem + row * 128 + (c << 4) + ((u & 1) << 3));
        }
    } else if (MODE == 2) {
        for (int kk = 0; kk < 4; ++kk) a[kk] = (unsigned)(uintptr_t)(smem + 16 * hh + 32 * kk);
    } else {
        for (int kk = 0; kk < 4; ++kk) a[kk] = (unsigned)(uintptr_t)(smem + lane * 8 + 512 * kk);
    }
    f32x4_t acc = {0, 0, 0, 0};
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            f32x4_t r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %9 offset:0\n\tds_read_b128 %2, %10 offset:0\n\tds_read_b128 %3, %11 offset:0\n\t"
                         "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %9 offset:4096\n\tds_read_b128 %6, %10 offset:4096\n\tds_read_b128 %7, %11 offset:4096\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
            acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        } else {
            f32x2_t r0, r1, r2, r3, r4, r5, r6, r7;
            if (MODE == 1)
                asm volatile("ds_read_b64_tr_b16 %0, %8 offset:0\n\tds_read_b64_tr_b16 %1, %9 offset:0\n\tds_read_b64_tr_b16 %2, %10 offset:0\n\tds_read_b64_tr_b16 %3, %11 offset:0\n\t"
                             "ds_read_b64_tr_b16 %4, %8 offset:2048\n\tds_read_b64_tr_b16 %5, %9 offset:2048\n\tds_read_b64_tr_b16 %6, %10 offset:2048\n\tds_read_b64_tr_b16 %7, %11 offset:2048\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
            else
                asm volatile("ds_read_b64 %0, %8 offset:0\n\tds_read_b64 %1, %9 offset:0\n\tds_read_b64 %2, %10 offset:0\n\tds_read_b64 %3, %11 offset:0\n\t"
                             "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %9 offset:2048\n\tds_read_b64 %6, %10 offset:2048\n\tds_read_b64 %7, %11 offset:2048\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
            const f32x2_t t = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
            acc[0] += t.x; acc[1] += t.y;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int MODE>
static void run(const char* name, int bytes_per_instr) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 1 << 20); hipMalloc(&sink, 64);
    const int iters = 2000;
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<256, threads>>>(iters, cyc, sink);
        hipEventRecord(e0);
        k<MODE><<<256, threads>>>(iters, cyc, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256 * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= h.size();
        const double bytes_cu = (double)iters * 8 * bytes_per_instr * (threads / 64);
        printf("%-34s %d waves/CU: %7.1f cycles per 8-read batch per wave, %6.1f B/clk/CU, %5.2f TB/s chip (wall %.3f ms)\n", name, threads / 64,
               mean / iters, bytes_cu / mean, bytes_cu * 256 / (ms * 1e-3) / 1e12, ms);
    }
}

int main() {
    run<0>("ds_read_b128 row fragments", 1024);
    run<1>("ds_read_b64_tr_b16", 512);
    run<2>("ds_read_b128 broadcast (2 addrs)", 1024);
    run<3>("ds_read_b64 contiguous", 512);
    return 0;
}
