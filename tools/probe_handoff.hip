// What does a flag-ordered hand-off of a partial-result tile between workgroups of ONE XCD cost?  (the price of a
// fused 5-pass attention backward that accumulates dQ in L2 in a fixed rotation order instead of running a separate dQ kernel.)
//
// A group = NM member workgroups on one XCD (block b runs on XCD b % 8; group members are the blocks x + 8 (NM g + k)).  Each group owns
// NT = NM tiles of TILE bytes per chain.  Member k visits the tiles in the rotation i = (k + s) % NT, s = 0 .. NT-1; a tile is therefore
// handed from member k (its visit number s) to member k-1 (visit s+1).  A visit = wait for flag[i] == visit number, load the tile through
// L2 (cache bits AUX_LD), add the member's contribution, burn `compute` x 64 clocks (s_sleep: the matrix work of one q block), store the
// tile, wait for the stores (vmcnt(0)), publish flag[i] = visit + 1.  With `chains` = 2 every member alternates between two independent
// groups' tiles (A0 B0 A1 B1 ...): the hand-off of chain A then has chain B's compute time to land -- the "skewed rotation".
// Reported: wall time per visit (100 MHz constant clock) minus the pure compute time = exposed hand-off cost; sum check of every tile.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_handoff.hip -o tools/probe_handoff && tools/probe_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int AUX_LD, int AUX_FLAG>
__global__ void __launch_bounds__(256) handoff(float* tiles, unsigned* flags, int NM, int tile_f4_per_thread, int compute, int chains, int spin_sleep,
                                               long long* wall, int* err) {
    const int b = blockIdx.x, x = b & 7, q = b >> 3, g = q / NM, k = q - g * NM;
    const int NT = NM;
    const int grp = x + 8 * g;                                          // group id
    const int tile_floats = tile_f4_per_thread * 4 * 256;
    const long long chain_stride = (long long)gridDim.x / NM * NT;     // tiles per chain (all groups)
    __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(tiles, 0, 0xffffffffu, 0x00020000);
    __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 0xffffffffu, 0x00020000);
    const long long w0 = wall_clock64();
    for (int s = 0; s < NT * chains; ++s) {
        const int c = s % chains, ls = s / chains;
        const int i = (k + ls) % NT;
        const long long tile_id = c * chain_stride + (long long)grp * NT + i;
        const unsigned fo = (unsigned)(tile_id * 4);
        u32x4_t v[8];
        if (ls > 0) {
            // wait for the previous visitor (lane 0 polls, the wave / block follows through the barrier)
            if (threadIdx.x == 0) {
                for (int spin_ = 0; ; ++spin_) { if (spin_ > (1 << 22)) { atomicAdd(err, 1 << 20); break; } unsigned f_; asm volatile("buffer_load_dword %0, %1, %2, 0 offen sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f_) : "v"(fo), "s"(rsF) : "memory"); if (f_ == (unsigned)ls) break; if (spin_sleep) __builtin_amdgcn_s_sleep(1); }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < tile_f4_per_thread)
                    v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsT, (unsigned)((r * 256 + threadIdx.x) * 16), (int)(tile_id * tile_floats * 4), AUX_LD);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < tile_f4_per_thread)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[r][e] = __float_as_uint(__uint_as_float(v[r][e]) + (float)(k + 1));
        for (int t = 0; t < compute; ++t) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < tile_f4_per_thread)
                __builtin_amdgcn_raw_buffer_store_b128(v[r], rsT, (unsigned)((r * 256 + threadIdx.x) * 16), (int)(tile_id * tile_floats * 4), 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(ls + 1), rsF, fo, 0, 0);
    }
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) wall[b] = w1 - w0;
    // the last visitor of a tile checks it: sum_k (k + 1) = NM (NM + 1) / 2
}

__global__ void check(const float* tiles, long long n, float want, int* err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && tiles[i] != want) atomicAdd(err, 1);
}

template <int AUX_LD, int AUX_FLAG>
static void run(const char* label, int groups_per_xcd, int NM, int f4, int compute, int chains, int spin_sleep) {
    const int blocks = 8 * groups_per_xcd * NM;
    const long long ntiles = (long long)chains * (blocks / NM) * NM;
    const long long tile_floats = (long long)f4 * 4 * 256;
    float* tiles; unsigned* flags; long long* wall; int* err;
    hipMalloc(&tiles, ntiles * tile_floats * 4); hipMalloc(&flags, ntiles * 4); hipMalloc(&wall, blocks * 8); hipMalloc(&err, 4);
    double best = 1e30, sum = 0; int nrun = 0, bad = 0;
    for (int it = 0; it < 6; ++it) {
        hipMemset(tiles, 0xff, ntiles * tile_floats * 4); hipMemset(flags, 0, ntiles * 4); hipMemset(err, 0, 4);
        hipLaunchKernelGGL((handoff<AUX_LD, AUX_FLAG>), dim3(blocks), dim3(256), 0, 0, tiles, flags, NM, f4, compute, chains, spin_sleep, wall, err);
        hipDeviceSynchronize();
        const long long n = ntiles * tile_floats;
        hipLaunchKernelGGL(check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tiles, n, (float)(NM * (NM + 1) / 2), err);
        int e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); bad += e;
        std::vector<long long> h(blocks); hipMemcpy(h.data(), wall, blocks * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        if (it > 0) { const double us = mx / 100.0; best = us < best ? us : best; sum += us; ++nrun; }
    }
    const int visits = NM * chains;
    printf("%-34s groups/XCD %d  tile %3d KB  compute %4d x64clk  chains %d : %8.2f us per visit (best %8.2f), wrong elements %d\n", label,
           groups_per_xcd, (int)(tile_floats * 4 / 1024), compute, chains, sum / nrun / visits, best / visits, bad);
    hipFree(tiles); hipFree(flags); hipFree(wall); hipFree(err);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    // pure hand-off latency (no compute), 12 members, 16 KB and 32 KB tiles, cache-bit variants of the loads (1 = sc0, 16 = sc1, 17 = both)
    for (int f4 = 4; f4 <= 8; f4 *= 2) {
        run<17, 17>("tile loads sc0 sc1", 1, 12, f4, 0, 1, 0);
        run<1, 1>("tile loads sc0", 1, 12, f4, 0, 1, 0);
        run<0, 17>("tile loads default policy", 1, 12, f4, 0, 1, 0);
        run<17, 17>("  same, poll with s_sleep", 1, 12, f4, 0, 1, 1);
        run<17, 17>("  two groups per XCD", 2, 12, f4, 0, 1, 0);
    }
    // with the compute of one q block (~8 us at 1.9 GHz = 240 x 64 clocks; s_sleep counts in 64-clock units), one chain vs two
    for (int comp : {60, 120, 240}) {
        run<17, 17>("compute, one chain", 2, 12, 4, comp, 1, 0);
        run<17, 17>("compute, two chains (skewed)", 2, 12, 4, comp, 2, 0);
        run<17, 17>("compute, one chain, 32 KB", 2, 12, 8, comp, 1, 0);
        run<17, 17>("compute, two chains, 32 KB", 2, 12, 8, comp, 2, 0);
    }
    return 0;
}
