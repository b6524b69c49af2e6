// Micro-benchmark: how many divergent (one distinct 128-B line per lane) vector loads per cycle a CU
// sustains out of an L2-resident table -- the unit cost the tokenizer's record fetch is priced in.
// build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ u32 mix(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: one dwordx4 per lane, random line            (1 request / lane / instr)
// MODE 1: four dwordx4 per lane from the SAME random line (the tokenizer's starts+ends fetch)
// MODE 2: 8 lanes share a line: lane&7 picks the 16-B quad (8 lines per wave instruction)
// MODE 3: one dword per lane, random line
// MODE 4: one dwordx2 per lane, random line
// MODE 5: 2 lanes share a line (64 B each half?) -> lane&1 picks quad 0/1
// MODE 6: 4 lanes share a line
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ tab, u32 n_lines, int iters, u32 *out) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    u32 acc = 0;
    u32 h = mix(tid * 2654435761u + 12345u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h = h * 1664525u + 1013904223u;
            u32 r = mix(h);
            if (MODE == 2) r = __shfl(r, lane & ~7u);
            if (MODE == 5) r = __shfl(r, lane & ~1u);
            if (MODE == 6) r = __shfl(r, lane & ~3u);
            const u32 line = r % n_lines;
            const uint4 *p = tab + (size_t)line * 8;
            if (MODE == 0) { uint4 v = p[0]; acc ^= v.x ^ v.w; }
            if (MODE == 1) { uint4 a = p[0], b = p[1], c = p[2], d = p[3]; acc ^= a.x ^ b.y ^ c.z ^ d.w; }
            if (MODE == 2) { uint4 v = p[lane & 7]; acc ^= v.x ^ v.w; }
            if (MODE == 3) { acc ^= reinterpret_cast<const u32 *>(p)[0]; }
            if (MODE == 4) { uint2 v = reinterpret_cast<const uint2 *>(p)[0]; acc ^= v.x ^ v.y; }
            if (MODE == 5) { uint4 v = p[lane & 1]; acc ^= v.x ^ v.w; }
            if (MODE == 14) { uint4 a = p[0], b = p[1], c = p[2]; acc ^= a.x ^ b.y ^ c.z; }
            if (MODE == 15) { const uint4 *q = tab + (size_t)(r % (n_lines * 2)) * 4; uint4 a = q[0], b = q[1], c = q[2]; acc ^= a.x ^ b.y ^ c.z; }
            if (MODE == 16) { const uint2 *q = reinterpret_cast<const uint2 *>(tab + (size_t)(r % (n_lines * 2)) * 4); uint2 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = q[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y; }
            if (MODE == 17) { const u32 *q = reinterpret_cast<const u32 *>(tab + (size_t)(r % (n_lines * 2)) * 4); u32 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = q[i];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc ^= v[i]; }
            if (MODE == 18) { const uint2 *q = reinterpret_cast<const uint2 *>(tab + (size_t)(r % (n_lines * 2)) * 4); uint2 v[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i] = q[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) acc ^= v[i].x ^ v[i].y; }
            if (MODE == 19) { const uint4 *q = tab + (size_t)(r % (n_lines * 2)) * 4; uint4 a = q[0], b = q[1], c = q[2], d = q[3]; acc ^= a.x ^ b.y ^ c.z ^ d.w; }
            if (MODE == 7) { uint4 a = p[0], b = p[1]; acc ^= a.x ^ b.y; }
            if (MODE == 8) { uint4 a = p[0]; acc ^= a.x; if ((a.x ^ r) % 100u < 65u) { uint4 b = p[1 + (r >> 20) % 7]; acc ^= b.y; } }
            if (MODE == 9) { uint4 a = p[0], c = p[1]; acc ^= a.x ^ c.z; if ((a.x ^ r) % 100u < 65u) { uint4 b = p[2 + (r >> 20) % 6]; acc ^= b.y; } }
            if (MODE == 10) { uint4 v; v.x = __builtin_nontemporal_load(&p[0].x); v.y = __builtin_nontemporal_load(&p[0].y); v.z = __builtin_nontemporal_load(&p[0].z); v.w = __builtin_nontemporal_load(&p[0].w); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            if (MODE == 11) { uint4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); acc ^= v.x ^ v.w; }
            if (MODE == 12) { uint4 v; asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); acc ^= v.x ^ v.w; }
            if (MODE == 13) { uint4 v; asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); acc ^= v.x ^ v.w; }
            if (MODE == 6) { uint4 v = p[lane & 3]; acc ^= v.x ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE>
static void run(const char *name, const uint4 *tab, u32 n_lines, int wg_per_cu, u32 *out, double req_per_lane_iter, double bytes_per_lane_iter) {
    const int iters = 256;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather<MODE>, dim3(grid), dim3(256), 0, 0, tab, n_lines, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gather<MODE>, dim3(grid), dim3(256), 0, 0, tab, n_lines, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double lane_iters = (double)grid * 256 * iters * 4;
    const double req = lane_iters * req_per_lane_iter;
    printf("%-34s lines=%7u wg/cu=%d  %8.1f us  %7.2f Greq/s  %5.2f req/clk/CU(2.4GHz)  %7.2f TB/s useful\n", name, n_lines, wg_per_cu,
           ms * 1e3, req / ms / 1e6, req / (ms * 1e-3) / 256 / 2.4e9, lane_iters * bytes_per_lane_iter / ms / 1e9);
}

int main() {
    u32 *out; hipMalloc(&out, 256 * 16 * 256 * 4);
    for (u32 n_lines : {16700u}) {  // 1.6 MB (L2), 12.8 MB (MALL), 205 MB (HBM/MALL)
        uint4 *tab; hipMalloc(&tab, (size_t)n_lines * 128);
        hipMemset(tab, 1, (size_t)n_lines * 128);
        for (int wg : {8}) {
            run<19>("4 x x4, 64-B record", tab, n_lines, wg, out, 4, 64);
            run<16>("8 x x2, 64-B record", tab, n_lines, wg, out, 8, 64);
            run<17>("16 x dword, 64-B record", tab, n_lines, wg, out, 16, 64);
            run<18>("6 x x2, 48 B of a 64-B record", tab, n_lines, wg, out, 6, 48);
            run<14>("3 x x4 same line", tab, n_lines, wg, out, 3, 48);
            run<15>("3 x x4, 64-B records (2 per line)", tab, n_lines, wg, out, 3, 48);
            run<7>("2 x x4 same line", tab, n_lines, wg, out, 2, 32);
            run<8>("x4 then 65%: x4 same line", tab, n_lines, wg, out, 1.65, 16);
            run<9>("2 x x4 then 65%: x4 same line", tab, n_lines, wg, out, 2.65, 32);
            run<10>("x4 nontemporal builtin", tab, n_lines, wg, out, 1, 16);
            run<13>("x4 asm plain (wait each)", tab, n_lines, wg, out, 1, 16);
            run<11>("x4 asm sc0 sc1 (wait each)", tab, n_lines, wg, out, 1, 16);
            run<12>("x4 asm nt (wait each)", tab, n_lines, wg, out, 1, 16);
            run<0>("x4 divergent (1 req/lane)", tab, n_lines, wg, out, 1, 16);
            run<1>("4 x x4 same line (4 req/lane)", tab, n_lines, wg, out, 4, 64);
            run<2>("x4, 8 lanes per line", tab, n_lines, wg, out, 1, 16);
            run<6>("x4, 4 lanes per line", tab, n_lines, wg, out, 1, 16);
            run<5>("x4, 2 lanes per line", tab, n_lines, wg, out, 1, 16);
            run<3>("dword divergent", tab, n_lines, wg, out, 1, 4);
            run<4>("dwordx2 divergent", tab, n_lines, wg, out, 1, 8);
        }
        hipFree(tab);
    }
    return 0;
}
