// Micro-benchmark of a CU's vector-memory (TA / L1) path in the shape the tokenizer uses it: 1024-thread
// workgroups, one per CU (16 waves), every wave streaming independent requests with several in flight.
// Prices, in shader cycles per wave-instruction per CU (all 16 waves issuing the same pattern):
//   coalesced loads of 4 / 8 / 12 / 16 B per lane from an L2-resident table, random n x 16 B gathers from 64-B and
//   32-B records, scattered dword stores in the shape of the id stream, wide coalesced stores.
// build: hipcc --offload-arch=gfx950 -O3 -o ta ta.hip ; run on the GPU box (prints one line per pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 mix(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ u64 now() {
    u64 t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

enum Mode {
    C1, C2, C3, C4,            // coalesced loads, 4/8/12/16 B per lane, wave-contiguous, random wave base
    G64_1, G64_2, G64_3, G64_4,  // random 64-B record per lane, n x dwordx4
    G32_1, G32_2,              // random 32-B record per lane
    G64_X2,                    // 64-B record as 8 x dwordx2
    G128_4,                    // 4 x dwordx4 from the first half of a random 128-B line
    GS64_4,                    // 64-B records, lanes of a wave sorted by record (neighbours share lines)
    G16A_2,                    // 2 x dwordx4 from a random 16-B-aligned address (a 32-byte span that is NOT 32-byte aligned: a
                               // record array without the look-ahead copies -- block b and block b + 1 -- at half the footprint)
    S1SC, S1CO, S2CO, S4CO,    // stores: scattered dword (stride ~2.3 words), coalesced dword / x2 / x4
    NMODES
};

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(1024) k_ta(const u32 *__restrict__ tab, u32 tab_bytes, int iters, u32 *out, u64 *cycles) {
    const u32 lane = threadIdx.x & 63;
    const u32 gw = blockIdx.x * 16 + (threadIdx.x >> 6);
    u32 acc = 0;
    u32 h = mix(gw * 2654435761u + 12345u);
    const u32 n64 = tab_bytes / 64, n32 = tab_bytes / 32, n128 = tab_bytes / 128;
    const u64 t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            h = h * 1664525u + 1013904223u;
            const u32 rw = mix(h);                 // wave-uniform random
            const u32 rl = mix(h ^ (lane * 0x9E3779B9u));  // per-lane random
            if (MODE == C1) { const u32 *p = tab + (rw % (tab_bytes / 256)) * 64 + lane; acc ^= __builtin_nontemporal_load(p); }
            if (MODE == C2) { const u32x2 *p = (const u32x2 *)(tab + (rw % (tab_bytes / 512)) * 128) + lane; u32x2 v = *p; acc ^= v.x ^ v.y; }
            if (MODE == C3) { const u32 *p = tab + (rw % (tab_bytes / 1024)) * 256 + lane * 3; u32x3 v = *(const u32x3 *)p; acc ^= v.x ^ v.y ^ v.z; }
            if (MODE == C4) { const u32x4 *p = (const u32x4 *)(tab + (rw % (tab_bytes / 1024)) * 256) + lane; u32x4 v = *p; acc ^= v.x ^ v.w; }
            if constexpr (MODE >= G64_1 && MODE <= G64_4) {
                const u32x4 *p = (const u32x4 *)(tab + (size_t)(rl % n64) * 16);
                constexpr int N = MODE - G64_1 + 1;
                u32x4 v[N];
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = p[i];
#pragma unroll
                for (int i = 0; i < N; ++i) acc ^= v[i].x ^ v[i].w;
            }
            if constexpr (MODE == G32_1 || MODE == G32_2) {
                const u32x4 *p = (const u32x4 *)(tab + (size_t)(rl % n32) * 8);
                constexpr int N = MODE - G32_1 + 1;
                u32x4 v[N];
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = p[i];
#pragma unroll
                for (int i = 0; i < N; ++i) acc ^= v[i].x ^ v[i].w;
            }
            if (MODE == G64_X2) {
                const u32x2 *p = (const u32x2 *)(tab + (size_t)(rl % n64) * 16);
                u32x2 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = p[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y;
            }
            if (MODE == G128_4) {
                const u32x4 *p = (const u32x4 *)(tab + (size_t)(rl % n128) * 32);
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = p[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].w;
            }
            if (MODE == G16A_2) {
                const u32x4 *p = (const u32x4 *)(tab + (size_t)(rl % (tab_bytes / 16 - 1)) * 4);
                u32x4 v[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) v[i] = p[i];
#pragma unroll
                for (int i = 0; i < 2; ++i) acc ^= v[i].x ^ v[i].w;
            }
            if (MODE == GS64_4) {
                // a position-sorted batch: the wave's 64 queries fall into ~3 consecutive records
                const u32 rec = (rw % (n64 - 4)) + (lane * 3 >> 6);
                const u32x4 *p = (const u32x4 *)(tab + (size_t)rec * 16);
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = p[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].w;
            }
            if (MODE == S1SC) { u32 *p = out + (size_t)(rw % (tab_bytes / 1024)) * 256 + (lane * 37 >> 4); __builtin_nontemporal_store(h, p); }
            if (MODE == S1CO) { u32 *p = out + (size_t)(rw % (tab_bytes / 256)) * 64 + lane; __builtin_nontemporal_store(h, p); }
            if (MODE == S2CO) { u32x2 *p = (u32x2 *)(out + (size_t)(rw % (tab_bytes / 512)) * 128) + lane; u32x2 v = {h, h}; __builtin_nontemporal_store(v, p); }
            if (MODE == S4CO) { u32x4 *p = (u32x4 *)(out + (size_t)(rw % (tab_bytes / 1024)) * 256) + lane; u32x4 v = {h, h, h, h}; __builtin_nontemporal_store(v, p); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 t1 = now();
    if (lane == 0) atomicMax(cycles, t1 - t0);
    if (acc == 0x12345678u) out[gw] = acc;
}

static const char *NAMES[NMODES] = {
    "coalesced dword", "coalesced dwordx2", "coalesced dwordx3", "coalesced dwordx4",
    "gather 64-B rec, 1 x4", "gather 64-B rec, 2 x4", "gather 64-B rec, 3 x4", "gather 64-B rec, 4 x4",
    "gather 32-B rec, 1 x4", "gather 32-B rec, 2 x4", "gather 64-B rec, 8 x2", "gather 128-B line, 4 x4 (first half)",
    "sorted wave, 64-B rec, 4 x4", "gather 2 x4 at a 16-B-aligned address", "store dword scattered (id stream)", "store dword coalesced", "store dwordx2 coalesced",
    "store dwordx4 coalesced"};
static const int INSTR[NMODES] = {1, 1, 1, 1, 1, 2, 3, 4, 1, 2, 8, 4, 4, 2, 1, 1, 1, 1};

template <int MODE, int UNROLL>
static void run(const u32 *tab, u32 tab_bytes, u32 *out, u64 *d_cyc, int wgs, const char *tag) {
    const int iters = 2048 / UNROLL;
    hipMemset(d_cyc, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_ta<MODE, UNROLL>), dim3(wgs), dim3(1024), 0, 0, tab, tab_bytes, iters, out, d_cyc);
    hipDeviceSynchronize();
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_ta<MODE, UNROLL>), dim3(wgs), dim3(1024), 0, 0, tab, tab_bytes, iters, out, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    u64 cyc = 0; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double groups = (double)iters * UNROLL;            // per wave
    const double per_cu_groups = groups * 16 * (wgs / 256.0); // wave-groups a CU serves
    printf("%-40s %-10s unroll %d  %8.1f us  %9.0f cyc  clk %.2f GHz  %7.1f cyc per wave-group per CU (%d instr)  = %5.2f cyc/lane-group\n", NAMES[MODE], tag,
           UNROLL, ms * 1e3, (double)cyc, cyc / (ms * 1e6), cyc / per_cu_groups, INSTR[MODE], cyc / per_cu_groups / 64.0);
}

template <int U>
static void suite(const u32 *tab, u32 bytes, u32 *out, u64 *d_cyc, const char *tag) {
    run<C1, U>(tab, bytes, out, d_cyc, 256, tag);
    run<C2, U>(tab, bytes, out, d_cyc, 256, tag);
    run<C3, U>(tab, bytes, out, d_cyc, 256, tag);
    run<C4, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G64_1, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G64_2, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G64_3, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G64_4, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G32_1, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G32_2, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G64_X2, U>(tab, bytes, out, d_cyc, 256, tag);
    run<G128_4, U>(tab, bytes, out, d_cyc, 256, tag);
    run<GS64_4, U>(tab, bytes, out, d_cyc, 256, tag);
}

int main(int argc, char **argv) {
    u32 *out; hipMalloc(&out, 64u << 20);
    u64 *d_cyc; hipMalloc(&d_cyc, 8);
    const u32 small = 2112 * 1024;  // 2.1 MB: the 100k-region index (L2-resident)
    u32 *tab; hipMalloc(&tab, 64u << 20);
    hipMemset(tab, 1, 64u << 20);
    suite<4>(tab, small, out, d_cyc, "2.1MB");
    suite<8>(tab, small, out, d_cyc, "2.1MB");
    run<G16A_2, 4>(tab, small, out, d_cyc, 256, "2.1MB");
    run<G16A_2, 4>(tab, small / 2, out, d_cyc, 256, "1.05MB");
    run<G64_4, 4>(tab, 64u << 20, out, d_cyc, 256, "64MB");
    run<G32_2, 4>(tab, 64u << 20, out, d_cyc, 256, "64MB");
    // a 1M-region universe: 16 MB of 32-byte records against 8 MB without the look-ahead copies (4 MB of L2 per XCD)
    run<G32_2, 4>(tab, 16u << 20, out, d_cyc, 256, "16MB");
    run<G16A_2, 4>(tab, 8u << 20, out, d_cyc, 256, "8MB");
    run<G32_2, 4>(tab, 4u << 20, out, d_cyc, 256, "4MB");
    run<G16A_2, 4>(tab, 2u << 20, out, d_cyc, 256, "2MB");
    run<S1SC, 4>(tab, 32u << 20, out, d_cyc, 256, "32MB");
    run<S1CO, 4>(tab, 32u << 20, out, d_cyc, 256, "32MB");
    run<S2CO, 4>(tab, 32u << 20, out, d_cyc, 256, "32MB");
    run<S4CO, 4>(tab, 32u << 20, out, d_cyc, 256, "32MB");
    return 0;
}
