// Does slow vector-memory traffic of ONE wave delay the L2-hit gathers of the other waves of its CU?
// One 1024-thread workgroup per CU.  Waves 0..14 gather 2 x 16 B from random 32-byte records of an L2-resident table
// (the tokenizer's record burst, 4 gathers in flight per lane); wave 15 runs a side pattern for as long as they do:
//   0 nothing   1 coalesced 16-B/lane loads streaming from HBM (fresh lines)   2 agent-scope 8-byte loads that miss L2's
//   neighbourhood (look-back polling)   3 coalesced 16-B/lane non-temporal stores   4 like 1, issued by EVERY gather
//   wave behind its gathers (a prefetch)
// Prints the gather waves' cycles per gather pair per CU.  build: hipcc --offload-arch=gfx950 -O3 -o hol hol.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
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

template <int SIDE>
__global__ void __launch_bounds__(1024) k_hol(const u32 *__restrict__ tab, u32 tab_bytes, const u32 *__restrict__ big, u64 big_words,
                                               u32 *sink_st, int iters, u32 *out, u64 *cycles) {
    __shared__ u32 s_done;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_done = 0;
    __syncthreads();
    u32 acc = 0;
    u32 h = mix((blockIdx.x * 16 + wave) * 2654435761u + 12345u);
    const u32 n32 = tab_bytes / 32;
    if (wave < 15) {
        const u64 t0 = now();
        for (int it = 0; it < iters; ++it) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h = h * 1664525u + 1013904223u;
                const u32 rl = mix(h ^ (lane * 0x9E3779B9u));
                const u32x4 *p = (const u32x4 *)(tab + (size_t)(rl % n32) * 8);
                v[2 * u] = p[0];
                v[2 * u + 1] = p[1];
            }
            u32x4 pf = {0, 0, 0, 0};
            if (SIDE == 4) {
                const u64 w = ((u64)mix(h + it) * 1024u + (u64)(blockIdx.x * 16 + wave) * 256u) % (big_words - 256);
                pf = __builtin_nontemporal_load((const u32x4 *)(big + (w & ~3ull)) + lane);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
            acc ^= pf.x;
        }
        const u64 t1 = now();
        if (lane == 0) {
            atomicAdd((unsigned long long *)&cycles[blockIdx.x], (unsigned long long)(t1 - t0));
            atomicAdd(&s_done, 1u);
        }
    } else {
        u64 k = 0;
        while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 15u) {
            h = h * 1664525u + 1013904223u;
            const u64 w = ((u64)mix(h) * 4096u + (u64)blockIdx.x * 256u) % (big_words - 256);
            if (SIDE == 1) {
                const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(big + (w & ~3ull)) + lane);
                acc ^= v.x;
            } else if (SIDE == 2) {
                const u64 v = __hip_atomic_load((const u64 *)(big + (w & ~1ull)) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc ^= (u32)v;
            } else if (SIDE == 3) {
                u32x4 v = {h, h, h, h};
                __builtin_nontemporal_store(v, (u32x4 *)(sink_st + ((size_t)blockIdx.x * 4096 + (k & 15) * 256)) + lane);
            } else {
                __builtin_amdgcn_s_sleep(8);
            }
            ++k;
        }
    }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <int SIDE>
static void run(const char *name, const u32 *tab, u32 tab_bytes, const u32 *big, u64 big_words, u32 *sink, u32 *out, u64 *cyc, int cus) {
    const int iters = 2000;
    hipMemset(cyc, 0, sizeof(u64) * cus);
    k_hol<SIDE><<<cus, 1024>>>(tab, tab_bytes, big, big_words, sink, 100, out, cyc);
    hipMemset(cyc, 0, sizeof(u64) * cus);
    k_hol<SIDE><<<cus, 1024>>>(tab, tab_bytes, big, big_words, sink, iters, out, cyc);
    hipDeviceSynchronize();
    std::vector<u64> h(cus);
    hipMemcpy(h.data(), cyc, sizeof(u64) * cus, hipMemcpyDeviceToHost);
    double s = 0;
    for (u64 x : h) s += (double)x;
    // s = sum over 15 waves of their elapsed cycles (s_memtime ticks at 100 MHz: convert with the shader clock)
    const double per_wave = s / cus / 15.0;
    printf("%-44s %8.1f memtime ticks per wave, %6.3f ticks per (4 gather pairs x 15 waves)\n", name, per_wave, per_wave / iters);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const u32 tab_bytes = 3200000;
    const u64 big_words = 1ull << 30;  // 4 GiB
    u32 *tab, *big, *sink, *out;
    u64 *cyc;
    hipMalloc(&tab, tab_bytes);
    hipMalloc(&big, big_words * 4);
    hipMalloc(&sink, (size_t)cus * 4096 * 4 + 65536);
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, sizeof(u64) * cus);
    hipMemset(tab, 1, tab_bytes);
    hipMemset(big, 2, big_words * 4);
    run<0>("side wave idle", tab, tab_bytes, big, big_words, sink, out, cyc, cus);
    run<1>("side wave streams 16 B/lane from HBM", tab, tab_bytes, big, big_words, sink, out, cyc, cus);
    run<2>("side wave polls agent-scope 8-byte loads", tab, tab_bytes, big, big_words, sink, out, cyc, cus);
    run<3>("side wave stores 16 B/lane non-temporal", tab, tab_bytes, big, big_words, sink, out, cyc, cus);
    run<4>("every wave prefetches 16 B/lane from HBM", tab, tab_bytes, big, big_words, sink, out, cyc, cus);
    run<0>("side wave idle (again)", tab, tab_bytes, big, big_words, sink, out, cyc, cus);
    return 0;
}
