// Does a cache-policy bit make a divergent 32-byte-record gather cheaper for a CU's vector-memory path?
// Same shape as ta.hip's "gather 32-B rec, 2 x4" (1024-thread workgroups, one per CU, 16 waves, 4 gathers in flight
// per lane, 2.1 MB L2-resident table), with the two 16-byte loads issued as inline asm carrying
//   0 (none)   1 nt   2 sc0   3 sc1   4 sc0 sc1   5 sc0 nt   6 sc1 nt   7 sc0 sc1 nt
// build: hipcc --offload-arch=gfx950 -O3 -o gather_policy gather_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
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

#define LD2(POL)                                                                                          \
    asm volatile("global_load_dwordx4 %0, %2, off " POL "\n\tglobal_load_dwordx4 %1, %2, off offset:16 " POL \
                 : "=&v"(va[u]), "=&v"(vb[u])                                                             \
                 : "v"(p)                                                                                 \
                 : "memory")

template <int POLICY>
__global__ void __launch_bounds__(1024) k_g(const u32 *__restrict__ tab, u32 tab_bytes, int iters, u32 *out, u64 *cycles) {
    const u32 lane = threadIdx.x & 63;
    const u32 gw = blockIdx.x * 16 + (threadIdx.x >> 6);
    u32 acc = 0;
    u32 h = mix(gw * 2654435761u + 12345u);
    const u32 n32 = tab_bytes / 32;
    const u64 t0 = now();
    for (int it = 0; it < iters; ++it) {
        u32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h = h * 1664525u + 1013904223u;
            const u32 rl = mix(h ^ (lane * 0x9E3779B9u));
            const u32 *p = tab + (size_t)(rl % n32) * 8;
            if (POLICY == 0) LD2("");
            if (POLICY == 1) LD2("nt");
            if (POLICY == 2) LD2("sc0");
            if (POLICY == 3) LD2("sc1");
            if (POLICY == 4) LD2("sc0 sc1");
            if (POLICY == 5) LD2("sc0 nt");
            if (POLICY == 6) LD2("sc1 nt");
            if (POLICY == 7) LD2("sc0 sc1 nt");
        }
        // the loaded registers are operands of the wait, so that no use of them can move in front of it
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3])
                     :
                     : "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= va[u].x ^ vb[u].w;
    }
    const u64 t1 = now();
    if (lane == 0) atomicMax((unsigned long long *)cycles, (unsigned long long)(t1 - t0));
    if (acc == 0x12345678u) out[gw] = acc;
}

template <int POLICY>
static void run(const char *name, const u32 *tab, u32 bytes, u32 *out, u64 *d_cyc) {
    const int iters = 512;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d_cyc, 0, 8);
        hipLaunchKernelGGL((k_g<POLICY>), dim3(256), dim3(1024), 0, 0, tab, bytes, iters, out, d_cyc);
        hipDeviceSynchronize();
    }
    u64 cyc = 0;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s %9.0f cyc  = %5.2f cyc per lane-gather (2 x 16 B of a random 32-B record) per CU\n", name, (double)cyc,
           (double)cyc / ((double)iters * 4 * 16 * 64));
}

int main() {
    u32 *out, *tab;
    u64 *d_cyc;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&d_cyc, 8);
    const u32 bytes = 1600 * 1024;  // the 100k-universe record array
    hipMalloc(&tab, bytes);
    hipMemset(tab, 1, bytes);
    run<0>("(none)", tab, bytes, out, d_cyc);
    run<1>("nt", tab, bytes, out, d_cyc);
    run<2>("sc0", tab, bytes, out, d_cyc);
    run<3>("sc1", tab, bytes, out, d_cyc);
    run<4>("sc0 sc1", tab, bytes, out, d_cyc);
    run<5>("sc0 nt", tab, bytes, out, d_cyc);
    run<6>("sc1 nt", tab, bytes, out, d_cyc);
    run<7>("sc0 sc1 nt", tab, bytes, out, d_cyc);
    return 0;
}
