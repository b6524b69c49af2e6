// Micro-benchmark: per-CU rate of fully coalesced vector loads/stores of different widths
// (L2-resident source), to price the tokenizer's query loads and offset stores in TA cycles.
// build: hipcc --offload-arch=gfx950 -O3 -o coalesced coalesced.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;

// MODE 0: dword  1: dwordx2  2: dwordx4  (loads, lane-consecutive)
// MODE 3: dwordx4 stores     4: dword stores   5: dwordx4 loads, lane stride 32 B (every other 16 B)
template <int MODE>
__global__ void __launch_bounds__(256) k_co(const u32 *__restrict__ src, u32 *__restrict__ dst, u32 n_words, int iters, u32 *out) {
    const u32 gt = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nthr = gridDim.x * blockDim.x;
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 k = (u32)(it * 4 + u);
            if (MODE == 0) { u32 i = (gt + k * nthr) % n_words; acc ^= src[i]; }
            if (MODE == 1) { u32 i = ((gt + k * nthr) * 2) % n_words; uint2 v = *(const uint2 *)(src + i); acc ^= v.x ^ v.y; }
            if (MODE == 2) { u32 i = ((gt + k * nthr) * 4) % n_words; uint4 v = *(const uint4 *)(src + i); acc ^= v.x ^ v.w; }
            if (MODE == 3) { u32 i = ((gt + k * nthr) * 4) % n_words; *(uint4 *)(dst + i) = make_uint4(k, gt, k, gt); }
            if (MODE == 4) { u32 i = (gt + k * nthr) % n_words; dst[i] = k; }
            if (MODE == 5) { u32 i = ((gt + k * nthr) * 8) % n_words; uint4 v = *(const uint4 *)(src + i); acc ^= v.x ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[gt] = acc;
}

template <int MODE>
static void run(const char *name, const u32 *src, u32 *dst, u32 n_words, u32 *out, double bytes_per_req) {
    const int iters = 256, grid = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_co<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, n_words, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_co<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, n_words, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double req = (double)grid * 256 * iters * 4;
    printf("%-34s words=%9u  %8.1f us  %7.2f Greq/s  %5.2f lane-req/clk/CU(2.4GHz)  %6.2f TB/s\n", name, n_words, ms * 1e3,
           req / ms / 1e6, req / (ms * 1e-3) / 256 / 2.4e9, req * bytes_per_req / ms / 1e9);
}

int main() {
    u32 *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (u32 n_words : {1u << 19, 1u << 28}) {  // 2 MB (L2) and 1 GB (HBM)
        u32 *src, *dst; hipMalloc(&src, (size_t)n_words * 4); hipMalloc(&dst, (size_t)n_words * 4);
        hipMemset(src, 1, (size_t)n_words * 4);
        run<0>("load dword coalesced", src, dst, n_words, out, 4);
        run<1>("load dwordx2 coalesced", src, dst, n_words, out, 8);
        run<2>("load dwordx4 coalesced", src, dst, n_words, out, 16);
        run<5>("load dwordx4 lane stride 32 B", src, dst, n_words, out, 16);
        run<3>("store dwordx4 coalesced", src, dst, n_words, out, 16);
        run<4>("store dword coalesced", src, dst, n_words, out, 4);
        hipFree(src); hipFree(dst);
    }
    return 0;
}
