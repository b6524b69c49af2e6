// Host <-> device copy rates for the host-pointer entry points: pageable vs page-locked buffers, hipHostRegister cost.
// build: hipcc --offload-arch=gfx950 -O3 -o pcie pcie.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (size_t mb : {4, 12, 48}) {
        const size_t n = mb << 20;
        void *d; hipMalloc(&d, n);
        char *pg = (char *)malloc(n); memset(pg, 1, n);
        char *pin; hipHostMalloc((void **)&pin, n, hipHostMallocDefault); memset(pin, 1, n);
        hipStream_t s; hipStreamCreate(&s);
        auto t = [&](const char *name, auto &&f) {
            f(); hipDeviceSynchronize();
            double t0 = now(); for (int i = 0; i < 10; ++i) f(); hipDeviceSynchronize();
            double dt = (now() - t0) / 10;
            printf("%2zu MB %-34s %8.1f us  %6.1f GB/s\n", mb, name, dt * 1e6, n / dt / 1e9);
        };
        t("H2D pageable hipMemcpy", [&] { hipMemcpy(d, pg, n, hipMemcpyHostToDevice); });
        t("D2H pageable hipMemcpy", [&] { hipMemcpy(pg, d, n, hipMemcpyDeviceToHost); });
        t("H2D pinned hipMemcpyAsync", [&] { hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); });
        t("D2H pinned hipMemcpyAsync", [&] { hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
        t("memcpy pageable -> pinned (1 thread)", [&] { memcpy(pin, pg, n); });
        t("hipHostRegister + Unregister", [&] { hipHostRegister(pg, n, hipHostRegisterDefault); hipHostUnregister(pg); });
        hipHostRegister(pg, n, hipHostRegisterDefault);
        t("H2D registered hipMemcpyAsync", [&] { hipMemcpyAsync(d, pg, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); });
        hipHostUnregister(pg);
        t("malloc+free device", [&] { void *x; hipMalloc(&x, n); hipFree(x); });
        hipFree(d); free(pg); hipHostFree(pin);
    }
    return 0;
}
