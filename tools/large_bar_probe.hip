// can the host store straight into fine-grained DEVICE memory on this box (large BAR)?  and how long does a small host write + kernel read take
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void sum_kernel(const float* in, int n, float* out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    atomicAdd(out, s);
}
int main() {
    float* dev = nullptr; float* out = nullptr;
    CHECK(hipExtMallocWithFlags((void**)&dev, 1 << 16, hipDeviceMallocFinegrained));
    CHECK(hipMalloc(&out, 4));
    hipPointerAttribute_t at;
    CHECK(hipPointerGetAttributes(&at, dev));
    printf("fine-grained device allocation: type %d device %d host ptr %p dev ptr %p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
    int large_bar = -1;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    fflush(stdout);
    if (large_bar != 1) { printf("no large BAR: host stores into device memory not possible\n"); return 0; }
    const int n = 2048;
    float host[n];
    for (int i = 0; i < n; ++i) host[i] = 1.0f;
    printf("about to store from the host...\n"); fflush(stdout);
    memcpy(dev, host, sizeof(host));          // faults if the mapping is not host-accessible
    printf("stored\n"); fflush(stdout);
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(out, 0, 4)); CHECK(hipDeviceSynchronize());
        for (int i = 0; i < n; ++i) host[i] = (float)(rep + 1);
        auto t0 = std::chrono::steady_clock::now();
        memcpy(dev, host, sizeof(host));
        __builtin_ia32_sfence();
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, dev, n, out);
        CHECK(hipDeviceSynchronize());
        auto t1 = std::chrono::steady_clock::now();
        float r = 0; CHECK(hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost));
        printf("rep %d: sum %.0f (expected %.0f)  host write + launch + sync %.1f us\n", rep, r, (double)n * (rep + 1), std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    return 0;
}
