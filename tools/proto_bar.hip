// can the host write straight into device memory (large BAR)?  fine-grained device allocation, host store, kernel reads it
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_read(const unsigned long long* p, unsigned long long* out) { out[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// one wave polls p until it sees `want`, then stamps the clock
__global__ void k_poll(const unsigned long long* p, unsigned long long want, unsigned long long* out) {
    long long spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want && spins < 400000000LL) ++spins;
    out[0] = (unsigned long long)spins;
}
int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("isLargeBar %d canMapHostMemory %d\n", prop.isLargeBar, prop.canMapHostMemory);
    unsigned long long *d = nullptr, *out = nullptr, *hout = nullptr;
    hipError_t rc = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags finegrained rc=%d ptr=%p\n", (int)rc, (void*)d);
    hipMalloc((void**)&out, 64);
    hipHostMalloc((void**)&hout, 64);
    hipMemset(d, 0, 4096);
    hipDeviceSynchronize();
    if (!prop.isLargeBar) { printf("no large BAR: not touching the pointer from the host\n"); return 0; }
    hipPointerAttribute_t at;
    rc = hipPointerGetAttributes(&at, d);
    printf("attr rc=%d type=%d hostPointer=%p devicePointer=%p\n", (int)rc, (int)at.type, at.hostPointer, at.devicePointer);
    volatile unsigned long long* hp = (volatile unsigned long long*)d;
    hp[0] = 0x1234567811223344ull;       // host store into device memory
    __sync_synchronize();
    hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, 0, d, out);
    hipMemcpy(hout, out, 8, hipMemcpyDeviceToHost);
    printf("kernel read back %llx (host wrote 1234567811223344)\n", hout[0]);
    printf("host reads its own store: %llx\n", (unsigned long long)hp[0]);
    // latency: kernel polls, host writes after a delay
    for (int rep = 0; rep < 3; ++rep) {
        hp[1] = 0;
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_poll, dim3(1), dim3(64), 0, 0, d + 1, 77ull + rep, out);
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.002) {}
        auto t1 = std::chrono::steady_clock::now();
        hp[1] = 77ull + rep;
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("host store -> polling kernel done and synchronised: %.1f us\n", 1e6 * std::chrono::duration<double>(t2 - t1).count());
    }
    // the same through pinned host memory (what the relay does)
    unsigned long long* hpin = nullptr;
    hipHostMalloc((void**)&hpin, 4096);
    for (int rep = 0; rep < 3; ++rep) {
        hpin[1] = 0;
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_poll, dim3(1), dim3(64), 0, 0, hpin + 1, 77ull + rep, out);
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.002) {}
        auto t1 = std::chrono::steady_clock::now();
        __atomic_store_n(hpin + 1, 77ull + rep, __ATOMIC_RELEASE);
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("pinned host store -> polling kernel done and synchronised: %.1f us\n", 1e6 * std::chrono::duration<double>(t2 - t1).count());
    }
    return 0;
}
