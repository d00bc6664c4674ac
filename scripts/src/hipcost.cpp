// HIP API costs on the target box (average of 200 calls):  hipcc --offload-arch=gfx950 -O2 -w -o scripts/bin/hipcost scripts/src/hipcost.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
template <class F> double ms(F f, int n) { auto t0 = std::chrono::steady_clock::now(); for (int i = 0; i < n; ++i) f(i); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n; }
int main() {
    (void)hipSetDevice(0); void* w; hipMalloc(&w, 1024); const int N = 200;
    std::vector<hipStream_t> st(N); std::vector<void*> p(N), q(N); std::vector<hipEvent_t> ev(N);
    printf("hipStreamCreateWithFlags %.3f ms\n", ms([&](int i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); }, N));
    printf("  first use (memcpyAsync+sync) %.3f ms\n", ms([&](int i) { int x = 1; hipMemcpyAsync(w, &x, 4, hipMemcpyHostToDevice, st[i]); hipStreamSynchronize(st[i]); }, N));
    printf("hipStreamDestroy %.3f ms\n", ms([&](int i) { hipStreamDestroy(st[i]); }, N));
    printf("hipMalloc 512K %.3f ms\n", ms([&](int i) { hipMalloc(&p[i], 512 << 10); }, N));
    printf("hipFree 512K %.3f ms\n", ms([&](int i) { hipFree(p[i]); }, N));
    printf("hipMalloc 8K x1 %.3f ms\n", ms([&](int i) { hipMalloc(&p[i], 8 << 10); }, N));
    printf("hipFree 8K %.3f ms\n", ms([&](int i) { hipFree(p[i]); }, N));
    printf("hipHostMalloc 512B %.3f ms\n", ms([&](int i) { hipHostMalloc(&q[i], 512, hipHostMallocDefault); }, N));
    printf("hipHostFree %.3f ms\n", ms([&](int i) { hipHostFree(q[i]); }, N));
    printf("hipEventCreate %.3f ms\n", ms([&](int i) { hipEventCreate(&ev[i]); }, N));
    std::vector<char> host(100 << 10);
    (void)hipMalloc(&p[0], 512 << 10);
    printf("hipMemcpy H2D 8K sync %.3f ms\n", ms([&](int i) { hipMemcpy(p[0], host.data(), 8 << 10, hipMemcpyHostToDevice); }, N));
    printf("hipMemset 8K %.3f ms\n", ms([&](int i) { hipMemset(p[0], 0, 8 << 10); }, N));
    return 0;
}
