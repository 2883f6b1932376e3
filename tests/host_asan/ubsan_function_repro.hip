// hipcc 7.2: with -fsanitize=undefined (which includes -fsanitize=function) the launch through `auto kern = kern_t<...>` below is silently dropped --
// "template via auto: b[5] = 0 (want 31)"; with -fno-sanitize=function it prints 31.  Build: hipcc --offload-arch=gfx950 -O3 -fPIC -fsanitize=undefined -fno-gpu-sanitize -c; link into a
// shared object and call run_k(1000) from an executable built with the same -fsanitize (profiles/archive/r4_01_host_asan.md).
#include <hip/hip_runtime.h>
#include <cstdio>
namespace nmfx {
struct P { const float *a; float *b; long n; float s; };
template <int MUL, bool FLAG> __global__ __launch_bounds__(256) void kern_t(const P p) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < p.n) p.b[i] = p.a[i] * p.s * MUL + (FLAG ? 1.0f : 0.0f); }
template <int MUL, bool FLAG> static int launch_t(hipStream_t st, const P &p) {
    auto kern = kern_t<MUL, FLAG>;
    static bool attr_done = false;
    if (!attr_done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 1024) != hipSuccess) return 3; attr_done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 1024, st, p);
    hipError_t e = hipGetLastError();
    printf("launch<%d,%d>: %s\n", MUL, (int)FLAG, hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
template <int MUL> __global__ void kern_d(float *b, long n, float s) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = s * MUL; }
}
extern "C" int run_k(int n) {
    using namespace nmfx;
    float *a, *b;
    (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    float *h = new float[n];
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    (void)hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice);
    (void)hipMemset(b, 0, n * 4);
    P p{a, b, n, 2.0f};
    launch_t<3, true>(nullptr, p);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, b, n * 4, hipMemcpyDeviceToHost);
    printf("template via auto: b[5] = %g (want 31)\n", h[5]);
    hipLaunchKernelGGL(kern_d<7>, dim3((n + 255) / 256), dim3(256), 0, 0, b, (long)n, 1.0f);
    printf("direct template launch: %s\n", hipGetErrorString(hipGetLastError()));
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, b, n * 4, hipMemcpyDeviceToHost);
    printf("template direct: b[5] = %g (want 7)\n", h[5]);
    delete[] h;
    return 0;
}
