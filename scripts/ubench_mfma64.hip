// Rate of v_mfma_f64_16x16x4_f64 on gfx950, alone: NACC independent accumulators per wave, NWAVE waves per SIMD, every CU busy -> TFLOP/s.  What gemm64.hip is priced against.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma64.hip -o scripts/ubench_mfma64 && ./scripts/ubench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu) {
    double *out;
    const int grid = 256 * wg_per_cu, iters = 2000;
    hipMalloc(&out, sizeof(double) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0, 1e-9);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 2048.0;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * NACC * wg_per_cu);   // cycles per MFMA per SIMD at 2.4 GHz
    printf("{\"acc_per_wave\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"cycles_per_mfma_at_2p4GHz\": %.1f}\n", NACC, wg_per_cu, ms, flops / ms / 1e9, cyc);
    hipFree(out);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<8>(2); run<4>(4);
    return 0;
}
