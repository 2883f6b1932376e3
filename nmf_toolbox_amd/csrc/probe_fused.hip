// DEV ONLY (not part of libnmfx): times ablated variants of the fused kernel to find where MFMA cycles are lost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. probe_fused.hip -o /tmp/probe_fused && /tmp/probe_fused
#include <cstdio>
#include <vector>

#include "fused_kernel.h"

namespace nmfx { void set_error(const char *, ...) {} }
using namespace nmfx;

template <int PROBE, int FUNC, bool DO_G2, bool D_RC = true>
static float run(const FusedParams &p, int nsplit, int reps) {
    constexpr int K = 256;
    const size_t ldsb = sizeof(float) * 2 * FT_C * (K + 4);
    auto kern = fused_kernel<K, D_RC, FUNC, DO_G2, 0, PROBE>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    dim3 grid((unsigned)(p.R / FT_ROWS), (unsigned)nsplit);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, 0, p);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const long m = 16384, n = 16384; const int K = 256, nsplit = 2;
    float *V, *W, *H, *out; double *cp;
    hipMalloc(&V, sizeof(float) * m * n); hipMalloc(&W, sizeof(float) * m * K); hipMalloc(&H, sizeof(float) * K * n);
    hipMalloc(&out, sizeof(float) * m * K * nsplit); hipMalloc(&cp, sizeof(double) * 4096);
    std::vector<float> h((size_t)m * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.25f + 0.5f * ((i * 2654435761u) % 1000) / 1000.0f;
    hipMemcpy(W, h.data(), sizeof(float) * m * K, hipMemcpyHostToDevice);
    hipMemcpy(H, h.data(), sizeof(float) * K * (size_t)n, hipMemcpyHostToDevice);
    for (long j = 0; j < n; j += (m * K) / m) hipMemcpy(V + j * m, h.data(), sizeof(float) * m * K, hipMemcpyHostToDevice);
    FusedParams p{};
    p.X = W; p.xs_r = 1; p.xs_k = m; p.Y = H; p.D = V; p.ldd = m; p.R = m; p.Cn = n; p.K = K; p.c_per_split = n / nsplit;
    p.out = out; p.slab_stride = m * (long)K; p.os_r = 1; p.os_k = m; p.cost_partials = cp;
    const double fl = 4.0 * m * n * K;
    auto rep = [&](const char *name, float ms, double flops) { printf("%-44s %8.3f ms  %7.1f TF  (%.1f%% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573); };
    rep("warm-up (clocks ramp: ignore)", run<0, 3, true>(p, nsplit, 5), fl);
    rep("KL, no cost (func 2)", run<0, 2, true>(p, nsplit, 5), fl);
    rep("no barrier/DMA after tile 0", run<1, 3, true>(p, nsplit, 5), fl);
    rep("no element map", run<2, 3, true>(p, nsplit, 5), fl);
        rep("no barrier/DMA, no emap, no V loads", run<7, 3, true>(p, nsplit, 5), fl);
    rep("full (KL + cost)", run<0, 3, true>(p, nsplit, 5), fl);
    rep("cost: log replaced by mul", run<8, 3, true>(p, nsplit, 5), fl);
    rep("cost: no (S-V) sum", run<16, 3, true>(p, nsplit, 5), fl);
    rep("cost: neither", run<24, 3, true>(p, nsplit, 5), fl);
    rep("second product only (func 0)", run<0, 0, true>(p, nsplit, 5), fl / 2);
    rep("second product only, no barrier/DMA", run<1, 0, true>(p, nsplit, 5), fl / 2);
    rep("second product only, no V loads", run<4, 0, true>(p, nsplit, 5), fl / 2);
    rep("second product only, no barrier/DMA/V", run<5, 0, true>(p, nsplit, 5), fl / 2);
    {
        FusedParams q = p;   // H-step form: stationary = columns of V, streamed = rows of W (the buffers are reused, the values do not matter)
        q.X = H; q.xs_r = K; q.xs_k = 1; q.Y = W; q.R = n; q.Cn = m; q.c_per_split = m / nsplit; q.os_r = K; q.os_k = 1; q.slab_stride = (long)K * n;
        rep("H form, second product only", run<0, 0, true, false>(q, nsplit, 5), fl / 2);
        rep("H form, second only, no barrier/DMA", run<1, 0, true, false>(q, nsplit, 5), fl / 2);
        rep("H form, second only, no V loads", run<4, 0, true, false>(q, nsplit, 5), fl / 2);
        rep("H form, KL (func 2)", run<0, 2, true, false>(q, nsplit, 5), fl);
        rep("H form, KL, no V loads", run<4, 2, true, false>(q, nsplit, 5), fl);
    }
    rep("cost-only pass", run<0, 3, false>(p, nsplit, 5), fl / 2);
    rep("cost-only, no barrier/emap/V", run<7, 3, false>(p, nsplit, 5), fl / 2);
    return 0;
}
