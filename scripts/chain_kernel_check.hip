// Debug harness (not part of libnmfx): ONE instantiation of the fused kernel's chain functors (7: partial S stored; 8: last block, R = V./S + KL cost) launched on
// its own against a float64 host reference, to look at the asm first product in these kernels outside the engine (round 6's first asm form failed here -- hipcc re-homed the
// loop-carried accumulator tuple with v_mov copies right behind an asm MFMA).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -I include -I nmf_toolbox_amd/csrc [-DNMFX_G1_ASM=0] scripts/chain_kernel_check.hip -o scripts/chain_check_<variant>
//   ./scripts/chain_check_<variant>  -> per case: worst |error| of the stored array, where the wrong elements sit (tile column, row within the 128-row block, half)
#include "fused_kernel.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace nmfx {
thread_local char g_err[8];
void set_error(const char *, ...) {}
}  // namespace nmfx

template <int KB, int FUNC>
static int run_case(const char *name, long m, long n, int Ktot, int k0, bool with_sin, int nsplit) {
    using namespace nmfx;
    std::vector<float> W((size_t)m * Ktot), H((size_t)Ktot * n), V((size_t)m * n), Sin((size_t)m * n), out((size_t)m * n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return 0.05f + (float)((s >> 8) & 0xffff) / 65536.0f; };
    for (auto &x : W) x = rnd();
    for (auto &x : H) x = rnd();
    for (auto &x : V) x = rnd();
    for (auto &x : Sin) x = rnd() * 3.0f;
    float *dW, *dH, *dV, *dS, *dO;
    double *dC;
    CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&dV, V.size() * 4)); CK(hipMalloc(&dS, Sin.size() * 4)); CK(hipMalloc(&dO, out.size() * 4));
    CK(hipMalloc(&dC, sizeof(double) * 4096));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dS, Sin.data(), Sin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dO, 0xff, out.size() * 4));
    FusedParams f;
    memset(&f, 0, sizeof(f));
    f.X = dW + (size_t)m * k0; f.xs_r = 1; f.xs_k = m;
    f.Y = dH + k0; f.y_stride = Ktot;
    f.D = dV; f.ldd = m; f.R = m; f.Cn = n; f.K = KB; f.c_per_split = ((n / 64 + nsplit - 1) / nsplit) * 64;
    f.Sin = with_sin ? dS : nullptr;
    f.Rout = dO;
    f.cost_partials = FUNC == 8 ? dC : nullptr;
    auto kern = fused_kernel<KB, true, FUNC, false, 0, false, 1>;
    const size_t ldsb = sizeof(float) * 2 * 64 * (KB + 4);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(kern, dim3((unsigned)(m / 128), (unsigned)nsplit), dim3(256), ldsb, 0, f);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    // reference: S = Sin + W(:, k0:k0+KB) * H(k0:k0+KB, :) in double; functor 8 stores V ./ S
    double worst = 0;
    long nbad = 0, first = -1;
    long hist_half[2] = {0, 0}, hist_tile[64] = {0}, hist_wave[4] = {0};
    for (long j = 0; j < n; ++j)
        for (long i = 0; i < m; ++i) {
            double acc = with_sin ? (double)Sin[i + m * j] : 0.0;
            for (int k = 0; k < KB; ++k) acc += (double)W[i + m * (k0 + k)] * (double)H[(k0 + k) + (size_t)Ktot * j];
            const double ref = FUNC == 8 ? (double)V[i + m * j] / acc : acc;
            const double e = std::fabs((double)out[i + m * j] - ref) / std::fabs(ref);
            if (e > worst) worst = e;
            if (!(e < 1e-4)) {
                ++nbad;
                if (first < 0) first = i + m * j;
                hist_half[(j % 64) / 32]++; hist_tile[(j / 64) % 64]++; hist_wave[(i % 128) / 32]++;
            }
        }
    printf("%s: K_block %d functor %d, %ld x %ld, Sin %d, split %d: worst rel error %.3g, %ld elements off by > 1e-4", name, KB, FUNC, m, n, (int)with_sin, nsplit, worst, nbad);
    if (nbad) {
        printf(" (first at row %ld col %ld; by half of the tile: %ld / %ld; by wave: %ld %ld %ld %ld; by tile index:", first % m, first / m, hist_half[0], hist_half[1], hist_wave[0], hist_wave[1], hist_wave[2], hist_wave[3]);
        for (int t = 0; t < 8 && t < (int)(n / 64); ++t) printf(" %ld", hist_tile[t]);
        printf(")");
    }
    printf("\n");
    CK(hipFree(dW)); CK(hipFree(dH)); CK(hipFree(dV)); CK(hipFree(dS)); CK(hipFree(dO)); CK(hipFree(dC));
    return nbad ? 1 : 0;
}

int main() {
    int bad = 0;
    printf("NMFX_G1_ASM %d\n", NMFX_G1_ASM);
    bad += run_case<256, 7>("a", 256, 512, 512, 0, false, 1);
    bad += run_case<256, 7>("b", 256, 512, 512, 256, true, 1);
    bad += run_case<256, 7>("c", 256, 64, 512, 256, true, 1);      // one tile
    bad += run_case<256, 7>("d", 256, 128, 512, 256, true, 1);     // two tiles
    bad += run_case<256, 8>("e", 256, 512, 512, 256, true, 1);
    bad += run_case<160, 7>("f", 256, 512, 320, 160, true, 2);
    bad += run_case<128, 7>("g", 256, 512, 384, 128, true, 1);
    bad += run_case<224, 8>("h", 384, 640, 448, 224, true, 1);
    bad += run_case<192, 7>("i", 128, 192, 384, 0, false, 1);
    printf("%s\n", bad ? "FAILURES" : "all cases match the float64 reference");
    return 0;
}
