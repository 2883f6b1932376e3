// Dev aid: a few deterministic calls through the C ABI with the results printed -- run once against libnmfx.so and once against libnmfx_asan.so to see
// where a host-sanitizer build starts to differ (tests/host_asan/fuzz_multi.cpp is the campaign; this is its microscope).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "nmfx.h"

static std::vector<double> rnd(size_t n, unsigned s) { std::vector<double> v(n); unsigned x = s * 2654435761u + 1; for (auto &e : v) { x = x * 1664525u + 1013904223u; e = ((x >> 8) + 1) / 16777217.0; } return v; }

int main() {
    if (nmfx_device_count() < 1) { printf("no device\n"); return 2; }
    {   // ReconstructFromDecomposition: upload, one GEMM, download
        const int m = 64, n = 96, K = 8;
        auto W = rnd((size_t)m * K, 1), H = rnd((size_t)K * n, 2);
        std::vector<double> Vh((size_t)m * n, -1.0);
        nmfx_status s = nmfx_reconstruct(m, n, K, 1, NMFX_F64, W.data(), H.data(), Vh.data(), 0);
        double err = 0;
        for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) { double r = 0; for (int k = 0; k < K; ++k) r += W[i + (size_t)m * k] * H[k + (size_t)K * j]; err = std::fmax(err, std::fabs(r - Vh[i + (size_t)m * j])); }
        printf("reconstruct rc %d max abs err %.3g (%s)\n", (int)s, err, nmfx_last_error());
    }
    {   // projfunc (float64 end to end)
        const int N = 200;
        auto sv = rnd(N, 3);
        std::vector<double> v(N, -1.0);
        int it = -1;
        nmfx_status s = nmfx_projfunc(N, 1, NMFX_F64, sv.data(), 8.0, 1.0, 1, v.data(), &it, 0);
        double l1 = 0, l2 = 0;
        for (double x : v) { l1 += std::fabs(x); l2 += x * x; }
        printf("projfunc rc %d iters %d L1 %.12g (want 8) L2 %.12g (want 1)\n", (int)s, it, l1, l2);
    }
    for (int path = 0; path <= 2; ++path)
        for (int div = 0; div < 2; ++div)
            for (int K : {8, 64}) {
                const int m = 128, n = 192, iters = 3;
                if (path == 2 && K == 8) continue;
                auto V = rnd((size_t)m * n, 4), W0 = rnd((size_t)m * K, 5), H0 = rnd((size_t)K * n, 6);
                std::vector<double> W((size_t)m * K), H((size_t)K * n), cost(iters + 1, -1.0);
                nmfx_problem p; memset(&p, 0, sizeof(p));
                p.m = m; p.n = n; p.K_total = K; p.T = 1; p.dtype = NMFX_F64; p.V = V.data(); p.W_init = W0.data(); p.H_init = H0.data();
                p.divergence = div; p.alpha = p.beta = 1; p.num_sources = 1; p.maxiter = iters; p.tolerance = 1e-300; p.path = path;
                nmfx_result r; memset(&r, 0, sizeof(r));
                r.W = W.data(); r.H = H.data(); r.cost = cost.data();
                nmfx_status s = nmfx_nmf(&p, &r);
                double sw = 0, sh = 0;
                for (double x : W) sw += x;
                for (double x : H) sh += x;
                printf("nmf path %d div %d K %d: rc %d cost %.9g %.9g %.9g sumW %.9g sumH %.9g (%s)\n", path, div, K, (int)s, cost[0], cost[1], cost[2], sw, sh, s ? nmfx_last_error() : "");
            }
    return 0;
}
