// Host-sanitizer campaign for the blocking C ABI (test infrastructure, not product).
//
// Built by nmf_toolbox_amd/build.py::build_sanitized() with -fsanitize=address,undefined on the HOST pass only (device code objects are the
// normal gfx950 ones: GPU ASan / xnack+ are not available on this pool) and linked against the equally instrumented libnmfx_asan.so.  It
// drives what scripts/fuzz_campaign_r3.py `multi_edge` drove when a rare host-heap corruption showed up in round 3
// (profiles/archive/r3_40_multi_edge_crash.md): nmf / cnmf / lnmf through nmfx_problem.n_gpus on awkward geometry at a high call rate, while other
// threads of the process churn the malloc heap the way the NumPy oracle's temporaries did.  Run with NMFX_NO_POOL=1 to put the per-call
// stream / event create + destroy of round 3 back.  Every multi-device result is compared with the one-device result of the same call
// (no oracle needed: the property is "sharding changes only the summation order").
//
//   fuzz_multi <seconds> <seed> [kinds: e = multi_edge, s = nmfsc on shards, x = error paths, 1 = one-device calls,
//                                 round 5: r = the RCCL backend (one shard: the 1-GPU box) against the plain call, c = cnmfsc / small nmfsc on one device (the
//                                 Gram-form W branch, the float64 nmfsc), p = nmf on path 1 (materialised V_hat, float64 W*(H*H'))]   e.g.  fuzz_multi 600 12 esx1rcp
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>
#include <pthread.h>
#include <signal.h>
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/common_interface_defs.h>
#define HAVE_SAN_STACK 1
#endif
#endif

#include "nmfx.h"

namespace {

std::atomic<bool> g_stop{false};
std::atomic<long> g_progress{0};     // library calls finished so far (watchdog)
char g_current[256] = "";            // the call in flight
pthread_t g_main;

// a call that does not come back: say which one it is and where the calling thread stands, then give up (exit code 3)
void on_usr1(int) {
    fprintf(stderr, "WATCHDOG: main thread stuck in: %s\n", g_current);
#ifdef HAVE_SAN_STACK
    __sanitizer_print_stack_trace();
#endif
    fflush(stdout); fflush(stderr);
    _exit(3);
}
void watchdog() {
    long last = -1;
    int quiet = 0;
    while (!g_stop.load()) {
        std::this_thread::sleep_for(std::chrono::seconds(5));
        const long now = g_progress.load();
        quiet = now == last ? quiet + 1 : 0;
        last = now;
        if (quiet >= 12) { pthread_kill(g_main, SIGUSR1); std::this_thread::sleep_for(std::chrono::seconds(30)); _exit(4); }   // 60 s without a finished call
    }
}

// what the float64 NumPy oracle did to the heap between two library calls: many short-lived blocks from a few bytes to a few MiB, written to
void churn(unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<std::vector<double>> live;
    while (!g_stop.load(std::memory_order_relaxed)) {
        const unsigned r = rng();
        const size_t n = (r & 7) == 0 ? (size_t)(rng() % (1u << 19)) + 1 : (size_t)(rng() % 4096) + 1;
        live.emplace_back(n, 1.0);
        double s = 0;
        for (size_t i = 0; i < n; i += 64) s += live.back()[i];
        if (s < 0) abort();
        if (live.size() > 24) { live.erase(live.begin() + (long)(rng() % live.size())); }
        if ((r & 1023) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

struct Case {
    int alg;   // 0 nmf, 1 cnmf, 2 lnmf, 4 nmfsc, 5 cnmfsc
    long m, n;
    int K, T, div, N, it;
    int backend = 0, path = 0;   // nmfx_problem.multi_backend (2 = RCCL), nmfx_problem.path (1 = two-operand GEMMs, 2 = fused passes by name)
    double lamW, lamH, sH, sW;
    double tol;
};

struct Out {
    std::vector<double> W, H, cost;
    int cost_len = 0;
    nmfx_status rc = NMFX_OK;
    std::string err;
};

void fill(std::vector<double> &v, std::mt19937 &rng, double lo = 2.220446049250313e-16) {
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (double &x : v) { x = u(rng); if (x < lo) x = lo; }
}

Out run(const Case &c, const std::vector<double> &V, const std::vector<double> &W0, const std::vector<double> &H0, int N, const int32_t *ids, bool f32) {
    Out o;
    const size_t mK = (size_t)c.m * c.K * c.T, Kn = (size_t)c.K * c.n;
    std::vector<float> Vf, Wf, Hf, Wo, Ho;
    o.W.assign(mK, 0.0); o.H.assign(Kn, 0.0); o.cost.assign((size_t)c.it + 1, 0.0);
    nmfx_problem p;
    memset(&p, 0, sizeof(p));
    p.m = c.m; p.n = c.n; p.K_total = c.K; p.T = c.T; p.dtype = f32 ? NMFX_F32 : NMFX_F64;
    if (f32) {
        Vf.assign(V.begin(), V.end()); Wf.assign(W0.begin(), W0.end()); Hf.assign(H0.begin(), H0.end());
        Wo.assign(mK, 0.f); Ho.assign(Kn, 0.f);
        p.V = Vf.data(); p.W_init = Wf.data(); p.H_init = Hf.data();
    } else { p.V = V.data(); p.W_init = W0.data(); p.H_init = H0.data(); }
    p.divergence = c.div; p.alpha = 1; p.beta = 1; p.num_sources = 1;
    const double lw = c.lamW, lh = c.lamH;
    if (c.lamW > 0) { p.W_sparsity = &lw; p.H_sparsity = &lh; }
    p.maxiter = c.it; p.tolerance = c.tol; p.device = 0;
    p.sc_W_sparsity = c.sW; p.sc_H_sparsity = c.sH;
    p.n_gpus = N; p.device_ids = ids;
    p.multi_backend = c.backend; p.path = c.path;
    snprintf(g_current, sizeof(g_current), "alg %d %ldx%ld K %d T %d div %d N %d it %d f32 %d lamW %g tol %g sW %g sH %g", c.alg, (long)c.m, (long)c.n, c.K, c.T, c.div, N, c.it,
             (int)f32, c.lamW, c.tol, c.sW, c.sH);
    nmfx_result r;
    memset(&r, 0, sizeof(r));
    r.W = f32 ? (void *)Wo.data() : (void *)o.W.data(); r.H = f32 ? (void *)Ho.data() : (void *)o.H.data(); r.cost = o.cost.data();
    std::vector<int32_t> tH((size_t)c.it, 0), tW((size_t)c.it * (size_t)(c.T > 1 ? c.T : 1), 0);   // (cnmfsc: one W search per time slice)
    r.tries_H = tH.data(); r.tries_W = tW.data();
    switch (c.alg) {
    case 0: o.rc = nmfx_nmf(&p, &r); break;
    case 1: o.rc = nmfx_cnmf(&p, &r); break;
    case 2: o.rc = nmfx_lnmf(&p, &r); break;
    case 5: o.rc = nmfx_cnmfsc(&p, &r); break;
    default: o.rc = nmfx_nmfsc(&p, &r); break;
    }
    g_progress.fetch_add(1);
    if (o.rc != NMFX_OK) o.err = nmfx_last_error();
    o.cost_len = r.cost_len;
    if (f32) { o.W.assign(Wo.begin(), Wo.end()); o.H.assign(Ho.begin(), Ho.end()); }
    return o;
}

double rel(const std::vector<double> &a, const std::vector<double> &b) {
    double d = 0, n = 0;
    for (size_t i = 0; i < a.size(); ++i) { d += (a[i] - b[i]) * (a[i] - b[i]); n += b[i] * b[i]; }
    return std::sqrt(d / (n > 0 ? n : 1e-300));
}

}  // namespace

int main(int argc, char **argv) {
    const double budget = argc > 1 ? atof(argv[1]) : 60.0;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 12u;
    const std::string kinds = argc > 3 ? argv[3] : "esx1";
    if (nmfx_device_count() < 1) { fprintf(stderr, "fuzz_multi: no MI355X visible (%s)\n", nmfx_last_error()); return 2; }
    g_main = pthread_self();
    signal(SIGUSR1, on_usr1);
    std::vector<std::thread> bg;
    for (unsigned t = 0; t < 4; ++t) bg.emplace_back(churn, seed * 131 + t);
    if (!getenv("FUZZ_NO_WATCHDOG")) bg.emplace_back(watchdog);
    std::mt19937 rng(seed);
    auto ri = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo)); };   // [lo, hi)
    const auto t0 = std::chrono::steady_clock::now();
    long ncase = 0, ncalls = 0, nbad = 0, nerr_expected = 0;
    double worst = 0;
    const int32_t ids0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget) {
        const char kind = kinds[(size_t)ri(0, (int)kinds.size())];
        Case c{};
        c.N = ri(2, 9); c.tol = 1e-300; c.T = 1;
        if (kind == 's') {   // nmfsc: one host thread per shard, the threaded peer all-reduce (fused kernels: K a multiple of 32 after padding, n_local >= 64)
            c.alg = 4; c.N = ri(2, 5); c.m = ri(64, 200); c.K = ri(3, 70); c.n = (long)c.N * 64 + ri(0, 300); c.it = ri(1, 4); c.div = NMFX_DIV_EUCLIDEAN;
            const int br = ri(0, 3);
            c.sH = br != 1 ? 0.3 + 0.4 * (ri(0, 100) / 100.0) : 0.0; c.sW = br != 0 ? 0.3 + 0.4 * (ri(0, 100) / 100.0) : 0.0;
        } else {
            c.alg = ri(0, 3);
            c.T = c.alg == 1 ? ri(2, 6) : 1;
            c.m = ri(8, 300);
            c.n = ri(c.N * std::max(c.T, 2), 700);
            c.K = ri(2, 70);
            c.div = c.alg == 2 ? NMFX_DIV_KL : (c.alg == 0 ? ri(0, 3) : ri(0, 2));
            c.it = ri(1, 7);
            if (c.alg == 0 && ri(0, 10) < 4) { c.lamW = 0.1 * ri(0, 100) / 100.0 + 1e-3; c.lamH = 0.1 * ri(0, 100) / 100.0; }
            if (ri(0, 10) == 0) c.tol = 1e-1;   // the stop rule with its W backup / restore
        }
        std::vector<double> V((size_t)c.m * c.n), W0((size_t)c.m * c.K * c.T), H0((size_t)c.K * c.n);
        fill(V, rng); fill(W0, rng); fill(H0, rng);
        if (c.alg == 2) for (int k = 0; k < c.K; ++k) { double s = 0; for (long i = 0; i < c.m; ++i) s += W0[(size_t)k * c.m + i]; for (long i = 0; i < c.m; ++i) W0[(size_t)k * c.m + i] /= s; }
        const bool f32 = ri(0, 4) == 0;
        ++ncase;
        if (ncase % 200 == 0) { printf("... %ld cases, %ld calls, bad %ld\n", ncase, ncalls, nbad); fflush(stdout); }
        if (kind == 'x') {   // error paths: every one of them must leave nothing in flight into host memory (ASan / the churn threads would see it)
            const int which = ri(0, 5);
            int32_t bad_ids[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            Case e = c;
            Out o;
            if (which == 0) { bad_ids[c.N - 1] = 7; o = run(e, V, W0, H0, c.N, bad_ids, f32); }               // no such device: fails after the first shards are set up
            else if (which == 1) { o = run(e, V, W0, H0, 99, ids0, f32); }          // too many devices
            else if (which == 2) { e.alg = 1; e.K = 2; e.div = NMFX_DIV_KL; e.T = (int)(c.n / c.N) + 3; W0.assign((size_t)e.m * e.K * e.T, 0.5); H0.assign((size_t)e.K * e.n, 0.5); o = run(e, V, W0, H0, c.N, ids0, f32); }   // context longer than a shard
            else if (which == 3) { e.alg = 4; e.sH = 0.5; e.N = 2; V[V.size() / 2] = -1.0; o = run(e, V, W0, H0, 2, ids0, f32); }   // "Negative values in data!"
            else { e.alg = 4; e.sH = 0.5; e.K = 300; e.N = 2; W0.assign((size_t)e.m * 300, 0.5); H0.assign((size_t)300 * e.n, 0.5); o = run(e, V, W0, H0, 2, ids0, f32); }   // nmfsc on shards above K = 256
            ++ncalls;
            if (o.rc == NMFX_OK) { ++nbad; printf("BAD: error case %d returned NMFX_OK\n", which); }
            else ++nerr_expected;
            continue;
        }
        if (kind == 'c') {   // cnmfsc (fused passes where the shape allows them, else the default) or a small nmfsc (float64 end to end): no reference run, ASan is the judge
            Case e = c;
            const bool conv = ri(0, 3) != 0;
            e.div = NMFX_DIV_EUCLIDEAN; e.lamW = e.lamH = 0; e.it = ri(1, 4); e.tol = 1e-300;
            if (conv) {
                static const int kt[][2] = {{32, 4}, {64, 2}, {32, 3}, {64, 3}, {6, 2}, {20, 3}};
                const int w = ri(0, 6);
                e.alg = 5; e.K = kt[w][0]; e.T = kt[w][1]; e.m = 4 * ri(16, 80); e.n = ri(64, 500);
                e.path = (w < 4 && ri(0, 2)) ? 2 : 0;
                e.sH = ri(0, 2) ? 0.4 : 0.0; e.sW = 0.0;
            } else { e.alg = 4; e.T = 1; e.K = ri(2, 40); e.m = ri(16, 120); e.n = ri(64, 300); e.sH = ri(0, 2) ? 0.5 : 0.0; e.sW = e.sH == 0.0 ? 0.4 : 0.0; }
            std::vector<double> V2((size_t)e.m * e.n), W2((size_t)e.m * e.K * e.T), H2((size_t)e.K * e.n);
            fill(V2, rng); fill(W2, rng); fill(H2, rng);
            Out a = run(e, V2, W2, H2, 1, ids0, f32);
            ++ncalls;
            if (a.rc != NMFX_OK) { ++nbad; printf("BAD: rc %d (%s) alg %d %ldx%ld K %d T %d path %d\n", (int)a.rc, a.err.c_str(), e.alg, e.m, e.n, e.K, e.T, e.path); continue; }
            for (double x : a.W) if (!std::isfinite(x)) { ++nbad; printf("BAD: non-finite W: alg %d %ldx%ld K %d T %d path %d\n", e.alg, e.m, e.n, e.K, e.T, e.path); break; }
            continue;
        }
        if (kind == 'r' || kind == 'p') {   // one device: the RCCL branch of the sharded driver with ONE shard / the materialised path, against the plain call
            Case e = c;
            if (kind == 'r') e.backend = 2; else { e.alg = 0; e.T = 1; e.div = NMFX_DIV_EUCLIDEAN; e.path = 1; W0.resize((size_t)e.m * e.K); }
            Out a = run(e, V, W0, H0, 1, ids0, f32);
            Case e0 = e; e0.backend = 0; e0.path = 0;
            Out b = run(e0, V, W0, H0, 1, ids0, f32);
            ncalls += 2;
            if (a.rc != NMFX_OK || b.rc != NMFX_OK) { ++nbad; printf("BAD: rc %d / %d (%s) kind %c alg %d %ldx%ld K %d T %d div %d\n", (int)a.rc, (int)b.rc, a.err.c_str(), kind, e.alg, e.m, e.n, e.K, e.T, e.div); continue; }
            const double eW = rel(a.W, b.W), eH = rel(a.H, b.H);
            worst = std::max(worst, std::max(eW, eH));
            if (!(eW < 2e-5 && eH < 2e-5) || a.cost_len != b.cost_len) { ++nbad; printf("BAD: kind %c vs the plain call W %.3g H %.3g len %d/%d alg %d %ldx%ld K %d T %d div %d\n", kind, eW, eH, a.cost_len, b.cost_len, e.alg, e.m, e.n, e.K, e.T, e.div); }
            continue;
        }
        const int N = kind == '1' ? 1 : c.N;
        Out a = run(c, V, W0, H0, N, ids0, f32);
        ++ncalls;
        if (a.rc != NMFX_OK) { ++nbad; printf("BAD: rc %d (%s) alg %d %ldx%ld K %d T %d div %d N %d it %d\n", (int)a.rc, a.err.c_str(), c.alg, c.m, c.n, c.K, c.T, c.div, N, c.it); continue; }
        if (N > 1 && c.alg != 4) {   // (nmfsc's line searches amplify summation-order differences: its shard parity lives in tests/test_gpu_sharded.py)
            Out b = run(c, V, W0, H0, 1, ids0, f32);
            ++ncalls;
            const double eW = rel(a.W, b.W), eH = rel(a.H, b.H);
            worst = std::max(worst, std::max(eW, eH));
            if (!(eW < 2e-5 && eH < 2e-5) || a.cost_len != b.cost_len) {
                ++nbad;
                printf("BAD: shards vs one device W %.3g H %.3g len %d/%d alg %d %ldx%ld K %d T %d div %d N %d it %d\n", eW, eH, a.cost_len, b.cost_len, c.alg, c.m, c.n, c.K, c.T, c.div, N, c.it);
            }
        }
        for (double x : a.W) if (!std::isfinite(x)) {
            ++nbad;
            printf("BAD: non-finite W: alg %d %ldx%ld K %d T %d div %d N %d it %d f32 %d lam %g tol %g cost0 %g\n", c.alg, c.m, c.n, c.K, c.T, c.div, N, c.it, (int)f32, c.lamW, c.tol, a.cost[0]);
            break;
        }
    }
    g_stop.store(true);
    for (auto &t : bg) t.join();
    printf("fuzz_multi seed %u kinds %s: %ld cases, %ld library calls, %ld expected errors, worst shard-vs-one-device deviation %.3g, bad %ld, pool %s\n", seed, kinds.c_str(), ncase,
           ncalls, nerr_expected, worst, nbad, (getenv("NMFX_NO_POOL") && getenv("NMFX_NO_POOL")[0] == '1') ? "OFF" : "on");
    fflush(stdout);
    // (no static destructors: the HSA runtime's own teardown trips an internal CHECK of the ROCm ASan runtime at exit -- "dev_runtime_unloaded_" -- which has
    // nothing to do with the calls above and would turn every clean run into exit code 1)
    _exit(nbad ? 1 : 0);
}
