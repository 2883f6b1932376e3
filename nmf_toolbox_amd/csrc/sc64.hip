// nmfsc.m:57-245 in float64 end to end, for the problem sizes at which results get compared with MATLAB by eye (m*n*K <= 2^27 multiply-adds per product,
// one GPU): V, W, H, V_hat, every gradient, the Hoyer projection and the objective are doubles on the device, so the accept tests of the line searches
// (nmfsc.m:164, :215) are decided on the same numbers the reference decides them on.  With fp32 storage of W a CONVERGED line search -- its objective moving by
// 1e-10 relative per try -- took a different number of tries than the reference (round 4: K = 3 with H fixed, profiles/archive/r4_19); parity here is 1e-12 and the
// try counts are the reference's.  Everything is VALU work (one thread per output element): these problems are a few hundred microseconds per product either way.
#include "api_common.h"

using namespace nmfx;

namespace {

constexpr double EPS64 = 2.220446049250313e-16;

__device__ __forceinline__ double bsum(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// V_hat = W*H (RFD.m:30) with H given transposed (HT: n x K) and, per workgroup, sum (V - V_hat).^2 (nmfsc.m:139,161,197,212,238)
__global__ __launch_bounds__(256) void sc64_recon_kernel(const double *__restrict__ W, const double *__restrict__ HT, long m, long n, int K, const double *__restrict__ V,
                                                          double *__restrict__ Vh, double *__restrict__ partials) {
    __shared__ double red[4];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    double d2 = 0.0;
    if (idx < m * n) {
        const long i = idx % m, j = idx / m;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s = fma(W[i + m * k], HT[j + n * k], s);
        Vh[idx] = s;
        const double d = V[idx] - s;
        d2 = d * d;
    }
    d2 = bsum(d2, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = d2;
}
__global__ __launch_bounds__(256) void sc64_sum_kernel(const double *partials, int count, double scale, double *out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += partials[i];
    s = bsum(s, red);
    if (threadIdx.x == 0) *out = scale * s;
}
// negT(j, k) = sum_i V(i, j) W(i, k), posT(j, k) = sum_i V_hat(i, j) W(i, k): (W'*V)' and (W'*V_hat)' (nmfsc.m:144-145), n x K
__global__ __launch_bounds__(256) void sc64_hterms_kernel(const double *__restrict__ V, const double *__restrict__ Vh, const double *__restrict__ W, long m, long n, int K,
                                                           double *__restrict__ negT, double *__restrict__ posT) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * K) return;
    const long j = idx % n;
    const int k = (int)(idx / n);
    const double *v = V + m * j, *vh = Vh + m * j, *w = W + m * k;
    double a = 0.0, b = 0.0;
    for (long i = 0; i < m; ++i) { a = fma(v[i], w[i], a); b = fma(vh[i], w[i], b); }
    negT[idx] = a; posT[idx] = b;
}
// neg(i, k) = sum_j V(i, j) H(k, j), pos likewise with V_hat: V*H' and V_hat*H' (nmfsc.m:194-195), m x K
__global__ __launch_bounds__(256) void sc64_wterms_kernel(const double *__restrict__ V, const double *__restrict__ Vh, const double *__restrict__ HT, long m, long n, int K,
                                                           double *__restrict__ neg, double *__restrict__ pos) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * K) return;
    const long i = idx % m;
    const int k = (int)(idx / m);
    const double *h = HT + n * k;
    double a = 0.0, b = 0.0;
    for (long j = 0; j < n; ++j) { a = fma(V[i + m * j], h[j], a); b = fma(Vh[i + m * j], h[j], b); }
    neg[idx] = a; pos[idx] = b;
}
__global__ void sc64_step_kernel(const double *X, const double *neg, const double *pos, double mu, long count, double *out) {   // X - mu*(pos - neg)   nmfsc.m:148,154 / :200,205
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < count) out[idx] = X[idx] - mu * (pos[idx] - neg[idx]);
}
__global__ void sc64_mu_kernel(double *X, const double *neg, const double *pos, long count) {   // X .* (neg ./ max(pos, eps))   nmfsc.m:182,232
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < count) X[idx] = X[idx] * (neg[idx] / fmax(pos[idx], EPS64));
}
__global__ void sc64_rescale_kernel(double *HT, long n, double *W, long m, int K, const double *ss) {   // nmfsc.m:185-187: rows of H to unit norm, columns of W take the norms
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n * K) { const double nrm = sqrt(ss[idx / n]); HT[idx] = (1.0 / nrm) * HT[idx]; }
    if (idx < m * K) W[idx] = W[idx] * sqrt(ss[idx / m]);
}
__global__ void sc64_scale_kernel(double *X, long count, double div) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < count) X[idx] = X[idx] / div;
}
__global__ __launch_bounds__(256) void sc64_transpose_kernel(const double *in, long rows, long cols, double *out) {   // out (cols x rows)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < rows * cols) { const long r = idx % rows, c = idx / rows; out[c + cols * r] = in[idx]; }
}
__global__ void sc64_widen_kernel(const float *in, double *out, long count) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < count) out[idx] = (double)in[idx];
}
__global__ void sc64_narrow_kernel(const double *in, float *out, long count) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < count) out[idx] = (float)in[idx];
}
inline dim3 g1(long count) { return dim3((unsigned)((count + 255) / 256)); }

}  // namespace

namespace nmfx {

bool nmfsc_f64_eligible(const nmfx_problem *p) {
    return p && p->path == 0 && p->T == 1 && p->n_gpus <= 1 && (double)p->m * (double)p->n * (double)p->K_total <= (double)(1 << 27);
}

nmfx_status run_nmfsc_f64(const nmfx_problem *p, nmfx_result *r) {
    TRY(validate_problem(p, r, true));
    const long m = p->m, n = p->n;
    const int K = p->K_total;
    const size_t mn = (size_t)m * n, mK = (size_t)m * K, Kn = (size_t)K * n;
    double vmin = INFINITY, vmax = -INFINITY;   // nmfsc.m:57-62
    host_minmax(p->V, p->dtype, mn, &vmin, &vmax);
    if (vmin < 0) { set_error("Negative values in data!"); return NMFX_ERR_NEGATIVE; }
    DeviceGuard dg_;
    TRY(check_device(p->device));
    // an own (pooled, non-blocking) stream: the legacy NULL stream would synchronise implicitly with every other blocking stream of the process, other threads'
    // engines included.  The measurement hooks of the nmfsc calls (nmfx_last_call_timing, nmfx_sc_iteration_seconds, nmfx_nmfsc_profile_read) describe THIS
    // call afterwards, not an earlier one: timing and iteration times are filled below, the per-tag profile is emptied (this path has no tagged launch groups)
    struct PooledStream {
        int dev; hipStream_t st = nullptr;
        ~PooledStream() { if (st) { (void)hipStreamSynchronize(st); unpool_stream(dev, st); } }
    } ps{p->device};
    TRY(pool_stream(p->device, &ps.st));
    hipStream_t st = ps.st;
    IoStats &io = io_stats();
    io = IoStats{};
    sc_hooks_reset();
    const auto t_in = std::chrono::steady_clock::now();
    double sW = p->sc_W_sparsity, sH = p->sc_H_sparsity, L1a = 0, L1s = 0;
    if (sW > 0) { if (sW > 1) sW = 1; L1a = std::sqrt((double)m) - (std::sqrt((double)m) - 1) * sW; }   // nmfsc.m:89-93
    if (sH > 0) { if (sH > 1) sH = 1; L1s = std::sqrt((double)n) - (std::sqrt((double)n) - 1) * sH; }   // nmfsc.m:102-106
    const bool fixW = p->W_fixed && p->W_fixed[0], fixH = p->H_fixed && p->H_fixed[0];
    DevBuf V, Vh, W, Wn, HT, HnT, Hk, negW, posW, negH, posH, parts, scal, stage;
    const int nparts = (int)((mn + 255) / 256);
    TRY(V.alloc(mn * 8)); TRY(Vh.alloc(mn * 8)); TRY(W.alloc(mK * 8)); TRY(Wn.alloc(mK * 8)); TRY(HT.alloc(Kn * 8)); TRY(HnT.alloc(Kn * 8)); TRY(Hk.alloc(Kn * 8));
    TRY(negW.alloc(mK * 8)); TRY(posW.alloc(mK * 8)); TRY(negH.alloc(Kn * 8)); TRY(posH.alloc(Kn * 8)); TRY(parts.alloc(sizeof(double) * nparts)); TRY(scal.alloc(64 + sizeof(double) * K));
    StreamDrain drain_(st);
    // host -> device as doubles (float32 host arrays are widened on the device)
    auto ingest = [&](const void *host, size_t count, double *dst) -> nmfx_status {
        if (p->dtype == NMFX_F64) { NMFX_HIP(hipMemcpyAsync(dst, host, count * 8, hipMemcpyHostToDevice, st)); return NMFX_OK; }
        if (!stage.p) TRY(stage.alloc(std::max(mn, std::max(mK, Kn)) * 4));
        NMFX_HIP(hipMemcpyAsync(stage.p, host, count * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sc64_widen_kernel, g1((long)count), dim3(256), 0, st, stage.as<float>(), dst, (long)count);
        NMFX_HIP(hipGetLastError());
        NMFX_HIP(hipStreamSynchronize(st));   // the staging buffer is reused
        return NMFX_OK;
    };
    TRY(ingest(p->V, mn, V.as<double>()));
    hipLaunchKernelGGL(sc64_scale_kernel, g1((long)mn), dim3(256), 0, st, V.as<double>(), (long)mn, vmax);   // V = V / max(V(:))   nmfsc.m:62
    TRY(ingest(p->W_init, mK, W.as<double>()));
    TRY(ingest(p->H_init, Kn, Hk.as<double>()));
    hipLaunchKernelGGL(sc64_transpose_kernel, g1((long)Kn), dim3(256), 0, st, Hk.as<double>(), (long)K, n, HT.as<double>());   // rows of H = columns of HT
    NMFX_HIP(hipGetLastError());
    double *Wd = W.as<double>(), *Wnew = Wn.as<double>(), *HTd = HT.as<double>(), *HnewT = HnT.as<double>();
    if (sW > 0) TRY(projfunc_cols_f64(st, Wd, m, K, L1a, 1.0, 1, nullptr));    // nmfsc.m:94-96
    if (sH > 0) TRY(projfunc_cols_f64(st, HTd, n, K, L1s, 1.0, 1, nullptr));   // nmfsc.m:107-109
    struct Pinned { double *p = nullptr; ~Pinned() { if (p) (void)hipHostFree(p); } } pin;   // (declared after drain_: freed before the stream is drained? no -- destructors run in reverse order, and every read of it is followed by a synchronise)
    NMFX_HIP(hipHostMalloc(reinterpret_cast<void **>(&pin.p), sizeof(double), hipHostMallocDefault));
    double *hobj = pin.p;
    // V_hat = Wx*Hx and 0.5*||V - V_hat||^2, read by the host (the line searches branch on it)
    auto recon_obj = [&](const double *Wx, const double *HxT, double *obj) -> nmfx_status {
        hipLaunchKernelGGL(sc64_recon_kernel, dim3(nparts), dim3(256), 0, st, Wx, HxT, m, n, K, V.as<double>(), Vh.as<double>(), parts.as<double>());
        hipLaunchKernelGGL(sc64_sum_kernel, dim3(1), dim3(256), 0, st, parts.as<double>(), nparts, 0.5, scal.as<double>());
        NMFX_HIP(hipGetLastError());
        NMFX_HIP(hipMemcpyAsync(hobj, scal.p, sizeof(double), hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));   // (the line search branches on it)
        *obj = *hobj;
        return NMFX_OK;
    };
    double stepH = p->sc_stepsize_H0 > 0 ? p->sc_stepsize_H0 : 1.0, stepW = p->sc_stepsize_W0 > 0 ? p->sc_stepsize_W0 : 1.0;   // nmfsc.m:133-134
    TRY(recon_obj(Wd, HTd, &r->cost[0]));   // nmfsc.m:138-139
    const auto t_it = std::chrono::steady_clock::now();
    io.ingest_s = std::chrono::duration<double>(t_it - t_in).count();
    int ncost = p->maxiter + 1, nH = 0, nW = 0;
    bool early = false;
    for (int it = 1; it <= p->maxiter && !early; ++it) {
        if (!fixH) {
            hipLaunchKernelGGL(sc64_hterms_kernel, g1((long)Kn), dim3(256), 0, st, V.as<double>(), Vh.as<double>(), Wd, m, n, K, negH.as<double>(), posH.as<double>());   // nmfsc.m:144-145
            if (sH > 0) {
                const double begobj = r->cost[it - 1];                                              // nmfsc.m:149
                int tries = 0;
                for (;;) {
                    ++tries;
                    hipLaunchKernelGGL(sc64_step_kernel, g1((long)Kn), dim3(256), 0, st, HTd, negH.as<double>(), posH.as<double>(), stepH, (long)Kn, HnewT);   // nmfsc.m:154
                    TRY(projfunc_cols_f64(st, HnewT, n, K, L1s, 1.0, 1, nullptr));                  // nmfsc.m:155-157
                    double newobj;
                    TRY(recon_obj(Wd, HnewT, &newobj));                                             // nmfsc.m:160-161
                    if (newobj <= begobj) break;                                                    // nmfsc.m:164
                    stepH /= 2;                                                                     // nmfsc.m:169
                    if (stepH < 1e-200) { early = true; break; }                                    // nmfsc.m:170-174
                }
                if (r->tries_H) r->tries_H[nH] = tries;
                ++nH;
                if (early) { ncost = it; break; }
                stepH *= 1.2;                                                                       // nmfsc.m:178
                std::swap(HTd, HnewT);                                                              // nmfsc.m:179
            } else {
                hipLaunchKernelGGL(sc64_mu_kernel, g1((long)Kn), dim3(256), 0, st, HTd, negH.as<double>(), posH.as<double>(), (long)Kn);   // nmfsc.m:182
                NMFX_HIP(hipGetLastError());
                TRY(col_reduce64(st, HTd, n, n, K, 1, scal.as<double>() + 8));                      // nmfsc.m:185
                hipLaunchKernelGGL(sc64_rescale_kernel, g1((long)std::max(Kn, mK)), dim3(256), 0, st, HTd, n, Wd, m, K, scal.as<double>() + 8);   // nmfsc.m:186-187
            }
            NMFX_HIP(hipGetLastError());
        }
        if (!fixW) {
            double begobj;
            TRY(recon_obj(Wd, HTd, &begobj));                                                       // nmfsc.m:193,197
            hipLaunchKernelGGL(sc64_wterms_kernel, g1((long)mK), dim3(256), 0, st, V.as<double>(), Vh.as<double>(), HTd, m, n, K, negW.as<double>(), posW.as<double>());   // nmfsc.m:194-195
            if (sW > 0) {
                int tries = 0;
                for (;;) {
                    ++tries;
                    hipLaunchKernelGGL(sc64_step_kernel, g1((long)mK), dim3(256), 0, st, Wd, negW.as<double>(), posW.as<double>(), stepW, (long)mK, Wnew);   // nmfsc.m:205
                    TRY(projfunc_cols_f64(st, Wnew, m, K, L1a, 1.0, 1, nullptr));                   // nmfsc.m:206-208
                    double newobj;
                    TRY(recon_obj(Wnew, HTd, &newobj));                                             // nmfsc.m:211-212
                    if (newobj <= begobj) break;                                                    // nmfsc.m:215
                    stepW /= 2;                                                                     // nmfsc.m:220
                    if (stepW < 1e-200) { early = true; break; }                                    // nmfsc.m:221-225
                }
                if (r->tries_W) r->tries_W[nW] = tries;
                ++nW;
                if (early) { ncost = it; break; }
                stepW *= 1.2;                                                                       // nmfsc.m:228
                std::swap(Wd, Wnew);                                                                // nmfsc.m:229
            } else hipLaunchKernelGGL(sc64_mu_kernel, g1((long)mK), dim3(256), 0, st, Wd, negW.as<double>(), posW.as<double>(), (long)mK);   // nmfsc.m:232
            NMFX_HIP(hipGetLastError());
        }
        TRY(recon_obj(Wd, HTd, &r->cost[it]));                                                      // nmfsc.m:237-238
        sc_hooks_iteration_done(t_it);
        if (p->tolerance >= 0 && it > 1 && r->cost[it] < r->cost[it - 1] && r->cost[it - 1] - r->cost[it] < p->tolerance) {   // nmfsc.m:241-244
            ncost = it + 1;
            break;
        }
    }
    r->cost_len = ncost;
    r->iters_run = ncost - 1;
    r->stepsize_H = stepH; r->stepsize_W = stepW;
    r->converged_early = early ? 1 : 0;
    if (r->tries_H) for (int i = nH; i < p->maxiter; ++i) r->tries_H[i] = 0;
    if (r->tries_W) for (int i = nW; i < p->maxiter; ++i) r->tries_W[i] = 0;
    const auto t_out = std::chrono::steady_clock::now();   // (the last objective was read on the host: the iterations are complete)
    io.iterate_s = std::chrono::duration<double>(t_out - t_it).count();
    hipLaunchKernelGGL(sc64_transpose_kernel, g1((long)Kn), dim3(256), 0, st, HTd, n, (long)K, Hk.as<double>());
    NMFX_HIP(hipGetLastError());
    auto egress = [&](const double *src, size_t count, void *host) -> nmfx_status {
        if (p->dtype == NMFX_F64) { NMFX_HIP(hipMemcpyAsync(host, src, count * 8, hipMemcpyDeviceToHost, st)); return NMFX_OK; }
        if (!stage.p) TRY(stage.alloc(std::max(mn, std::max(mK, Kn)) * 4));
        hipLaunchKernelGGL(sc64_narrow_kernel, g1((long)count), dim3(256), 0, st, src, stage.as<float>(), (long)count);
        NMFX_HIP(hipGetLastError());
        NMFX_HIP(hipMemcpyAsync(host, stage.p, count * 4, hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));
        return NMFX_OK;
    };
    TRY(egress(Wd, mK, r->W));
    TRY(egress(Hk.as<double>(), Kn, r->H));
    NMFX_HIP(hipStreamSynchronize(st));
    io.egress_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_out).count();
    return NMFX_OK;
}

}  // namespace nmfx
