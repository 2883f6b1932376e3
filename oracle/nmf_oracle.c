/* TEST INFRASTRUCTURE ONLY -- parity unpinned by the reference (see oracle/__init__.py).
 *
 * Second, INDEPENDENT float64 restatement of the reference's hot path in plain C
 * (no BLAS, naive loops).  It is written from the algebraically reduced form that the
 * HIP engine also uses -- one concatenated problem with per-column / per-row sparsity
 * offsets and fixed masks, column sums instead of the diag(H*B'*W) GEMM chains
 *  -- so agreement with the literal NumPy restatement
 * (oracle/nmf_oracle.py) to <=1e-12 checks both the restatement and the identities.
 *
 * All matrices are column-major (MATLAB order): V[i+m*j], W[i+m*k+m*K*t], H[k+K*j].
 * Reference lines followed: nmf.m:130-225, cnmf.m:137-258, nmfsc.m:57-245,
 * projfunc.m:13-65, ReconstructFromDecomposition.m:30-38, lnmf.m:49-92, cnmfsc.m:67-277.
 *
 * Build: make -C oracle   (-> oracle/liboracle.so, loaded with ctypes by tests)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 2.220446049250313e-16 /* MATLAB eps = 2^-52 */

enum { DIV_EUCLID = 0, DIV_KL = 1, DIV_IS = 2, DIV_FROB_NOCOST = 4 };

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static double fmax_nan(double x, double e) { return (x > e) ? x : e; } /* MATLAB max(x,eps): NaN -> eps */

/* ReconstructFromDecomposition.m:30-38.  T==1 is the plain W*H branch. */
void oracle_reconstruct(int m, int n, int K, int T, const double *W, const double *H, double *Vh) {
    memset(Vh, 0, sizeof(double) * (size_t)m * n);
    for (int t = 0; t < T; ++t) {
        const double *Wt = W + (size_t)m * K * t;
        for (int j = t; j < n; ++j)
            for (int k = 0; k < K; ++k) {
                double h = H[k + (size_t)K * (j - t)];
                const double *w = Wt + (size_t)m * k;
                double *v = Vh + (size_t)m * j;
                for (int i = 0; i < m; ++i) v[i] += w[i] * h;
            }
    }
}

/* element maps (V, Vhat) -> (A, B):  nmf.m:149-156 / cnmf.m:191-192 with (alpha,beta) per cnmf.m:137-147 */
static void div_maps(int div, size_t cnt, const double *V, const double *Vh, double *A, double *B, int cnmf_style) {
    for (size_t e = 0; e < cnt; ++e) {
        double v = V[e], s = Vh[e];
        switch (div) {
        case DIV_KL:
            A[e] = cnmf_style ? v * (1.0 / s) : v / s;   /* cnmf: V.^1 .* V_hat.^(-1) */
            B[e] = 1.0;
            break;
        case DIV_IS:
            A[e] = cnmf_style ? v * pow(s, -2.0) : v / (s * s);
            B[e] = 1.0 / s;
            break;
        default:
            A[e] = v;
            B[e] = s;
        }
    }
}

static double div_cost(int div, size_t cnt, const double *V, const double *Vh) {
    double c = 0.0;
    if (div == DIV_FROB_NOCOST) return 0.0;
    for (size_t e = 0; e < cnt; ++e) {
        double v = V[e], s = Vh[e];
        if (div == DIV_EUCLID) c += (v - s) * (v - s);
        else if (div == DIV_KL) c += v * log(v / s) - v + s;
        else c += log(s / v) + v / s - 1.0;
    }
    return div == DIV_EUCLID ? 0.5 * c : c;
}

/* X (m x n) * rshift_t(H)' -> out (m x K):  out[i,k] = sum_{j>=t} X[i,j] H[k,j-t] */
static void x_times_ht(int m, int n, int K, int t, const double *X, const double *H, double *out) {
    memset(out, 0, sizeof(double) * (size_t)m * K);
    for (int j = t; j < n; ++j)
        for (int k = 0; k < K; ++k) {
            double h = H[k + (size_t)K * (j - t)];
            const double *x = X + (size_t)m * j;
            double *o = out + (size_t)m * k;
            for (int i = 0; i < m; ++i) o[i] += x[i] * h;
        }
}

/* out (K x n) += Wt' * lshift_t(X):  out[k,j] += sum_i Wt[i,k] X[i,j+t]  (j+t<n) */
static void wt_times_x_acc(int m, int n, int K, int t, const double *Wt, const double *X, double *out) {
    for (int j = 0; j + t < n; ++j)
        for (int k = 0; k < K; ++k) {
            const double *w = Wt + (size_t)m * k;
            const double *x = X + (size_t)m * (j + t);
            double a = 0.0;
            for (int i = 0; i < m; ++i) a += w[i] * x[i];
            out[k + (size_t)K * j] += a;
        }
}

static double l1_terms(int m, int n, int K, int T, const double *W, const double *H, const double *lamW, const double *lamH) {
    double c = 0.0;
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < K; ++k) {
            if (lamW[k] == 0.0) continue;
            double a = 0.0;
            const double *w = W + (size_t)m * k + (size_t)m * K * t;
            for (int i = 0; i < m; ++i) a += fabs(w[i]);
            c += lamW[k] * a;
        }
    for (int k = 0; k < K; ++k) {
        if (lamH[k] == 0.0) continue;
        double a = 0.0;
        for (int j = 0; j < n; ++j) a += fabs(H[k + (size_t)K * j]);
        c += lamH[k] * a;
    }
    return c;
}

/* nmf.m:130-225 on the concatenated problem.  lamW/fixW are per COLUMN of W (source value repeated),
 * lamH/fixH per ROW of H.  W, H: in = init, out = result.  cost has maxiter entries. */
int oracle_nmf(int m, int n, int K, const double *V, double *W, double *H, int div, const double *lamW,
               const double *lamH, const unsigned char *fixW, const unsigned char *fixH, int maxiter, double tol,
               double *cost, int *iters_run) {
    size_t mn = (size_t)m * n;
    double *Vh = dalloc(mn), *A = dalloc(mn), *B = dalloc(mn);
    double *N = dalloc((size_t)m * K), *P = dalloc((size_t)m * K);
    double *Gn = dalloc((size_t)K * n), *Gp = dalloc((size_t)K * n);
    for (int k = 0; k < K; ++k) { /* nmf.m:132-134: every W{s} is L2-column-normalised, fixed or not */
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += W[i + (size_t)m * k] * W[i + (size_t)m * k];
        s = 1.0 / sqrt(s);
        for (int i = 0; i < m; ++i) W[i + (size_t)m * k] *= s;
    }
    oracle_reconstruct(m, n, K, 1, W, H, Vh); /* nmf.m:139 */
    int it;
    *iters_run = maxiter;
    for (it = 0; it < maxiter; ++it) {
        /* W step, nmf.m:145-171 (V_hat not refreshed between sources) */
        div_maps(div, mn, V, Vh, A, B, 0);
        x_times_ht(m, n, K, 0, A, H, N);
        x_times_ht(m, n, K, 0, B, H, P);
        for (int k = 0; k < K; ++k) {
            if (fixW[k]) continue;
            double *w = W + (size_t)m * k, *nn = N + (size_t)m * k, *pp = P + (size_t)m * k;
            double dn = 0.0, dp = 0.0, ss = 0.0;
            for (int i = 0; i < m; ++i) { dn += w[i] * pp[i]; dp += w[i] * nn[i]; }
            for (int i = 0; i < m; ++i) {
                double neg = nn[i] + w[i] * dn, pos = pp[i] + w[i] * dp;
                w[i] = w[i] * (neg / fmax_nan(pos + lamW[k], EPS)); /* nmf.m:168 */
                ss += w[i] * w[i];
            }
            ss = 1.0 / sqrt(ss); /* nmf.m:169 */
            for (int i = 0; i < m; ++i) w[i] *= ss;
        }
        oracle_reconstruct(m, n, K, 1, W, H, Vh); /* nmf.m:173 */
        /* H step, nmf.m:176-201 */
        div_maps(div, mn, V, Vh, A, B, 0);
        memset(Gn, 0, sizeof(double) * (size_t)K * n);
        memset(Gp, 0, sizeof(double) * (size_t)K * n);
        wt_times_x_acc(m, n, K, 0, W, A, Gn);
        wt_times_x_acc(m, n, K, 0, W, B, Gp);
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < K; ++k) {
                if (fixH[k]) continue;
                size_t e = k + (size_t)K * j;
                H[e] = H[e] * (Gn[e] / fmax_nan(Gp[e] + lamH[k], EPS)); /* nmf.m:199 */
            }
        oracle_reconstruct(m, n, K, 1, W, H, Vh); /* nmf.m:203 */
        cost[it] = div_cost(div, mn, V, Vh) + l1_terms(m, n, K, 1, W, H, lamW, lamH); /* nmf.m:206-218 */
        if (it > 0 && cost[it] < cost[it - 1] && cost[it - 1] - cost[it] < tol) { /* nmf.m:221-224 */
            *iters_run = it + 1;
            break;
        }
    }
    free(Vh); free(A); free(B); free(N); free(P); free(Gn); free(Gp);
    return 0;
}

static void slab_norms(int m, int K, int T, const double *W, double *nrm) {
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int t = 0; t < T; ++t) {
            const double *w = W + (size_t)m * k + (size_t)m * K * t;
            for (int i = 0; i < m; ++i) s += w[i] * w[i];
        }
        nrm[k] = sqrt(s) / T; /* norm(squeeze(W(:,k,:)),'fro') / context_len */
    }
}

/* cnmf.m:155-258 on the concatenated problem (non-dual forms: euclid, frobenius, kl, is). */
int oracle_cnmf(int m, int n, int K, int T, const double *V, double *W, double *H, int div, const double *lamW,
                const double *lamH, const unsigned char *fixW, const unsigned char *fixH, int maxiter, double tol,
                double *cost, int *iters_run) {
    size_t mn = (size_t)m * n, mK = (size_t)m * K;
    int mdiv = (div == DIV_FROB_NOCOST) ? DIV_EUCLID : div;
    double *Vh = dalloc(mn), *A = dalloc(mn), *B = dalloc(mn);
    double *N = dalloc(mK), *P = dalloc(mK), *nrm = dalloc(K);
    double *Gn = dalloc((size_t)K * n), *Gp = dalloc((size_t)K * n);
    slab_norms(m, K, T, W, nrm); /* cnmf.m:157-166: all sources, fixed or not; H is rescaled here only */
    for (int k = 0; k < K; ++k) {
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < m; ++i) W[i + (size_t)m * k + mK * t] /= nrm[k];
        for (int j = 0; j < n; ++j) H[k + (size_t)K * j] *= nrm[k];
    }
    oracle_reconstruct(m, n, K, T, W, H, Vh); /* cnmf.m:171 */
    *iters_run = maxiter;
    for (int it = 0; it < maxiter; ++it) {
        /* W step, cnmf.m:187-194: every t uses the same V_hat and the not-yet-updated slice t */
        div_maps(mdiv, mn, V, Vh, A, B, 1);
        for (int t = 0; t < T; ++t) {
            double *Wt = W + mK * t;
            x_times_ht(m, n, K, t, A, H, N);
            x_times_ht(m, n, K, t, B, H, P);
            for (int k = 0; k < K; ++k) {
                if (fixW[k]) continue;
                double *w = Wt + (size_t)m * k, *nn = N + (size_t)m * k, *pp = P + (size_t)m * k;
                double dn = 0.0, dp = 0.0;
                for (int i = 0; i < m; ++i) { dn += w[i] * pp[i]; dp += w[i] * nn[i]; }
                for (int i = 0; i < m; ++i) {
                    double neg = nn[i] + w[i] * dn, pos = pp[i] + w[i] * dp;
                    w[i] = w[i] * (neg / fmax_nan(pos + lamW[k], EPS)); /* cnmf.m:193 */
                }
            }
        }
        slab_norms(m, K, T, W, nrm); /* cnmf.m:196-199 (inside `if ~W_fixed`) */
        for (int k = 0; k < K; ++k) {
            if (fixW[k]) continue;
            for (int t = 0; t < T; ++t)
                for (int i = 0; i < m; ++i) W[i + (size_t)m * k + mK * t] /= nrm[k];
        }
        oracle_reconstruct(m, n, K, T, W, H, Vh); /* cnmf.m:204 */
        /* H step, cnmf.m:209-232 */
        div_maps(mdiv, mn, V, Vh, A, B, 1);
        memset(Gn, 0, sizeof(double) * (size_t)K * n);
        memset(Gp, 0, sizeof(double) * (size_t)K * n);
        for (int t = 0; t < T; ++t) {
            wt_times_x_acc(m, n, K, t, W + mK * t, A, Gn);
            if (mdiv == DIV_KL) { /* cnmf.m:220-221: V_pos NOT shifted for KL */
                for (int k = 0; k < K; ++k) {
                    double cs = 0.0;
                    const double *w = W + mK * t + (size_t)m * k;
                    for (int i = 0; i < m; ++i) cs += w[i];
                    for (int j = 0; j < n; ++j) Gp[k + (size_t)K * j] += cs;
                }
            } else
                wt_times_x_acc(m, n, K, t, W + mK * t, B, Gp);
        }
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < K; ++k) {
                if (fixH[k]) continue;
                size_t e = k + (size_t)K * j;
                H[e] = H[e] * (Gn[e] / fmax_nan(Gp[e] + lamH[k], EPS)); /* cnmf.m:231 */
            }
        oracle_reconstruct(m, n, K, T, W, H, Vh); /* cnmf.m:236 */
        cost[it] = div_cost(div, mn, V, Vh) + l1_terms(m, n, K, T, W, H, lamW, lamH); /* cnmf.m:239-251 */
        if (it > 0 && cost[it] < cost[it - 1] && cost[it - 1] - cost[it] < tol) { /* cnmf.m:254-257 */
            *iters_run = it + 1;
            break;
        }
    }
    free(Vh); free(A); free(B); free(N); free(P); free(nrm); free(Gn); free(Gp);
    return 0;
}

/* projfunc.m:13-65.  s, v have stride `inc` (lets callers project rows of a column-major matrix). */
int oracle_projfunc(int N, const double *s, int inc, double k1, double k2, int nn, double *v, int vinc, int *usediters) {
    unsigned char *isneg = (unsigned char *)calloc(N, 1), *Z = (unsigned char *)calloc(N, 1);
    double *x = dalloc(N);
    double sum = 0.0;
    int nz = 0, j = 0;
    for (int i = 0; i < N; ++i) {
        double e = s[(size_t)i * inc];
        if (!nn) { isneg[i] = e < 0; e = fabs(e); }
        x[i] = e;
        sum += e;
    }
    for (int i = 0; i < N; ++i) x[i] += (k1 - sum) / N; /* projfunc.m:22 */
    for (;;) {
        double mid = k1 / (N - nz), a = 0.0, b = 0.0, c = 0.0; /* projfunc.m:31-36 */
        for (int i = 0; i < N; ++i) {
            double w = x[i] - (Z[i] ? 0.0 : mid);
            a += w * w; b += w * x[i]; c += x[i] * x[i];
        }
        b *= 2.0; c -= k2;
        double disc = b * b - 4.0 * a * c;
        double al = (-b + (disc > 0 ? sqrt(disc) : 0.0)) / (2.0 * a); /* projfunc.m:37 real(sqrt()) */
        int allnn = 1;
        for (int i = 0; i < N; ++i) {
            double w = x[i] - (Z[i] ? 0.0 : mid);
            x[i] = al * w + x[i]; /* projfunc.m:38 */
            if (!(x[i] >= 0)) allnn = 0;
        }
        if (allnn) { *usediters = j + 1; break; } /* projfunc.m:40-44 */
        ++j;
        nz = 0; sum = 0.0;
        for (int i = 0; i < N; ++i) { /* projfunc.m:49-51 */
            Z[i] = (x[i] <= 0);
            if (Z[i]) { x[i] = 0.0; ++nz; }
            sum += x[i];
        }
        for (int i = 0; i < N; ++i) x[i] = Z[i] ? 0.0 : x[i] + (k1 - sum) / (N - nz); /* projfunc.m:52-53 */
    }
    for (int i = 0; i < N; ++i) v[(size_t)i * vinc] = (!nn && isneg[i]) ? -x[i] : x[i];
    free(isneg); free(Z); free(x);
    return 0;
}

static double half_sq_resid(size_t cnt, const double *V, const double *Vh) {
    double c = 0.0;
    for (size_t e = 0; e < cnt; ++e) c += (V[e] - Vh[e]) * (V[e] - Vh[e]);
    return 0.5 * c;
}

/* nmfsc.m:57-245.  V is rescaled by its max into an internal copy.  cost has maxiter+1 entries;
 * triesH/triesW (maxiter entries each, may be NULL) get the line-search try counts; steps[2]={stepH,stepW}.
 * returns 0 ok, 1 negative data */
int oracle_nmfsc(int m, int n, int K, const double *Vin, double *W, double *H, double sW, double sH, int fixW,
                 int fixH, int maxiter, double tol, double *cost, int *ncost, int *triesH, int *triesW, double *steps) {
    size_t mn = (size_t)m * n, mK = (size_t)m * K, Kn = (size_t)K * n;
    double vmax = -INFINITY, vmin = INFINITY;
    for (size_t e = 0; e < mn; ++e) { if (Vin[e] > vmax) vmax = Vin[e]; if (Vin[e] < vmin) vmin = Vin[e]; }
    if (vmin < 0) return 1; /* nmfsc.m:57-59 */
    double *V = dalloc(mn), *Vh = dalloc(mn), *neg = dalloc(Kn > mK ? Kn : mK), *pos = dalloc(Kn > mK ? Kn : mK);
    double *Xn = dalloc(Kn > mK ? Kn : mK);
    for (size_t e = 0; e < mn; ++e) V[e] = Vin[e] / vmax; /* nmfsc.m:62 */
    double L1a = 0, L1s = 0, stepW = 1.0, stepH = 1.0;
    int it_used;
    if (sW > 0) { /* nmfsc.m:89-97 */
        if (sW > 1) sW = 1;
        L1a = sqrt((double)m) - (sqrt((double)m) - 1) * sW;
        for (int k = 0; k < K; ++k) oracle_projfunc(m, W + (size_t)m * k, 1, L1a, 1.0, 1, W + (size_t)m * k, 1, &it_used);
    }
    if (sH > 0) { /* nmfsc.m:102-110 */
        if (sH > 1) sH = 1;
        L1s = sqrt((double)n) - (sqrt((double)n) - 1) * sH;
        for (int k = 0; k < K; ++k) oracle_projfunc(n, H + k, K, L1s, 1.0, 1, H + k, K, &it_used);
    }
    oracle_reconstruct(m, n, K, 1, W, H, Vh);
    cost[0] = half_sq_resid(mn, V, Vh); /* nmfsc.m:138-139 */
    *ncost = maxiter + 1;
    int nH = 0, nW = 0, early = 0;
    for (int it = 1; it <= maxiter && !early; ++it) {
        if (!fixH) {
            memset(neg, 0, sizeof(double) * Kn); memset(pos, 0, sizeof(double) * Kn);
            wt_times_x_acc(m, n, K, 0, W, V, neg);  /* nmfsc.m:144 */
            wt_times_x_acc(m, n, K, 0, W, Vh, pos); /* nmfsc.m:145 */
            if (sH > 0) {
                double begobj = cost[it - 1]; /* nmfsc.m:149 */
                int tries = 0;
                for (;;) {
                    ++tries;
                    for (size_t e = 0; e < Kn; ++e) Xn[e] = H[e] - stepH * (pos[e] - neg[e]); /* nmfsc.m:154 */
                    for (int k = 0; k < K; ++k) oracle_projfunc(n, Xn + k, K, L1s, 1.0, 1, Xn + k, K, &it_used);
                    oracle_reconstruct(m, n, K, 1, W, Xn, Vh); /* nmfsc.m:160 */
                    double newobj = half_sq_resid(mn, V, Vh);
                    if (newobj <= begobj) break; /* nmfsc.m:164 */
                    stepH /= 2;
                    if (stepH < 1e-200) { early = 1; break; } /* nmfsc.m:170-174 */
                }
                if (triesH) triesH[nH] = tries;
                ++nH;
                if (early) { *ncost = it; break; }
                stepH *= 1.2; /* nmfsc.m:178 */
                memcpy(H, Xn, sizeof(double) * Kn);
            } else {
                for (size_t e = 0; e < Kn; ++e) H[e] = H[e] * (neg[e] / fmax_nan(pos[e], EPS)); /* nmfsc.m:182 */
                for (int k = 0; k < K; ++k) { /* nmfsc.m:185-187 */
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += H[k + (size_t)K * j] * H[k + (size_t)K * j];
                    s = sqrt(s);
                    for (int j = 0; j < n; ++j) H[k + (size_t)K * j] *= 1.0 / s;
                    for (int i = 0; i < m; ++i) W[i + (size_t)m * k] *= s;
                }
            }
        }
        if (!fixW) {
            oracle_reconstruct(m, n, K, 1, W, H, Vh); /* nmfsc.m:193 */
            x_times_ht(m, n, K, 0, V, H, neg);        /* nmfsc.m:194 */
            x_times_ht(m, n, K, 0, Vh, H, pos);       /* nmfsc.m:195 */
            if (sW > 0) {
                double begobj = half_sq_resid(mn, V, Vh); /* nmfsc.m:197 */
                int tries = 0;
                for (;;) {
                    ++tries;
                    for (size_t e = 0; e < mK; ++e) Xn[e] = W[e] - stepW * (pos[e] - neg[e]); /* nmfsc.m:205 */
                    for (int k = 0; k < K; ++k) oracle_projfunc(m, Xn + (size_t)m * k, 1, L1a, 1.0, 1, Xn + (size_t)m * k, 1, &it_used);
                    oracle_reconstruct(m, n, K, 1, Xn, H, Vh);
                    double newobj = half_sq_resid(mn, V, Vh);
                    if (newobj <= begobj) break;
                    stepW /= 2;
                    if (stepW < 1e-200) { early = 1; break; }
                }
                if (triesW) triesW[nW] = tries;
                ++nW;
                if (early) { *ncost = it; break; }
                stepW *= 1.2;
                memcpy(W, Xn, sizeof(double) * mK);
            } else {
                for (size_t e = 0; e < mK; ++e) W[e] = W[e] * (neg[e] / fmax_nan(pos[e], EPS)); /* nmfsc.m:232 */
            }
        }
        oracle_reconstruct(m, n, K, 1, W, H, Vh); /* nmfsc.m:237 */
        cost[it] = half_sq_resid(mn, V, Vh);
        if (it > 1 && cost[it] < cost[it - 1] && cost[it - 1] - cost[it] < tol) { /* nmfsc.m:241-244 */
            *ncost = it + 1;
            break;
        }
    }
    if (steps) { steps[0] = stepH; steps[1] = stepW; }
    free(V); free(Vh); free(neg); free(pos); free(Xn);
    return 0;
}

/* lnmf.m:49-92 (independent restatement; KL only, one source).  cost has maxiter entries and is NOT trimmed on break. */
int oracle_lnmf(int m, int n, int K, const double *V, double *W, double *H, int fixW, int fixH, int maxiter, double tol,
                double *cost, int *iters_run) {
    size_t mn = (size_t)m * n;
    double *Vh = dalloc(mn), *A = dalloc(mn), *N = dalloc((size_t)m * K), *G = dalloc((size_t)K * n);
    for (int k = 0; k < K; ++k) { /* lnmf.m:59 */
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += W[i + (size_t)m * k];
        for (int i = 0; i < m; ++i) W[i + (size_t)m * k] *= 1.0 / s;
    }
    oracle_reconstruct(m, n, K, 1, W, H, Vh);
    for (int it = 0; it < maxiter; ++it) cost[it] = 0.0;
    *iters_run = maxiter;
    for (int it = 0; it < maxiter; ++it) {
        if (!fixW) { /* lnmf.m:68-72 */
            for (size_t e = 0; e < mn; ++e) A[e] = V[e] / Vh[e];
            x_times_ht(m, n, K, 0, A, H, N);
            for (int k = 0; k < K; ++k) {
                double rs = 0.0, s = 0.0;
                for (int j = 0; j < n; ++j) rs += H[k + (size_t)K * j]; /* ones(m,n)*H' */
                for (int i = 0; i < m; ++i) { double *w = &W[i + (size_t)m * k]; *w = *w * (N[i + (size_t)m * k] / fmax_nan(rs, EPS)); s += *w; }
                for (int i = 0; i < m; ++i) W[i + (size_t)m * k] *= 1.0 / s;
            }
            oracle_reconstruct(m, n, K, 1, W, H, Vh);
        }
        if (!fixH) { /* lnmf.m:75-78 */
            for (size_t e = 0; e < mn; ++e) A[e] = V[e] / Vh[e];
            memset(G, 0, sizeof(double) * (size_t)K * n);
            wt_times_x_acc(m, n, K, 0, W, A, G);
            for (size_t e = 0; e < (size_t)K * n; ++e) H[e] = sqrt(H[e] * G[e]);
            oracle_reconstruct(m, n, K, 1, W, H, Vh);
        }
        cost[it] = div_cost(DIV_KL, mn, V, Vh); /* lnmf.m:81 */
        if (it > 0 && cost[it] <= cost[it - 1] && cost[it - 1] - cost[it] <= tol) { *iters_run = it + 1; break; } /* lnmf.m:84-86 */
    }
    free(Vh); free(A); free(N); free(G);
    return 0;
}

/* cnmfsc.m:67-277 (independent restatement, quirks included: see oracle/nmf_oracle.py::cnmfsc).
 * W in/out is m x K x T; tries arrays may be NULL; triesW needs maxiter*T entries.  returns 0 ok, 1 negative data */
int oracle_cnmfsc(int m, int n, int K, int T, const double *Vin, double *W, double *H, double sW, double sH, int fixW, int fixH,
                  int maxiter, double tol, double *cost, int *ncost, int *triesH, int *triesW) {
    size_t mn = (size_t)m * n, mK = (size_t)m * K, mKT = mK * T, Kn = (size_t)K * n;
    double vmax = -INFINITY, vmin = INFINITY;
    for (size_t e = 0; e < mn; ++e) { if (Vin[e] > vmax) vmax = Vin[e]; if (Vin[e] < vmin) vmin = Vin[e]; }
    if (vmin < 0) return 1;
    double *V = dalloc(mn), *Vh = dalloc(mn), *W0 = dalloc(mKT), *neg = dalloc(Kn > mK ? Kn : mK), *pos = dalloc(Kn > mK ? Kn : mK);
    double *Xn = dalloc(Kn > mK ? Kn : mK), *stepW = dalloc(T);
    for (size_t e = 0; e < mn; ++e) V[e] = Vin[e] / vmax;
    memcpy(W0, W, sizeof(double) * mKT); /* W0 = W_init; W = W0 (cnmfsc.m:93-94) */
    double L1a = 0, L1s = 0, stepH = 1.0;
    int dummy, nH = 0, nW = 0, early = 0;
    for (int t = 0; t < T; ++t) stepW[t] = 1.0;
    if (sW > 0) { /* only W is projected, W0 is not (cnmfsc.m:105-109) */
        if (sW > 1) sW = 1;
        L1a = sqrt((double)m) - (sqrt((double)m) - 1) * sW;
        for (int c = 0; c < K * T; ++c) oracle_projfunc(m, W + (size_t)m * c, 1, L1a, 1.0, 1, W + (size_t)m * c, 1, &dummy);
    }
    if (sH > 0) {
        if (sH > 1) sH = 1;
        L1s = sqrt((double)n) - (sqrt((double)n) - 1) * sH;
        for (int k = 0; k < K; ++k) oracle_projfunc(n, H + k, K, L1s, 1.0, 1, H + k, K, &dummy);
    }
    oracle_reconstruct(m, n, K, T, W, H, Vh);
    cost[0] = half_sq_resid(mn, V, Vh);
    *ncost = maxiter + 1;
    for (int it = 1; it <= maxiter && !early; ++it) {
        if (!fixH) {
            memset(neg, 0, sizeof(double) * Kn); memset(pos, 0, sizeof(double) * Kn);
            for (int t = 0; t < T; ++t) { /* cnmfsc.m:160-165 */
                wt_times_x_acc(m, n, K, t, W0 + mK * t, V, neg);
                wt_times_x_acc(m, n, K, t, W0 + mK * t, Vh, pos);
            }
            if (sH > 0) {
                double begobj = cost[it - 1];
                int tries = 0;
                for (;;) {
                    ++tries;
                    for (size_t e = 0; e < Kn; ++e) Xn[e] = H[e] - stepH * (pos[e] - neg[e]);
                    for (int k = 0; k < K; ++k) oracle_projfunc(n, Xn + k, K, L1s, 1.0, 1, Xn + k, K, &dummy);
                    oracle_reconstruct(m, n, K, T, W0, Xn, Vh);
                    if (half_sq_resid(mn, V, Vh) <= begobj) break;
                    stepH /= 2;
                    if (stepH < 1e-200) { early = 1; break; }
                }
                if (triesH) triesH[nH] = tries;
                ++nH;
                if (early) { *ncost = it; break; }
                stepH *= 1.2;
                memcpy(H, Xn, sizeof(double) * Kn);
            } else {
                for (size_t e = 0; e < Kn; ++e) H[e] = H[e] * (neg[e] / (pos[e] + EPS)); /* cnmfsc.m:202 */
                for (int k = 0; k < K; ++k) {
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += H[k + (size_t)K * j] * H[k + (size_t)K * j];
                    s = sqrt(s);
                    for (int j = 0; j < n; ++j) H[k + (size_t)K * j] *= 1.0 / s;
                    for (int t = 0; t < T; ++t)
                        for (int i = 0; i < m; ++i) W0[i + (size_t)m * k + mK * t] *= s; /* cnmfsc.m:207-209 */
                }
            }
        }
        if (!fixW) {
            oracle_reconstruct(m, n, K, T, W0, H, Vh); /* cnmfsc.m:215 */
            for (int t = 0; t < T && !early; ++t) {
                double *W0t = W0 + mK * t, *Wt = W + mK * t;
                x_times_ht(m, n, K, t, V, H, neg);
                x_times_ht(m, n, K, t, Vh, H, pos);
                if (sW > 0) {
                    double begobj = half_sq_resid(mn, V, Vh);
                    int tries = 0;
                    for (;;) {
                        ++tries;
                        for (size_t e = 0; e < mK; ++e) Xn[e] = W0t[e] - stepW[t] * (pos[e] - neg[e]);
                        for (int k = 0; k < K; ++k) oracle_projfunc(m, Xn + (size_t)m * k, 1, L1a, 1.0, 1, Xn + (size_t)m * k, 1, &dummy);
                        oracle_reconstruct(m, n, K, 1, Xn, H, Vh); /* 2-D slice: plain Wnew*H (cnmfsc.m:235) */
                        if (half_sq_resid(mn, V, Vh) <= begobj) break;
                        stepW[t] /= 2;
                        if (stepW[t] < 1e-200) { early = 1; break; }
                    }
                    if (triesW) triesW[nW] = tries;
                    ++nW;
                    if (early) { *ncost = it; break; }
                    stepW[t] *= 1.2;
                    memcpy(Wt, Xn, sizeof(double) * mK);
                } else {
                    for (size_t e = 0; e < mK; ++e) { Wt[e] = W0t[e] * (neg[e] / fmax_nan(pos[e], EPS)); Xn[e] = Wt[e] - W0t[e]; }
                    for (int j = t; j < n; ++j) /* V_hat = max(V_hat + dW*rshift_t(H), 0)  (cnmfsc.m:262) */
                        for (int k = 0; k < K; ++k) {
                            double h = H[k + (size_t)K * (j - t)];
                            for (int i = 0; i < m; ++i) Vh[i + (size_t)m * j] += Xn[i + (size_t)m * k] * h;
                        }
                    for (size_t e = 0; e < mn; ++e) if (!(Vh[e] > 0.0)) Vh[e] = 0.0;
                }
            }
            if (early) break;
        }
        memcpy(W0, W, sizeof(double) * mKT); /* cnmfsc.m:266 */
        oracle_reconstruct(m, n, K, T, W0, H, Vh);
        cost[it] = half_sq_resid(mn, V, Vh);
        if (it > 1 && cost[it] < cost[it - 1] && cost[it - 1] - cost[it] < tol) { *ncost = it + 1; break; }
    }
    free(V); free(Vh); free(W0); free(neg); free(pos); free(Xn); free(stepW);
    return 0;
}

/* constrainednmf.m:183-258 on label-SORTED samples (the caller has done lines 147-170: processed labels, stable sort, permuted
 * V); seg[0..nz] are the column ranges of the non-zeros of A's rows (A = [I 0; 0 C], constrainednmf.m:166-170), so
 * X*A' is a segmented column sum and Z*A a segment broadcast.  div: 0 euclidean, 1 kl, 2 is.  W (m x K), Z (K x nz) in/out,
 * H (K x n) out = Z*A in sorted order. */
int oracle_constrainednmf(int m, int n, int K, const double *V, double *W, double *Z, double *H, const long *seg, int nz, int div,
                          double lamW, double lamZ, int fixW, int fixZ, int maxiter, double tol, double *cost, int *iters_run) {
    size_t mn = (size_t)m * n;
    double *Vh = dalloc(mn), *A = dalloc(mn), *B = dalloc(mn);
    double *N = dalloc((size_t)m * K), *P = dalloc((size_t)m * K);
    double *Gn = dalloc((size_t)K * n), *Gp = dalloc((size_t)K * n);
    for (int k = 0; k < K; ++k) { /* constrainednmf.m:144-145 */
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += W[i + (size_t)m * k] * W[i + (size_t)m * k];
        s = 1.0 / sqrt(s);
        for (int i = 0; i < m; ++i) W[i + (size_t)m * k] *= s;
    }
    for (int c = 0; c < nz; ++c) /* H = Z*A, constrainednmf.m:177 */
        for (long j = seg[c]; j < seg[c + 1]; ++j)
            for (int k = 0; k < K; ++k) H[k + (size_t)K * j] = Z[k + (size_t)K * c];
    oracle_reconstruct(m, n, K, 1, W, H, Vh); /* 179 */
    *iters_run = maxiter;
    for (int it = 0; it < maxiter; ++it) {
        if (!fixW) { /* 185-209: nmf's W step */
            div_maps(div, mn, V, Vh, A, B, 0);
            x_times_ht(m, n, K, 0, A, H, N);
            x_times_ht(m, n, K, 0, B, H, P);
            for (int k = 0; k < K; ++k) {
                double *w = W + (size_t)m * k, *nn = N + (size_t)m * k, *pp = P + (size_t)m * k;
                double dn = 0.0, dp = 0.0, ss = 0.0;
                for (int i = 0; i < m; ++i) { dn += w[i] * pp[i]; dp += w[i] * nn[i]; }
                for (int i = 0; i < m; ++i) {
                    double neg = nn[i] + w[i] * dn, pos = pp[i] + w[i] * dp;
                    w[i] = w[i] * (neg / fmax_nan(pos + lamW, EPS)); /* 207 */
                    ss += w[i] * w[i];
                }
                ss = 1.0 / sqrt(ss); /* 208 */
                for (int i = 0; i < m; ++i) w[i] *= ss;
            }
        }
        oracle_reconstruct(m, n, K, 1, W, H, Vh); /* 210 */
        if (!fixZ) { /* 213-236 */
            div_maps(div, mn, V, Vh, A, B, 0);
            memset(Gn, 0, sizeof(double) * (size_t)K * n);
            memset(Gp, 0, sizeof(double) * (size_t)K * n);
            wt_times_x_acc(m, n, K, 0, W, A, Gn);
            wt_times_x_acc(m, n, K, 0, W, B, Gp);
            for (int c = 0; c < nz; ++c)
                for (int k = 0; k < K; ++k) {
                    double neg = 0.0, pos = 0.0;
                    for (long j = seg[c]; j < seg[c + 1]; ++j) { neg += Gn[k + (size_t)K * j]; pos += Gp[k + (size_t)K * j]; } /* ... * A' */
                    Z[k + (size_t)K * c] *= neg / fmax_nan(pos + lamZ, EPS); /* 235 */
                }
        }
        for (int c = 0; c < nz; ++c) /* 237 */
            for (long j = seg[c]; j < seg[c + 1]; ++j)
                for (int k = 0; k < K; ++k) H[k + (size_t)K * j] = Z[k + (size_t)K * c];
        oracle_reconstruct(m, n, K, 1, W, H, Vh); /* 238 */
        double l1w = 0.0, l1z = 0.0;
        for (size_t e = 0; e < (size_t)m * K; ++e) l1w += fabs(W[e]);
        for (size_t e = 0; e < (size_t)K * nz; ++e) l1z += fabs(Z[e]);
        cost[it] = div_cost(div, mn, V, Vh) + lamW * l1w + lamZ * l1z; /* 241-251 */
        if (it > 0 && cost[it] < cost[it - 1] && cost[it - 1] - cost[it] < tol) { /* 254-257 */
            *iters_run = it + 1;
            break;
        }
    }
    free(Vh); free(A); free(B); free(N); free(P); free(Gn); free(Gp);
    return 0;
}

/* SortDictionary.m:33-43: 0-based stable order of the columns of W (m x K) by centre of gravity */
void oracle_sort_dictionary_order(int m, int K, const double *W, int *order) {
    int *cog = (int *)calloc((size_t)K, sizeof(int));
    for (int j = 0; j < K; ++j) {
        const double *w = W + (size_t)m * j;
        double total = 0.0, cs = 0.0;
        for (int i = 0; i < m; ++i) total += w[i]; /* W_sum(end, j): the sequential cumulative sum */
        int idx = 0;
        for (int i = 0; i < m; ++i) { cs += w[i]; if (cs <= total / 2) idx = i + 1; } /* find(..., 1, 'last') */
        cog[j] = idx ? idx : 1;
    }
    for (int j = 0; j < K; ++j) order[j] = j;
    for (int a = 1; a < K; ++a) { /* insertion sort: stable, like MATLAB's sort */
        int v = order[a], b = a - 1;
        while (b >= 0 && cog[order[b]] > cog[v]) { order[b + 1] = order[b]; --b; }
        order[b + 1] = v;
    }
    free(cog);
}
