/* nmfx_mex.c -- thin MEX gateway from MATLAB to the C ABI of libnmfx (include/nmfx.h).
 *
 * Neither MATLAB nor Octave exists in the build image, so this file has never met a real mex.h; it is compiled with -Wall -Wextra
 * -Werror and EXECUTED end to end against the minimal MEX runtime of tests/mock_mex/ (tests/test_mex_gateway.py: argument
 * checking on the CPU, nmf / cnmf / nmfsc / ... against the ctypes path on the GPU).  It is the binding a maintainer of the toolbox
 * would add.  No numerics and no validation beyond types and shapes: the .m wrappers next to it keep the reference
 * signatures and run the reference's local ValidateParameters logic before calling in here.
 *
 *   [W, H, cost, info] = nmfx_mex(algo, V, W_init, H_init, K_s, T, opts)           algo = 'nmf' | 'cnmf' | 'lnmf' | 'nmfsc' | 'cnmfsc'
 *   [W, H, cost, Z]    = nmfx_mex('constrainednmf', V, W_init, Z_init, segments, opts)   segments: int64 1 x (nz+1)  (see nmfx.h)
 *   V_hat              = nmfx_mex('reconstruct', W, H)                              ReconstructFromDecomposition.m:1
 *   [v, usediters]     = nmfx_mex('projfunc', s, k1, k2, nn)                        projfunc.m:1 (columns of s are projected independently)
 *   [Ws, Hs, order]    = nmfx_mex('sortdictionary', W, H)                           SortDictionary.m:1 (H may be [])
 *
 *     V      : m x n double          W_init : m x K x T double        H_init : K x n double
 *     K_s    : 1 x S int32 (basis elements per source, sum = K)       T : scalar
 *     opts   : struct; fields (all optional): divergence (0 euclidean, 1 kl, 2 is, 3 ab, 4 cnmf's 'frobenius'), alpha, beta,
 *              W_sparsity, H_sparsity (1 x S double), W_fixed, H_fixed (1 x S uint8), maxiter, tolerance, device, path,
 *              sc_W_sparsity, sc_H_sparsity (nmfsc / cnmfsc), device_ids (1 x N int32: V column-sharded over N GPUs; nmf / cnmf / lnmf / nmfsc)
 *     info   : struct iters_run, stepsize_H, stepsize_W, converged_early, tries_H, tries_W (nmfsc / cnmfsc line searches)
 *
 * build (on a machine with MATLAB):  mex -I../include nmfx_mex.c -L../nmf_toolbox_amd -lnmfx
 */
#include <string.h>

#include "mex.h"
#include "nmfx.h"

#define FAIL(id, ...) do { mexErrMsgIdAndTxt("nmfx:" id, __VA_ARGS__); return; } while (0)

static const mxArray *opt(const mxArray *o, const char *f) {
    const mxArray *a = o ? mxGetField(o, 0, f) : NULL;
    return (a && !mxIsEmpty(a)) ? a : NULL;
}
static double opt_d(const mxArray *o, const char *f, double dflt) {
    const mxArray *a = opt(o, f);
    return a ? mxGetScalar(a) : dflt;
}
/* typed vector option: NULL when absent; *bad is set when present with the wrong class or length */
static const void *opt_vec(const mxArray *o, const char *f, const char *cls, size_t len, int *bad) {
    const mxArray *a = opt(o, f);
    if (!a) return NULL;
    if (!mxIsClass(a, cls) || (len && mxGetNumberOfElements(a) != len)) { *bad = 1; return NULL; }
    return mxGetData(a);
}
static int is_real_double(const mxArray *a) { return a && mxIsDouble(a) && !mxIsComplex(a); }

static void factorise(const char *algo, int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    nmfx_problem p;
    nmfx_result r;
    nmfx_status st = NMFX_ERR_INVALID;
    mwSize dimsW[3];
    const mxArray *o;
    const mwSize *dw;
    int bad = 0, is_sc, tw_len;
    size_t S;
    mxArray *cost, *tH = NULL, *tW = NULL;
    if (nrhs != 7) FAIL("usage", "nmfx_mex(algo, V, W_init, H_init, K_s, T, opts)");
    if (nlhs > 4) FAIL("usage", "at most four outputs: [W, H, cost, info]");
    if (!is_real_double(prhs[1]) || !is_real_double(prhs[2]) || !is_real_double(prhs[3])) FAIL("type", "V, W_init and H_init must be real double arrays");
    if (!mxIsClass(prhs[4], "int32") || mxIsEmpty(prhs[4])) FAIL("type", "K_s must be a non-empty int32 vector");
    if (!mxIsStruct(prhs[6])) FAIL("type", "opts must be a struct");
    o = prhs[6];
    memset(&p, 0, sizeof(p));
    memset(&r, 0, sizeof(r));
    p.m = (int64_t)mxGetM(prhs[1]);
    p.n = (int64_t)mxGetN(prhs[1]);
    p.K_total = (int32_t)mxGetM(prhs[3]);
    p.T = (int32_t)mxGetScalar(prhs[5]);
    if (p.m < 1 || p.n < 1 || p.K_total < 1 || p.T < 1) FAIL("shape", "V, H_init must be non-empty and T >= 1");
    dw = mxGetDimensions(prhs[2]);
    if ((int64_t)dw[0] != p.m || (int64_t)mxGetNumberOfElements(prhs[2]) != p.m * p.K_total * p.T)
        FAIL("shape", "W_init must be %d x %d x %d", (int)p.m, (int)p.K_total, (int)p.T);
    if ((int64_t)mxGetN(prhs[3]) != p.n) FAIL("shape", "H_init must be %d x %d", (int)p.K_total, (int)p.n);
    p.dtype = NMFX_F64;
    p.V = mxGetPr(prhs[1]);
    p.W_init = mxGetPr(prhs[2]);
    p.H_init = mxGetPr(prhs[3]);
    S = mxGetNumberOfElements(prhs[4]);
    p.num_sources = (int32_t)S;
    p.K_s = (const int32_t *)mxGetData(prhs[4]);
    p.divergence = (int32_t)opt_d(o, "divergence", NMFX_DIV_EUCLIDEAN);
    p.alpha = opt_d(o, "alpha", 1.0);
    p.beta = opt_d(o, "beta", 1.0);
    p.W_sparsity = (const double *)opt_vec(o, "W_sparsity", "double", S, &bad);
    p.H_sparsity = (const double *)opt_vec(o, "H_sparsity", "double", S, &bad);
    p.W_fixed = (const uint8_t *)opt_vec(o, "W_fixed", "uint8", S, &bad);
    p.H_fixed = (const uint8_t *)opt_vec(o, "H_fixed", "uint8", S, &bad);
    p.device_ids = (const int32_t *)opt_vec(o, "device_ids", "int32", 0, &bad);
    if (bad) FAIL("type", "opts: W_sparsity / H_sparsity must be double, W_fixed / H_fixed uint8 (one entry per source), device_ids int32");
    if (p.device_ids) p.n_gpus = (int32_t)mxGetNumberOfElements(opt(o, "device_ids"));
    p.maxiter = (int32_t)opt_d(o, "maxiter", 100);
    p.tolerance = opt_d(o, "tolerance", 1e-3);
    p.device = (int32_t)opt_d(o, "device", 0);
    p.path = (int32_t)opt_d(o, "path", 0);
    p.multi_backend = (int32_t)opt_d(o, "multi_backend", 0);   /* 0 auto | 1 peer exchange | 2 RCCL (device_ids) */
    p.sc_W_sparsity = opt_d(o, "sc_W_sparsity", 0.0);
    p.sc_H_sparsity = opt_d(o, "sc_H_sparsity", 0.0);
    if (p.maxiter < 1) FAIL("value", "maxiter must be positive (the .m wrapper applies the reference default)");

    is_sc = !strcmp(algo, "nmfsc") || !strcmp(algo, "cnmfsc");
    tw_len = !strcmp(algo, "cnmfsc") ? p.maxiter * p.T : p.maxiter;      /* cnmfsc: one W line search per time slice (cnmfsc.m:216-246) */
    dimsW[0] = (mwSize)p.m; dimsW[1] = (mwSize)p.K_total; dimsW[2] = (mwSize)p.T;
    plhs[0] = mxCreateNumericArray(p.T > 1 ? 3 : 2, dimsW, mxDOUBLE_CLASS, mxREAL);
    if (nlhs > 1) plhs[1] = mxCreateDoubleMatrix((mwSize)p.K_total, (mwSize)p.n, mxREAL);
    cost = mxCreateDoubleMatrix((mwSize)p.maxiter + 1, 1, mxREAL);        /* nmfsc keeps the initial cost too (nmfsc.m:137-139) */
    r.W = mxGetPr(plhs[0]);
    r.H = nlhs > 1 ? mxGetPr(plhs[1]) : mxGetPr(mxCreateDoubleMatrix((mwSize)p.K_total, (mwSize)p.n, mxREAL));
    r.cost = mxGetPr(cost);
    if (is_sc) {
        tH = mxCreateNumericMatrix((mwSize)p.maxiter, 1, mxINT32_CLASS, mxREAL);
        tW = mxCreateNumericMatrix((mwSize)tw_len, 1, mxINT32_CLASS, mxREAL);
        r.tries_H = (int32_t *)mxGetData(tH);
        r.tries_W = (int32_t *)mxGetData(tW);
    }

    if (!strcmp(algo, "nmf")) st = nmfx_nmf(&p, &r);
    else if (!strcmp(algo, "cnmf")) st = nmfx_cnmf(&p, &r);
    else if (!strcmp(algo, "lnmf")) st = nmfx_lnmf(&p, &r);
    else if (!strcmp(algo, "nmfsc")) st = nmfx_nmfsc(&p, &r);
    else if (!strcmp(algo, "cnmfsc")) st = nmfx_cnmfsc(&p, &r);
    if (st != NMFX_OK) FAIL("error", "%s", nmfx_last_error());            /* same text the reference's error() uses */
    mxSetM(cost, (mwSize)r.cost_len);                                      /* cost = cost(1:iter) trim (nmf.m:222) */
    if (nlhs > 2) plhs[2] = cost;
    if (nlhs > 3) {
        const char *f[] = {"iters_run", "stepsize_H", "stepsize_W", "converged_early", "tries_H", "tries_W"};
        plhs[3] = mxCreateStructMatrix(1, 1, 6, f);
        mxSetField(plhs[3], 0, "iters_run", mxCreateDoubleScalar(r.iters_run));
        mxSetField(plhs[3], 0, "stepsize_H", mxCreateDoubleScalar(r.stepsize_H));
        mxSetField(plhs[3], 0, "stepsize_W", mxCreateDoubleScalar(r.stepsize_W));
        mxSetField(plhs[3], 0, "converged_early", mxCreateDoubleScalar(r.converged_early));
        if (tH) mxSetField(plhs[3], 0, "tries_H", tH);
        if (tW) mxSetField(plhs[3], 0, "tries_W", tW);
    }
}

/* [W, H, cost, Z] = nmfx_mex('constrainednmf', Vsorted, W_init, Z_init, segments, opts)   -- constrainednmf.m:183-258 */
static void constrained(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    nmfx_problem p;
    nmfx_result r;
    const mxArray *o;
    mxArray *cost, *Z;
    int64_t nz;
    double lamW, lamZ;
    uint8_t fixW, fixZ;
    if (nrhs != 6) FAIL("usage", "nmfx_mex('constrainednmf', V, W_init, Z_init, segments, opts)");
    if (nlhs > 4) FAIL("usage", "at most four outputs: [W, H, cost, Z]");
    if (!is_real_double(prhs[1]) || !is_real_double(prhs[2]) || !is_real_double(prhs[3])) FAIL("type", "V, W_init and Z_init must be real double arrays");
    if (!mxIsClass(prhs[4], "int64") || mxGetNumberOfElements(prhs[4]) < 2) FAIL("type", "segments must be an int64 vector with nz + 1 entries");
    if (!mxIsStruct(prhs[5])) FAIL("type", "opts must be a struct");
    o = prhs[5];
    memset(&p, 0, sizeof(p));
    memset(&r, 0, sizeof(r));
    p.m = (int64_t)mxGetM(prhs[1]); p.n = (int64_t)mxGetN(prhs[1]);
    p.K_total = (int32_t)mxGetM(prhs[3]); p.T = 1; p.dtype = NMFX_F64;
    nz = (int64_t)mxGetNumberOfElements(prhs[4]) - 1;
    if ((int64_t)mxGetN(prhs[3]) != nz || (int64_t)mxGetM(prhs[2]) != p.m || (int64_t)mxGetN(prhs[2]) != p.K_total) FAIL("shape", "W_init must be m x K and Z_init K x nz");
    p.V = mxGetPr(prhs[1]); p.W_init = mxGetPr(prhs[2]);
    p.num_sources = 1;
    p.divergence = (int32_t)opt_d(o, "divergence", NMFX_DIV_EUCLIDEAN);
    p.alpha = opt_d(o, "alpha", 1.0); p.beta = opt_d(o, "beta", 1.0);
    lamW = opt_d(o, "W_sparsity", 0.0); lamZ = opt_d(o, "Z_sparsity", 0.0);
    fixW = (uint8_t)(opt_d(o, "W_fixed", 0.0) != 0.0); fixZ = (uint8_t)(opt_d(o, "Z_fixed", 0.0) != 0.0);
    p.W_sparsity = &lamW; p.H_sparsity = &lamZ; p.W_fixed = &fixW; p.H_fixed = &fixZ;
    p.maxiter = (int32_t)opt_d(o, "maxiter", 100); p.tolerance = opt_d(o, "tolerance", 1e-3);
    p.device = (int32_t)opt_d(o, "device", 0); p.path = (int32_t)opt_d(o, "path", 0);
    if (p.maxiter < 1) FAIL("value", "maxiter must be positive");
    plhs[0] = mxCreateDoubleMatrix((mwSize)p.m, (mwSize)p.K_total, mxREAL);
    if (nlhs > 1) plhs[1] = mxCreateDoubleMatrix((mwSize)p.K_total, (mwSize)p.n, mxREAL);
    cost = mxCreateDoubleMatrix((mwSize)p.maxiter, 1, mxREAL);
    Z = mxCreateDoubleMatrix((mwSize)p.K_total, (mwSize)nz, mxREAL);
    r.W = mxGetPr(plhs[0]);
    r.H = nlhs > 1 ? mxGetPr(plhs[1]) : mxGetPr(mxCreateDoubleMatrix((mwSize)p.K_total, (mwSize)p.n, mxREAL));
    r.cost = mxGetPr(cost);
    if (nmfx_constrainednmf(&p, (const int64_t *)mxGetData(prhs[4]), nz, mxGetPr(prhs[3]), &r, mxGetPr(Z)) != NMFX_OK) FAIL("error", "%s", nmfx_last_error());
    mxSetM(cost, (mwSize)r.cost_len);
    if (nlhs > 2) plhs[2] = cost;
    if (nlhs > 3) plhs[3] = Z;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char algo[24];
    int32_t sz_p = 0, sz_r = 0;
    /* the library reads every field of nmfx_problem: a gateway compiled against another nmfx.h must not call in (nmfx.h, NMFX_VERSION) */
    nmfx_abi_sizes(&sz_p, &sz_r, NULL);
    if (nmfx_version() != NMFX_VERSION || sz_p != (int32_t)sizeof(nmfx_problem) || sz_r != (int32_t)sizeof(nmfx_result))
        FAIL("abi", "libnmfx is ABI version %d (nmfx_problem %d bytes), this gateway was compiled against %d (%d bytes): rebuild nmfx_mex",
             (int)nmfx_version(), (int)sz_p, (int)NMFX_VERSION, (int)sizeof(nmfx_problem));
    if (nrhs < 1 || !mxIsChar(prhs[0]) || mxGetString(prhs[0], algo, sizeof(algo)) != 0) FAIL("usage", "nmfx_mex(algo, ...): algo must be a string");
    if (!strcmp(algo, "nmf") || !strcmp(algo, "cnmf") || !strcmp(algo, "lnmf") || !strcmp(algo, "nmfsc") || !strcmp(algo, "cnmfsc")) {
        factorise(algo, nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(algo, "constrainednmf")) {
        constrained(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(algo, "reconstruct")) {               /* V_hat = ReconstructFromDecomposition(W, H) */
        const mwSize *dw;
        mwSize m, K, T, n;
        if (nrhs != 3 || nlhs > 1) FAIL("usage", "V_hat = nmfx_mex('reconstruct', W, H)");
        if (!is_real_double(prhs[1]) || !is_real_double(prhs[2])) FAIL("type", "W and H must be real double arrays");
        dw = mxGetDimensions(prhs[1]);
        m = dw[0]; K = mxGetM(prhs[2]); n = mxGetN(prhs[2]);
        if (m < 1 || K < 1 || n < 1 || mxGetNumberOfElements(prhs[1]) % (m * K) != 0) FAIL("shape", "W must be m x K (x T) with K = size(H, 1)");
        T = mxGetNumberOfElements(prhs[1]) / (m * K);
        plhs[0] = mxCreateDoubleMatrix(m, n, mxREAL);
        if (nmfx_reconstruct((int64_t)m, (int64_t)n, (int32_t)K, (int32_t)T, NMFX_F64, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(plhs[0]), 0) != NMFX_OK)
            FAIL("error", "%s", nmfx_last_error());
    } else if (!strcmp(algo, "projfunc")) {                  /* [v, usediters] = projfunc(s, k1, k2, nn) */
        mxArray *it;
        mwSize N, count;
        if (nrhs != 5 || nlhs > 2) FAIL("usage", "[v, usediters] = nmfx_mex('projfunc', s, k1, k2, nn)");
        if (!is_real_double(prhs[1]) || mxIsEmpty(prhs[1])) FAIL("type", "s must be a non-empty real double array");
        N = mxGetM(prhs[1]); count = mxGetN(prhs[1]);
        plhs[0] = mxCreateDoubleMatrix(N, count, mxREAL);
        it = mxCreateNumericMatrix(count, 1, mxINT32_CLASS, mxREAL);
        if (nmfx_projfunc((int64_t)N, (int32_t)count, NMFX_F64, mxGetPr(prhs[1]), mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), mxGetScalar(prhs[4]) != 0.0,
                          mxGetPr(plhs[0]), (int32_t *)mxGetData(it), 0) != NMFX_OK)
            FAIL("error", "%s", nmfx_last_error());
        if (nlhs > 1) plhs[1] = it;
    } else if (!strcmp(algo, "sortdictionary")) {            /* [W_sorted, H_sorted, order] = SortDictionary(W, H) */
        mxArray *ord;
        mwSize m, K, n = 0;
        int has_H;
        if (nrhs != 3 || nlhs > 3) FAIL("usage", "[Ws, Hs, order] = nmfx_mex('sortdictionary', W, H)");
        if (!is_real_double(prhs[1])) FAIL("type", "W must be a real double matrix");
        has_H = !mxIsEmpty(prhs[2]);
        if (has_H && !is_real_double(prhs[2])) FAIL("type", "H must be a real double matrix or []");
        m = mxGetM(prhs[1]); K = mxGetN(prhs[1]);
        if (has_H) { n = mxGetN(prhs[2]); if (mxGetM(prhs[2]) != K) FAIL("shape", "size(H, 1) must equal size(W, 2)"); }
        plhs[0] = mxCreateDoubleMatrix(m, K, mxREAL);
        if (nlhs > 1) plhs[1] = mxCreateDoubleMatrix(has_H ? K : 0, has_H ? n : 0, mxREAL);
        ord = mxCreateNumericMatrix(1, K, mxINT32_CLASS, mxREAL);
        if (nmfx_sort_dictionary((int64_t)m, (int32_t)K, (int64_t)n, NMFX_F64, mxGetPr(prhs[1]), (has_H && nlhs > 1) ? mxGetPr(prhs[2]) : NULL, mxGetPr(plhs[0]),
                                 (has_H && nlhs > 1) ? mxGetPr(plhs[1]) : NULL, (int32_t *)mxGetData(ord), 0) != NMFX_OK)
            FAIL("error", "%s", nmfx_last_error());
        if (nlhs > 2) plhs[2] = ord;
    } else {
        FAIL("algo", "unknown algorithm %s", algo);
    }
}
