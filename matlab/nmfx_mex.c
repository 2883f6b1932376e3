/* nmfx_mex.c -- thin MEX gateway from MATLAB to the C ABI of libnmfx (include/nmfx.h).
 *
 * SOURCE ONLY: neither MATLAB nor Octave exists in the build image, so this file has never been compiled; it is the
 * binding a maintainer of the toolbox would add.  It does no numerics and no validation beyond
 * shapes: the .m wrappers next to it keep the reference signatures and run the reference's local ValidateParameters
 * logic before calling in here.
 *
 *   [W, H, cost, info] = nmfx_mex(algo, V, W_init, H_init, K_s, T, opts)
 *     algo   : 'nmf' | 'cnmf' | 'nmfsc' | 'cnmfsc' | 'lnmf'
 *              (constrainednmf, SortDictionary, projfunc and ReconstructFromDecomposition take other argument lists: their
 *               gateways are the same dozen lines around nmfx_constrainednmf / nmfx_sort_dictionary / nmfx_projfunc /
 *               nmfx_reconstruct)
 *     V      : m x n double          W_init : m x K x T double        H_init : K x n double
 *     K_s    : 1 x S int32 (basis elements per source, sum = K)
 *     opts   : struct with fields divergence (int32 nmfx_divergence), alpha, beta, W_sparsity, H_sparsity (1 x S double),
 *              W_fixed, H_fixed (1 x S uint8), maxiter (int32), tolerance (double), device (int32),
 *              sc_W_sparsity, sc_H_sparsity (double, nmfsc only)
 *
 * build (on a machine with MATLAB):  mex -I../include nmfx_mex.c -L../nmf_toolbox_amd -lnmfx
 */
#include <string.h>

#include "mex.h"
#include "nmfx.h"

static double opt_d(const mxArray *o, const char *f, double dflt) {
    const mxArray *a = mxGetField(o, 0, f);
    return a ? mxGetScalar(a) : dflt;
}
static const void *opt_p(const mxArray *o, const char *f) {
    const mxArray *a = mxGetField(o, 0, f);
    return (a && !mxIsEmpty(a)) ? mxGetData(a) : NULL;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char algo[16];
    nmfx_problem p;
    nmfx_result r;
    nmfx_status st;
    mwSize dimsW[3];
    if (nrhs != 7) mexErrMsgIdAndTxt("nmfx:usage", "nmfx_mex(algo, V, W_init, H_init, K_s, T, opts)");
    mxGetString(prhs[0], algo, sizeof(algo));
    memset(&p, 0, sizeof(p));
    memset(&r, 0, sizeof(r));
    p.m = (int64_t)mxGetM(prhs[1]);
    p.n = (int64_t)mxGetN(prhs[1]);
    p.K_total = (int32_t)mxGetM(prhs[3]);
    p.T = (int32_t)mxGetScalar(prhs[5]);
    p.dtype = NMFX_F64;
    p.V = mxGetPr(prhs[1]);
    p.W_init = mxGetPr(prhs[2]);
    p.H_init = mxGetPr(prhs[3]);
    p.num_sources = (int32_t)mxGetNumberOfElements(prhs[4]);
    p.K_s = (const int32_t *)mxGetData(prhs[4]);
    p.divergence = (int32_t)opt_d(prhs[6], "divergence", NMFX_DIV_EUCLIDEAN);
    p.alpha = opt_d(prhs[6], "alpha", 1.0);
    p.beta = opt_d(prhs[6], "beta", 1.0);
    p.W_sparsity = (const double *)opt_p(prhs[6], "W_sparsity");
    p.H_sparsity = (const double *)opt_p(prhs[6], "H_sparsity");
    p.W_fixed = (const uint8_t *)opt_p(prhs[6], "W_fixed");
    p.H_fixed = (const uint8_t *)opt_p(prhs[6], "H_fixed");
    p.maxiter = (int32_t)opt_d(prhs[6], "maxiter", 100);
    p.tolerance = opt_d(prhs[6], "tolerance", 1e-3);
    p.device = (int32_t)opt_d(prhs[6], "device", 0);
    p.sc_W_sparsity = opt_d(prhs[6], "sc_W_sparsity", 0.0);
    p.sc_H_sparsity = opt_d(prhs[6], "sc_H_sparsity", 0.0);

    dimsW[0] = (mwSize)p.m; dimsW[1] = (mwSize)p.K_total; dimsW[2] = (mwSize)p.T;
    plhs[0] = mxCreateNumericArray(p.T > 1 ? 3 : 2, dimsW, mxDOUBLE_CLASS, mxREAL);
    plhs[1] = mxCreateDoubleMatrix((mwSize)p.K_total, (mwSize)p.n, mxREAL);
    plhs[2] = mxCreateDoubleMatrix((mwSize)p.maxiter + 1, 1, mxREAL);
    r.W = mxGetPr(plhs[0]);
    r.H = mxGetPr(plhs[1]);
    r.cost = mxGetPr(plhs[2]);

    if (!strcmp(algo, "nmf")) st = nmfx_nmf(&p, &r);
    else if (!strcmp(algo, "cnmf")) st = nmfx_cnmf(&p, &r);
    else if (!strcmp(algo, "nmfsc")) st = nmfx_nmfsc(&p, &r);
    else if (!strcmp(algo, "cnmfsc")) st = nmfx_cnmfsc(&p, &r);
    else if (!strcmp(algo, "lnmf")) st = nmfx_lnmf(&p, &r);
    else { mexErrMsgIdAndTxt("nmfx:algo", "unknown algorithm %s", algo); return; }
    if (st != NMFX_OK) mexErrMsgIdAndTxt("nmfx:error", "%s", nmfx_last_error());   /* same text the reference's error() uses */
    mxSetM(plhs[2], (mwSize)r.cost_len);                                         /* cost = cost(1:iter) trim (nmf.m:222) */
    if (nlhs > 3) {
        const char *f[] = {"iters_run", "stepsize_H", "stepsize_W", "converged_early"};
        plhs[3] = mxCreateStructMatrix(1, 1, 4, f);
        mxSetField(plhs[3], 0, "iters_run", mxCreateDoubleScalar(r.iters_run));
        mxSetField(plhs[3], 0, "stepsize_H", mxCreateDoubleScalar(r.stepsize_H));
        mxSetField(plhs[3], 0, "stepsize_W", mxCreateDoubleScalar(r.stepsize_W));
        mxSetField(plhs[3], 0, "converged_early", mxCreateDoubleScalar(r.converged_early));
    }
}
