/* Minimal MEX runtime for tests/test_mex_gateway.py -- TEST INFRASTRUCTURE ONLY (see mex.h next to this file).
 * Arrays are never freed individually; mock_reset() releases everything a test created. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

#define MAXF 16
struct mxArray_tag {
    mxClassID cls;
    mwSize ndim;
    mwSize dims[4];
    void *data;                 /* numeric / char payload (column-major) */
    int nfields;                /* struct */
    char fname[MAXF][32];
    mxArray *fval[MAXF];
    struct mxArray_tag *next;   /* allocation list */
};

static mxArray *g_all = NULL;
static jmp_buf g_jmp;
static int g_armed = 0;
static char g_err[1024];

static size_t esize(mxClassID c) {
    switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;
    case mxINT8_CLASS: case mxUINT8_CLASS: case mxLOGICAL_CLASS: return 1;
    default: return 0;
    }
}
static mxArray *alloc_array(mxClassID c, mwSize ndim, const mwSize *dims) {
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    size_t n = 1;
    mwSize i;
    a->cls = c;
    a->ndim = ndim < 2 ? 2 : ndim;
    a->dims[0] = a->dims[1] = a->dims[2] = a->dims[3] = 1;
    for (i = 0; i < ndim && i < 4; ++i) { a->dims[i] = dims[i]; }
    for (i = 0; i < a->ndim; ++i) n *= a->dims[i];
    a->data = (n && esize(c)) ? calloc(n, esize(c)) : NULL;
    a->next = g_all;
    g_all = a;
    return a;
}

size_t mxGetM(const mxArray *a) { return a->dims[0]; }
size_t mxGetN(const mxArray *a) { size_t n = 1; mwSize i; for (i = 1; i < a->ndim; ++i) n *= a->dims[i]; return n; }   /* product of the trailing dims, as documented */
size_t mxGetNumberOfElements(const mxArray *a) { size_t n = 1; mwSize i; for (i = 0; i < a->ndim; ++i) n *= a->dims[i]; return n; }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->ndim; }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims; }
double *mxGetPr(const mxArray *a) { return a->cls == mxDOUBLE_CLASS ? (double *)a->data : NULL; }
void *mxGetData(const mxArray *a) { return a->data; }
double mxGetScalar(const mxArray *a) {
    if (!a->data) return 0.0;
    switch (a->cls) {
    case mxDOUBLE_CLASS: return *(double *)a->data;
    case mxSINGLE_CLASS: return *(float *)a->data;
    case mxINT32_CLASS: return *(int32_t *)a->data;
    case mxUINT32_CLASS: return *(uint32_t *)a->data;
    case mxINT64_CLASS: return (double)*(int64_t *)a->data;
    case mxUINT8_CLASS: case mxLOGICAL_CLASS: return *(uint8_t *)a->data;
    default: return 0.0;
    }
}
int mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
int mxIsComplex(const mxArray *a) { (void)a; return 0; }
int mxIsClass(const mxArray *a, const char *name) {
    static const struct { const char *n; mxClassID c; } t[] = {{"double", mxDOUBLE_CLASS}, {"single", mxSINGLE_CLASS}, {"int32", mxINT32_CLASS}, {"uint32", mxUINT32_CLASS},
        {"int64", mxINT64_CLASS}, {"uint64", mxUINT64_CLASS}, {"uint8", mxUINT8_CLASS}, {"int8", mxINT8_CLASS}, {"logical", mxLOGICAL_CLASS}, {"char", mxCHAR_CLASS},
        {"struct", mxSTRUCT_CLASS}, {"cell", mxCELL_CLASS}};
    size_t i;
    for (i = 0; i < sizeof(t) / sizeof(t[0]); ++i) if (!strcmp(name, t[i].n)) return a->cls == t[i].c;
    return 0;
}
int mxIsEmpty(const mxArray *a) { return mxGetNumberOfElements(a) == 0; }
int mxIsStruct(const mxArray *a) { return a->cls == mxSTRUCT_CLASS; }
int mxIsChar(const mxArray *a) { return a->cls == mxCHAR_CLASS; }
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {
    size_t n = mxGetNumberOfElements(a), i;
    if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
    for (i = 0; i < n && i + 1 < buflen; ++i) buf[i] = (char)((uint16_t *)a->data)[i];
    buf[i] = 0;
    return n + 1 > buflen;      /* 1 = truncated, as documented */
}
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *f) {
    int i;
    if (a->cls != mxSTRUCT_CLASS || index != 0) return NULL;
    for (i = 0; i < a->nfields; ++i) if (!strcmp(a->fname[i], f)) return a->fval[i];
    return NULL;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) { mwSize d[2]; (void)flag; d[0] = m; d[1] = n; return alloc_array(mxDOUBLE_CLASS, 2, d); }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity flag) { mwSize d[2]; (void)flag; d[0] = m; d[1] = n; return alloc_array(c, 2, d); }
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID c, mxComplexity flag) { (void)flag; return alloc_array(c, ndim, dims); }
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL); *(double *)a->data = v; return a; }
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names) {
    mwSize d[2];
    mxArray *a;
    int i;
    d[0] = m; d[1] = n;
    a = alloc_array(mxSTRUCT_CLASS, 2, d);
    for (i = 0; i < nfields && i < MAXF; ++i) { strncpy(a->fname[i], names[i], 31); a->fval[i] = NULL; }
    a->nfields = nfields < MAXF ? nfields : MAXF;
    return a;
}
void mxSetField(mxArray *a, mwIndex index, const char *f, mxArray *v) {
    int i;
    if (a->cls != mxSTRUCT_CLASS || index != 0) return;
    for (i = 0; i < a->nfields; ++i) if (!strcmp(a->fname[i], f)) { a->fval[i] = v; return; }
    if (a->nfields < MAXF) { strncpy(a->fname[a->nfields], f, 31); a->fval[a->nfields++] = v; }   /* harness convenience: add the field */
}
void mxSetM(mxArray *a, mwSize m) { a->dims[0] = m; }
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    va_list ap;
    size_t k = (size_t)snprintf(g_err, sizeof(g_err), "%s: ", id);
    va_start(ap, fmt);
    vsnprintf(g_err + k, sizeof(g_err) - k, fmt, ap);
    va_end(ap);
    if (g_armed) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside mock_call: %s\n", g_err);
    abort();
}

/* ---- harness API (driven from Python through ctypes) ---- */
mxArray *mock_numeric(int classid, int ndim, const size_t *dims, const void *data) {
    mxArray *a = alloc_array((mxClassID)classid, (mwSize)ndim, dims);
    size_t n = mxGetNumberOfElements(a);
    if (data && n) memcpy(a->data, data, n * esize(a->cls));
    return a;
}
mxArray *mock_string(const char *s) {
    mwSize d[2];
    mxArray *a;
    size_t i;
    d[0] = 1; d[1] = strlen(s);
    a = alloc_array(mxCHAR_CLASS, 2, d);
    for (i = 0; i < d[1]; ++i) ((uint16_t *)a->data)[i] = (uint16_t)(unsigned char)s[i];
    return a;
}
mxArray *mock_struct(void) { return mxCreateStructMatrix(1, 1, 0, NULL); }
int mock_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {   /* 0 = returned normally, 1 = mexErrMsgIdAndTxt */
    int rc;
    g_err[0] = 0;
    g_armed = 1;
    if (setjmp(g_jmp) == 0) { mexFunction(nlhs, plhs, nrhs, prhs); rc = 0; }
    else rc = 1;
    g_armed = 0;
    return rc;
}
const char *mock_error(void) { return g_err; }
int mock_class(const mxArray *a) { return (int)a->cls; }
void mock_reset(void) {
    while (g_all) { mxArray *nx = g_all->next; free(g_all->data); free(g_all); g_all = nx; }
}
