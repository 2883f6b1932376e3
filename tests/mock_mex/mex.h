/* Minimal stand-in for MATLAB's mex.h / matrix.h -- TEST INFRASTRUCTURE ONLY (tests/test_mex_gateway.py).
 * It declares exactly the subset of the documented MEX C API that matlab/nmfx_mex.c uses, with the documented signatures, so the
 * gateway can be compiled with -Wall -Wextra -Werror and executed here; mock_mex.c implements it over a plain struct.
 * It is NOT MATLAB: a green run shows the gateway is well-formed C that drives libnmfx correctly, not that MATLAB accepts it. */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef enum {
    mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS, mxSINGLE_CLASS,
    mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS
} mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
int mxIsDouble(const mxArray *a);
int mxIsComplex(const mxArray *a);
int mxIsClass(const mxArray *a, const char *classname);
int mxIsEmpty(const mxArray *a);
int mxIsStruct(const mxArray *a);
int mxIsChar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *fieldname);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames);
void mxSetField(mxArray *a, mwIndex index, const char *fieldname, mxArray *value);
void mxSetM(mxArray *a, mwSize m);
void mexErrMsgIdAndTxt(const char *errorid, const char *errormsg, ...);   /* does not return (longjmp into the harness) */

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

#ifdef __cplusplus
}
#endif
#endif
