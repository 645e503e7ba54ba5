/*
 * tests/stubs/mex.h -- TEST INFRASTRUCTURE, not MATLAB's header.
 *
 * A mock of the small subset of MATLAB's documented C Matrix / MEX API (R2018a interleaved-complex API: mxGetDoubles ...)
 * that THIS repository's gateways use (the .c files under sph-poiseuille-flow_amd/matlab).  It exists so that those three files can be
 * compiled with -Wall -Werror and driven by tests/test_matlab_gateways.py in an image that has no MATLAB: argument
 * unpacking, arity / shape checks, error identifiers and the calls into libsphx are then exercised for real, through
 * tests/stubs/mex_mock.c.  Nothing of the reference is built with it, and passing these tests says nothing about MATLAB
 * itself -- a maintainer still builds the gateways with `mex` (ensure_sphx_mex_compiled.m).
 * Prototypes follow the public documentation of the API (names, argument order and types).
 */
#ifndef SPHX_TEST_MEX_H
#define SPHX_TEST_MEX_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum {
    mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
    mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
    mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS
} mxClassID;

double *mxGetDoubles(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
bool mxIsDouble(const mxArray *a);
bool mxIsChar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames);
void mxSetField(mxArray *a, mwIndex index, const char *fieldname, mxArray *value);
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *fieldname);
void mexErrMsgIdAndTxt(const char *identifier, const char *fmt, ...);
void mexLock(void);
void mexUnlock(void);

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_TEST_MEX_H */
