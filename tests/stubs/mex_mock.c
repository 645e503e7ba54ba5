/*
 * tests/stubs/mex_mock.c -- TEST INFRASTRUCTURE: a minimal runtime behind tests/stubs/mex.h (see there), linked into one
 * shared object per gateway by tests/mex_mock.py.  Arrays are heap blocks on one list (mock_free_all); mexErrMsgIdAndTxt
 * records identifier + message and long-jumps back to mock_call, as MATLAB returns to the prompt.
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

struct mxArray_tag {
    mxClassID cls;
    size_t m, n;
    void *data;            /* doubles, chars or one uint64 */
    int nfields;
    char **names;
    struct mxArray_tag **fields;
    struct mxArray_tag *next;
};

static mxArray *g_all = NULL;
static jmp_buf g_jmp;
static int g_in_call = 0, g_locks = 0;
static char g_id[256], g_msg[2048];

static mxArray *fresh(mxClassID cls, size_t m, size_t n, size_t elem)
{
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    a->cls = cls; a->m = m; a->n = n;
    a->data = calloc(m * n > 0 ? m * n : 1, elem);
    a->next = g_all; g_all = a;
    return a;
}

/* ---- the API of mex.h ---- */
double *mxGetDoubles(const mxArray *a) { return (double *)a->data; }
void *mxGetData(const mxArray *a) { return a->data; }
double mxGetScalar(const mxArray *a)
{
    if (a->cls == mxUINT64_CLASS) return (double)*(uint64_t *)a->data;
    if (a->cls == mxCHAR_CLASS) return (double)((char *)a->data)[0];
    return ((double *)a->data)[0];
}
size_t mxGetM(const mxArray *a) { return a->m; }
size_t mxGetN(const mxArray *a) { return a->n; }
size_t mxGetNumberOfElements(const mxArray *a) { return a->m * a->n; }
bool mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsChar(const mxArray *a) { return a->cls == mxCHAR_CLASS; }
int mxGetString(const mxArray *a, char *buf, mwSize buflen)
{
    const size_t len = a->m * a->n;
    if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
    if (len + 1 > buflen) { memcpy(buf, a->data, buflen - 1); buf[buflen - 1] = 0; return 1; }
    memcpy(buf, a->data, len); buf[len] = 0;
    return 0;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) { (void)flag; return fresh(mxDOUBLE_CLASS, m, n, sizeof(double)); }
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = fresh(mxDOUBLE_CLASS, 1, 1, sizeof(double)); *(double *)a->data = v; return a; }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag)
{
    (void)flag;
    return fresh(cls, m, n, cls == mxDOUBLE_CLASS || cls == mxUINT64_CLASS || cls == mxINT64_CLASS ? 8 : 4);
}
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names)
{
    mxArray *a = fresh(mxSTRUCT_CLASS, m, n, 1);
    int k;
    a->nfields = nfields;
    a->names = (char **)calloc((size_t)nfields, sizeof(char *));
    a->fields = (mxArray **)calloc((size_t)nfields, sizeof(mxArray *));
    for (k = 0; k < nfields; ++k) { a->names[k] = (char *)malloc(strlen(names[k]) + 1); strcpy(a->names[k], names[k]); }
    return a;
}
void mxSetField(mxArray *a, mwIndex index, const char *name, mxArray *value)
{
    int k;
    (void)index;
    for (k = 0; k < a->nfields; ++k) if (strcmp(a->names[k], name) == 0) { a->fields[k] = value; return; }
}
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *name)
{
    int k;
    (void)index;
    if (a->cls != mxSTRUCT_CLASS) return NULL;
    for (k = 0; k < a->nfields; ++k) if (strcmp(a->names[k], name) == 0) return a->fields[k];
    return NULL;
}
void mexErrMsgIdAndTxt(const char *identifier, const char *fmt, ...)
{
    va_list ap;
    snprintf(g_id, sizeof(g_id), "%s", identifier ? identifier : "");
    va_start(ap, fmt);
    vsnprintf(g_msg, sizeof(g_msg), fmt, ap);
    va_end(ap);
    if (g_in_call) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside mock_call: [%s] %s\n", g_id, g_msg);
    abort();
}
void mexLock(void) { ++g_locks; }
void mexUnlock(void) { --g_locks; }

/* ---- what the Python harness calls ---- */
mxArray *mock_doubles(size_t m, size_t n, const double *src)
{
    mxArray *a = fresh(mxDOUBLE_CLASS, m, n, sizeof(double));
    if (src && m * n > 0) memcpy(a->data, src, m * n * sizeof(double));
    return a;
}
mxArray *mock_string(const char *s)
{
    const size_t len = strlen(s);
    mxArray *a = fresh(mxCHAR_CLASS, 1, len, 1);
    memcpy(a->data, s, len);
    return a;
}
mxArray *mock_uint64(uint64_t v) { mxArray *a = fresh(mxUINT64_CLASS, 1, 1, 8); *(uint64_t *)a->data = v; return a; }
mxArray *mock_struct(int nfields, const char **names, const double *values)
{
    mxArray *a = mxCreateStructMatrix(1, 1, nfields, names);
    int k;
    for (k = 0; k < nfields; ++k) a->fields[k] = mxCreateDoubleScalar(values[k]);
    return a;
}
int mock_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs)
{
    g_id[0] = 0; g_msg[0] = 0;
    if (setjmp(g_jmp)) { g_in_call = 0; return 1; }
    g_in_call = 1;
    mexFunction(nlhs, plhs, nrhs, prhs);
    g_in_call = 0;
    return 0;
}
const char *mock_error_id(void) { return g_id; }
const char *mock_error_msg(void) { return g_msg; }
int mock_class(const mxArray *a) { return (int)a->cls; }
size_t mock_m(const mxArray *a) { return a->m; }
size_t mock_n(const mxArray *a) { return a->n; }
void *mock_data(const mxArray *a) { return a->data; }
int mock_nfields(const mxArray *a) { return a->nfields; }
const char *mock_field_name(const mxArray *a, int k) { return a->names[k]; }
mxArray *mock_field(const mxArray *a, int k) { return a->fields[k]; }
int mock_lock_count(void) { return g_locks; }
void mock_free_all(void)
{
    while (g_all) {
        mxArray *a = g_all;
        int k;
        g_all = a->next;
        for (k = 0; k < a->nfields; ++k) free(a->names[k]);
        free(a->names); free(a->fields); free(a->data); free(a);
    }
}
