/* Minimal stand-in for MATLAB's mex.h -- TEST INFRASTRUCTURE ONLY (tests/test_host_cpu.py uses it to
 * syntax-check matlab/mbfir_mex.c and matlab/mbfir_slr_mex.c where no MATLAB exists).  Declares only what the gateway calls; it is
 * not used to build anything that runs. */
#ifndef MBFIR_STUB_MEX_H
#define MBFIR_STUB_MEX_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef struct { double real, imag; } mxComplexDouble;
#ifndef MX_HAS_INTERLEAVED_COMPLEX
#define MX_HAS_INTERLEAVED_COMPLEX 0
#endif
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
double mxGetScalar(const mxArray*);
int mxIsComplex(const mxArray*);
int mxIsEmpty(const mxArray*);
double* mxGetPr(const mxArray*);
double* mxGetPi(const mxArray*);
double* mxGetDoubles(const mxArray*);
mxComplexDouble* mxGetComplexDoubles(const mxArray*);
mxArray* mxCreateDoubleMatrix(size_t, size_t, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
void* mxCalloc(size_t, size_t);
void mxFree(void*);
void mexErrMsgTxt(const char*);
int mexAtExit(void (*)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#endif
