/* TEST INFRASTRUCTURE.  Declarations of the part of the MATLAB / Octave MEX C API that a MEX shell of this library
 * uses (published API: mex.h / matrix.h of MATLAB's extern/include; names, argument and return types only, no
 * definitions).  tests/test_mex_shell.py feeds the reference's own iLQG_mex.c to `gcc -fsyntax-only` against
 * include/ and this header: the shell compiles unchanged against the headers the product ships.  Never linked. */
#ifndef TEST_MEX_API_H
#define TEST_MEX_API_H
#include <stddef.h>
#include <math.h>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef ptrdiff_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
int mexPrintf(const char *fmt, ...);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);

double *mxGetPr(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
int mxIsStruct(const mxArray *a);
int mxIsDouble(const mxArray *a);
int mxIsSparse(const mxArray *a);
int mxGetNumberOfFields(const mxArray *a);
mxArray *mxGetFieldByNumber(const mxArray *a, mwIndex i, int field);
const char *mxGetFieldNameByNumber(const mxArray *a, int field);
mxArray *mxGetField(const mxArray *a, mwIndex i, const char *name);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
void *mxMalloc(size_t n);
void mxFree(void *p);
int mxIsNaN(double v);
int mxIsInf(double v);
double mxGetInf(void);
#endif
