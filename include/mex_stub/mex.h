/* Minimal stand-in for the MATLAB/Octave "mex.h" so that generated problem
 * files (iLQG_problem.h includes "mex.h", see reference iLQG_problem.tem:6)
 * compile when no MATLAB/Octave is installed.  Only the three calls that the
 * generated header actually uses are provided (iLQG_problem.tem:11-12).
 * When building a real MEX target, put the genuine mex.h earlier on the
 * include path and this file is never seen. */
#ifndef ILQG_MEX_STUB_H
#define ILQG_MEX_STUB_H

#include <math.h>
#include <stdio.h>

#ifndef HAVE_OCTAVE
#define HAVE_OCTAVE 1 /* suppresses the '#include "matrix.h"' in generated code */
#endif

#define mxIsNaN(v) (isnan(v))
#define mxIsInf(v) (isinf(v))
#define mxGetInf() (HUGE_VAL)

#endif
