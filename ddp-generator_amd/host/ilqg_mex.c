/* MEX shell of the MI355X iLQG library: [success, x, u, cost] = iLQG<Problem>(x0, u_nom, params, opts).
 *
 * Replaces the reference's iLQG_mex.c:19-144 for a build against libilqg_<problem>_fd<k>_hip.so: the same four
 * inputs and four outputs, the same argument checks and message identifiers.  Everything between argument parsing and
 * the output copies — options by name, parameters by name with their lengths checked, trajectory buffers, init_opt,
 * the initial roll-out, iLQG() — is the library's ilqg_solve_single() (include/ilqg_batch.h), which restates
 * iLQG_mex.c:55-137 behind plain pointers and reports the MEX entry's own messages through `err`.
 * (The reference's iLQG_mex.c itself also compiles unchanged against include/ and links to the same library: the
 * drop-in symbols iLQG(), standard_parameters(), setOptParam(), ... are exported)
 *
 * Build: `make -C ddp-generator_amd/csrc mex` (needs mkoctfile or mex on PATH; make_iLQG.m:61-86 is the reference's
 * compile step).  Syntax-checked without MATLAB/Octave by tests/test_mex_shell.py. */
#include <string.h>

#include "mex.h"
#ifndef HAVE_OCTAVE
#include "matrix.h"
#endif

#include "ilqg_batch.h"

#define ILQG_MEX_MAX_FIELDS 256

static int named_from_struct(const mxArray *s, ilqg_named_t *out, int cap) {
    int i, n = mxGetNumberOfFields(s);
    if(n > cap) n = cap;
    for(i = 0; i < n; i++) {
        const mxArray *v = mxGetFieldByNumber(s, 0, i);
        out[i].name = mxGetFieldNameByNumber(s, i);
        out[i].value = mxGetPr(v);
        out[i].n = (int)mxGetNumberOfElements(v);
    }
    return n;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    int dims[8], n, m, N, i, np, no, rc, iters = 0;
    double secs = 0.0;
    char err[512];
    static ilqg_named_t par[ILQG_MEX_MAX_FIELDS], opt[ILQG_MEX_MAX_FIELDS];

    if(nrhs != 4) { mexErrMsgIdAndTxt("MATLAB:minrhs", "wrong number of arguments (expected: x0, u_nom, params, opt_params)"); return; }
    if(nlhs != 4) { mexErrMsgIdAndTxt("MATLAB:minlhs", "wrong number of return values (expected: success, x_new, u_new, new_cost)"); return; }

    ilqg_problem_dims(dims); /* N_X, N_U, FULL_DDP, ... of the library this shell is linked to */
    n = (int)mxGetNumberOfElements(prhs[0]);
    m = (int)mxGetM(prhs[1]);
    N = (int)mxGetN(prhs[1]); /* time steps (the reference's N - 1) */
    if(n != dims[0]) { mexErrMsgIdAndTxt("MATLAB:dimagree", "wrong number of states (%d expected)", dims[0]); return; }
    if(m != dims[1]) { mexErrMsgIdAndTxt("MATLAB:dimagree", "wrong number of inputs (%d expected)", dims[1]); return; }
    if((int)mxGetNumberOfElements(prhs[1]) != m * N) { mexErrMsgIdAndTxt("MATLAB:dimagree", "wrong number of elements in u_nom (%dx%d expected)", m, N); return; }
    if(!mxIsStruct(prhs[2]) || mxGetNumberOfElements(prhs[2]) != 1) { mexErrMsgIdAndTxt("MATLAB:dimagree", "Input 3 must be a scalar struct.\n"); return; }
    if(!mxIsStruct(prhs[3]) || mxGetNumberOfElements(prhs[3]) != 1) { mexErrMsgIdAndTxt("MATLAB:dimagree", "Input 4 must be a scalar struct of optimization parameters.\n"); return; }
    for(i = 0; i < mxGetNumberOfFields(prhs[2]); i++) { /* iLQG_mex.c:78: parameters are real dense vectors */
        const mxArray *v = mxGetFieldByNumber(prhs[2], 0, i);
        if(mxIsSparse(v) || !mxIsDouble(v) || (mxGetM(v) != 1 && mxGetN(v) != 1)) {
            mexErrMsgIdAndTxt("MATLAB:dimagree", "Parameter name '%s' must be a vector.\n", mxGetFieldNameByNumber(prhs[2], i));
            return;
        }
    }
    np = named_from_struct(prhs[2], par, ILQG_MEX_MAX_FIELDS);
    no = named_from_struct(prhs[3], opt, ILQG_MEX_MAX_FIELDS);

    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    plhs[1] = mxCreateDoubleMatrix(n, N + 1, mxREAL);
    plhs[2] = mxCreateDoubleMatrix(m, N, mxREAL);
    plhs[3] = mxCreateDoubleMatrix(1, 1, mxREAL);

    err[0] = 0;
    rc = ilqg_solve_single(N, mxGetPr(prhs[0]), mxGetPr(prhs[1]), par, np, opt, no, mxGetPr(plhs[1]), mxGetPr(plhs[2]),
                           mxGetPr(plhs[3]), &iters, &secs, err, (int)sizeof err);
    if(rc < 0) { mexErrMsgIdAndTxt("MATLAB:dimagree", "%s\n", err); return; } /* the reference's messages, iLQG_mex.c:62-84 */
    mxGetPr(plhs[0])[0] = (double)rc;
    mexPrintf("Time for iLQG: %f seconds\n", secs);
}
