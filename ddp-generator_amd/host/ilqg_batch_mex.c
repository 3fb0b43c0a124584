/* MEX shell of the BATCH entry (additive; no counterpart in the reference):
 *     [success, x, u, cost] = iLQGbatch<Problem>(x0[n,B], u_nom[m,N,B], params, opts)
 * B independent trajectories, all state resident on the GPU (include/ilqg_batch.h).  MATLAB's column-major x0(n,B),
 * u_nom(m,N,B) are exactly the trajectory-major host layout the C-ABI takes: no transposition.  Options and
 * parameters by name as in the reference's MEX entry (iLQG_mex.c:60-84), same keys, same messages.
 * success(b) is the reference's iLQG() return value for trajectory b (iLQG.c:365-378).
 * Build: `make -C ddp-generator_amd/csrc mex`. */
#include "mex.h"
#ifndef HAVE_OCTAVE
#include "matrix.h"
#endif

#include "ilqg_batch.h"

static void fail(ilqg_batch_t *c, const char *id) {
    static char msg[512];
    const char *e = ilqg_batch_error(c);
    int i = 0;
    for(; e && e[i] && i < 510; i++) msg[i] = e[i];
    msg[i] = 0;
    if(c) ilqg_batch_destroy(c);
    mexErrMsgIdAndTxt(id, "%s\n", msg);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    int dims[8], n, m, N, B, i, b;
    ilqg_batch_t *c;
    int *status, *iters;
    double *succ;

    if(nrhs != 4) { mexErrMsgIdAndTxt("MATLAB:minrhs", "wrong number of arguments (expected: x0, u_nom, params, opt_params)"); return; }
    if(nlhs != 4) { mexErrMsgIdAndTxt("MATLAB:minlhs", "wrong number of return values (expected: success, x_new, u_new, new_cost)"); return; }
    ilqg_problem_dims(dims);
    n = dims[0];
    m = dims[1];
    if((int)mxGetM(prhs[0]) != n) { mexErrMsgIdAndTxt("MATLAB:dimagree", "wrong number of states (%d expected)", n); return; }
    if((int)mxGetM(prhs[1]) != m) { mexErrMsgIdAndTxt("MATLAB:dimagree", "wrong number of inputs (%d expected)", m); return; }
    B = (int)mxGetN(prhs[0]);
    if(B < 1 || mxGetNumberOfElements(prhs[1]) % ((size_t)m * B) != 0) { mexErrMsgIdAndTxt("MATLAB:dimagree", "u_nom must be %d x N x %d", m, B); return; }
    N = (int)(mxGetNumberOfElements(prhs[1]) / ((size_t)m * B));
    if(!mxIsStruct(prhs[2]) || !mxIsStruct(prhs[3])) { mexErrMsgIdAndTxt("MATLAB:dimagree", "Inputs 3 and 4 must be scalar structs.\n"); return; }

    c = ilqg_batch_create(0, B, N);
    if(!c) { fail(NULL, "iLQG:device"); return; }
    for(i = 0; i < mxGetNumberOfFields(prhs[3]); i++) { /* options by name, iLQG_mex.c:60-67 */
        const mxArray *v = mxGetFieldByNumber(prhs[3], 0, i);
        if(ilqg_batch_set_option(c, mxGetFieldNameByNumber(prhs[3], i), mxGetPr(v), (int)mxGetNumberOfElements(v))) { fail(c, "MATLAB:dimagree"); return; }
    }
    for(i = 0; i < dims[6]; i++) { /* every parameter of the problem, by name, iLQG_mex.c:70-84 */
        const char *name = ilqg_problem_param_name(i);
        const mxArray *v = mxGetField(prhs[2], 0, name);
        if(!v) {
            ilqg_batch_destroy(c);
            mexErrMsgIdAndTxt("MATLAB:dimagree", "Parameter name '%s' is not member of parameters struct.\n", name);
            return;
        }
        if(mxIsSparse(v) || !mxIsDouble(v) || ilqg_batch_set_param(c, name, mxGetPr(v), (int)mxGetNumberOfElements(v))) { fail(c, "MATLAB:dimagree"); return; }
    }
    if(ilqg_batch_set_x0(c, mxGetPr(prhs[0])) || ilqg_batch_set_u(c, mxGetPr(prhs[1]))) { fail(c, "iLQG:device"); return; }
    if(ilqg_batch_init(c) || ilqg_batch_solve(c)) { fail(c, "iLQG:device"); return; }

    plhs[0] = mxCreateDoubleMatrix(1, B, mxREAL);
    plhs[1] = mxCreateDoubleMatrix(n, (mwSize)(N + 1) * B, mxREAL); /* reshape(x, n, N+1, B) */
    plhs[2] = mxCreateDoubleMatrix(m, (mwSize)N * B, mxREAL);
    plhs[3] = mxCreateDoubleMatrix(1, B, mxREAL);
    status = (int *)mxMalloc(sizeof(int) * B);
    iters = (int *)mxMalloc(sizeof(int) * B);
    if(ilqg_batch_get_x(c, mxGetPr(plhs[1])) || ilqg_batch_get_u(c, mxGetPr(plhs[2])) ||
       ilqg_batch_get_scalar(c, "cost", mxGetPr(plhs[3])) || ilqg_batch_get_int(c, "status", status) ||
       ilqg_batch_get_int(c, "iterations", iters)) {
        mxFree(status);
        mxFree(iters);
        fail(c, "iLQG:device");
        return;
    }
    succ = mxGetPr(plhs[0]);
    for(b = 0; b < B; b++) succ[b] = (double)ilqg_reference_success(status[b], iters[b]);
    mxFree(status);
    mxFree(iters);
    ilqg_batch_destroy(c);
}
