/* Problem header for 'BrachiHli' emitted by tools/gen_problem.py. Do not edit.
 * Layout contract: reference iLQG_problem.tem:16-89. */
#ifndef ILQG_PROBLEM_H
#define ILQG_PROBLEM_H

#include <math.h>
#include "mex.h"
#ifndef  HAVE_OCTAVE
#include "matrix.h"
#endif

#define isNANorINF(v) (mxIsNaN(v) || mxIsInf(v))
#define INF mxGetInf()

#define N_X 1
#define N_U 1

#define sizeofQxx 1
#define sizeofQuu 1
#define sizeofQxu 1

/* additive hints for the batched backend (absent in Maxima-generated headers,
 * which are then treated as the general case) */
#define ILQG_PROBLEM_NAME "BrachiHli"
#define ILQG_STATE_DEPENDENT_LIMITS 0
#define ILQG_TENSOR_NBASIS 0  /* > 0: iLQG_func.c has the factored tensor tables */
#define ILQG_TENSOR_INIT_WRITES 1  /* init_running() writes constant entries of fxx / fuu / fxu */
/* the derivative entries bp_derivsL() writes, X(member, index) each: all others are written once, by init_running() */
#define ILQG_TIME_VARYING(X) X(cx, 0) X(cxx, 0) X(cu, 0) X(cuu, 0) X(cxu, 0)
#if FULL_DDP
#define ILQG_TIME_VARYING_FULL(X) 
#else
#define ILQG_TIME_VARYING_FULL(X)
#endif
/* among the others: the entries that are identically 0 (what a dense back_pass multiplies by zero, matMult.c:3-72) */
#define ILQG_STRUCTURAL_ZERO(X) 
#if FULL_DDP
#define ILQG_STRUCTURAL_ZERO_FULL(X) X(fxx, 0) X(fuu, 0) X(fxu, 0)
#else
#define ILQG_STRUCTURAL_ZERO_FULL(X)
#endif

typedef struct {
    double x[N_X];
    double u[N_U];
    double lower[N_U];
    double upper[N_U];
    double lower_sign[N_U];
    double upper_sign[N_U];
    double lower_hx[N_X*N_U];
    double upper_hx[N_X*N_U];

    double l[N_U];
    double L[N_U*N_X];
    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
    double cu[N_U];
    double cuu[sizeofQuu];
    double cxu[sizeofQxu];
    double fx[N_X*N_X];
    double fu[N_X*N_U];
#if FULL_DDP
    double fxx[N_X*sizeofQxx];
    double fuu[N_X*sizeofQuu];
    double fxu[N_X*sizeofQxu];
#endif
    double hli_1;
    double pli_1;
    double dpli_1_x0;
    double dpli_1_x0x0;
#if FULL_DDP
#endif
} trajEl_t;

typedef struct {
    double x[N_X];

    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
    double hfe_1;
    double pfe_1;
    double dpfe_1_x0;
    double dpfe_1_x0x0;
} trajFin_t;

typedef struct {
    trajEl_t* t;
    trajFin_t f;
} traj_t;

typedef struct {
    double mu_li[1];
    double last_hli[1];
} multipliersEl_t;

typedef struct {
    double mu_fe[1];
    double last_hfe[1];
} multipliersFin_t;

typedef struct {
    multipliersEl_t* t;
    multipliersFin_t f;
} multipliers_t;

#endif // ILQG_PROBLEM_H
