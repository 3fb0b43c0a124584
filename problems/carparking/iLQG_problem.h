/* Problem header for 'CarParking' emitted by tools/gen_problem.py. Do not edit.
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

#define N_X 4
#define N_U 2

#define sizeofQxx 10
#define sizeofQuu 3
#define sizeofQxu 8

/* additive hints for the batched backend (absent in Maxima-generated headers,
 * which are then treated as the general case) */
#define ILQG_PROBLEM_NAME "CarParking"
#define ILQG_STATE_DEPENDENT_LIMITS 0
#define ILQG_TENSOR_NBASIS 0  /* > 0: iLQG_func.c has the factored tensor tables */
#define ILQG_TENSOR_INIT_WRITES 1  /* init_running() writes constant entries of fxx / fuu / fxu */
/* the derivative entries bp_derivsL() writes, X(member, index) each: all others are written once, by init_running() */
#define ILQG_TIME_VARYING(X) X(fx, 8) X(fx, 9) X(fx, 12) X(fx, 13) X(fx, 14) X(fu, 0) X(fu, 1) X(fu, 2) X(cx, 0) X(cx, 1) X(cxx, 0) X(cxx, 2) X(cu, 0) X(cu, 1)
#if FULL_DDP
#define ILQG_TIME_VARYING_FULL(X) X(fxx, 5) X(fxx, 8) X(fxx, 9) X(fxx, 15) X(fxx, 18) X(fxx, 19) X(fxx, 29) X(fuu, 0) X(fuu, 3) X(fuu, 6) X(fxu, 2) X(fxu, 3) X(fxu, 10) X(fxu, 11) X(fxu, 19)
#else
#define ILQG_TIME_VARYING_FULL(X)
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
    double s;
    double ds_x3;
    double ds_u0;
#if FULL_DDP
    double ds_x3x3;
    double ds_u0u0;
    double ds_u0x3;
#endif
} trajEl_t;

typedef struct {
    double x[N_X];

    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
} trajFin_t;

typedef struct {
    trajEl_t* t;
    trajFin_t f;
} traj_t;

typedef struct {
} multipliersEl_t;

typedef struct {
} multipliersFin_t;

typedef struct {
    multipliersEl_t* t;
    multipliersFin_t f;
} multipliers_t;

#endif // ILQG_PROBLEM_H
