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
/* among the others: the entries that are identically 0 (what a dense back_pass multiplies by zero, matMult.c:3-72) */
#define ILQG_STRUCTURAL_ZERO(X) X(fx, 1) X(fx, 2) X(fx, 3) X(fx, 4) X(fx, 6) X(fx, 7) X(fx, 11) X(fu, 3) X(fu, 4) X(fu, 5) X(fu, 6) X(cx, 2) X(cx, 3) X(cxx, 1) X(cxx, 3) X(cxx, 4) X(cxx, 5) X(cxx, 6) X(cxx, 7) X(cxx, 8) X(cxx, 9) X(cuu, 1) X(cxu, 0) X(cxu, 1) X(cxu, 2) X(cxu, 3) X(cxu, 4) X(cxu, 5) X(cxu, 6) X(cxu, 7)
#if FULL_DDP
#define ILQG_STRUCTURAL_ZERO_FULL(X) X(fxx, 0) X(fxx, 1) X(fxx, 2) X(fxx, 3) X(fxx, 4) X(fxx, 6) X(fxx, 7) X(fxx, 10) X(fxx, 11) X(fxx, 12) X(fxx, 13) X(fxx, 14) X(fxx, 16) X(fxx, 17) X(fxx, 20) X(fxx, 21) X(fxx, 22) X(fxx, 23) X(fxx, 24) X(fxx, 25) X(fxx, 26) X(fxx, 27) X(fxx, 28) X(fxx, 30) X(fxx, 31) X(fxx, 32) X(fxx, 33) X(fxx, 34) X(fxx, 35) X(fxx, 36) X(fxx, 37) X(fxx, 38) X(fxx, 39) X(fuu, 1) X(fuu, 2) X(fuu, 4) X(fuu, 5) X(fuu, 7) X(fuu, 8) X(fuu, 9) X(fuu, 10) X(fuu, 11) X(fxu, 0) X(fxu, 1) X(fxu, 4) X(fxu, 5) X(fxu, 6) X(fxu, 7) X(fxu, 8) X(fxu, 9) X(fxu, 12) X(fxu, 13) X(fxu, 14) X(fxu, 15) X(fxu, 16) X(fxu, 17) X(fxu, 18) X(fxu, 20) X(fxu, 21) X(fxu, 22) X(fxu, 23) X(fxu, 24) X(fxu, 25) X(fxu, 26) X(fxu, 27) X(fxu, 28) X(fxu, 29) X(fxu, 30) X(fxu, 31)
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
