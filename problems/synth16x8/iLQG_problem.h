/* Problem header for 'Synth16x8' emitted by tools/gen_problem.py. Do not edit.
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

#define N_X 16
#define N_U 8

#define sizeofQxx 136
#define sizeofQuu 36
#define sizeofQxu 128

/* additive hints for the batched backend (absent in Maxima-generated headers,
 * which are then treated as the general case) */
#define ILQG_PROBLEM_NAME "Synth16x8"
#define ILQG_STATE_DEPENDENT_LIMITS 0
#define ILQG_TENSOR_NBASIS 32  /* > 0: iLQG_func.c has the factored tensor tables */
#define ILQG_TENSOR_INIT_WRITES 0  /* init_running() writes constant entries of fxx / fuu / fxu */

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
    double s1_0;
    double s1_1;
    double s1_2;
    double s1_3;
    double s1_4;
    double s1_5;
    double s1_6;
    double s1_7;
    double s1_8;
    double s1_9;
    double s1_10;
    double s1_11;
    double s1_12;
    double s1_13;
    double s1_14;
    double s1_15;
    double s2_0;
    double s2_1;
    double s2_2;
    double s2_3;
    double s2_4;
    double s2_5;
    double s2_6;
    double s2_7;
    double s2_8;
    double s2_9;
    double s2_10;
    double s2_11;
    double s2_12;
    double s2_13;
    double s2_14;
    double s2_15;
#if FULL_DDP
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
