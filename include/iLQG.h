/* Solver-side interface of the batched iLQG library.
 *
 * This header is the link-time contract between the solver and a generated
 * problem file (iLQG_problem.h + iLQG_func.c).  It replaces reference
 * iLQG.h:1-108: same macro names, same type and field names, same function
 * prototypes, so a problem file emitted by the reference's Maxima generator
 * (which does `#include "iLQG.h"`, iLQG_func.tem:2) compiles against it
 * unchanged.  Field ORDER of tOptSet is not part of the contract (everything
 * is compiled per problem against one header); INIT_OPTSET zero-initialises.
 */
#ifndef ILQG_H
#define ILQG_H

/* second-order dynamics terms in the backward pass (reference iLQG.h:4-6) */
#ifndef FULL_DDP
#define FULL_DDP 1
#endif

#include "iLQG_problem.h"

#ifndef PRNT
#define PRNT printf
#endif

/* The reference's experimental 2-thread pipeline (iLQG.h:16-25) is superseded
 * by batch parallelism on the device; only the single-threaded layout exists. */
#ifndef MULTI_THREADED
#define MULTI_THREADED 0
#endif
#if MULTI_THREADED
#error "MULTI_THREADED=1 is not supported: trajectories are parallelised across the GPU batch instead"
#endif
#ifndef NUMBER_OF_THREADS
#define NUMBER_OF_THREADS 1
#endif

/* lets `-DDEBUG_X` (empty) mean 1, reference iLQG.h:27-28 */
#define DO_PREFIX1(VAL) 1##VAL
#define PREFIX1(VAL) DO_PREFIX1(VAL)

#ifdef __cplusplus
extern "C" {
#endif

/* problem-parameter descriptor emitted by generated code (reference iLQG.h:31-35):
 * size == -1 means one value per time step (n_hor+1 values) */
typedef struct paramDesc {
    char *name;
    int size;
    int is_var;
} tParamDesc;

/* All solver state of ONE trajectory optimisation (reference iLQG.h:37-76). */
typedef struct optSet {
    /* problem instance */
    int n_hor;            /* number of control steps (trajectory has n_hor+1 states) */
    double *x0;           /* initial state, borrowed */
    double **p;           /* problem parameters, borrowed, indexed as paramdesc[] */

    /* options (setOptParam / standard_parameters) */
    const double *alpha;  /* line-search step sizes, borrowed */
    int n_alpha;
    double tolFun, tolConstraint, tolGrad;
    int max_iter;
    double lambdaInit, dlambdaInit, lambdaFactor, lambdaMax, lambdaMin;
    int regType;
    double zMin;
    int debug_level;
    double w_pen_init_l, w_pen_init_f;
    double w_pen_max_l, w_pen_max_f;
    double w_pen_fact1, w_pen_fact2;

    /* iteration state / results */
    double cost, new_cost, dcost, expected;
    double lambda, g_norm;
    double dV[2];
    int iterations;
    double w_pen_l, w_pen_f;

    /* optional per-iteration logs (NULL = off) */
    int *log_linesearch;
    double *log_z;
    double *log_cost;

    /* trajectory storage: caller allocates trajectories[i].t (n_hor elements) */
    traj_t *nominal;
    traj_t *candidates[NUMBER_OF_THREADS];
    traj_t trajectories[NUMBER_OF_THREADS + 1];
    multipliers_t multipliers;

    /* device backend handle, created lazily by back_pass()/line_search(),
     * released by ilqg_release(); not present in the reference */
    void *backend;
} tOptSet;

#define INIT_OPTSET {0}

/* ---- solver side (defined by this library) ------------------------------ */
void printParams(double **p, int k);                     /* reference iLQG.h:78 */
void standard_parameters(tOptSet *o);                    /* reference iLQG.h:79 */
int iLQG(tOptSet *o);                                    /* reference iLQG.h:80: 1 converged, 0 otherwise */
/* NULL = ok, else a static error string (reference iLQG.h:81, iLQG.c:91-216) */
char *setOptParam(tOptSet *o, const char *name, const double *value, const int n);
void makeCandidateNominal(tOptSet *o, int idx);          /* reference iLQG.h:83 */
/* frees the lazily created device backend of `o` (additive) */
void ilqg_release(tOptSet *o);

/* ---- generated side (defined by iLQG_func.c; reference iLQG.h:82-88) ---- */
int forward_pass(traj_t *c, tOptSet *o, double alpha, double *csum, int cost_only);
int calc_derivs(tOptSet *o);
int init_opt(tOptSet *o);
int update_multipliers(tOptSet *o, int init);
int get_g_size();
int calcG(double g[], trajEl_t *t, int k, double *p[]);

extern int n_params;
extern int n_vars;
extern tParamDesc *paramdesc[];

#ifdef __cplusplus
}
#endif

/* scalar min/max used by generated code (reference iLQG.h:90-96).  HIP device
 * translation units get them from the HIP math headers instead. */
#if !defined(__HIPCC__)
static inline double max(double a, double b) { return (a > b) ? a : b; }
static inline double min(double a, double b) { return (a < b) ? a : b; }
#endif

#endif /* ILQG_H */
