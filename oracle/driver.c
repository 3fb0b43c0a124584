/* TEST INFRASTRUCTURE — not part of the product.
 *
 * Mex-free harness around one `tOptSet`, following the call sequence of the
 * reference's MEX entry (iLQG_mex.c:55-137): standard_parameters, options by
 * name, parameters by name through paramdesc[], caller-allocated trajectory
 * buffers, init_opt, initial roll-out with alpha = 0, swap, iLQG().
 *
 * The same file is linked three ways (see oracle/Makefile):
 *   _ref/libref_<problem>.so     reference solver sources from /root/reference
 *   liboracle_<problem>.so       the CPU restatement in oracle/ilqg_oracle.c
 *   libilqg_<problem>_hip.so     the HIP product (drop-in symbols)
 * and exposes flat-array entry points so Python (ctypes) can drive all three
 * identically.  Hooks on back_pass/line_search are installed with the linker's
 * --wrap option so that per-iteration traces can be recorded without touching
 * solver code.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"
#include "iLQG.h"
#include "back_pass.h"
#include "line_search.h"
#include "matMult.h"

#define DRV_MAX_TRACE 4096

typedef struct drv {
    tOptSet o;
    int n_hor;
    double x0[N_X];
    double **pstore;   /* owned copies of the parameter vectors */
    double alpha_store[64];
    /* trace of the last drv_solve() */
    int n_trace;
    double tr_lambda[DRV_MAX_TRACE];   /* lambda seen by line_search (after back pass) */
    double tr_gnorm[DRV_MAX_TRACE];
    double tr_dV0[DRV_MAX_TRACE], tr_dV1[DRV_MAX_TRACE];
    double tr_cost[DRV_MAX_TRACE];     /* nominal cost entering the iteration */
    double tr_newcost[DRV_MAX_TRACE];
    int tr_alpha[DRV_MAX_TRACE];       /* 1-based index of accepted alpha, n_alpha+1 = none */
    int tr_bp_calls[DRV_MAX_TRACE];    /* back_pass calls in this iteration */
    int bp_calls_pending;
} drv_t;

static drv_t *g_tracing = NULL;

/* ---- --wrap hooks ------------------------------------------------------ */
#ifndef DRV_NO_WRAP
int __real_back_pass(tOptSet *o);
int __real_line_search(tOptSet *o, int iter);

int __wrap_back_pass(tOptSet *o) {
    if(g_tracing && &g_tracing->o == o) g_tracing->bp_calls_pending++;
    return __real_back_pass(o);
}

int __wrap_line_search(tOptSet *o, int iter) {
    int r, i;
    drv_t *d = (g_tracing && &g_tracing->o == o) ? g_tracing : NULL;
    double cost_before = o->cost;
    r = __real_line_search(o, iter);
    if(d && d->n_trace < DRV_MAX_TRACE) {
        i = d->n_trace++;
        d->tr_lambda[i] = o->lambda;
        d->tr_gnorm[i] = o->g_norm;
        d->tr_dV0[i] = o->dV[0];
        d->tr_dV1[i] = o->dV[1];
        d->tr_cost[i] = cost_before;
        d->tr_newcost[i] = o->new_cost;
        d->tr_alpha[i] = o->log_linesearch ? o->log_linesearch[iter] : (r ? 0 : o->n_alpha + 1);
        d->tr_bp_calls[i] = d->bp_calls_pending;
        d->bp_calls_pending = 0;
    }
    return r;
}
#endif /* DRV_NO_WRAP */

/* ---- problem facts ------------------------------------------------------ */
void drv_dims(int *out) {
    out[0] = N_X;
    out[1] = N_U;
    out[2] = FULL_DDP;
    out[3] = (int)sizeof(trajEl_t);
    out[4] = n_params;
#ifdef ILQG_STATE_DEPENDENT_LIMITS
    out[5] = ILQG_STATE_DEPENDENT_LIMITS;
#else
    out[5] = 1;
#endif
}

const char *drv_param_name(int i) { return paramdesc[i]->name; }
int drv_param_size(int i) { return paramdesc[i]->size; }

/* ---- lifecycle ------------------------------------------------------------ */
drv_t *drv_create(int n_hor) {
    int i;
    drv_t *d = (drv_t *)calloc(1, sizeof(drv_t));
    tOptSet init = INIT_OPTSET;
    d->o = init;
    d->n_hor = n_hor;
    d->o.n_hor = n_hor;
    d->o.x0 = d->x0;
    standard_parameters(&d->o);
    d->o.debug_level = 0;
    d->pstore = (double **)calloc(n_params, sizeof(double *));
    for(i = 0; i < n_params; i++) {
        int sz = paramdesc[i]->size == -1 ? n_hor + 1 : paramdesc[i]->size;
        d->pstore[i] = (double *)calloc(sz, sizeof(double));
    }
    d->o.p = d->pstore;
    for(i = 0; i < NUMBER_OF_THREADS + 1; i++)
        d->o.trajectories[i].t = (trajEl_t *)calloc(n_hor, sizeof(trajEl_t));
    d->o.multipliers.t = (multipliersEl_t *)calloc(n_hor + 1, sizeof(multipliersEl_t) + 1);
    d->o.log_linesearch = (int *)calloc(DRV_MAX_TRACE, sizeof(int));
    d->o.log_z = (double *)calloc(DRV_MAX_TRACE, sizeof(double));
    d->o.log_cost = (double *)calloc(DRV_MAX_TRACE, sizeof(double));
    return d;
}

#ifdef DRV_HAVE_RELEASE
void ilqg_release(tOptSet *o);
#endif

void drv_destroy(drv_t *d) {
    int i;
    if(!d) return;
#ifdef DRV_HAVE_RELEASE
    ilqg_release(&d->o);
#endif
    for(i = 0; i < n_params; i++) free(d->pstore[i]);
    free(d->pstore);
    for(i = 0; i < NUMBER_OF_THREADS + 1; i++) free(d->o.trajectories[i].t);
    free(d->o.multipliers.t);
    free(d->o.log_linesearch);
    free(d->o.log_z);
    free(d->o.log_cost);
    free(d);
}

/* parameters are looked up by name, as iLQG_mex.c:70-84 does; returns 0 ok,
 * -1 unknown name, -2 wrong length */
int drv_set_param(drv_t *d, const char *name, const double *v, int n) {
    int i;
    for(i = 0; i < n_params; i++) {
        if(strcmp(paramdesc[i]->name, name) == 0) {
            int sz = paramdesc[i]->size == -1 ? d->n_hor + 1 : paramdesc[i]->size;
            if(sz != n) return -2;
            memcpy(d->pstore[i], v, sizeof(double) * n);
            return 0;
        }
    }
    return -1;
}

/* returns NULL or the solver's static error string (iLQG.c:91-216) */
const char *drv_set_opt(drv_t *d, const char *name, const double *v, int n) {
    if(strcmp(name, "alpha") == 0) { /* alpha is borrowed by the solver: keep a copy */
        if(n > 64) return "too many alpha";
        memcpy(d->alpha_store, v, sizeof(double) * n);
        v = d->alpha_store;
    }
    return setOptParam(&d->o, name, v, n);
}

/* init_opt + nominal controls + alpha=0 roll-out + swap  (iLQG_mex.c:108-120).
 * u0 is [n_hor][N_U] (input index fastest, MATLAB's u_nom(:,k)).  1 ok, 0 failed */
int drv_init(drv_t *d, const double *x0, const double *u0) {
    int k, i;
    memcpy(d->x0, x0, sizeof(double) * N_X);
    if(!init_opt(&d->o)) return 0;
    for(k = 0; k < d->n_hor; k++)
        for(i = 0; i < N_U; i++)
            d->o.nominal->t[k].u[i] = u0[MAT_IDX(i, k, N_U)];
    if(!forward_pass(d->o.candidates[0], &d->o, 0.0, &d->o.cost, 0)) return 0;
    makeCandidateNominal(&d->o, 0);
    /* state the solver sets on entry (iLQG.c:228-237) so that single stages can be driven */
    d->o.lambda = d->o.lambdaInit;
    d->o.w_pen_l = d->o.w_pen_init_l;
    d->o.w_pen_f = d->o.w_pen_init_f;
    update_multipliers(&d->o, 1);
    return 1;
}

/* ---- single stages -------------------------------------------------------- */
int drv_calc_derivs(drv_t *d) { return calc_derivs(&d->o); }
int drv_back_pass(drv_t *d) { return back_pass(&d->o); }
int drv_line_search(drv_t *d, int iter) { return line_search(&d->o, iter); }
void drv_accept(drv_t *d) { /* iLQG.c:325-327 */
    makeCandidateNominal(&d->o, 0);
    d->o.cost = d->o.new_cost;
}
int drv_forward_pass(drv_t *d, double alpha, double *cost) {
    return forward_pass(d->o.candidates[0], &d->o, alpha, cost, 0);
}
void drv_set_lambda(drv_t *d, double lambda) { d->o.lambda = lambda; }

int drv_solve(drv_t *d) {
    int r;
    d->n_trace = 0;
    d->bp_calls_pending = 0;
    g_tracing = d;
    r = iLQG(&d->o);
    g_tracing = NULL;
    return r;
}

/* ---- read-back ------------------------------------------------------------ */
/* which: 0 nominal, 1 candidate.  x is [n_hor+1][N_X], u is [n_hor][N_U] */
void drv_get_traj(drv_t *d, int which, double *x, double *u) {
    int k, i;
    traj_t *tr = which ? d->o.candidates[0] : d->o.nominal;
    for(k = 0; k < d->n_hor; k++) {
        for(i = 0; i < N_X; i++) x[k * N_X + i] = tr->t[k].x[i];
        for(i = 0; i < N_U; i++) u[k * N_U + i] = tr->t[k].u[i];
    }
    for(i = 0; i < N_X; i++) x[d->n_hor * N_X + i] = tr->f.x[i];
}

/* l is [n_hor][N_U], L is [n_hor][N_U*N_X] (column-major m x n per step) */
void drv_get_gains(drv_t *d, double *l, double *L) {
    int k;
    for(k = 0; k < d->n_hor; k++) {
        memcpy(l + k * N_U, d->o.nominal->t[k].l, sizeof(double) * N_U);
        memcpy(L + k * N_U * N_X, d->o.nominal->t[k].L, sizeof(double) * N_U * N_X);
    }
}

void drv_set_gains(drv_t *d, const double *l, const double *L) {
    int k;
    for(k = 0; k < d->n_hor; k++) {
        memcpy(d->o.nominal->t[k].l, l + k * N_U, sizeof(double) * N_U);
        memcpy(d->o.nominal->t[k].L, L + k * N_U * N_X, sizeof(double) * N_U * N_X);
    }
}

/* number of doubles in one packed derivative record:
 * cx cxx cu cuu cxu fx fu lower upper [fxx fuu fxu] lower_sign upper_sign lower_hx upper_hx */
int drv_record_size(void) {
    int s = N_X + sizeofQxx + N_U + sizeofQuu + sizeofQxu + N_X * N_X + N_X * N_U + 2 * N_U;
#if FULL_DDP
    s += N_X * sizeofQxx + N_X * sizeofQuu + N_X * sizeofQxu;
#endif
    s += 2 * N_U + 2 * N_X * N_U;
    return s;
}

#define PUT(field, cnt) do { memcpy(r, (field), sizeof(double) * (cnt)); r += (cnt); } while(0)
#define GET(field, cnt) do { memcpy((field), r, sizeof(double) * (cnt)); r += (cnt); } while(0)

/* rec is [n_hor][drv_record_size()], fin is [N_X + sizeofQxx] */
void drv_get_derivs(drv_t *d, double *rec, double *fin) {
    int k;
    double *r = rec;
    for(k = 0; k < d->n_hor; k++) {
        trajEl_t *t = &d->o.nominal->t[k];
        PUT(t->cx, N_X); PUT(t->cxx, sizeofQxx); PUT(t->cu, N_U); PUT(t->cuu, sizeofQuu);
        PUT(t->cxu, sizeofQxu); PUT(t->fx, N_X * N_X); PUT(t->fu, N_X * N_U);
        PUT(t->lower, N_U); PUT(t->upper, N_U);
#if FULL_DDP
        PUT(t->fxx, N_X * sizeofQxx); PUT(t->fuu, N_X * sizeofQuu); PUT(t->fxu, N_X * sizeofQxu);
#endif
        PUT(t->lower_sign, N_U); PUT(t->upper_sign, N_U);
        PUT(t->lower_hx, N_X * N_U); PUT(t->upper_hx, N_X * N_U);
    }
    r = fin;
    PUT(d->o.nominal->f.cx, N_X); PUT(d->o.nominal->f.cxx, sizeofQxx);
}

void drv_set_derivs(drv_t *d, const double *rec, const double *fin) {
    int k;
    const double *r = rec;
    for(k = 0; k < d->n_hor; k++) {
        trajEl_t *t = &d->o.nominal->t[k];
        GET(t->cx, N_X); GET(t->cxx, sizeofQxx); GET(t->cu, N_U); GET(t->cuu, sizeofQuu);
        GET(t->cxu, sizeofQxu); GET(t->fx, N_X * N_X); GET(t->fu, N_X * N_U);
        GET(t->lower, N_U); GET(t->upper, N_U);
#if FULL_DDP
        GET(t->fxx, N_X * sizeofQxx); GET(t->fuu, N_X * sizeofQuu); GET(t->fxu, N_X * sizeofQxu);
#endif
        GET(t->lower_sign, N_U); GET(t->upper_sign, N_U);
        GET(t->lower_hx, N_X * N_U); GET(t->upper_hx, N_X * N_U);
    }
    r = fin;
    GET(d->o.nominal->f.cx, N_X); GET(d->o.nominal->f.cxx, sizeofQxx);
}

/* out[0..8] = cost new_cost dcost expected lambda g_norm dV0 dV1 iterations */
void drv_get_scalars(drv_t *d, double *out) {
    out[0] = d->o.cost;
    out[1] = d->o.new_cost;
    out[2] = d->o.dcost;
    out[3] = d->o.expected;
    out[4] = d->o.lambda;
    out[5] = d->o.g_norm;
    out[6] = d->o.dV[0];
    out[7] = d->o.dV[1];
    out[8] = (double)d->o.iterations;
}

/* out[0..1] = doubles in multipliersEl_t / multipliersFin_t (0 for empty structs) */
void drv_multiplier_dims(int *out) {
    out[0] = (int)(sizeof(multipliersEl_t) / sizeof(double));
    out[1] = (int)(sizeof(multipliersFin_t) / sizeof(double));
}

/* the multiplier structs member by member: el [n_hor][dims[0]], fin [dims[1]], w_pen = {w_pen_l, w_pen_f} */
void drv_get_multipliers(drv_t *d, double *el, double *fin, double *w_pen) {
    if(sizeof(multipliersEl_t) > 0) memcpy(el, d->o.multipliers.t, sizeof(multipliersEl_t) * d->n_hor);
    if(sizeof(multipliersFin_t) > 0) memcpy(fin, &d->o.multipliers.f, sizeof(multipliersFin_t));
    w_pen[0] = d->o.w_pen_l;
    w_pen[1] = d->o.w_pen_f;
}

/* teacher forcing (lock-step parity tests): overwrite the nominal trajectory and the scalar state a single stage
 * reads.  x is [n_hor+1][N_X], u is [n_hor][N_U]; st = {cost, lambda, w_pen_l, w_pen_f} */
void drv_set_state(drv_t *d, const double *x, const double *u, const double *st) {
    int k, i;
    traj_t *tr = d->o.nominal;
    for(k = 0; k < d->n_hor; k++) {
        for(i = 0; i < N_X; i++) tr->t[k].x[i] = x[k * N_X + i];
        for(i = 0; i < N_U; i++) tr->t[k].u[i] = u[k * N_U + i];
    }
    for(i = 0; i < N_X; i++) tr->f.x[i] = x[d->n_hor * N_X + i];
    d->o.cost = st[0];
    d->o.lambda = st[1];
    d->o.w_pen_l = st[2];
    d->o.w_pen_f = st[3];
}

void drv_set_multipliers(drv_t *d, const double *el, const double *fin) {
    if(sizeof(multipliersEl_t) > 0) memcpy(d->o.multipliers.t, el, sizeof(multipliersEl_t) * d->n_hor);
    if(sizeof(multipliersFin_t) > 0) memcpy(&d->o.multipliers.f, fin, sizeof(multipliersFin_t));
}

int drv_get_log_linesearch(drv_t *d, int iter) { return d->o.log_linesearch[iter]; }

/* trace of the last drv_solve(): returns count; each array has room for `cap` */
int drv_get_trace(drv_t *d, int cap, double *lambda, double *gnorm, double *dV0, double *dV1,
                  double *cost, double *newcost, int *alpha_idx, int *bp_calls) {
    int i, n = d->n_trace < cap ? d->n_trace : cap;
    for(i = 0; i < n; i++) {
        lambda[i] = d->tr_lambda[i];
        gnorm[i] = d->tr_gnorm[i];
        dV0[i] = d->tr_dV0[i];
        dV1[i] = d->tr_dV1[i];
        cost[i] = d->tr_cost[i];
        newcost[i] = d->tr_newcost[i];
        alpha_idx[i] = d->tr_alpha[i];
        bp_calls[i] = d->tr_bp_calls[i];
    }
    return d->n_trace;
}

/* ---- many independent solves on host threads (CPU baseline of bench.py) ----
 * Trajectories i = 0..n-1 with x0 [n][N_X], u0 [n][n_hor][N_U]; options and
 * parameters are taken from `tmpl`.  Static split over `n_threads` threads, each
 * with its own tOptSet, exactly as independent runs of the reference would be. */
typedef struct {
    const drv_t *tmpl;
    const double *x0, *u0;
    double *cost;
    int *iters, *rc;
    int first, last;
} many_job_t;

static void *many_worker(void *arg) {
    many_job_t *j = (many_job_t *)arg;
    const drv_t *t = j->tmpl;
    int i, q;
    drv_t *d = drv_create(t->n_hor);
    for(q = 0; q < n_params; q++) {
        int sz = paramdesc[q]->size == -1 ? t->n_hor + 1 : paramdesc[q]->size;
        memcpy(d->pstore[q], t->pstore[q], sizeof(double) * sz);
    }
    for(i = j->first; i < j->last; i++) {
        /* options: copy the scalar settings of the template */
        d->o.max_iter = t->o.max_iter; d->o.tolFun = t->o.tolFun; d->o.tolGrad = t->o.tolGrad;
        d->o.lambdaInit = t->o.lambdaInit; d->o.dlambdaInit = t->o.dlambdaInit; d->o.lambdaFactor = t->o.lambdaFactor;
        d->o.lambdaMax = t->o.lambdaMax; d->o.lambdaMin = t->o.lambdaMin; d->o.regType = t->o.regType;
        d->o.zMin = t->o.zMin; d->o.debug_level = 0;
        if(!drv_init(d, j->x0 + (size_t)i * N_X, j->u0 + (size_t)i * t->n_hor * N_U)) {
            j->rc[i] = -1;
            continue;
        }
        j->rc[i] = iLQG(&d->o);
        j->cost[i] = d->o.cost;
        j->iters[i] = d->o.iterations;
    }
    drv_destroy(d);
    return NULL;
}

int drv_solve_many(drv_t *tmpl, int n, const double *x0, const double *u0, int n_threads, double *cost, int *iters,
                   int *rc) {
    int t, per;
    pthread_t *th;
    many_job_t *jobs;
    if(n_threads < 1) n_threads = 1;
    if(n_threads > n) n_threads = n;
    th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    jobs = (many_job_t *)calloc(n_threads, sizeof(many_job_t));
    per = (n + n_threads - 1) / n_threads;
    for(t = 0; t < n_threads; t++) {
        jobs[t].tmpl = tmpl; jobs[t].x0 = x0; jobs[t].u0 = u0;
        jobs[t].cost = cost; jobs[t].iters = iters; jobs[t].rc = rc;
        jobs[t].first = t * per;
        jobs[t].last = (t + 1) * per < n ? (t + 1) * per : n;
        pthread_create(&th[t], NULL, many_worker, &jobs[t]);
    }
    for(t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}
