/* Host side of the batched iLQG library — plain C, as in the reference.
 *
 * Part 1: the reference's own link-time symbols (include/iLQG.h, back_pass.h,
 *         line_search.h, boxQP.h), operating on one tOptSet.  The hot path —
 *         back_pass() and line_search() — runs on the GPU through the shim
 *         (ilqg_shim.h) with a batch of one; the outer loop iLQG() stays on the
 *         host exactly where the reference has it (iLQG.c:224-379) and keeps
 *         calling the generated host callbacks calc_derivs()/forward_pass().
 *         There is NO CPU implementation of the hot path in this library: with
 *         no usable HIP device these entry points print an error and abort().
 * Part 2: the batch interface of include/ilqg_batch.h: all trajectories and the
 *         per-trajectory solver state live in HBM, one launch sequence per
 *         lock-step iteration, no host round trip inside an iteration.
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "mex.h"
#include "iLQG.h"
#include "back_pass.h"
#include "line_search.h"
#include "boxQP.h"
#include "matMult.h"
#include "printMat.h"

#include "ilqg_batch.h"
#include "ilqg_shim.h"

#define REC_HOST_SIZE_BASE (N_X + sizeofQxx + N_U + sizeofQuu + sizeofQxu + N_X * N_X + N_X * N_U + 2 * N_U)
#if FULL_DDP
#define REC_HOST_SIZE (REC_HOST_SIZE_BASE + N_X * sizeofQxx + N_X * sizeofQuu + N_X * sizeofQxu + 2 * N_U + 2 * N_X * N_U)
#else
#define REC_HOST_SIZE (REC_HOST_SIZE_BASE + 2 * N_U + 2 * N_X * N_U)
#endif
#define FIN_SIZE (N_X + sizeofQxx)

/* The reference's console output of the outer loop and the line search (its TRACE macros: iLQG.c:24-33 on by
 * default, line_search.c:19-28 on by default; back_pass.c:26-34 and boxQP.c:33 off by default and not reproduced —
 * their messages name a time step inside a device sweep).  Same switches, same defaults, same texts, through PRNT
 * and under the same o->debug_level thresholds. */
#ifndef DEBUG_ILQG
#define DEBUG_ILQG 1
#endif
#ifndef DEBUG_FORWARDPASS
#define DEBUG_FORWARDPASS 1
#endif
#define SAY_LOOP(level, args) do { if((DEBUG_ILQG) && o->debug_level >= (level)) PRNT args; } while(0)
#define SAY_SEARCH(level, args) do { if((DEBUG_FORWARDPASS) && o->debug_level >= (level)) PRNT args; } while(0)

/* The batch is held as up to ILQG_MAX_GROUPS independent device contexts ("groups") of consecutive trajectories,
 * each with its own HIP stream, advanced alternately.  Every kernel of an iteration is either a chain of n_hor
 * dependent time steps with one wavefront per 64 trajectories — too few wavefronts to keep a SIMD busy — or wide
 * and throughput bound; with two groups in flight the chain of one overlaps with the wide kernel of the other
 * (measured at 65 536 CarParking trajectories, 20 iterations, round 3 with the line search that keeps its roll-outs:
 * 167-170 it/s with one group, 171-174 with two, 162-167 with three, 178-183 with four; five and more share the
 * process's four hardware queues and collapse to 118, with GPU_MAX_HW_QUEUES=8 they reach 159-171).  Trajectories are
 * independent, so results do not depend on the grouping. */
#define ILQG_MAX_GROUPS 4
#define ILQG_TRACE_MAX 4096
struct ilqg_batch {
    ilqg_dev_t *dev[ILQG_MAX_GROUPS];
    int first[ILQG_MAX_GROUPS], count[ILQG_MAX_GROUPS];
    int groups;
    int device, B, N;
    tOptSet opt;                     /* option holder, filled through setOptParam() */
    double alpha_store[ILQG_MAX_ALPHA];
    int resweep, fuse_derivs, ls_split, ls_keep, bw_split;
    double **p;                      /* owned copies of the problem parameters */
    char *p_given;                   /* which of them the caller has set */
    int params_pushed;
    /* full solves: finished trajectories are retired (ilqg_batch_solve) */
    int compact;                     /* 0: never; else the smallest number of live trajectories still worth a smaller context */
    int trace_n;                     /* the last solve, poll by poll: iterations done, trajectories active, slots iterated */
    int trace_it[ILQG_TRACE_MAX], trace_active[ILQG_TRACE_MAX], trace_slots[ILQG_TRACE_MAX];
    int compactions;
    double *scratch;                 /* host arrays of the drop-in entry points, kept between calls */
    size_t scratch_doubles;
    char err[512];
};

static char g_create_err[512];

static void fatal_no_device(const char *what, const char *msg) {
    fprintf(stderr,
            "ilqg: %s needs the HIP backend, which is unavailable: %s\n"
            "ilqg: this library has no CPU fallback for the hot path.\n",
            what, msg ? msg : "");
    abort();
}

/* =========================================================================
 * options                                      reference iLQG.c:36-216
 * ========================================================================= */
static double ilqg_default_alpha[8] = {1.0, 0.3727594, 0.1389495, 0.0517947,
                                       0.0193070, 0.0071969, 0.0026827, 0.0010000};

void standard_parameters(tOptSet *o) {
    o->alpha = ilqg_default_alpha;
    o->n_alpha = 8;
    o->tolFun = 1e-7;
    o->tolConstraint = 1e-7;
    o->tolGrad = 1e-5;
    o->max_iter = 20;
    o->lambdaInit = 1;
    o->dlambdaInit = 1;
    o->lambdaFactor = 1.6;
    o->lambdaMax = 1e10;
    o->lambdaMin = 1e-6;
    o->regType = 1;
    o->zMin = 0.0;
    o->debug_level = 2;
    o->w_pen_init_l = 1.0;
    o->w_pen_init_f = 1.0;
    o->w_pen_max_l = INF;
    o->w_pen_max_f = INF;
    o->w_pen_fact1 = 4.0;
    o->w_pen_fact2 = 1.0;
}

static char err_scalar[] = "parameter must be scalar";
static char err_alpha_range[] = "all alpha must be in the range [1.0..0.0)";
static char err_alpha_mono[] = "all alpha must be monotonically decreasing";
static char err_pos[] = "parameter must be positive";
static char err_gt_one[] = "parameter must be > 1";
static char err_one_two[] = "parameter must be in range [1..2]";
static char err_zero_one[] = "parameter must be in range [0..1)";
static char err_debug[] = "parameter must be in range [0..6]";
static char err_unknown[] = "no such parameter";

/* one row per scalar option: where it lives, its type, the lowest/highest
 * admissible value and the message of the reference for a violation */
enum { CHK_GT0, CHK_GE0, CHK_GE1, CHK_1_2, CHK_0_1, CHK_0_6 };
typedef struct {
    const char *name;
    size_t offset;
    int is_int;
    int check;
} opt_row_t;

#define ROW_D(field, chk) {#field, offsetof(tOptSet, field), 0, chk}
#define ROW_I(field, chk) {#field, offsetof(tOptSet, field), 1, chk}
static const opt_row_t opt_rows[] = {
    ROW_D(tolFun, CHK_GT0),       ROW_D(tolConstraint, CHK_GT0), ROW_D(tolGrad, CHK_GT0),
    ROW_I(max_iter, CHK_GE0),     ROW_D(lambdaInit, CHK_GE0),    ROW_D(dlambdaInit, CHK_GE0),
    ROW_D(lambdaFactor, CHK_GE1), ROW_D(lambdaMax, CHK_GE0),     ROW_D(lambdaMin, CHK_GE0),
    ROW_I(regType, CHK_1_2),      ROW_D(zMin, CHK_0_1),          ROW_I(debug_level, CHK_0_6),
    ROW_D(w_pen_init_l, CHK_GE0), ROW_D(w_pen_init_f, CHK_GE0),  ROW_D(w_pen_max_l, CHK_GE0),
    ROW_D(w_pen_max_f, CHK_GE0),  ROW_D(w_pen_fact1, CHK_GE1),   ROW_D(w_pen_fact2, CHK_GE1),
};

char *setOptParam(tOptSet *o, const char *name, const double *value, const int n) {
    size_t r;
    int i;
    if(strcmp(name, "alpha") == 0) {
        for(i = 0; i < n; i++) {
            if(value[i] < 0.0 || value[i] > 1.0) return err_alpha_range;
            if(i > 0 && value[i] >= value[i - 1]) return err_alpha_mono;
        }
        o->alpha = value; /* borrowed, as in the reference (iLQG.c:101) */
        o->n_alpha = n;
        return NULL;
    }
    for(r = 0; r < sizeof(opt_rows) / sizeof(opt_rows[0]); r++) {
        const opt_row_t *row = &opt_rows[r];
        double v;
        if(strcmp(name, row->name) != 0) continue;
        if(n != 1) return err_scalar;
        v = value[0];
        switch(row->check) {
            case CHK_GT0: if(v <= 0.0) return err_pos; break;
            case CHK_GE0: if(v < 0.0) return err_pos; break;
            case CHK_GE1: if(v < 1.0) return err_gt_one; break;
            case CHK_1_2: if(v < 1.0 || v > 2.0) return err_one_two; break;
            case CHK_0_1: if(v < 0.0 || v >= 1.0) return err_zero_one; break;
            case CHK_0_6: if(v < 0.0 || v > 6.0) return err_debug; break;
        }
        if(row->is_int)
            *(int *)((char *)o + row->offset) = (int)v;
        else
            *(double *)((char *)o + row->offset) = v;
        return NULL;
    }
    return err_unknown;
}

void makeCandidateNominal(tOptSet *o, int idx) {
    traj_t *swap = o->candidates[idx];
    o->candidates[idx] = o->nominal;
    o->nominal = swap;
}

void printVec(const double *A, const int n, const char *nm) {
    int i;
    PRNT("%s= [", nm);
    for(i = 0; i < n; i++) PRNT(i ? ", %g" : "%g", A[i]);
    PRNT("]\n");
}

void printTri(const double *A, const int n, const char *nm) {
    int r, c;
    PRNT("%s= [\n", nm);
    for(r = 0; r < n; r++) {
        for(c = 0; c < n; c++) PRNT(c ? ", %g" : "  %g", A[SYMTRI_MAT_IDX(r, c)]);
        PRNT("\n");
    }
    PRNT("]\n");
}

void printMat(const double *A, const int n, const int m, const char *nm) {
    int r, c;
    PRNT("%s= [\n", nm);
    for(r = 0; r < n; r++) {
        for(c = 0; c < m; c++) PRNT(c ? ", %g" : "  %g", A[r + n * c]);
        PRNT("\n");
    }
    PRNT("]\n");
}

void printParams(double **p, int k) {
    int i;
    for(i = 0; i < n_params; i++) {
        if(paramdesc[i]->size == -1)
            PRNT("%s[k]= %g\n", paramdesc[i]->name, p[i][k]);
        else if(paramdesc[i]->size == 1)
            PRNT("%s= %g\n", paramdesc[i]->name, p[i][0]);
        else
            printVec(p[i], paramdesc[i]->size, paramdesc[i]->name);
    }
}

/* =========================================================================
 * batch interface
 * ========================================================================= */
static int fail(ilqg_batch_t *c, const char *what) {
    snprintf(c->err, sizeof(c->err), "%s: %s", what, ilqg_dev_error());
    return 1;
}

static int fail_msg(ilqg_batch_t *c, const char *msg) {
    snprintf(c->err, sizeof(c->err), "%s", msg);
    return 1;
}

void ilqg_problem_dims(int *out) {
    ilqg_dev_dims(out);
    out[6] = n_params;
}

const char *ilqg_problem_param_name(int i) { return (i >= 0 && i < n_params) ? paramdesc[i]->name : NULL; }
int ilqg_problem_param_size(int i) { return (i >= 0 && i < n_params) ? paramdesc[i]->size : 0; }
int ilqg_device_count(void) { return ilqg_dev_count(); }

int ilqg_reference_success(int status, int iterations) {
    switch(status) {
    case ILQG_ST_CONVERGED_GRAD:
    case ILQG_ST_CONVERGED_FUN:
    case ILQG_ST_LAMBDA_MAX: return 1;
    case ILQG_ST_DERIVS_FAILED: return iterations > 0; /* backPassDone of the previous iteration is still set */
    default: return 0;
    }
}

const char *ilqg_batch_error(const ilqg_batch_t *c) { return c ? c->err : g_create_err; }

static int param_len(const ilqg_batch_t *c, int i) { return paramdesc[i]->size == -1 ? c->N + 1 : paramdesc[i]->size; }

/* groups = 0: automatic (ILQG_GROUPS in the environment, else 4 for large batches in the lane mapping, see above) */
ilqg_batch_t *ilqg_batch_create_groups(int device, int batch, int n_hor, int groups) {
    int i, g, per, dims[8];
    ilqg_batch_t *c = (ilqg_batch_t *)calloc(1, sizeof(*c));
    if(!c) return NULL;
    c->device = device;
    c->B = batch;
    c->N = n_hor;
    /* the cost-only sweep after an accepted step (iLQG.c:338) only changes the cost when multipliers or
     * penalty weights changed; a problem without multipliers (empty structs in the generated header)
     * gets bit-identical costs from it, so it is skipped unless asked for */
    c->resweep = (sizeof(multipliersEl_t) > 0 || sizeof(multipliersFin_t) > 0) ? 1 : 0;
    /* derivatives inside the backward kernel, except where measured slower: the one-state multiplier demos
     * (65 536 Brachistochrone solves 0.137 s with stored records, 0.156 s fused) */
    c->fuse_derivs = c->resweep ? 0 : 1;
    c->ls_keep = 1;
    /* measured: no gain (the fused backward kernel alone 2.65 ms on one wavefront per tile, 2.73 ms on two; headline
     * 135.6 against 130-133 it/s) — the Riccati update with its box QP is the chain that bounds a step, the derivative
     * evaluation already hides behind it in one instruction stream */
    c->bw_split = 0;
    standard_parameters(&c->opt);
    ilqg_dev_dims(dims);
    /* first line-search stage: 3 step sizes in the lane mapping (CarParking accepts 85 % of the steps there); 1 in
     * the wave mapping, whose roll-outs are latency bound whatever their number (n = 16 problem: 89 % accepted at
     * the first step size; measured 2.55 it/s with 1, 2.41 with 2, 2.48 with 3, 2.53 with 4) */
    c->ls_split = dims[7] ? 1 : 4;
    /* lane mapping: every roll-out of the search is kept and the accepted one becomes the current trajectory by a change
     * of its location index (k_search / k_commit), with four step sizes in the first stage (one whole cache line per
     * store of a step size's 16 lanes); wave mapping: second stage beside the winner pass (ROLL_SECOND) */
    c->ls_keep = 2; /* (wave mapping: where the generated file offers the step in parts, both stages keep their roll-outs
                     * and the accepted ones are copied into the records; else as ls_keep = 1) */
    if(groups <= 0) {
        /* lane mapping: 4 (see above).  Wave mapping: 1 — two groups whose backward passes take turns (they share the
         * device's derivative work buffer) so that the roll-outs of one run beside the backward pass of the other were
         * measured SLOWER on the n = 16 problem (1.98 -> 1.53 it/s): a workgroup of the backward kernel takes all the
         * registers of its CU (8 wavefronts x 250) and the whole LDS, so roll-out wavefronts (256 registers) find no
         * room beside it, while each group's roll-outs keep the full chain latency of n_hor steps. */
        const char *e = getenv("ILQG_GROUPS");
        groups = e ? atoi(e) : ((batch >= 8192 && !dims[7]) ? 4 : 1);
    }

    if(groups < 1) groups = 1;
    if(groups > ILQG_MAX_GROUPS) groups = ILQG_MAX_GROUPS;
    /* whole tiles of 64 trajectories per group */
    per = ((batch + groups - 1) / groups + 63) / 64 * 64;
    c->groups = 0;
    for(g = 0; g < groups && (g == 0 || g * per < batch); g++) {  /* an empty batch is refused by group 0 */
        c->first[g] = g * per;
        c->count[g] = (batch - g * per < per) ? batch - g * per : per;
        if(ilqg_dev_create(&c->dev[g], device, c->count[g], n_hor)) {
            snprintf(g_create_err, sizeof(g_create_err), "ilqg_batch_create: %s", ilqg_dev_error());
            for(i = 0; i < g; i++) ilqg_dev_destroy(c->dev[i]);
            free(c);
            return NULL;
        }
        c->groups = g + 1;
    }
    c->p = (double **)calloc(n_params > 0 ? n_params : 1, sizeof(double *));
    c->p_given = (char *)calloc(n_params > 0 ? n_params : 1, 1);
    for(i = 0; i < n_params; i++) c->p[i] = (double *)calloc(param_len(c, i), sizeof(double));
    return c;
}

ilqg_batch_t *ilqg_batch_create(int device, int batch, int n_hor) {
    return ilqg_batch_create_groups(device, batch, n_hor, 0);
}

int ilqg_batch_groups(const ilqg_batch_t *c) { return c->groups; }

void ilqg_batch_destroy(ilqg_batch_t *c) {
    int i;
    if(!c) return;
    for(i = 0; i < c->groups; i++) ilqg_dev_destroy(c->dev[i]);
    for(i = 0; i < n_params; i++) free(c->p[i]);
    free(c->p);
    free(c->p_given);
    free(c->scratch);
    free(c);
}

/* host arrays are [trajectory][steps][width]: group g starts first[g] trajectories in */
#define EACH_GROUP(g) for(g = 0; g < c->groups; g++)
static int each_write(ilqg_batch_t *c, int field, const double *host, const char *what) {
    int g;
    EACH_GROUP(g) {
        const size_t per = (size_t)ilqg_dev_field_steps(c->dev[g], field) * ilqg_dev_field_width(field);
        if(ilqg_dev_write(c->dev[g], field, host + per * c->first[g])) return fail(c, what);
    }
    return 0;
}
static int each_write_steps(ilqg_batch_t *c, int field, const double *host, int steps, const char *what) {
    int g;
    EACH_GROUP(g) {
        const size_t per = (size_t)steps * ilqg_dev_field_width(field);
        if(ilqg_dev_write_steps(c->dev[g], field, host + per * c->first[g], steps)) return fail(c, what);
    }
    return 0;
}
static int each_read(ilqg_batch_t *c, int field, double *host, const char *what) {
    int g;
    EACH_GROUP(g) {
        const size_t per = (size_t)ilqg_dev_field_steps(c->dev[g], field) * ilqg_dev_field_width(field);
        if(ilqg_dev_read(c->dev[g], field, host + per * c->first[g])) return fail(c, what);
    }
    return 0;
}
static int int_width(int field) { return field == ILQG_I_ALPHA_OK ? ILQG_MAX_ALPHA : 1; }
static int each_read_int(ilqg_batch_t *c, int field, int *host, const char *what) {
    int g;
    EACH_GROUP(g) if(ilqg_dev_read_int(c->dev[g], field, host + (size_t)int_width(field) * c->first[g])) return fail(c, what);
    return 0;
}
static int each_write_int(ilqg_batch_t *c, int field, const int *host, const char *what) {
    int g;
    EACH_GROUP(g) if(ilqg_dev_write_int(c->dev[g], field, host + (size_t)int_width(field) * c->first[g])) return fail(c, what);
    return 0;
}

int ilqg_batch_set_option(ilqg_batch_t *c, const char *name, const double *value, int n) {
    char *e;
    if(strcmp(name, "resweep") == 0) {
        if(n != 1) return fail_msg(c, err_scalar);
        c->resweep = value[0] != 0.0;
        return 0;
    }
    if(strcmp(name, "ls_split") == 0) {
        if(n != 1) return fail_msg(c, err_scalar);
        if(value[0] < 0.0) return fail_msg(c, err_pos);
        c->ls_split = (int)value[0];
        return 0;
    }
    if(strcmp(name, "bw_split") == 0) {
        if(n != 1) return fail_msg(c, err_scalar);
        c->bw_split = value[0] != 0.0;
        return 0;
    }
    if(strcmp(name, "ls_keep") == 0) {
        if(n != 1) return fail_msg(c, err_scalar);
        if(value[0] != 0.0 && value[0] != 1.0 && value[0] != 2.0) return fail_msg(c, "ls_keep must be 0, 1 or 2");
        c->ls_keep = (int)value[0];
        return 0;
    }
    if(strcmp(name, "fuse_derivs") == 0) {
        if(n != 1) return fail_msg(c, err_scalar);
        c->fuse_derivs = value[0] != 0.0;
        return 0;
    }
    if(strcmp(name, "compact") == 0) {
        if(n != 1) return fail_msg(c, err_scalar);
        if(value[0] < 0.0) return fail_msg(c, err_pos);
        c->compact = (int)value[0];
        return 0;
    }
    if(strcmp(name, "alpha") == 0) {
        /* the option set only borrows the array (iLQG.c:101): it is validated on the caller's values first, and only
         * an accepted set replaces the stored one */
        tOptSet probe = c->opt;
        if(n > ILQG_MAX_ALPHA) return fail_msg(c, "at most 16 alpha values");
        e = setOptParam(&probe, name, value, n);
        if(e) return fail_msg(c, e);
        memcpy(c->alpha_store, value, sizeof(double) * n);
        value = c->alpha_store;
    }
    e = setOptParam(&c->opt, name, value, n);
    if(e) return fail_msg(c, e);
    return 0;
}

int ilqg_batch_set_param(ilqg_batch_t *c, const char *name, const double *value, int n) {
    int i;
    for(i = 0; i < n_params; i++) {
        if(strcmp(paramdesc[i]->name, name) != 0) continue;
        if(param_len(c, i) != n) {
            snprintf(c->err, sizeof(c->err), "Parameter name '%s' must be a vector length %d.", name, param_len(c, i));
            return 1;
        }
        memcpy(c->p[i], value, sizeof(double) * n);
        c->p_given[i] = 1;
        c->params_pushed = 0;
        return 0;
    }
    snprintf(c->err, sizeof(c->err), "Parameter name '%s' is not a parameter of this problem.", name);
    return 1;
}

/* options and parameters are pushed lazily before any stage */
static int push_config(ilqg_batch_t *c) {
    ilqg_dev_opts_t d;
    int i;
    const tOptSet *o = &c->opt;
    memset(&d, 0, sizeof(d));
    if(o->n_alpha < 1 || o->n_alpha > ILQG_MAX_ALPHA) return fail_msg(c, "n_alpha must be in 1..16");
    d.n_alpha = o->n_alpha;
    for(i = 0; i < o->n_alpha; i++) d.alpha[i] = o->alpha[i];
    d.tolFun = o->tolFun; d.tolGrad = o->tolGrad; d.tolConstraint = o->tolConstraint;
    d.lambdaInit = o->lambdaInit; d.dlambdaInit = o->dlambdaInit; d.lambdaFactor = o->lambdaFactor;
    d.lambdaMax = o->lambdaMax; d.lambdaMin = o->lambdaMin;
    d.zMin = o->zMin; d.regType = o->regType; d.max_iter = o->max_iter;
    d.w_pen_init_l = o->w_pen_init_l; d.w_pen_init_f = o->w_pen_init_f;
    d.w_pen_max_l = o->w_pen_max_l; d.w_pen_max_f = o->w_pen_max_f;
    d.w_pen_fact1 = o->w_pen_fact1; d.w_pen_fact2 = o->w_pen_fact2;
    d.resweep = c->resweep;
    d.fuse_derivs = c->fuse_derivs;
    d.ls_split = c->ls_split;
    d.ls_keep = c->ls_keep;
    d.bw_split = c->bw_split;
    { int g; EACH_GROUP(g) if(ilqg_dev_set_opts(c->dev[g], &d)) return fail(c, "options"); }
    if(!c->params_pushed) {
        int sizes[64];
        for(i = 0; i < n_params; i++)  /* every parameter must be given, as in iLQG_mex.c:73-76 */
            if(!c->p_given[i]) {
                snprintf(c->err, sizeof(c->err), "Parameter name '%s' was not set.", paramdesc[i]->name);
                return 1;
            }
        if(n_params > 64) return fail_msg(c, "more than 64 problem parameters");
        for(i = 0; i < n_params; i++) sizes[i] = paramdesc[i]->size;
        { int g; EACH_GROUP(g) if(ilqg_dev_set_params(c->dev[g], n_params, sizes, (const double *const *)c->p)) return fail(c, "parameters"); }
        c->params_pushed = 1;
    }
    return 0;
}

int ilqg_batch_set_x0(ilqg_batch_t *c, const double *x0) { return each_write_steps(c, ILQG_F_X, x0, 1, "set_x0"); }
int ilqg_batch_set_u(ilqg_batch_t *c, const double *u) { return each_write(c, ILQG_F_U, u, "set_u"); }
int ilqg_batch_set_x(ilqg_batch_t *c, const double *x) { return each_write(c, ILQG_F_X, x, "set_x"); }

int ilqg_batch_init(ilqg_batch_t *c) {
    int g;
    if(push_config(c)) return 1;
    EACH_GROUP(g) {
        if(ilqg_dev_rollout_init(c->dev[g])) return fail(c, "initial roll-out");
        if(ilqg_dev_reset(c->dev[g])) return fail(c, "reset");
    }
    return 0;
}

/* the groups advance alternately, one iteration at a time: their launches interleave on the device */
static int iterate_groups(ilqg_batch_t *c, int n) {
    int it, g;
    for(it = 0; it < n; it++) EACH_GROUP(g) if(ilqg_dev_iterate(c->dev[g], 1)) return fail(c, "iterate");
    return 0;
}

int ilqg_batch_iterate(ilqg_batch_t *c, int n) {
    if(push_config(c)) return 1;
    return iterate_groups(c, n);
}

int ilqg_batch_active(ilqg_batch_t *c, int *n) {
    int g, a;
    *n = 0;
    EACH_GROUP(g) {
        if(ilqg_dev_count_active(c->dev[g], &a)) return fail(c, "active count");
        *n += a;
    }
    return 0;
}

/* ---- full solves: retiring finished trajectories -----------------------------------------------------------------
 * The reference's product is a solve to convergence (iLQG.c:224-379): a trajectory leaves the loop through the gradient
 * test (:297-303), the cost test (:331), lambda > lambdaMax (:273, :356) or max_iter (:372).  In a lock-step batch the
 * kernels skip a finished trajectory, but a wavefront with ONE live lane costs what a full one costs, and CarParking
 * starts converge anywhere between 50 and 550 iterations: most lane-iterations of a solve's tail are idle lanes in
 * resident wavefronts.  So, between iterations, once at most half of the slots being iterated are live, the live
 * trajectories are gathered into a smaller context (ilqg_dev_move: current x / u, records, scalars, integers,
 * multipliers, and the stored derivative records of the unfused path), iterated there, and written back into their own
 * slots of the caller's batch when they are gathered again or the solve ends.  Every getter therefore reads the caller's
 * batch as if nothing had moved; a trajectory's iterations depend on nothing but its own state, so every result is
 * bit for bit that of the uncompacted solve (tests/test_gpu_solve.py).  Option "compact" = n > 0 switches it on:
 * contexts of fewer than n trajectories are not made (below a few thousand trajectories an iteration costs the latency
 * of its chains of n_hor steps whatever the size). */
static double now_s(void) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}
static int group_of(const ilqg_batch_t *c, int b) {
    int g;
    for(g = c->groups - 1; g > 0; g--)
        if(b >= c->first[g]) break;
    return g;
}
/* state of trajectories src_idx[0..n) of `src` -> trajectories dst_idx[0..n) of `dst` (indices of the whole batches) */
static int move_between(ilqg_batch_t *dst, ilqg_batch_t *src, int n, const int *dst_idx, const int *src_idx) {
    int gd, gs, j, rc = 0;
    int *to = (int *)malloc(sizeof(int) * (n > 0 ? n : 1)), *from = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for(gd = 0; gd < dst->groups && !rc; gd++)
        for(gs = 0; gs < src->groups && !rc; gs++) {
            int m = 0;
            for(j = 0; j < n; j++)
                if(group_of(dst, dst_idx[j]) == gd && group_of(src, src_idx[j]) == gs) {
                    to[m] = dst_idx[j] - dst->first[gd];
                    from[m] = src_idx[j] - src->first[gs];
                    m++;
                }
            if(m && ilqg_dev_move(dst->dev[gd], src->dev[gs], m, to, from, !src->fuse_derivs)) rc = fail(dst, "moving trajectories");
        }
    free(to);
    free(from);
    return rc;
}
/* a context of `n` trajectories with the options and parameters of c */
static ilqg_batch_t *working_copy(ilqg_batch_t *c, int n) {
    int i;
    /* (MEASURED: four stream groups for the small contexts too — so that their chains of n_hor dependent steps overlap —
     * changed nothing: an iteration of 3 738 trajectories takes the 5 ms an iteration of 65 536 takes, in one group or four) */
    int dims[8], groups;
    ilqg_batch_t *w;
    ilqg_dev_dims(dims);
    (void)dims;
    groups = 0;  /* the library's choice for that size (ILQG_COMPACT_GROUPS: experiments) */
    if(getenv("ILQG_COMPACT_GROUPS")) groups = atoi(getenv("ILQG_COMPACT_GROUPS"));
    w = ilqg_batch_create_groups(c->device, n, c->N, groups);
    if(!w) return NULL;
    w->opt = c->opt;
    memcpy(w->alpha_store, c->opt.alpha, sizeof(double) * c->opt.n_alpha);
    w->opt.alpha = w->alpha_store;
    w->resweep = c->resweep; w->fuse_derivs = c->fuse_derivs; w->ls_split = c->ls_split; w->ls_keep = c->ls_keep; w->bw_split = c->bw_split;
    for(i = 0; i < n_params; i++) {
        memcpy(w->p[i], c->p[i], sizeof(double) * param_len(c, i));
        w->p_given[i] = c->p_given[i];
    }
    if(push_config(w)) {
        snprintf(c->err, sizeof(c->err), "%s", w->err);
        ilqg_batch_destroy(w);
        return NULL;
    }
    return w;
}

int ilqg_batch_solve(ilqg_batch_t *c) {
    int it, active = 1, rc = 0, j;
    ilqg_batch_t *cur = c;  /* the context being iterated; map[j]: the trajectory of c in its slot j */
    int *map = NULL, *status = NULL, *ident = NULL;
    const int dbg = getenv("ILQG_SOLVE_DEBUG") != NULL;
    double t_dbg = 0.0, t_poll = 0.0;
    if(push_config(c)) return 1;
    c->err[0] = 0;
    c->trace_n = 0;
    t_poll = dbg ? now_s() : 0.0;
    c->compactions = 0;
    /* poll the active count every few iterations: one small D2H copy */
    for(it = 0; it < c->opt.max_iter && active; ) {
        int n = c->opt.max_iter - it < 4 ? c->opt.max_iter - it : 4;
        if(iterate_groups(cur, n)) { rc = 1; break; }
        it += n;
        if(ilqg_batch_active(cur, &active)) { rc = 1; break; }
        if(dbg && it % 40 == 0) {
            fprintf(stderr, "ilqg solve: iteration %d, %d live of %d slots, %.2f ms per iteration\n", it, active, cur->B, 1e3 * (now_s() - t_poll) / n);
        }
        t_poll = dbg ? now_s() : 0.0;
        if(c->trace_n < ILQG_TRACE_MAX) {
            c->trace_it[c->trace_n] = it;
            c->trace_active[c->trace_n] = active;
            c->trace_slots[c->trace_n] = cur->B;
            c->trace_n++;
        }
        if(c->compact > 0 && active > 0 && it < c->opt.max_iter && 2 * active <= cur->B && active >= c->compact) {
            ilqg_batch_t *w;
            int *nmap, m = 0;
            /* who is live (in cur's numbering), and where each lives in c */
            status = (int *)realloc(status, sizeof(int) * cur->B);
            if(each_read_int(cur, ILQG_I_STATUS, status, "status")) { rc = 1; break; }
            if(cur != c) {  /* everything cur holds goes home first: its finished trajectories are results */
                ident = (int *)realloc(ident, sizeof(int) * cur->B);
                for(j = 0; j < cur->B; j++) ident[j] = j;
                if(move_between(c, cur, cur->B, map, ident)) { rc = 1; break; }
            }
            nmap = (int *)malloc(sizeof(int) * active);
            for(j = 0; j < cur->B && m < active; j++)
                if(status[j] == ILQG_ST_ACTIVE) nmap[m++] = (cur == c) ? j : map[j];
            t_dbg = dbg ? now_s() : 0.0;
            w = working_copy(c, m);
            if(!w) { free(nmap); rc = 1; break; }
            if(dbg) fprintf(stderr, "ilqg solve: iteration %d, %d live of %d slots: new context %.1f ms", it, m, cur->B, 1e3 * (now_s() - t_dbg));
            t_dbg = dbg ? now_s() : 0.0;
            ident = (int *)realloc(ident, sizeof(int) * (m > cur->B ? m : cur->B));
            for(j = 0; j < m; j++) ident[j] = j;
            if(move_between(w, c, m, ident, nmap)) { snprintf(c->err, sizeof(c->err), "%s", w->err); ilqg_batch_destroy(w); free(nmap); rc = 1; break; }
            if(dbg) fprintf(stderr, ", gather %.1f ms", 1e3 * (now_s() - t_dbg));
            t_dbg = dbg ? now_s() : 0.0;
            if(cur != c) ilqg_batch_destroy(cur);
            if(dbg) fprintf(stderr, ", release %.1f ms\n", 1e3 * (now_s() - t_dbg));
            free(map);
            cur = w;
            map = nmap;
            c->compactions++;
        }
    }
    if(cur != c) {
        if(!rc) {
            ident = (int *)realloc(ident, sizeof(int) * cur->B);
            for(j = 0; j < cur->B; j++) ident[j] = j;
            if(move_between(c, cur, cur->B, map, ident)) rc = 1;
        } else if(!c->err[0]) {  /* (a failure already recorded in c — a move, a new context — stays; else the iterated context's) */
            snprintf(c->err, sizeof(c->err), "%s", cur->err);
        }
        ilqg_batch_destroy(cur);
    }
    free(map);
    free(status);
    free(ident);
    return rc;
}

/* ---- a STREAM of starts through the resident batch ------------------------------------------------------------------
 * An iteration of the lane mapping costs what its chains of n_hor dependent steps cost, for 600 trajectories as for 65 536
 * (one wavefront per SIMD either way): a solve's rate is decided by how many of the batch's slots hold a live trajectory.
 * Here finished trajectories do not just leave — their slots are given to the next starts of the stream: every
 * ILQG_STREAM_ROUND iterations the finished slots are harvested (cost, exit reason, iteration count; the trajectory if the
 * caller wants it) and, once enough are free, refilled: the new starts are initialised in a small staging context
 * (initial roll-out + solver entry state, exactly ilqg_batch_init) and moved into the free slots (ilqg_dev_move).  A
 * trajectory's iterations depend on nothing but its own state, so every start gets the result a plain ilqg_batch_solve
 * of a batch holding it would give, bit for bit (tests/test_gpu_solve.py). */
#define ILQG_STREAM_ROUND 8
#define ILQG_STREAM_STAGE 8192
int ilqg_batch_solve_stream(ilqg_batch_t *c, int total, const double *x0, const double *u0, double *cost, int *status,
                            int *iterations, double *x, double *u) {
    int dims[8], NXd, NUd, B = c->B, N = c->N, j, rc = 0, next = 0, n_done = 0, it = 0, active = 0, n_free;
    int *slot = NULL, *st = NULL, *its = NULL, *freel = NULL, *ident = NULL, *fin = NULL;
    double *cst = NULL, *xs = NULL, *us = NULL, *hx = NULL, *hu = NULL;
    ilqg_batch_t *stage = NULL;
    const int S = B < ILQG_STREAM_STAGE ? B : ILQG_STREAM_STAGE;
    if(total < 1) return fail_msg(c, "ilqg_batch_solve_stream: no starts");
    if(push_config(c)) return 1;
    ilqg_dev_dims(dims);
    NXd = dims[0];
    NUd = dims[1];
    slot = (int *)malloc(sizeof(int) * B); st = (int *)malloc(sizeof(int) * B); its = (int *)malloc(sizeof(int) * B);
    freel = (int *)malloc(sizeof(int) * B); ident = (int *)malloc(sizeof(int) * B); fin = (int *)malloc(sizeof(int) * B);
    cst = (double *)malloc(sizeof(double) * B);
    xs = (double *)calloc((size_t)S * NXd, sizeof(double));
    us = (double *)calloc((size_t)S * N * NUd, sizeof(double));
    for(j = 0; j < B; j++) { slot[j] = -1; st[j] = ILQG_ST_MAX_ITER; ident[j] = j; }
    /* every slot is free and says so to the kernels */
    if(each_write_int(c, ILQG_I_STATUS, st, "status")) { rc = 1; goto done; }
    stage = working_copy(c, S);
    if(!stage) { rc = 1; goto done; }
    c->trace_n = 0;
    c->compactions = 0;
    for(;;) {
        /* harvest what has finished */
        if(each_read_int(c, ILQG_I_STATUS, st, "status")) { rc = 1; break; }
        {
            int m = 0;
            active = 0;
            for(j = 0; j < B; j++) {
                if(slot[j] < 0) continue;
                if(st[j] == ILQG_ST_ACTIVE) active++;
                else fin[m++] = j;
            }
            if(m) {
                if(each_read(c, ILQG_F_COST, cst, "cost") || each_read_int(c, ILQG_I_ITER, its, "iterations")) { rc = 1; break; }
                if(x || u) {  /* the finished trajectories, through a context of their own */
                    ilqg_batch_t *h = working_copy(c, m);
                    if(!h) { rc = 1; break; }
                    if(move_between(h, c, m, ident, fin)) { ilqg_batch_destroy(h); rc = 1; break; }
                    hx = (double *)realloc(hx, sizeof(double) * (size_t)m * (N + 1) * NXd);
                    hu = (double *)realloc(hu, sizeof(double) * (size_t)m * N * NUd);
                    if((x && ilqg_batch_get_x(h, hx)) || (u && ilqg_batch_get_u(h, hu))) { snprintf(c->err, sizeof(c->err), "%s", h->err); ilqg_batch_destroy(h); rc = 1; break; }
                    ilqg_batch_destroy(h);
                }
                for(j = 0; j < m; j++) {
                    const int b = fin[j], s0 = slot[b];
                    if(cost) cost[s0] = cst[b];
                    if(status) status[s0] = st[b];
                    if(iterations) iterations[s0] = its[b];
                    if(x) memcpy(x + (size_t)s0 * (N + 1) * NXd, hx + (size_t)j * (N + 1) * NXd, sizeof(double) * (N + 1) * NXd);
                    if(u) memcpy(u + (size_t)s0 * N * NUd, hu + (size_t)j * N * NUd, sizeof(double) * N * NUd);
                    slot[b] = -1;
                }
                n_done += m;
            }
        }
        if(c->trace_n < ILQG_TRACE_MAX) {
            c->trace_it[c->trace_n] = it;
            c->trace_active[c->trace_n] = active;
            c->trace_slots[c->trace_n] = B;
            c->trace_n++;
        }
        if(n_done >= total) break;
        /* refill: once a sixteenth of the slots is free, or nothing is left to iterate */
        n_free = 0;
        for(j = 0; j < B; j++)
            if(slot[j] < 0) freel[n_free++] = j;
        while(next < total && n_free > 0 && (n_free >= B / 16 || active == 0 || total - next <= n_free)) {
            int m = n_free < S ? n_free : S;
            if(m > total - next) m = total - next;
            memcpy(xs, x0 + (size_t)next * NXd, sizeof(double) * (size_t)m * NXd);
            memcpy(us, u0 + (size_t)next * N * NUd, sizeof(double) * (size_t)m * N * NUd);
            if(ilqg_batch_set_x0(stage, xs) || ilqg_batch_set_u(stage, us) || ilqg_batch_init(stage)) { snprintf(c->err, sizeof(c->err), "%s", stage->err); rc = 1; break; }
            if(move_between(c, stage, m, freel + (n_free - m), ident)) { rc = 1; break; }
            for(j = 0; j < m; j++) slot[freel[n_free - m + j]] = next + j;
            next += m;
            n_free -= m;
            active += m;  /* (those whose initial roll-out failed are harvested at the next poll) */
            c->compactions++;
        }
        if(rc) break;
        if(iterate_groups(c, ILQG_STREAM_ROUND)) { rc = 1; break; }
        it += ILQG_STREAM_ROUND;
    }
done:
    if(stage) ilqg_batch_destroy(stage);
    free(slot); free(st); free(its); free(freel); free(ident); free(fin); free(cst); free(xs); free(us); free(hx); free(hu);
    return rc;
}

/* the last ilqg_batch_solve, poll by poll (every 4 iterations): iterations done so far, trajectories still active, slots the
 * iterations ran over (the batch, or the smaller context the live trajectories had been gathered into); returns the number
 * of polls (at most cap are written), *compactions = how often the live set was gathered */
int ilqg_batch_solve_trace(ilqg_batch_t *c, int *iterations, int *active, int *slots, int cap, int *compactions) {
    int i;
    for(i = 0; i < c->trace_n && i < cap; i++) {
        if(iterations) iterations[i] = c->trace_it[i];
        if(active) active[i] = c->trace_active[i];
        if(slots) slots[i] = c->trace_slots[i];
    }
    if(compactions) *compactions = c->compactions;
    return c->trace_n;
}

int ilqg_batch_sync(ilqg_batch_t *c) {
    int g;
    EACH_GROUP(g) if(ilqg_dev_sync(c->dev[g])) return fail(c, "sync");
    return 0;
}

int ilqg_batch_calc_derivs(ilqg_batch_t *c) {
    int g;
    if(push_config(c)) return 1;
    EACH_GROUP(g) if(ilqg_dev_derivs(c->dev[g])) return fail(c, "calc_derivs");
    return 0;
}

int ilqg_batch_back_pass(ilqg_batch_t *c, int mode) {
    int g;
    if(push_config(c)) return 1;
    EACH_GROUP(g) if(ilqg_dev_backward(c->dev[g], mode)) return fail(c, "back_pass");
    return 0;
}

int ilqg_batch_line_search(ilqg_batch_t *c) {
    int g;
    if(push_config(c)) return 1;
    EACH_GROUP(g) {
        if(ilqg_dev_search(c->dev[g])) return fail(c, "line_search");
        if(ilqg_dev_winner(c->dev[g])) return fail(c, "line_search (winner)");
    }
    return 0;
}

int ilqg_batch_update(ilqg_batch_t *c) {
    int g;
    if(push_config(c)) return 1;
    EACH_GROUP(g) if(ilqg_dev_update(c->dev[g])) return fail(c, "update");
    return 0;
}

int ilqg_batch_get_x(ilqg_batch_t *c, double *x) { return each_read(c, ILQG_F_X, x, "get_x"); }
int ilqg_batch_get_u(ilqg_batch_t *c, double *u) { return each_read(c, ILQG_F_U, u, "get_u"); }

int ilqg_batch_get_gains(ilqg_batch_t *c, double *l, double *L) {
    return each_read(c, ILQG_F_LG, l, "get_gains") || each_read(c, ILQG_F_KG, L, "get_gains");
}

int ilqg_batch_set_gains(ilqg_batch_t *c, const double *l, const double *L) {
    return each_write(c, ILQG_F_LG, l, "set_gains") || each_write(c, ILQG_F_KG, L, "set_gains");
}

int ilqg_batch_get_derivs(ilqg_batch_t *c, double *rec, double *fin) {
    return each_read(c, ILQG_F_DER, rec, "get_derivs") || each_read(c, ILQG_F_FIN, fin, "get_derivs");
}

/* multipliers as arrays of doubles in the member order of multipliersEl_t / multipliersFin_t
 * (iLQG_problem.tem:70-89): running [B][n_hor][el], final [B][fin] */
void ilqg_problem_multiplier_dims(int *out) { ilqg_dev_multiplier_dims(out); }

int ilqg_batch_get_multipliers(ilqg_batch_t *c, double *running, double *final) {
    int dims[2];
    ilqg_dev_multiplier_dims(dims);
    if(dims[0] > 0 && running && each_read(c, ILQG_F_MUL, running, "get_multipliers")) return 1;
    if(dims[1] > 0 && final && each_read(c, ILQG_F_MULF, final, "get_multipliers")) return 1;
    return 0;
}

int ilqg_batch_set_multipliers(ilqg_batch_t *c, const double *running, const double *final) {
    int dims[2];
    ilqg_dev_multiplier_dims(dims);
    if(dims[0] > 0 && running && each_write(c, ILQG_F_MUL, running, "set_multipliers")) return 1;
    if(dims[1] > 0 && final && each_write(c, ILQG_F_MULF, final, "set_multipliers")) return 1;
    return 0;
}

int ilqg_batch_set_derivs(ilqg_batch_t *c, const double *rec, const double *fin) {
    return each_write(c, ILQG_F_DER, rec, "set_derivs") || each_write(c, ILQG_F_FIN, fin, "set_derivs");
}

static const struct { const char *name; int field; } scalar_names[] = {
    {"cost", ILQG_F_COST},         {"new_cost", ILQG_F_NEW_COST}, {"dcost", ILQG_F_DCOST},
    {"expected", ILQG_F_EXPECTED}, {"lambda", ILQG_F_LAMBDA},     {"dlambda", ILQG_F_DLAMBDA},
    {"g_norm", ILQG_F_GNORM},      {"dV0", ILQG_F_DV0},           {"dV1", ILQG_F_DV1},
    {"alpha_cost", ILQG_F_ALPHA_COST}, {"w_pen_l", ILQG_F_WPEN_L},  {"w_pen_f", ILQG_F_WPEN_F},
};
static const struct { const char *name; int field; } int_names[] = {
    {"status", ILQG_I_STATUS},     {"iterations", ILQG_I_ITER},   {"alpha_idx", ILQG_I_ALPHA_IDX},
    {"accepted", ILQG_I_ACCEPTED}, {"bp_calls", ILQG_I_BP_CALLS}, {"bp_rc", ILQG_I_BP_RC},
    {"need_derivs", ILQG_I_NEED_DERIVS}, {"alpha_ok", ILQG_I_ALPHA_OK},
};

static int find_scalar(const char *name) {
    size_t i;
    for(i = 0; i < sizeof(scalar_names) / sizeof(scalar_names[0]); i++)
        if(strcmp(scalar_names[i].name, name) == 0) return scalar_names[i].field;
    return -1;
}

static int find_int(const char *name) {
    size_t i;
    for(i = 0; i < sizeof(int_names) / sizeof(int_names[0]); i++)
        if(strcmp(int_names[i].name, name) == 0) return int_names[i].field;
    return -1;
}

int ilqg_batch_get_scalar(ilqg_batch_t *c, const char *name, double *out) {
    int f = find_scalar(name);
    if(f < 0) return fail_msg(c, "no such scalar field");
    return each_read(c, f, out, name);
}

int ilqg_batch_set_scalar(ilqg_batch_t *c, const char *name, const double *in) {
    int f = find_scalar(name);
    if(f < 0) return fail_msg(c, "no such scalar field");
    return each_write(c, f, in, name);
}

int ilqg_batch_get_int(ilqg_batch_t *c, const char *name, int *out) {
    int f = find_int(name);
    if(f < 0) return fail_msg(c, "no such int field");
    return each_read_int(c, f, out, name);
}

int ilqg_batch_set_int(ilqg_batch_t *c, const char *name, const int *in) {
    int f = find_int(name);
    if(f < 0) return fail_msg(c, "no such int field");
    return each_write_int(c, f, in, name);
}

/* device address of the cost vector: contiguous only while the batch is one group (NULL otherwise: read the costs
 * with ilqg_batch_get_scalar) */
void *ilqg_batch_cost_device_ptr(ilqg_batch_t *c) { return c->groups == 1 ? ilqg_dev_field_ptr(c->dev[0], ILQG_F_COST) : NULL; }
void *ilqg_batch_stream(ilqg_batch_t *c) { return ilqg_dev_stream(c->dev[0]); }

/* a per-trajectory scalar ("cost", ...) of the whole batch into contiguous device memory of the caller (batch
 * doubles), without a host copy; synchronises.  This is what a collective over the costs is given. */
int ilqg_batch_scalar_to_device(ilqg_batch_t *c, const char *name, void *dst_device) {
    int g, f = find_scalar(name);
    if(f < 0) return fail_msg(c, "no such scalar field");
    EACH_GROUP(g) if(ilqg_dev_copy_scalar_to(c->dev[g], f, (double *)dst_device + c->first[g])) return fail(c, name);
    return ilqg_batch_sync(c);
}

int ilqg_batch_timing(ilqg_batch_t *c, int enable) {
    int g;
    EACH_GROUP(g) if(ilqg_dev_timing(c->dev[g], enable)) return fail(c, "timing");
    return 0;
}
int ilqg_batch_kernel_count(void) { return ILQG_K_COUNT; }
const char *ilqg_batch_kernel_name(int k) { return ilqg_dev_kernel_name(k); }
/* summed over the groups (whose kernels overlap in time) */
int ilqg_batch_get_timing(ilqg_batch_t *c, int kernel, int *launches, double *total_ms) {
    int g, n;
    double ms;
    *launches = 0;
    *total_ms = 0.0;
    EACH_GROUP(g) {
        if(ilqg_dev_get_timing(c->dev[g], kernel, &n, &ms)) return fail(c, "get_timing");
        *launches += n;
        *total_ms += ms;
    }
    return 0;
}

/* wall-clock time a kernel occupied (union of its launch intervals), summed over the groups */
int ilqg_batch_get_busy(ilqg_batch_t *c, int kernel, double *busy_ms) {
    int g;
    double ms;
    *busy_ms = 0.0;
    EACH_GROUP(g) {
        if(ilqg_dev_get_busy(c->dev[g], kernel, &ms)) return fail(c, "get_busy");
        *busy_ms += ms;
    }
    return 0;
}

int ilqg_boxqp_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                     const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    return ilqg_dev_boxqp_batch(device, n, count, H, g, lower, upper, x, clamp, n_free, invH, rc);
}

int ilqg_boxqp_wave_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                          const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    return ilqg_dev_boxqp_wave_batch(device, n, count, H, g, lower, upper, x, clamp, n_free, invH, rc);
}

int ilqg_boxqp_table_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                           const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    return ilqg_dev_boxqp_table_batch(device, n, count, H, g, lower, upper, x, clamp, n_free, invH, rc);
}

int ilqg_sincos_batch(int device, int n, const double *x, double *s, double *c) {
    return ilqg_dev_sincos_batch(device, n, x, s, c);
}

/* =========================================================================
 * drop-in single-trajectory entry points (batch of one on the device)
 * ========================================================================= */
static int env_device(void) {
    const char *e = getenv("ILQG_DEVICE");
    return e ? atoi(e) : 0;
}

static ilqg_batch_t *backend_of(tOptSet *o, const char *who) {
    ilqg_batch_t *c = (ilqg_batch_t *)o->backend;
    int i;
    if(c && c->N != o->n_hor) {
        ilqg_batch_destroy(c);
        c = NULL;
    }
    if(!c) {
        c = ilqg_batch_create(env_device(), 1, o->n_hor);
        if(!c) fatal_no_device(who, ilqg_batch_error(NULL));
        o->backend = c;
    }
    /* mirror the caller's options and (borrowed) parameters */
    c->opt.alpha = o->alpha;
    c->opt.n_alpha = o->n_alpha;
    c->opt.tolFun = o->tolFun; c->opt.tolGrad = o->tolGrad; c->opt.tolConstraint = o->tolConstraint;
    c->opt.lambdaInit = o->lambdaInit; c->opt.dlambdaInit = o->dlambdaInit; c->opt.lambdaFactor = o->lambdaFactor;
    c->opt.lambdaMax = o->lambdaMax; c->opt.lambdaMin = o->lambdaMin;
    c->opt.zMin = o->zMin; c->opt.regType = o->regType; c->opt.max_iter = o->max_iter;
    c->opt.w_pen_init_l = o->w_pen_l; c->opt.w_pen_init_f = o->w_pen_f; /* current penalty weights */
    c->opt.w_pen_max_l = o->w_pen_max_l; c->opt.w_pen_max_f = o->w_pen_max_f;
    c->opt.w_pen_fact1 = o->w_pen_fact1; c->opt.w_pen_fact2 = o->w_pen_fact2;
    for(i = 0; i < n_params; i++) {
        if(!c->p_given[i] || memcmp(c->p[i], o->p[i], sizeof(double) * param_len(c, i)) != 0) {
            memcpy(c->p[i], o->p[i], sizeof(double) * param_len(c, i));
            c->p_given[i] = 1;
            c->params_pushed = 0;
        }
    }
    if(push_config(c)) fatal_no_device(who, c->err);
    return c;
}

void ilqg_release(tOptSet *o) {
    if(o && o->backend) {
        ilqg_batch_destroy((ilqg_batch_t *)o->backend);
        o->backend = NULL;
    }
}

#define DEV_OK(call, who)                                   \
    do {                                                    \
        if(call) fatal_no_device(who, ilqg_dev_error());    \
    } while(0)

static void pack_xu(const traj_t *tr, int N, double *x, double *u) {
    int k;
    for(k = 0; k < N; k++) {
        memcpy(x + k * N_X, tr->t[k].x, sizeof(double) * N_X);
        memcpy(u + k * N_U, tr->t[k].u, sizeof(double) * N_U);
    }
    memcpy(x + N * N_X, tr->f.x, sizeof(double) * N_X);
}

/* host arrays of the drop-in entry points: one block owned by the backend, grown on demand */
static double *dropin_scratch(ilqg_batch_t *c, size_t doubles, const char *who) {
    if(doubles > c->scratch_doubles) {
        free(c->scratch);
        c->scratch = (double *)malloc(sizeof(double) * doubles);
        c->scratch_doubles = c->scratch ? doubles : 0;
        if(!c->scratch) fatal_no_device(who, "out of host memory");
    }
    return c->scratch;
}

/* Backward Riccati sweep of the nominal trajectory on the GPU.
 * Same contract as reference back_pass.c:38-257: reads the derivative fields
 * calc_derivs() left in o->nominal, writes t[k].l, t[k].L, o->dV, o->g_norm;
 * returns 0, or 1 when the box QP failed at some step.
 * All transfers of the call are queued on the stream and waited for once (ilqg_dev_io_begin / _end). */
int back_pass(tOptSet *o) {
    static const char who[] = "back_pass()";
    ilqg_batch_t *c = backend_of(o, who);
    ilqg_dev_t *d = c->dev[0];
    const int N = o->n_hor;
    const size_t n_rec = (size_t)N * REC_HOST_SIZE + FIN_SIZE, n_u = (size_t)N * N_U, n_l = (size_t)N * N_U, n_L = (size_t)N * N_U * N_X;
    double *rec = dropin_scratch(c, n_rec + n_u + n_l + n_L, who);
    double *fin = rec + (size_t)N * REC_HOST_SIZE, *u = rec + n_rec, *l = u + n_u, *L = l + n_l;
    double *r = rec, g_norm = 0.0;
    int k, rc = 1, zero = 0;

    for(k = 0; k < N; k++) {
        const trajEl_t *t = &o->nominal->t[k];
#define PUT(field, cnt) do { memcpy(r, (field), sizeof(double) * (cnt)); r += (cnt); } while(0)
        PUT(t->cx, N_X); PUT(t->cxx, sizeofQxx); PUT(t->cu, N_U); PUT(t->cuu, sizeofQuu);
        PUT(t->cxu, sizeofQxu); PUT(t->fx, N_X * N_X); PUT(t->fu, N_X * N_U);
        PUT(t->lower, N_U); PUT(t->upper, N_U);
#if FULL_DDP
        PUT(t->fxx, N_X * sizeofQxx); PUT(t->fuu, N_X * sizeofQuu); PUT(t->fxu, N_X * sizeofQxu);
#endif
        PUT(t->lower_sign, N_U); PUT(t->upper_sign, N_U);
        PUT(t->lower_hx, N_X * N_U); PUT(t->upper_hx, N_X * N_U);
        memcpy(u + (size_t)k * N_U, t->u, sizeof(double) * N_U);
    }
    r = fin;
    PUT(o->nominal->f.cx, N_X); PUT(o->nominal->f.cxx, sizeofQxx);
#undef PUT

    DEV_OK(ilqg_dev_io_begin(d), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_DER, rec), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_FIN, fin), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_U, u), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_LAMBDA, &o->lambda), who);
    DEV_OK(ilqg_dev_write_int(d, ILQG_I_STATUS, &zero), who);
    DEV_OK(ilqg_dev_backward(d, 1), who);
    DEV_OK(ilqg_dev_read_int(d, ILQG_I_BP_RC, &rc), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_LG, l), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_KG, L), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_DV0, &o->dV[0]), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_DV1, &o->dV[1]), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_GNORM, &g_norm), who);
    DEV_OK(ilqg_dev_io_end(d), who);
    if(!rc) o->g_norm = g_norm;  /* an abandoned sweep leaves it alone (back_pass.c:168-171,254) */
    for(k = 0; k < N; k++) {
        memcpy(o->nominal->t[k].l, l + (size_t)k * N_U, sizeof(double) * N_U);
        memcpy(o->nominal->t[k].L, L + (size_t)k * N_U * N_X, sizeof(double) * N_U * N_X);
    }
    return rc;
}

/* Line search on the GPU: all step sizes of o->alpha are rolled out in
 * parallel, the first acceptable one (lowest index, reference
 * line_search.c:37-60) is re-rolled and stored.  Same contract as reference
 * line_search.c:33-78: candidate left in o->candidates[0], o->new_cost /
 * dcost / expected and the optional logs written; returns 1 if accepted. */
int line_search(tOptSet *o, int iter) {
    static const char who[] = "line_search()";
    ilqg_batch_t *c = backend_of(o, who);
    ilqg_dev_t *d = c->dev[0];
    const int N = o->n_hor;
    const size_t n_x = (size_t)(N + 1) * N_X, n_u = (size_t)N * N_U, n_L = (size_t)N * N_U * N_X;
    double *x = dropin_scratch(c, 2 * n_x + 3 * n_u + n_L, who);
    double *u = x + n_x, *l = u + n_u, *L = l + n_u, *xc = L + n_L, *uc = xc + n_x;
    int k, accepted = 0, idx = 0, zero = 0;
    int ok[ILQG_MAX_ALPHA];  /* finite-flag of every step size's roll-out (the replay of the reference's console walk below) */
    double cnew = 0.0, dcost = 0.0, expected = 0.0, z, tmp;

    pack_xu(o->nominal, N, x, u);
    for(k = 0; k < N; k++) {
        memcpy(l + (size_t)k * N_U, o->nominal->t[k].l, sizeof(double) * N_U);
        memcpy(L + (size_t)k * N_U * N_X, o->nominal->t[k].L, sizeof(double) * N_U * N_X);
    }
    DEV_OK(ilqg_dev_io_begin(d), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_X, x), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_U, u), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_LG, l), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_KG, L), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_COST, &o->cost), who);
    if(sizeof(multipliersEl_t) > 0) { /* structs of doubles (iLQG_problem.tem:70-89): the array as it is */
        DEV_OK(ilqg_dev_write(d, ILQG_F_MUL, (const double *)o->multipliers.t), who);
    }
    if(sizeof(multipliersFin_t) > 0) {
        DEV_OK(ilqg_dev_write(d, ILQG_F_MULF, (const double *)&o->multipliers.f), who);
    }
    DEV_OK(ilqg_dev_write(d, ILQG_F_WPEN_L, &o->w_pen_l), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_WPEN_F, &o->w_pen_f), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_DV0, &o->dV[0]), who);
    DEV_OK(ilqg_dev_write(d, ILQG_F_DV1, &o->dV[1]), who);
    DEV_OK(ilqg_dev_write_int(d, ILQG_I_STATUS, &zero), who);
    DEV_OK(ilqg_dev_search(d), who);
    DEV_OK(ilqg_dev_winner(d), who);
    DEV_OK(ilqg_dev_read_int(d, ILQG_I_ACCEPTED, &accepted), who);
    DEV_OK(ilqg_dev_read_int(d, ILQG_I_ALPHA_IDX, &idx), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_NEW_COST, &cnew), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_DCOST, &dcost), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_EXPECTED, &expected), who);
    DEV_OK(ilqg_dev_read(d, ILQG_F_X, xc), who);  /* the stored winner; unchanged nominal if nothing was accepted */
    DEV_OK(ilqg_dev_read(d, ILQG_F_U, uc), who);
    if(DEBUG_FORWARDPASS) DEV_OK(ilqg_dev_read_int(d, ILQG_I_ALPHA_OK, ok), who);  /* (same batch: no second round trip) */
    DEV_OK(ilqg_dev_io_end(d), who);
    if(accepted) {
        traj_t *cand = o->candidates[0];
        for(k = 0; k < N; k++) {
            memcpy(cand->t[k].x, xc + (size_t)k * N_X, sizeof(double) * N_X);
            memcpy(cand->t[k].u, uc + (size_t)k * N_U, sizeof(double) * N_U);
        }
        memcpy(cand->f.x, xc + (size_t)N * N_X, sizeof(double) * N_X);
        /* let the generated code refresh the members it caches per step (auxiliaries, c)
         * from the stored x,u: its cost-only mode touches nothing else (iLQG_func.tem:160-176) */
        forward_pass(cand, o, 0.0, &tmp, 1);
    }
    z = (expected > 0) ? dcost / expected : 0;
    if(DEBUG_FORWARDPASS) {
        /* what the reference says while it walks the step sizes one by one (line_search.c:44-66): the device has
         * tried them all at once and kept every cost and finite-flag, so the same walk is replayed here */
        int tried = accepted ? idx : o->n_alpha, i;
        for(i = 0; i < tried && i < o->n_alpha; i++) {
            if(!ok[i])
                SAY_SEARCH(2, ("line search: %-3d: prediction or objective failed with inf or nan\n", i + 1));
            else if(!(-o->alpha[i] * (o->dV[0] + o->alpha[i] * o->dV[1]) > 0))
                SAY_SEARCH(-1000, ("non-positive expected reduction: should not occur (dV[0]= %g, dV[1]= %g)\n", o->dV[0], o->dV[1]));
        }
        if(!accepted) SAY_SEARCH(2, ("max number of line searches reached\n"));
    }
    if(o->log_linesearch != NULL) o->log_linesearch[iter] = idx;
    if(o->log_z != NULL) o->log_z[iter] = z;
    if(o->log_cost != NULL) o->log_cost[iter] = cnew;
    o->new_cost = cnew;
    o->dcost = dcost;
    o->expected = expected;
    return accepted;
}

/* reference boxQP.c:39-238, executed by the device routine the backward kernel uses */
int boxQP(double *H, const double *g, const double *lower, const double *upper, double *x, double *Hfree,
          double *L, double *grad, double *grad_clamped, double *search, int *is_clamped, int *n_free_,
          double *invHfree, const int n) {
    int rc = 0, i, j, fi, fj;
    double inv_full[(N_U * (N_U + 1) / 2 > 36) ? N_U * (N_U + 1) / 2 : 36];  /* n <= max(8, N_U) */
    (void)Hfree; (void)L; (void)grad; (void)grad_clamped; (void)search;
    if(n != 2 && n != 8 && n != N_U) {
        fprintf(stderr, "ilqg: boxQP on the device supports n in {2, 8, N_U}, got %d\n", n);
        abort();
    }
    if(ilqg_dev_boxqp_batch(env_device(), n, 1, H, g, lower, upper, x, is_clamped, n_free_, inv_full, &rc))
        fatal_no_device("boxQP()", ilqg_dev_error());
    /* compact the full-index inverse to the reference's free-block numbering */
    for(j = 0, fj = 0; j < n; j++) {
        if(is_clamped[j]) continue;
        for(i = 0, fi = 0; i <= j; i++) {
            if(is_clamped[i]) continue;
            invHfree[UTRI_MAT_IDX(fi, fj)] = inv_full[UTRI_MAT_IDX(i, j)];
            fi++;
        }
        fj++;
    }
    return rc;
}

/* ---- the reference's small dense helpers, executed on the device (matMult.h:11-14, cholesky.h:4-6) ---- */
#include "cholesky.h"

static void dense_call(const char *who, int op, int shape, const double *in0, int n0, const double *in1, int n1,
                       const double *in2, int n2, double *out, int nout, int *flag) {
    if(ilqg_dev_dense(env_device(), op, shape, in0, n0, in1, n1, in2, n2, out, nout, flag))
        fatal_no_device(who, ilqg_dev_error());
    if(*flag == -1) {
        fprintf(stderr, "ilqg: %s: this size is not built into the device library of this problem\n", who);
        abort();
    }
}

#define TRI(n) (((n) * ((n) + 1)) / 2)

void addMulVec(double base[], const double a[], const double b[], const int n_r, const int n_c) {
    int flag, shape = (n_r == N_X && n_c == N_U) ? 0 : (n_r == N_X && n_c == N_X) ? 1 : -1;
    if(shape < 0) { fprintf(stderr, "ilqg: addMulVec: sizes (%d,%d) not built\n", n_r, n_c); abort(); }
    dense_call("addMulVec()", 0, shape, a, n_r, b, n_r * n_c, NULL, 0, base, n_c, &flag);
}

void addSquareTri(double base[], const double b[], const double a[], const int n_r, const int n_c, double ba[]) {
    int flag, shape = (n_r == N_X && n_c == N_U) ? 0 : (n_r == N_X && n_c == N_X) ? 1 : (n_r == N_U && n_c == N_X) ? 2 : -1;
    (void)ba;
    if(shape < 0) { fprintf(stderr, "ilqg: addSquareTri: sizes (%d,%d) not built\n", n_r, n_c); abort(); }
    dense_call("addSquareTri()", 1, shape, b, TRI(n_r), a, n_r * n_c, NULL, 0, base, TRI(n_c), &flag);
}

void addMul2Tri(double base[], const double b[], const double a[], const int n_ra, const int n_ca, const double c[],
                const int n_rc, const int n_cc, double bc[]) {
    int flag, shape = (n_ra == N_X && n_ca == N_X && n_rc == N_X && n_cc == N_U) ? 0
                    : (n_ra == N_U && n_ca == N_X && n_rc == N_U && n_cc == 1) ? 2 : -1;
    (void)bc;
    if(shape < 0) { fprintf(stderr, "ilqg: addMul2Tri: sizes not built\n"); abort(); }
    dense_call("addMul2Tri()", 2, shape, b, TRI(n_ra), a, n_ra * n_ca, c, n_rc * n_cc, base, n_ca * n_cc, &flag);
}

int cholesky_tri(const double *A, int n, double *L) {
    int flag;
    dense_call("cholesky_tri()", 3, n, A, TRI(n), NULL, 0, NULL, 0, L, TRI(n), &flag);
    return flag;
}

void cholesky_tri_inv(const double *L_, double *invA, const int n, double *x) {
    int flag;
    (void)x;
    dense_call("cholesky_tri_inv()", 4, n, L_, TRI(n), NULL, 0, NULL, 0, invA, TRI(n), &flag);
}

static void lambda_increase(tOptSet *o, double *dlambda) {
    *dlambda = max(*dlambda * o->lambdaFactor, o->lambdaFactor);
    o->lambda = max(o->lambda * *dlambda, o->lambdaMin);
}

static void lambda_decrease(tOptSet *o, double *dlambda) {
    *dlambda = min(*dlambda / o->lambdaFactor, 1.0 / o->lambdaFactor);
    o->lambda = o->lambda * *dlambda * (o->lambda > o->lambdaMin);
}

/* Outer iteration for ONE trajectory, where the reference has it: on the host
 * (iLQG.c:224-379), with the two hot stages on the GPU.  `done` starts at 0, so
 * a calc_derivs failure in the first iteration returns 0 instead of reading an
 * uninitialised flag. */
int iLQG(tOptSet *o) {
    int iter, done = 0, stepped, fresh = 1;
    double dlambda = o->dlambdaInit;

    o->lambda = o->lambdaInit;
    o->w_pen_l = o->w_pen_init_l;
    o->w_pen_f = o->w_pen_init_f;
    update_multipliers(o, 1);

    for(iter = 0; iter < o->max_iter; iter++) {
        if(fresh) {
            if(!calc_derivs(o)) {
                SAY_LOOP(-1000, ("Calculating derivatives failed.\n"));
                break;
            }
            fresh = 0;
        }
        for(done = 0; !done;) {
            if(!back_pass(o)) {
                done = 1;
            } else {
                SAY_LOOP(1, ("Back pass failed.\n"));
                lambda_increase(o, &dlambda);
                if(o->lambda > o->lambdaMax) break;
            }
        }
        if(o->g_norm < o->tolGrad && o->lambda < 1e-5) {
            lambda_decrease(o, &dlambda);
            SAY_LOOP(1, ("\nSUCCESS: gradient norm < tolGrad\n"));
            break;
        }
        if(!done) break;

        stepped = line_search(o, iter);
        if(stepped) {
            SAY_LOOP(1, ("iter: %-3d  cost: %-9.6g  reduction: %-9.3g  gradient: %-9.3g  z: %-5.3g log10(lam): %3.1f "
                         "w_pen_l: %-9.3g w_pen_f: %-9.3g\n", iter + 1, o->cost, o->dcost, o->g_norm,
                         o->dcost / o->expected, log10(o->lambda), o->w_pen_l, o->w_pen_f));
            lambda_decrease(o, &dlambda);
            makeCandidateNominal(o, 0);
            o->cost = o->new_cost;
            fresh = 1;
            if(o->dcost < o->tolFun) {
                SAY_LOOP(1, ("\nSUCCESS: cost change < tolFun\n"));
                break;
            }
            update_multipliers(o, 0);
            forward_pass(o->nominal, o, 0.0, &o->cost, 1);
        } else {
            lambda_increase(o, &dlambda);
            if(o->w_pen_fact2 > 1.0) {
                o->w_pen_l = min(o->w_pen_max_l, o->w_pen_l * o->w_pen_fact2);
                o->w_pen_f = min(o->w_pen_max_f, o->w_pen_f * o->w_pen_fact2);
                forward_pass(o->nominal, o, 0.0, &o->cost, 1);
            }
            /* (the second label reads w_pen_l in the reference too, iLQG.c:353) */
            SAY_LOOP(1, ("iter: %-3d  REJECTED    expected: %-11.3g    actual: %-11.3g    log10lam: %3.1f "
                         "w_pen_l: %-9.3g w_pen_l: %-9.3g\n", iter + 1, o->expected, o->dcost, log10(o->lambda),
                         o->w_pen_l, o->w_pen_f));
            if(o->lambda > o->lambdaMax) {
                SAY_LOOP(1, ("\nEXIT: lambda > lambdaMax\n"));
                break;
            }
        }
    }
    o->iterations = iter;
    if(!done) {
        SAY_LOOP(1, ("\nEXIT: no descent direction found.\n"));
        return 0;
    }
    if(iter >= o->max_iter) {
        SAY_LOOP(1, ("\nEXIT: Maximum iterations reached.\n"));
        return 0;
    }
    return 1;
}

/* =========================================================================
 * several GPUs of one node, one process                         SURVEY 8(e)
 * =========================================================================
 * The reference is single-threaded and has no counterpart.  Trajectories are independent: device g owns the
 * contiguous block [first_g, first_g + count_g) of the batch (blocks of ceil(B / G)), advances it with the batch
 * interface above and never talks to the others; the one exchange is a single RCCL gather of the per-trajectory
 * costs (ilqg_multi_gather_costs).  All launches are asynchronous, so one host thread keeps every device busy:
 * the devices are served one iteration at a time in turn. */
struct ilqg_multi {
    int n, B, N, per;
    ilqg_batch_t *shard[ILQG_MULTI_MAX];
    int device[ILQG_MULTI_MAX], first[ILQG_MULTI_MAX], count[ILQG_MULTI_MAX];
    ilqg_comm_t *comm;
    char err[512];
};

static char g_multi_err[512];
const char *ilqg_multi_error(const ilqg_multi_t *m) { return m ? m->err : g_multi_err; }

static int multi_fail(ilqg_multi_t *m, int g) {
    snprintf(m->err, sizeof(m->err), "device %d: %s", m->device[g], ilqg_batch_error(m->shard[g]));
    return 1;
}

void ilqg_multi_destroy(ilqg_multi_t *m) {
    int g;
    if(!m) return;
    if(m->comm) ilqg_comm_destroy(m->comm);
    for(g = 0; g < m->n; g++) ilqg_batch_destroy(m->shard[g]);
    free(m);
}

ilqg_multi_t *ilqg_multi_create(int n_devices, const int *devices, int batch, int n_hor) {
    ilqg_multi_t *m;
    int g;
    if(n_devices < 1 || n_devices > ILQG_MULTI_MAX || batch < n_devices) {
        snprintf(g_multi_err, sizeof(g_multi_err), "ilqg_multi_create: 1..%d devices, at least one trajectory each", ILQG_MULTI_MAX);
        return NULL;
    }
    m = (ilqg_multi_t *)calloc(1, sizeof(*m));
    if(!m) return NULL;
    m->B = batch;
    m->N = n_hor;
    m->per = (batch + n_devices - 1) / n_devices;
    for(g = 0; g < n_devices && g * m->per < batch; g++) {
        m->device[g] = devices ? devices[g] : g;
        m->first[g] = g * m->per;
        m->count[g] = (batch - m->first[g] < m->per) ? batch - m->first[g] : m->per;
        m->shard[g] = ilqg_batch_create(m->device[g], m->count[g], n_hor);
        if(!m->shard[g]) {
            snprintf(g_multi_err, sizeof(g_multi_err), "device %d: %s", m->device[g], ilqg_batch_error(NULL));
            ilqg_multi_destroy(m);
            return NULL;
        }
        m->n = g + 1;
    }
    if(ilqg_comm_create(&m->comm, m->n, m->device, m->per)) {
        snprintf(g_multi_err, sizeof(g_multi_err), "RCCL communicator: %s", ilqg_dev_error());
        ilqg_multi_destroy(m);
        return NULL;
    }
    return m;
}

int ilqg_multi_devices(const ilqg_multi_t *m) { return m->n; }
ilqg_batch_t *ilqg_multi_shard(ilqg_multi_t *m, int g, int *first, int *count) {
    if(g < 0 || g >= m->n) return NULL;
    if(first) *first = m->first[g];
    if(count) *count = m->count[g];
    return m->shard[g];
}

#define EACH_SHARD(g) for(g = 0; g < m->n; g++)
int ilqg_multi_set_option(ilqg_multi_t *m, const char *name, const double *value, int n) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_set_option(m->shard[g], name, value, n)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_set_param(ilqg_multi_t *m, const char *name, const double *value, int n) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_set_param(m->shard[g], name, value, n)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_set_x0(ilqg_multi_t *m, const double *x0) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_set_x0(m->shard[g], x0 + (size_t)m->first[g] * N_X)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_set_u(ilqg_multi_t *m, const double *u) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_set_u(m->shard[g], u + (size_t)m->first[g] * m->N * N_U)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_init(ilqg_multi_t *m) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_init(m->shard[g])) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_iterate(ilqg_multi_t *m, int n) {
    int it, g;
    for(it = 0; it < n; it++) EACH_SHARD(g) if(ilqg_batch_iterate(m->shard[g], 1)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_sync(ilqg_multi_t *m) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_sync(m->shard[g])) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_active(ilqg_multi_t *m, int *n_active) {
    int g, a;
    *n_active = 0;
    EACH_SHARD(g) {
        if(ilqg_batch_active(m->shard[g], &a)) return multi_fail(m, g);
        *n_active += a;
    }
    return 0;
}
int ilqg_multi_solve(ilqg_multi_t *m) {
    int it, active = 1, max_iter = m->shard[0]->opt.max_iter;
    for(it = 0; it < max_iter && active; it += 4) {
        if(ilqg_multi_iterate(m, max_iter - it < 4 ? max_iter - it : 4)) return 1;
        if(ilqg_multi_active(m, &active)) return 1;
    }
    return 0;
}
int ilqg_multi_get_x(ilqg_multi_t *m, double *x) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_get_x(m->shard[g], x + (size_t)m->first[g] * (m->N + 1) * N_X)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_get_u(ilqg_multi_t *m, double *u) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_get_u(m->shard[g], u + (size_t)m->first[g] * m->N * N_U)) return multi_fail(m, g);
    return 0;
}
int ilqg_multi_get_int(ilqg_multi_t *m, const char *name, int *out) {
    int g;
    EACH_SHARD(g) if(ilqg_batch_get_int(m->shard[g], name, out + (size_t)m->first[g] * (strcmp(name, "alpha_ok") ? 1 : ILQG_MAX_ALPHA))) return multi_fail(m, g);
    return 0;
}

/* the path's single collective: every device copies the costs of its shard (all its groups of trajectories) into its
 * send buffer, then ONE ncclGather moves them to device 0 and on to the host */
int ilqg_multi_gather_costs(ilqg_multi_t *m, double *cost) {
    ilqg_dev_t *devs[ILQG_MULTI_MAX];
    int g;
    EACH_SHARD(g) {
        if(ilqg_batch_scalar_to_device(m->shard[g], "cost", ilqg_comm_send_buffer(m->comm, g))) return multi_fail(m, g);
        devs[g] = m->shard[g]->dev[0];
    }
    if(ilqg_comm_gather(m->comm, devs, m->first, m->count, cost)) {
        snprintf(m->err, sizeof(m->err), "gather of costs: %s", ilqg_dev_error());
        return 1;
    }
    return 0;
}

/* =========================================================================
 * the reference's MEX entry without MEX                 iLQG_mex.c:19-144
 * =========================================================================
 * [success, x, u, cost] = iLQG<Problem>(x0, u_nom, params, opts) for a C caller: same sequence — options by
 * name through setOptParam, every parameter of paramdesc[] by name with its length checked, trajectory buffers,
 * init_opt, the initial roll-out, iLQG() — and the same messages.  x is [n_hor+1][N_X], u is [n_hor][N_U] (the MEX
 * entry's column-major x_new(n,N), u_new(m,N-1)).  Returns iLQG()'s 1 / 0, or -1 with a message in err when an
 * argument is refused (where the MEX entry raises an error).  seconds: wall time of the iLQG() call alone, as the
 * reference times it (iLQG_mex.c:123-126). */
#include <time.h>

static const ilqg_named_t *find_named(const ilqg_named_t *list, int n, const char *name) {
    int i;
    for(i = 0; i < n; i++)
        if(strcmp(list[i].name, name) == 0) return &list[i];
    return NULL;
}

int ilqg_solve_single(int n_hor, const double *x0, const double *u_nom, const ilqg_named_t *params, int n_given,
                      const ilqg_named_t *opts, int n_opts, double *x, double *u, double *cost, int *iterations,
                      double *seconds, char *err, int err_len) {
    tOptSet o = INIT_OPTSET;
    struct timespec t0, t1;
    int i, k, ok = 0, success = 0;
    double x0_copy[N_X];

    if(err && err_len > 0) err[0] = 0;
    if(seconds) *seconds = 0.0;
    if(n_hor < 2) {
        if(err) snprintf(err, err_len, "There must be more than one time step.");
        return -1;
    }
    memcpy(x0_copy, x0, sizeof(x0_copy));
    o.x0 = x0_copy;
    o.n_hor = n_hor;
    standard_parameters(&o);
    for(i = 0; i < n_opts; i++) {
        char *msg = setOptParam(&o, opts[i].name, opts[i].value, opts[i].n);
        if(msg) {
            if(err) snprintf(err, err_len, "Error setting optimization parameter '%s': %s.", opts[i].name, msg);
            return -1;
        }
    }
    o.p = (double **)calloc(n_params > 0 ? n_params : 1, sizeof(double *));
    for(i = 0; i < n_params; i++) {
        const int want = (paramdesc[i]->size == -1) ? n_hor + 1 : paramdesc[i]->size;
        const ilqg_named_t *g = find_named(params, n_given, paramdesc[i]->name);
        if(!g) {
            if(err) snprintf(err, err_len, "Parameter name '%s' is not member of parameters struct.", paramdesc[i]->name);
            free(o.p);
            return -1;
        }
        if(g->n != want) {
            if(err) snprintf(err, err_len, "Parameter name '%s' must be a vector length %d.", paramdesc[i]->name, want);
            free(o.p);
            return -1;
        }
        o.p[i] = (double *)g->value;  /* borrowed, as in the reference */
    }
    for(i = 0; i < NUMBER_OF_THREADS + 1; i++) o.trajectories[i].t = (trajEl_t *)calloc(n_hor, sizeof(trajEl_t));
    o.multipliers.t = (multipliersEl_t *)calloc(n_hor + 1, sizeof(multipliersEl_t) > 0 ? sizeof(multipliersEl_t) : 1);

    if(init_opt(&o)) {
        for(k = 0; k < n_hor; k++) memcpy(o.nominal->t[k].u, u_nom + (size_t)k * N_U, sizeof(double) * N_U);
        if(forward_pass(o.candidates[0], &o, 0.0, &o.cost, 0)) {
            makeCandidateNominal(&o, 0);
            clock_gettime(CLOCK_MONOTONIC, &t0);
            success = iLQG(&o);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if(seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            for(k = 0; k < n_hor; k++) {
                memcpy(x + (size_t)k * N_X, o.nominal->t[k].x, sizeof(double) * N_X);
                memcpy(u + (size_t)k * N_U, o.nominal->t[k].u, sizeof(double) * N_U);
            }
            memcpy(x + (size_t)n_hor * N_X, o.nominal->f.x, sizeof(double) * N_X);
            ok = 1;
        }
    }
    if(cost) *cost = o.cost;
    if(iterations) *iterations = o.iterations;
    ilqg_release(&o);
    free(o.p);
    for(i = 0; i < NUMBER_OF_THREADS + 1; i++) free(o.trajectories[i].t);
    free(o.multipliers.t);
    return ok ? success : 0;
}

