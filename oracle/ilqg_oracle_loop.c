/* TEST INFRASTRUCTURE — CPU restatement of the reference's outer loop.
 *
 * Restates reference iLQG.c (outer iteration, regularisation schedule, option
 * handling) behind the same C symbols.  Kept in its own translation unit so
 * the harness can interpose back_pass()/line_search() with the linker's
 * --wrap option to record per-iteration traces.  See ilqg_oracle.c for the
 * role and pinning status of the oracle; the same rules apply here.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "mex.h"
#include "iLQG.h"
#include "back_pass.h"
#include "line_search.h"
#include "printMat.h"

/* ========================================================================
 * outer loop and options                         reference iLQG.c:36-386
 * ======================================================================== */
static double oracle_default_alpha[] = {1.0, 0.3727594, 0.1389495, 0.0517947,
                                        0.0193070, 0.0071969, 0.0026827, 0.0010000}; /* iLQG.c:36 */

void standard_parameters(tOptSet *o) { /* iLQG.c:57-78 */
    o->alpha = oracle_default_alpha;
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

static char E_scalar[] = "parameter must be scalar";
static char E_arange[] = "all alpha must be in the range [1.0..0.0)";
static char E_amono[] = "all alpha must be monotonically decreasing";
static char E_pos[] = "parameter must be positive";
static char E_gt1[] = "parameter must be > 1";
static char E_12[] = "parameter must be in range [1..2]";
static char E_01[] = "parameter must be in range [0..1)";
static char E_dbg[] = "parameter must be in range [0..6]";
static char E_none[] = "no such parameter";

/* validation rules of iLQG.c:91-216, one row per key */
#define OPT_D(key, field, bad, msg)                         \
    if(strcmp(name, key) == 0) {                            \
        if(n != 1) return E_scalar;                         \
        if(bad) return msg;                                 \
        o->field = value[0];                                \
        return NULL;                                        \
    }

char *setOptParam(tOptSet *o, const char *name, const double *value, const int n) {
    if(strcmp(name, "alpha") == 0) {
        for(int i = 0; i < n; i++) {
            if(value[i] < 0.0 || value[i] > 1.0) return E_arange;
            if(i > 0 && value[i] >= value[i - 1]) return E_amono;
        }
        o->alpha = value;
        o->n_alpha = n;
        return NULL;
    }
    OPT_D("tolFun", tolFun, value[0] <= 0.0, E_pos)
    OPT_D("tolConstraint", tolConstraint, value[0] <= 0.0, E_pos)
    OPT_D("tolGrad", tolGrad, value[0] <= 0.0, E_pos)
    OPT_D("max_iter", max_iter, value[0] < 0.0, E_pos)
    OPT_D("lambdaInit", lambdaInit, value[0] < 0.0, E_pos)
    OPT_D("dlambdaInit", dlambdaInit, value[0] < 0.0, E_pos)
    OPT_D("lambdaFactor", lambdaFactor, value[0] < 1.0, E_gt1)
    OPT_D("lambdaMax", lambdaMax, value[0] < 0.0, E_pos)
    OPT_D("lambdaMin", lambdaMin, value[0] < 0.0, E_pos)
    OPT_D("regType", regType, value[0] < 1.0 || value[0] > 2.0, E_12)
    OPT_D("zMin", zMin, value[0] < 0.0 || value[0] >= 1.0, E_01)
    OPT_D("debug_level", debug_level, value[0] < 0.0 || value[0] > 6.0, E_dbg)
    OPT_D("w_pen_init_l", w_pen_init_l, value[0] < 0.0, E_pos)
    OPT_D("w_pen_init_f", w_pen_init_f, value[0] < 0.0, E_pos)
    OPT_D("w_pen_max_l", w_pen_max_l, value[0] < 0.0, E_pos)
    OPT_D("w_pen_max_f", w_pen_max_f, value[0] < 0.0, E_pos)
    OPT_D("w_pen_fact1", w_pen_fact1, value[0] < 1.0, E_gt1)
    OPT_D("w_pen_fact2", w_pen_fact2, value[0] < 1.0, E_gt1)
    return E_none;
}

void makeCandidateNominal(tOptSet *o, int idx) { /* iLQG.c:381-386 */
    traj_t *was_nominal = o->nominal;
    o->nominal = o->candidates[idx];
    o->candidates[idx] = was_nominal;
}

void printParams(double **p, int k) { /* iLQG.c:45-55 */
    for(int i = 0; i < n_params; i++) {
        if(paramdesc[i]->size == -1)
            PRNT("%s[k]= %g\n", paramdesc[i]->name, p[i][k]);
        else if(paramdesc[i]->size == 1)
            PRNT("%s= %g\n", paramdesc[i]->name, p[i][0]);
        else
            printVec(p[i], paramdesc[i]->size, paramdesc[i]->name);
    }
}

static void lambda_up(tOptSet *o, double *dl) { /* iLQG.c:271-272, 342-343 */
    *dl = max(*dl * o->lambdaFactor, o->lambdaFactor);
    o->lambda = max(o->lambda * *dl, o->lambdaMin);
}

static void lambda_down(tOptSet *o, double *dl) { /* iLQG.c:298-299, 317-318: snaps to 0 below lambdaMin */
    *dl = min(*dl / o->lambdaFactor, 1.0 / o->lambdaFactor);
    o->lambda = o->lambda * *dl * (o->lambda > o->lambdaMin);
}

/* iLQG.c:224-379.  One deliberate difference: the reference reads
 * `backPassDone` uninitialised when calc_derivs fails in the very first
 * iteration; here it starts at 0, so that case returns 0. */
int iLQG(tOptSet *o) {
    int iter, bp_done = 0, accepted, need_derivs = 1;
    double dlambda = o->dlambdaInit;

    o->lambda = o->lambdaInit;
    o->w_pen_l = o->w_pen_init_l;
    o->w_pen_f = o->w_pen_init_f;
    update_multipliers(o, 1);

    for(iter = 0; iter < o->max_iter; iter++) {
        if(need_derivs) {
            if(!calc_derivs(o)) break;
            need_derivs = 0;
        }

        bp_done = 0;
        while(!bp_done) {
            if(back_pass(o)) {
                lambda_up(o, &dlambda);
                if(o->lambda > o->lambdaMax) break;
            } else {
                bp_done = 1;
            }
        }

        if(o->g_norm < o->tolGrad && o->lambda < 1e-5) {
            lambda_down(o, &dlambda);
            break;
        }
        if(!bp_done) break;

        accepted = line_search(o, iter);
        if(accepted) {
            lambda_down(o, &dlambda);
            makeCandidateNominal(o, 0);
            o->cost = o->new_cost;
            need_derivs = 1;
            if(o->dcost < o->tolFun) break;
            update_multipliers(o, 0);
            forward_pass(o->nominal, o, 0.0, &o->cost, 1);
        } else {
            lambda_up(o, &dlambda);
            if(o->w_pen_fact2 > 1.0) {
                o->w_pen_l = min(o->w_pen_max_l, o->w_pen_l * o->w_pen_fact2);
                o->w_pen_f = min(o->w_pen_max_f, o->w_pen_f * o->w_pen_fact2);
                forward_pass(o->nominal, o, 0.0, &o->cost, 1);
            }
            if(o->lambda > o->lambdaMax) break;
        }
    }

    o->iterations = iter;
    if(!bp_done) return 0;
    if(iter >= o->max_iter) return 0;
    return 1;
}

