/* Problem functions for 'Brachi' emitted by tools/gen_problem.py. Do not edit.
 * Function set, signatures and evaluation order: reference iLQG_func.tem:40-521. */
#include "iLQG.h"
#include "matMult.h"

#define mcond(cond, a, dummy, b) ((cond)? a: b)
#define sec(x) (1.0/cos(x))
#define csc(x) (1.0/sin(x))

int n_params= 3;

tParamDesc p_name1= {"dx", 1, 0};
tParamDesc p_name2= {"g", 1, 0};
tParamDesc p_name3= {"yf", 1, 0};
int n_vars= 0;

tParamDesc *paramdesc[]= {&p_name1, &p_name2, &p_name3};

#define aux_hfe_1 t->hfe_1
#define aux_pfe_1 t->pfe_1
#define daux_dpfe_1_x0 t->dpfe_1_x0
#define daux_dpfe_1_x0x0 t->dpfe_1_x0x0

static int calcXVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o);
static int calcXUVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o);
static int calcFVariableAux(trajFin_t *t, multipliersFin_t *m, tOptSet *o);
static int calcLAuxDeriv(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o);
static int calcFAuxDeriv(trajFin_t *t, multipliersFin_t *m, tOptSet *o);
static int bp_derivsL(trajEl_t *t, int k, double **p);
static int bp_derivsF(trajFin_t *t, int k, double **p);

/* running cost of one step, final cost, one step of the dynamics */
static int ddpL(trajEl_t *t, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    double **const p= o->p;

    t->c= -sqrt(2.0)*sqrt((u[0]*u[0]) + 1.0)*(-sqrt(-x[0]) + sqrt(-p[0][0]*u[0] - x[0]))*sqrt(1.0/p[1][0])/u[0];
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpF(trajFin_t *t, tOptSet *o) {
    const double *const x= t->x;
    const int k= o->n_hor;
    double **const p= o->p;

    t->c= aux_pfe_1;
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpf(double x_next[], trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    const double *const u= t->u;

    x_next[0]= p[0][0]*u[0] + x[0];
    if(isNANorINF(x_next[0])) { PRNT("    @k %d: x_next[0] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[0]); return 0; }
    return 1;
}

void clampU(double *u, trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    double bound;

}

static void limitsU(trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    int active[2][N_U];  /* constraint that bounds input iu from below [0] / from above [1]; -1: none */
    double bound;
    int iu, side;

    for(iu= 0; iu<N_U; iu++) {
        active[0][iu]= active[1][iu]= -1;
        t->lower[iu]= -INF;
        t->upper[iu]= INF;
    }


    /* the solver works with the change of u */
    for(iu= 0; iu<N_U; iu++) {
        t->lower[iu]-= t->u[iu];
        t->upper[iu]-= t->u[iu];
    }

    /* additive: a back-end that will not read *_sign / *_hx of this element (limits that do not depend on the
     * state: constants) may say so through a condition of its own */
#ifndef ILQG_LIMIT_GRADIENTS_WANTED
#define ILQG_LIMIT_GRADIENTS_WANTED 1
#endif
    if(ILQG_LIMIT_GRADIENTS_WANTED)
    for(side= 0; side<2; side++) {
        double *const sign= side? t->upper_sign: t->lower_sign;
        double *const grad= side? t->upper_hx: t->lower_hx;
        for(iu= 0; iu<N_U; iu++) {
            double *const hx_= grad + iu*N_X;
            switch(active[side][iu]) {
                default:  /* unbounded on this side: the gradient is not used */
                    sign[iu]= 0.0;
            }
        }
    }
}

/* Roll-out of candidate trajectory c (line_search.c:40, iLQG.c:338, iLQG_mex.c:116).
 * alpha != 0: u = u_nom + alpha*l + L (x - x_nom) with the gains of the nominal trajectory, accumulated state by
 * state; alpha == 0: the nominal inputs as they are.  cost_only: x and u of c are kept, only the cost is summed.
 * csum[0] holds the cost summed so far also when a NaN/Inf guard ends the sweep (return 0). */
int forward_pass(traj_t *c, tOptSet *o, double alpha, double *csum, int cost_only) {
    const int n_steps= o->n_hor;
    const int rollout= !cost_only;
    int k, ix, iu;

    csum[0]= 0.0;
    if(rollout)
        for(ix= 0; ix<N_X; ix++) c->t[0].x[ix]= o->x0[ix];

    for(k= 0; k<n_steps; k++) {
        const trajEl_t *const ref= o->nominal->t + k;
        trajEl_t *const cur= c->t + k;
        multipliersEl_t *const mul= o->multipliers.t + k;

        if(rollout) {
            if(alpha) {
                for(iu= 0; iu<N_U; iu++)
                    cur->u[iu]= ref->u[iu] + ref->l[iu]*alpha;
                for(ix= 0; ix<N_X; ix++) {
                    const double dev= cur->x[ix] - ref->x[ix];
                    for(iu= 0; iu<N_U; iu++)
                        cur->u[iu]+= ref->L[MAT_IDX(iu, ix, N_U)]*dev;
                }
            } else {
                for(iu= 0; iu<N_U; iu++)
                    cur->u[iu]= ref->u[iu];
            }
        }
        if(!calcXVariableAux(cur, mul, k, o)) return 0;
        if(rollout) clampU(cur->u, cur, k, o->p, n_steps);
        if(!calcXUVariableAux(cur, mul, k, o)) return 0;
        if(rollout && !ddpf((k+1<n_steps)? c->t[k+1].x: c->f.x, cur, k, o->p, n_steps)) return 0;
        if(!ddpL(cur, k, o)) return 0;
        csum[0]+= cur->c;
    }

    if(!calcFVariableAux(&c->f, &o->multipliers.f, o)) return 0;
    if(!ddpF(&c->f, o)) return 0;
    csum[0]+= c->f.c;
    return 1;
}

/* Derivatives along the nominal trajectory (iLQG.c:247): the final step, then the running steps from the end of
 * the horizon to its start, each with the box its input constraints leave around the nominal input. */
int calc_derivs(tOptSet *o) {
    const int n_steps= o->n_hor;
    traj_t *const nom= o->nominal;
    int k;

    if(!calcFAuxDeriv(&nom->f, &o->multipliers.f, o)) return 0;
    if(!bp_derivsF(&nom->f, n_steps, o->p)) return 0;

    for(k= n_steps; k-->0; ) {
        trajEl_t *const el= nom->t + k;
        if(!calcLAuxDeriv(el, o->multipliers.t + k, k, o)) return 0;
        if(!bp_derivsL(el, k, o->p)) return 0;
        limitsU(el, k, o->p, n_steps);
    }
    return 1;
}

/* auxiliary variables: members of the step's element, evaluated once and reused by everything that follows */
static int calcXVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    double **const p= o->p;
    const double w_pen= o->w_pen_l;

    return 1;
}

static int calcXUVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    double **const p= o->p;
    const double w_pen= o->w_pen_l;

    return 1;
}

static int calcFVariableAux(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {
    const double *const x= t->x;
    double **const p= o->p;
    const double w_pen= o->w_pen_f;
    const int k= o->n_hor;

    aux_hfe_1= -p[2][0] + x[0];
    if(isNANorINF(aux_hfe_1)) { PRNT("    @k %d: aux_hfe_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hfe_1); return 0; }
    aux_pfe_1= 0.5*(aux_hfe_1*aux_hfe_1)*w_pen + aux_hfe_1*m->mu_fe[0];
    if(isNANorINF(aux_pfe_1)) { PRNT("    @k %d: aux_pfe_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_pfe_1); return 0; }
    return 1;
}

static int calcLAuxDeriv(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    const double w_pen= o->w_pen_l;
    double **const p= o->p;

#if FULL_DDP
#endif
    return 1;
}

static int bp_derivsL(trajEl_t *t, int k, double **p) {
    const double *const x= t->x;
    const double *const u= t->u;

    /* dynamics */


#if FULL_DDP
#endif
    /* cost */
    t->cx[0]= -sqrt(2.0)*sqrt((u[0]*u[0]) + 1.0)*(-(1.0/2.0)/sqrt(-p[0][0]*u[0] - x[0]) - 1.0/2.0*sqrt(-x[0])/x[0])*sqrt(1.0/p[1][0])/u[0];
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }

    t->cxx[0]= -sqrt(2.0)*sqrt((u[0]*u[0]) + 1.0)*(-(1.0/4.0)/((-p[0][0]*u[0] - x[0])*sqrt(-p[0][0]*u[0] - x[0])) + (1.0/4.0)*sqrt(-x[0])/(x[0]*x[0]))*sqrt(1.0/p[1][0])/u[0];
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }

    t->cu[0]= (1.0/2.0)*sqrt(2.0)*p[0][0]*sqrt((u[0]*u[0]) + 1.0)*sqrt(1.0/p[1][0])/(u[0]*sqrt(-p[0][0]*u[0] - x[0])) - sqrt(2.0)*(-sqrt(-x[0]) + sqrt(-p[0][0]*u[0] - x[0]))*sqrt(1.0/p[1][0])/sqrt((u[0]*u[0]) + 1.0) + sqrt(2.0)*sqrt((u[0]*u[0]) + 1.0)*(-sqrt(-x[0]) + sqrt(-p[0][0]*u[0] - x[0]))*sqrt(1.0/p[1][0])/(u[0]*u[0]);
    if(isNANorINF(t->cu[0])) { PRNT("    @k %d: t->cu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cu[0]); return 0; }

    t->cuu[0]= (1.0/4.0)*sqrt(2.0)*(p[0][0]*p[0][0])*sqrt((u[0]*u[0]) + 1.0)*sqrt(1.0/p[1][0])/(u[0]*((-p[0][0]*u[0] - x[0])*sqrt(-p[0][0]*u[0] - x[0]))) + sqrt(2.0)*p[0][0]*sqrt(1.0/p[1][0])/(sqrt((u[0]*u[0]) + 1.0)*sqrt(-p[0][0]*u[0] - x[0])) - sqrt(2.0)*p[0][0]*sqrt((u[0]*u[0]) + 1.0)*sqrt(1.0/p[1][0])/((u[0]*u[0])*sqrt(-p[0][0]*u[0] - x[0])) + sqrt(2.0)*u[0]*(-sqrt(-x[0]) + sqrt(-p[0][0]*u[0] - x[0]))*sqrt(1.0/p[1][0])/(((u[0]*u[0]) + 1.0)*sqrt((u[0]*u[0]) + 1.0)) + sqrt(2.0)*(-sqrt(-x[0]) + sqrt(-p[0][0]*u[0] - x[0]))*sqrt(1.0/p[1][0])/(u[0]*sqrt((u[0]*u[0]) + 1.0)) - 2.0*sqrt(2.0)*sqrt((u[0]*u[0]) + 1.0)*(-sqrt(-x[0]) + sqrt(-p[0][0]*u[0] - x[0]))*sqrt(1.0/p[1][0])/(u[0]*u[0]*u[0]);
    if(isNANorINF(t->cuu[0])) { PRNT("    @k %d: t->cuu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[0]); return 0; }

    t->cxu[0]= (1.0/4.0)*sqrt(2.0)*p[0][0]*sqrt((u[0]*u[0]) + 1.0)*sqrt(1.0/p[1][0])/(u[0]*((-p[0][0]*u[0] - x[0])*sqrt(-p[0][0]*u[0] - x[0]))) - sqrt(2.0)*(-(1.0/2.0)/sqrt(-p[0][0]*u[0] - x[0]) - 1.0/2.0*sqrt(-x[0])/x[0])*sqrt(1.0/p[1][0])/sqrt((u[0]*u[0]) + 1.0) + sqrt(2.0)*sqrt((u[0]*u[0]) + 1.0)*(-(1.0/2.0)/sqrt(-p[0][0]*u[0] - x[0]) - 1.0/2.0*sqrt(-x[0])/x[0])*sqrt(1.0/p[1][0])/(u[0]*u[0]);
    if(isNANorINF(t->cxu[0])) { PRNT("    @k %d: t->cxu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxu[0]); return 0; }

    return 1;
}

static int calcFAuxDeriv(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {
    const double *const x= t->x;
    const double w_pen= o->w_pen_f;
    double **const p= o->p;
    const int k= o->n_hor;

    daux_dpfe_1_x0= 1.0*aux_hfe_1*w_pen + m->mu_fe[0];
    if(isNANorINF(daux_dpfe_1_x0)) { PRNT("    @k %d: daux_dpfe_1_x0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_1_x0); return 0; }
    daux_dpfe_1_x0x0= 1.0*w_pen;
    if(isNANorINF(daux_dpfe_1_x0x0)) { PRNT("    @k %d: daux_dpfe_1_x0x0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_1_x0x0); return 0; }
    return 1;
}

static int bp_derivsF(trajFin_t *t, int k, double **p) {
    const double *const x= t->x;

    t->cx[0]= daux_dpfe_1_x0;
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }

    t->cxx[0]= daux_dpfe_1_x0x0;
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }
    return 1;
}

/* constant entries of every element of a trajectory buffer */
static int init_running(trajEl_t *t, tOptSet *o) {
    double **const p= o->p;
    trajEl_t *const end= t + o->n_hor;
    int k= 0;

    for(; t<end; t++, k++) {
#if FULL_DDP
#endif
        /* cost */





        /* dynamics */
        t->fx[0]= 1.0;

        t->fu[0]= p[0][0];
        if(isNANorINF(t->fu[0])) { PRNT("    @k %d: t->fu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[0]); return 0; }

#if FULL_DDP
        { int e_; for(e_= 0; e_<N_X*sizeofQxx; e_++) t->fxx[e_]= 0.0; }

        { int e_; for(e_= 0; e_<N_X*sizeofQuu; e_++) t->fuu[e_]= 0.0; }

        { int e_; for(e_= 0; e_<N_X*sizeofQxu; e_++) t->fxu[e_]= 0.0; }

#endif
    }
    return 1;
}

static int init_final(trajFin_t *t, tOptSet *o) {
    double **const p= o->p;
    const int k= o->n_hor;


    return 1;
}

int init_trajectory(traj_t *t, tOptSet *o) {
    return init_running(t->t, o) && init_final(&t->f, o);
}

static int init_multipliers_running(tOptSet *o) {
    return 1;
}

static int init_multipliers_final(tOptSet *o) {
    multipliersFin_t *const m= &o->multipliers.f;
    int i;

    for(i= 0; i<1; i++) { m->mu_fe[i]= 0.0; m->last_hfe[i]= 0.0; }
    return 1;
}

int init_multipliers(tOptSet *o) {
    return init_multipliers_running(o) && init_multipliers_final(o);
}

/* iLQG_mex.c:108: constants of every trajectory buffer; buffer 0 starts as the nominal trajectory, the
 * others as line-search candidates; multipliers at their start values */
int init_opt(tOptSet *o) {
    int b;

    for(b= 0; b<=NUMBER_OF_THREADS; b++) {
        if(!init_trajectory(&o->trajectories[b], o)) return 0;
        if(b==0) o->nominal= &o->trajectories[0];
        else o->candidates[b-1]= &o->trajectories[b];
    }
    return init_multipliers(o);
}

/* a violation v stalls: above the tolerance and not smaller than 1/w_pen_fact1 of the one remembered */
static int violation_stalls(double v, double last, const tOptSet *o) {
    return v>o->tolConstraint && o->w_pen_fact1*v>last;
}

static int update_multipliers_running(tOptSet *o, int init) {
    return 1;
}

static int update_multipliers_final(tOptSet *o, int init) {
    trajFin_t *const t= &o->nominal->f;
    multipliersFin_t *const m= &o->multipliers.f;
    const double w_pen= o->w_pen_f;
    double **const p= o->p;
    const int k= o->n_hor;
    int stalled= 0;

    stalled|= violation_stalls(fabs(aux_hfe_1), fabs(m->last_hfe[0]), o);
    m->last_hfe[0]= aux_hfe_1;
    if(!init && stalled)
        o->w_pen_f= min(o->w_pen_max_f, o->w_pen_f*o->w_pen_fact1);
    if(init) return 1;
    m->mu_fe[0]= aux_hfe_1*w_pen + m->mu_fe[0];
    if(isNANorINF(m->mu_fe[0])) { PRNT("    @k %d: m->mu_fe[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_fe[0]); return 0; }
    return 1;
}

/* iLQG.c:236,337: multipliers of the running constraints, then of the final ones */
int update_multipliers(tOptSet *o, int init) {
    return update_multipliers_running(o, init) && update_multipliers_final(o, init);
}

/* no outputs g are defined by this generator (iLQG_func.tem:511-521) */
int get_g_size() { return 0; }

int calcG(double g[], trajEl_t *t, int k, double **p) { return 1; }
