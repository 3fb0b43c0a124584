/* Problem functions for 'CarParking' emitted by tools/gen_problem.py. Do not edit.
 * Function set, signatures and evaluation order: reference iLQG_func.tem:40-521. */
#include "iLQG.h"
#include "matMult.h"

#define mcond(cond, a, dummy, b) ((cond)? a: b)
#define sec(x) (1.0/cos(x))
#define csc(x) (1.0/sin(x))

int n_params= 9;

tParamDesc p_name1= {"cf", 4, 0};
tParamDesc p_name2= {"cu", 2, 0};
tParamDesc p_name3= {"cx", 2, 0};
tParamDesc p_name4= {"d", 1, 0};
tParamDesc p_name5= {"h", 1, 0};
tParamDesc p_name6= {"limA", 2, 0};
tParamDesc p_name7= {"limW", 2, 0};
tParamDesc p_name8= {"pf", 4, 0};
tParamDesc p_name9= {"px", 2, 0};
int n_vars= 0;

tParamDesc *paramdesc[]= {&p_name1, &p_name2, &p_name3, &p_name4, &p_name5, &p_name6, &p_name7, &p_name8, &p_name9};

#define aux_s t->s
#define daux_ds_x3 t->ds_x3
#define daux_ds_u0 t->ds_u0
#define daux_ds_x3x3 t->ds_x3x3
#define daux_ds_u0u0 t->ds_u0u0
#define daux_ds_u0x3 t->ds_u0x3

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

    t->c= p[1][0]*(u[0]*u[0]) + p[1][1]*(u[1]*u[1]) + p[2][0]*(-p[8][0] + sqrt((p[8][0]*p[8][0]) + (x[0]*x[0]))) + p[2][1]*(-p[8][1] + sqrt((p[8][1]*p[8][1]) + (x[1]*x[1])));
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpF(trajFin_t *t, tOptSet *o) {
    const double *const x= t->x;
    const int k= o->n_hor;
    double **const p= o->p;

    t->c= p[0][0]*(-p[7][0] + sqrt((p[7][0]*p[7][0]) + (x[0]*x[0]))) + p[0][1]*(-p[7][1] + sqrt((p[7][1]*p[7][1]) + (x[1]*x[1]))) + p[0][2]*(-p[7][2] + sqrt((p[7][2]*p[7][2]) + (x[2]*x[2]))) + p[0][3]*(-p[7][3] + sqrt((p[7][3]*p[7][3]) + (x[3]*x[3]))) + p[2][0]*(-p[8][0] + sqrt((p[8][0]*p[8][0]) + (x[0]*x[0]))) + p[2][1]*(-p[8][1] + sqrt((p[8][1]*p[8][1]) + (x[1]*x[1])));
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpf(double x_next[], trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    const double *const u= t->u;

    x_next[0]= aux_s*cos(x[2]) + x[0];
    if(isNANorINF(x_next[0])) { PRNT("    @k %d: x_next[0] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[0]); return 0; }
    x_next[1]= aux_s*sin(x[2]) + x[1];
    if(isNANorINF(x_next[1])) { PRNT("    @k %d: x_next[1] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[1]); return 0; }
    x_next[2]= x[2] + asin(p[4][0]*x[3]*sin(u[0])/p[3][0]);
    if(isNANorINF(x_next[2])) { PRNT("    @k %d: x_next[2] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[2]); return 0; }
    x_next[3]= p[4][0]*u[1] + x[3];
    if(isNANorINF(x_next[3])) { PRNT("    @k %d: x_next[3] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[3]); return 0; }
    return 1;
}

void clampU(double *u, trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    double bound;

    /* h[1]= limW[0] - w */
    bound= p[6][0];
    if(u[0]<bound) u[0]= bound;
    /* h[2]= -limW[1] + w */
    bound= p[6][1];
    if(u[0]>bound) u[0]= bound;
    /* h[3]= -a + limA[0] */
    bound= p[5][0];
    if(u[1]<bound) u[1]= bound;
    /* h[4]= a - limA[1] */
    bound= p[5][1];
    if(u[1]>bound) u[1]= bound;
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

    /* h[1]= limW[0] - w */
    bound= p[6][0];
    if(t->lower[0]<bound) { t->lower[0]= bound; active[0][0]= 0; }
    /* h[2]= -limW[1] + w */
    bound= p[6][1];
    if(t->upper[0]>bound) { t->upper[0]= bound; active[1][0]= 1; }
    /* h[3]= -a + limA[0] */
    bound= p[5][0];
    if(t->lower[1]<bound) { t->lower[1]= bound; active[0][1]= 2; }
    /* h[4]= a - limA[1] */
    bound= p[5][1];
    if(t->upper[1]>bound) { t->upper[1]= bound; active[1][1]= 3; }

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
                case 0:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    sign[iu]= -1.0;
                    break;
                case 1:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    sign[iu]= 1.0;
                    break;
                case 2:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    sign[iu]= -1.0;
                    break;
                case 3:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    sign[iu]= 1.0;
                    break;
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

    aux_s= p[3][0] + p[4][0]*x[3]*cos(u[0]) - sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])));
    if(isNANorINF(aux_s)) { PRNT("    @k %d: aux_s in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s); return 0; }
    return 1;
}

static int calcFVariableAux(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {
    const double *const x= t->x;
    double **const p= o->p;
    const double w_pen= o->w_pen_f;
    const int k= o->n_hor;

    return 1;
}

static int calcLAuxDeriv(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    const double w_pen= o->w_pen_l;
    double **const p= o->p;

    daux_ds_x3= p[4][0]*(p[4][0]*x[3]*(sin(u[0])*sin(u[0]))/sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))) + cos(u[0]));
    if(isNANorINF(daux_ds_x3)) { PRNT("    @k %d: daux_ds_x3 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_ds_x3); return 0; }
    daux_ds_u0= p[4][0]*x[3]*(sqrt(2.0)*p[4][0]*x[3]*sin(u[0])*cos(u[0])/sqrt(2.0*(p[3][0]*p[3][0]) + (p[4][0]*p[4][0])*(x[3]*x[3])*(2.0*(cos(u[0])*cos(u[0])) - 1.0) - (p[4][0]*p[4][0])*(x[3]*x[3])) - sin(u[0]));
    if(isNANorINF(daux_ds_u0)) { PRNT("    @k %d: daux_ds_u0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_ds_u0); return 0; }
#if FULL_DDP
    daux_ds_x3x3= (p[3][0]*p[3][0])*(p[4][0]*p[4][0])*(sin(u[0])*sin(u[0]))/(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))));
    if(isNANorINF(daux_ds_x3x3)) { PRNT("    @k %d: daux_ds_x3x3 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_ds_x3x3); return 0; }
    daux_ds_u0u0= (1.0/4.0)*p[4][0]*x[3]*((1.0/4.0)*sqrt(2.0)*(p[4][0]*p[4][0]*p[4][0])*(x[3]*x[3]*x[3])*(-8.0*(cos(u[0])*cos(u[0])*cos(u[0])*cos(u[0])) + 8.0*(cos(u[0])*cos(u[0])))*sqrt(2.0*(p[3][0]*p[3][0]) + (p[4][0]*p[4][0])*(x[3]*x[3])*(2.0*(cos(u[0])*cos(u[0])) - 1.0) - (p[4][0]*p[4][0])*(x[3]*x[3])) + 4.0*p[4][0]*x[3]*(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))))*(2.0*(cos(u[0])*cos(u[0])) - 1.0) - 4.0*(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))))*cos(u[0]))/(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))));
    if(isNANorINF(daux_ds_u0u0)) { PRNT("    @k %d: daux_ds_u0u0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_ds_u0u0); return 0; }
    daux_ds_u0x3= (p[4][0]*p[4][0]*p[4][0]*p[4][0])*(x[3]*x[3]*x[3])*(sin(u[0])*sin(u[0])*sin(u[0]))*cos(u[0])/(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))) + 2.0*(p[4][0]*p[4][0])*x[3]*sin(u[0])*cos(u[0])/sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))) - p[4][0]*sin(u[0]);
    if(isNANorINF(daux_ds_u0x3)) { PRNT("    @k %d: daux_ds_u0x3 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_ds_u0x3); return 0; }
#endif
    return 1;
}

static int bp_derivsL(trajEl_t *t, int k, double **p) {
    const double *const x= t->x;
    const double *const u= t->u;

    /* dynamics */
    t->fx[8]= -aux_s*sin(x[2]);
    if(isNANorINF(t->fx[8])) { PRNT("    @k %d: t->fx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[8]); return 0; }
    t->fx[9]= aux_s*cos(x[2]);
    if(isNANorINF(t->fx[9])) { PRNT("    @k %d: t->fx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[9]); return 0; }
    t->fx[12]= daux_ds_x3*cos(x[2]);
    if(isNANorINF(t->fx[12])) { PRNT("    @k %d: t->fx[12] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[12]); return 0; }
    t->fx[13]= daux_ds_x3*sin(x[2]);
    if(isNANorINF(t->fx[13])) { PRNT("    @k %d: t->fx[13] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[13]); return 0; }
    t->fx[14]= p[4][0]*sin(u[0])/(p[3][0]*sqrt(1.0 - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))/(p[3][0]*p[3][0])));
    if(isNANorINF(t->fx[14])) { PRNT("    @k %d: t->fx[14] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[14]); return 0; }

    t->fu[0]= daux_ds_u0*cos(x[2]);
    if(isNANorINF(t->fu[0])) { PRNT("    @k %d: t->fu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[0]); return 0; }
    t->fu[1]= daux_ds_u0*sin(x[2]);
    if(isNANorINF(t->fu[1])) { PRNT("    @k %d: t->fu[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[1]); return 0; }
    t->fu[2]= p[4][0]*x[3]*cos(u[0])/(p[3][0]*sqrt(1.0 - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))/(p[3][0]*p[3][0])));
    if(isNANorINF(t->fu[2])) { PRNT("    @k %d: t->fu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[2]); return 0; }

#if FULL_DDP
    t->fxx[5]= -aux_s*cos(x[2]);
    if(isNANorINF(t->fxx[5])) { PRNT("    @k %d: t->fxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[5]); return 0; }
    t->fxx[8]= -daux_ds_x3*sin(x[2]);
    if(isNANorINF(t->fxx[8])) { PRNT("    @k %d: t->fxx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[8]); return 0; }
    t->fxx[9]= daux_ds_x3x3*cos(x[2]);
    if(isNANorINF(t->fxx[9])) { PRNT("    @k %d: t->fxx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[9]); return 0; }
    t->fxx[15]= -aux_s*sin(x[2]);
    if(isNANorINF(t->fxx[15])) { PRNT("    @k %d: t->fxx[15] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[15]); return 0; }
    t->fxx[18]= daux_ds_x3*cos(x[2]);
    if(isNANorINF(t->fxx[18])) { PRNT("    @k %d: t->fxx[18] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[18]); return 0; }
    t->fxx[19]= daux_ds_x3x3*sin(x[2]);
    if(isNANorINF(t->fxx[19])) { PRNT("    @k %d: t->fxx[19] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[19]); return 0; }
    t->fxx[29]= (p[4][0]*p[4][0]*p[4][0])*x[3]*(sin(u[0])*sin(u[0])*sin(u[0]))/((p[3][0]*p[3][0]*p[3][0])*((1.0 - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))/(p[3][0]*p[3][0]))*sqrt(1.0 - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))/(p[3][0]*p[3][0]))));
    if(isNANorINF(t->fxx[29])) { PRNT("    @k %d: t->fxx[29] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[29]); return 0; }

    t->fuu[0]= daux_ds_u0u0*cos(x[2]);
    if(isNANorINF(t->fuu[0])) { PRNT("    @k %d: t->fuu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[0]); return 0; }
    t->fuu[3]= daux_ds_u0u0*sin(x[2]);
    if(isNANorINF(t->fuu[3])) { PRNT("    @k %d: t->fuu[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[3]); return 0; }
    t->fuu[6]= p[3][0]*p[4][0]*x[3]*sqrt(1.0 - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))/(p[3][0]*p[3][0]))*(-(p[3][0]*p[3][0]) + (p[4][0]*p[4][0])*(x[3]*x[3]))*sin(u[0])/(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))));
    if(isNANorINF(t->fuu[6])) { PRNT("    @k %d: t->fuu[6] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[6]); return 0; }

    t->fxu[2]= -daux_ds_u0*sin(x[2]);
    if(isNANorINF(t->fxu[2])) { PRNT("    @k %d: t->fxu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[2]); return 0; }
    t->fxu[3]= daux_ds_u0x3*cos(x[2]);
    if(isNANorINF(t->fxu[3])) { PRNT("    @k %d: t->fxu[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[3]); return 0; }
    t->fxu[10]= daux_ds_u0*cos(x[2]);
    if(isNANorINF(t->fxu[10])) { PRNT("    @k %d: t->fxu[10] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[10]); return 0; }
    t->fxu[11]= daux_ds_u0x3*sin(x[2]);
    if(isNANorINF(t->fxu[11])) { PRNT("    @k %d: t->fxu[11] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[11]); return 0; }
    t->fxu[19]= (p[3][0]*p[3][0]*p[3][0])*p[4][0]*sqrt(1.0 - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))/(p[3][0]*p[3][0]))*cos(u[0])/(((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0])))*((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(sin(u[0])*sin(u[0]))));
    if(isNANorINF(t->fxu[19])) { PRNT("    @k %d: t->fxu[19] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[19]); return 0; }

#endif
    /* cost */
    t->cx[0]= p[2][0]*x[0]/sqrt((p[8][0]*p[8][0]) + (x[0]*x[0]));
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }
    t->cx[1]= p[2][1]*x[1]/sqrt((p[8][1]*p[8][1]) + (x[1]*x[1]));
    if(isNANorINF(t->cx[1])) { PRNT("    @k %d: t->cx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[1]); return 0; }

    t->cxx[0]= p[2][0]*(p[8][0]*p[8][0])/(((p[8][0]*p[8][0]) + (x[0]*x[0]))*sqrt((p[8][0]*p[8][0]) + (x[0]*x[0])));
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }
    t->cxx[2]= p[2][1]*(p[8][1]*p[8][1])/(((p[8][1]*p[8][1]) + (x[1]*x[1]))*sqrt((p[8][1]*p[8][1]) + (x[1]*x[1])));
    if(isNANorINF(t->cxx[2])) { PRNT("    @k %d: t->cxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[2]); return 0; }

    t->cu[0]= 2.0*p[1][0]*u[0];
    if(isNANorINF(t->cu[0])) { PRNT("    @k %d: t->cu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cu[0]); return 0; }
    t->cu[1]= 2.0*p[1][1]*u[1];
    if(isNANorINF(t->cu[1])) { PRNT("    @k %d: t->cu[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cu[1]); return 0; }



    return 1;
}

static int calcFAuxDeriv(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {
    const double *const x= t->x;
    const double w_pen= o->w_pen_f;
    double **const p= o->p;
    const int k= o->n_hor;

    return 1;
}

static int bp_derivsF(trajFin_t *t, int k, double **p) {
    const double *const x= t->x;

    t->cx[0]= p[0][0]*x[0]/sqrt((p[7][0]*p[7][0]) + (x[0]*x[0])) + p[2][0]*x[0]/sqrt((p[8][0]*p[8][0]) + (x[0]*x[0]));
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }
    t->cx[1]= p[0][1]*x[1]/sqrt((p[7][1]*p[7][1]) + (x[1]*x[1])) + p[2][1]*x[1]/sqrt((p[8][1]*p[8][1]) + (x[1]*x[1]));
    if(isNANorINF(t->cx[1])) { PRNT("    @k %d: t->cx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[1]); return 0; }
    t->cx[2]= p[0][2]*x[2]/sqrt((p[7][2]*p[7][2]) + (x[2]*x[2]));
    if(isNANorINF(t->cx[2])) { PRNT("    @k %d: t->cx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[2]); return 0; }
    t->cx[3]= p[0][3]*x[3]/sqrt((p[7][3]*p[7][3]) + (x[3]*x[3]));
    if(isNANorINF(t->cx[3])) { PRNT("    @k %d: t->cx[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[3]); return 0; }

    t->cxx[0]= -p[0][0]*(x[0]*x[0])/(((p[7][0]*p[7][0]) + (x[0]*x[0]))*sqrt((p[7][0]*p[7][0]) + (x[0]*x[0]))) + p[0][0]/sqrt((p[7][0]*p[7][0]) + (x[0]*x[0])) - p[2][0]*(x[0]*x[0])/(((p[8][0]*p[8][0]) + (x[0]*x[0]))*sqrt((p[8][0]*p[8][0]) + (x[0]*x[0]))) + p[2][0]/sqrt((p[8][0]*p[8][0]) + (x[0]*x[0]));
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }
    t->cxx[2]= -p[0][1]*(x[1]*x[1])/(((p[7][1]*p[7][1]) + (x[1]*x[1]))*sqrt((p[7][1]*p[7][1]) + (x[1]*x[1]))) + p[0][1]/sqrt((p[7][1]*p[7][1]) + (x[1]*x[1])) - p[2][1]*(x[1]*x[1])/(((p[8][1]*p[8][1]) + (x[1]*x[1]))*sqrt((p[8][1]*p[8][1]) + (x[1]*x[1]))) + p[2][1]/sqrt((p[8][1]*p[8][1]) + (x[1]*x[1]));
    if(isNANorINF(t->cxx[2])) { PRNT("    @k %d: t->cxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[2]); return 0; }
    t->cxx[5]= p[0][2]*(p[7][2]*p[7][2])/(((p[7][2]*p[7][2]) + (x[2]*x[2]))*sqrt((p[7][2]*p[7][2]) + (x[2]*x[2])));
    if(isNANorINF(t->cxx[5])) { PRNT("    @k %d: t->cxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[5]); return 0; }
    t->cxx[9]= p[0][3]*(p[7][3]*p[7][3])/(((p[7][3]*p[7][3]) + (x[3]*x[3]))*sqrt((p[7][3]*p[7][3]) + (x[3]*x[3])));
    if(isNANorINF(t->cxx[9])) { PRNT("    @k %d: t->cxx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[9]); return 0; }
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
        t->cx[2]= 0.0;
        t->cx[3]= 0.0;

        t->cxx[1]= 0.0;
        t->cxx[3]= 0.0;
        t->cxx[4]= 0.0;
        t->cxx[5]= 0.0;
        t->cxx[6]= 0.0;
        t->cxx[7]= 0.0;
        t->cxx[8]= 0.0;
        t->cxx[9]= 0.0;


        t->cuu[0]= 2.0*p[1][0];
        if(isNANorINF(t->cuu[0])) { PRNT("    @k %d: t->cuu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[0]); return 0; }
        t->cuu[1]= 0.0;
        t->cuu[2]= 2.0*p[1][1];
        if(isNANorINF(t->cuu[2])) { PRNT("    @k %d: t->cuu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[2]); return 0; }

        t->cxu[0]= 0.0;
        t->cxu[1]= 0.0;
        t->cxu[2]= 0.0;
        t->cxu[3]= 0.0;
        t->cxu[4]= 0.0;
        t->cxu[5]= 0.0;
        t->cxu[6]= 0.0;
        t->cxu[7]= 0.0;

        /* dynamics */
        t->fx[0]= 1.0;
        t->fx[1]= 0.0;
        t->fx[2]= 0.0;
        t->fx[3]= 0.0;
        t->fx[4]= 0.0;
        t->fx[5]= 1.0;
        t->fx[6]= 0.0;
        t->fx[7]= 0.0;
        t->fx[10]= 1.0;
        t->fx[11]= 0.0;
        t->fx[15]= 1.0;

        t->fu[3]= 0.0;
        t->fu[4]= 0.0;
        t->fu[5]= 0.0;
        t->fu[6]= 0.0;
        t->fu[7]= p[4][0];
        if(isNANorINF(t->fu[7])) { PRNT("    @k %d: t->fu[7] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[7]); return 0; }

#if FULL_DDP
        t->fxx[0]= 0.0;
        t->fxx[1]= 0.0;
        t->fxx[2]= 0.0;
        t->fxx[3]= 0.0;
        t->fxx[4]= 0.0;
        t->fxx[6]= 0.0;
        t->fxx[7]= 0.0;
        t->fxx[10]= 0.0;
        t->fxx[11]= 0.0;
        t->fxx[12]= 0.0;
        t->fxx[13]= 0.0;
        t->fxx[14]= 0.0;
        t->fxx[16]= 0.0;
        t->fxx[17]= 0.0;
        t->fxx[20]= 0.0;
        t->fxx[21]= 0.0;
        t->fxx[22]= 0.0;
        t->fxx[23]= 0.0;
        t->fxx[24]= 0.0;
        t->fxx[25]= 0.0;
        t->fxx[26]= 0.0;
        t->fxx[27]= 0.0;
        t->fxx[28]= 0.0;
        t->fxx[30]= 0.0;
        t->fxx[31]= 0.0;
        t->fxx[32]= 0.0;
        t->fxx[33]= 0.0;
        t->fxx[34]= 0.0;
        t->fxx[35]= 0.0;
        t->fxx[36]= 0.0;
        t->fxx[37]= 0.0;
        t->fxx[38]= 0.0;
        t->fxx[39]= 0.0;

        t->fuu[1]= 0.0;
        t->fuu[2]= 0.0;
        t->fuu[4]= 0.0;
        t->fuu[5]= 0.0;
        t->fuu[7]= 0.0;
        t->fuu[8]= 0.0;
        t->fuu[9]= 0.0;
        t->fuu[10]= 0.0;
        t->fuu[11]= 0.0;

        t->fxu[0]= 0.0;
        t->fxu[1]= 0.0;
        t->fxu[4]= 0.0;
        t->fxu[5]= 0.0;
        t->fxu[6]= 0.0;
        t->fxu[7]= 0.0;
        t->fxu[8]= 0.0;
        t->fxu[9]= 0.0;
        t->fxu[12]= 0.0;
        t->fxu[13]= 0.0;
        t->fxu[14]= 0.0;
        t->fxu[15]= 0.0;
        t->fxu[16]= 0.0;
        t->fxu[17]= 0.0;
        t->fxu[18]= 0.0;
        t->fxu[20]= 0.0;
        t->fxu[21]= 0.0;
        t->fxu[22]= 0.0;
        t->fxu[23]= 0.0;
        t->fxu[24]= 0.0;
        t->fxu[25]= 0.0;
        t->fxu[26]= 0.0;
        t->fxu[27]= 0.0;
        t->fxu[28]= 0.0;
        t->fxu[29]= 0.0;
        t->fxu[30]= 0.0;
        t->fxu[31]= 0.0;

#endif
    }
    return 1;
}

static int init_final(trajFin_t *t, tOptSet *o) {
    double **const p= o->p;
    const int k= o->n_hor;


    t->cxx[1]= 0.0;
    t->cxx[3]= 0.0;
    t->cxx[4]= 0.0;
    t->cxx[6]= 0.0;
    t->cxx[7]= 0.0;
    t->cxx[8]= 0.0;
    return 1;
}

int init_trajectory(traj_t *t, tOptSet *o) {
    return init_running(t->t, o) && init_final(&t->f, o);
}

static int init_multipliers_running(tOptSet *o) {
    return 1;
}

static int init_multipliers_final(tOptSet *o) {
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

static int update_multipliers_running(tOptSet *o, int init) {
    return 1;
}

static int update_multipliers_final(tOptSet *o, int init) {
    return 1;
}

/* iLQG.c:236,337: multipliers of the running constraints, then of the final ones */
int update_multipliers(tOptSet *o, int init) {
    return update_multipliers_running(o, init) && update_multipliers_final(o, init);
}

/* no outputs g are defined by this generator (iLQG_func.tem:511-521) */
int get_g_size() { return 0; }

int calcG(double g[], trajEl_t *t, int k, double **p) { return 1; }

/* ---- additive: one step of forward_pass in ILQG_ROLLOUT_PARTS independent parts (batched back-ends that put
 * several wavefronts on a trajectory's step; the reference's solver never calls this).  Part r: component r of the
 * dynamics and the summands r, r + N_X, ... of the running cost, term[] indexed by their place in ddpL's sum:
 * t->c == ((term[0] + term[1]) + term[2]) + ...  A NaN or Inf in a guarded value sets bad[0]. */
#define ILQG_ROLLOUT_PARTS 4
#define ILQG_ROLLOUT_TERMS 4
#ifndef ILQG_PART_SIN  /* a back-end may define these two before including this file */
#define ILQG_PART_SIN(v) sin(v)
#define ILQG_PART_COS(v) cos(v)
#endif
#ifndef ILQG_PART_FN  /* ... and the function's storage class / attributes */
#define ILQG_PART_FN static
#endif
typedef struct {
    double s;
} ilqg_step_aux_t;
ILQG_PART_FN void ilqg_step_part(int part, double x_next[], double term[], int bad[], const double *x, const double *u, int k, double **p, int N) {
    ilqg_step_aux_t aux_, *const t= &aux_;

    switch(part) {
    case 0:
        aux_s= p[3][0] + p[4][0]*x[3]*ILQG_PART_COS(u[0]) - sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(ILQG_PART_SIN(u[0])*ILQG_PART_SIN(u[0])));
        if(!(fabs(aux_s) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[0]= aux_s*ILQG_PART_COS(x[2]) + x[0];
        if(!(fabs(x_next[0]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[0]= p[1][0]*(u[0]*u[0]);
        break;
    case 1:
        aux_s= p[3][0] + p[4][0]*x[3]*ILQG_PART_COS(u[0]) - sqrt((p[3][0]*p[3][0]) - (p[4][0]*p[4][0])*(x[3]*x[3])*(ILQG_PART_SIN(u[0])*ILQG_PART_SIN(u[0])));
        if(!(fabs(aux_s) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[1]= aux_s*ILQG_PART_SIN(x[2]) + x[1];
        if(!(fabs(x_next[1]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[1]= p[1][1]*(u[1]*u[1]);
        break;
    case 2:
        x_next[2]= x[2] + asin(p[4][0]*x[3]*ILQG_PART_SIN(u[0])/p[3][0]);
        if(!(fabs(x_next[2]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[2]= p[2][0]*(-p[8][0] + sqrt((p[8][0]*p[8][0]) + (x[0]*x[0])));
        break;
    case 3:
        x_next[3]= p[4][0]*u[1] + x[3];
        if(!(fabs(x_next[3]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[3]= p[2][1]*(-p[8][1] + sqrt((p[8][1]*p[8][1]) + (x[1]*x[1])));
        break;
    default: break;
    }
}
