/* Problem functions for 'Synth10Hx' emitted by tools/gen_problem.py. Do not edit.
 * Function set, signatures and evaluation order: reference iLQG_func.tem:40-521. */
#include "iLQG.h"
#include "matMult.h"

#define mcond(cond, a, dummy, b) ((cond)? a: b)
#define sec(x) (1.0/cos(x))
#define csc(x) (1.0/sin(x))

int n_params= 7;

tParamDesc p_name1= {"c", 1, 0};
tParamDesc p_name2= {"h", 1, 0};
tParamDesc p_name3= {"lim", 1, 0};
tParamDesc p_name4= {"px", 1, 0};
tParamDesc p_name5= {"qf", 10, 0};
tParamDesc p_name6= {"qx", 10, 0};
tParamDesc p_name7= {"ru", 3, 0};
int n_vars= 0;

tParamDesc *paramdesc[]= {&p_name1, &p_name2, &p_name3, &p_name4, &p_name5, &p_name6, &p_name7};

#define aux_s1_0 t->s1_0
#define aux_s1_1 t->s1_1
#define aux_s1_2 t->s1_2
#define aux_s1_3 t->s1_3
#define aux_s1_4 t->s1_4
#define aux_s1_5 t->s1_5
#define aux_s1_6 t->s1_6
#define aux_s1_7 t->s1_7
#define aux_s1_8 t->s1_8
#define aux_s1_9 t->s1_9
#define aux_s2_0 t->s2_0
#define aux_s2_1 t->s2_1
#define aux_s2_2 t->s2_2
#define aux_s2_3 t->s2_3
#define aux_s2_4 t->s2_4
#define aux_s2_5 t->s2_5
#define aux_s2_6 t->s2_6
#define aux_s2_7 t->s2_7
#define aux_s2_8 t->s2_8
#define aux_s2_9 t->s2_9

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

    t->c= p[5][0]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[0]*x[0]))) + p[5][1]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[1]*x[1]))) + p[5][2]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[2]*x[2]))) + p[5][3]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[3]*x[3]))) + p[5][4]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[4]*x[4]))) + p[5][5]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[5]*x[5]))) + p[5][6]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[6]*x[6]))) + p[5][7]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[7]*x[7]))) + p[5][8]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[8]*x[8]))) + p[5][9]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[9]*x[9]))) + p[6][0]*(u[0]*u[0]) + p[6][1]*(u[1]*u[1]) + p[6][2]*(u[2]*u[2]);
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpF(trajFin_t *t, tOptSet *o) {
    const double *const x= t->x;
    const int k= o->n_hor;
    double **const p= o->p;

    t->c= p[4][0]*(x[0]*x[0]) + p[4][1]*(x[1]*x[1]) + p[4][2]*(x[2]*x[2]) + p[4][3]*(x[3]*x[3]) + p[4][4]*(x[4]*x[4]) + p[4][5]*(x[5]*x[5]) + p[4][6]*(x[6]*x[6]) + p[4][7]*(x[7]*x[7]) + p[4][8]*(x[8]*x[8]) + p[4][9]*(x[9]*x[9]);
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpf(double x_next[], trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    const double *const u= t->u;

    x_next[0]= p[1][0]*(p[0][0]*sin(aux_s1_0)*cos(aux_s2_0) - 0.69999999999999996*u[0] + 0.30599999999999999*u[1] + 0.151*u[2] - 1.417*x[0] + 0.089999999999999997*x[1] - 0.094*x[2] + 0.096000000000000002*x[3] + 0.29999999999999999*x[4] + 0.309*x[5] - 0.041000000000000002*x[6] + 0.17199999999999999*x[7] - 0.34599999999999997*x[8] + 0.025999999999999999*x[9]) + x[0];
    if(isNANorINF(x_next[0])) { PRNT("    @k %d: x_next[0] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[0]); return 0; }
    x_next[1]= p[1][0]*(p[0][0]*sin(aux_s1_1)*cos(aux_s2_1) + 0.51000000000000001*u[0] - 0.027*u[1] + 0.122*u[2] - 0.20000000000000001*x[0] - 0.95899999999999996*x[1] - 0.10000000000000001*x[2] - 0.029000000000000001*x[3] - 0.081000000000000003*x[4] - 0.22600000000000001*x[5] - 0.059999999999999998*x[6] - 0.248*x[7] - 0.095000000000000001*x[8] - 0.097000000000000003*x[9]) + x[1];
    if(isNANorINF(x_next[1])) { PRNT("    @k %d: x_next[1] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[1]); return 0; }
    x_next[2]= p[1][0]*(p[0][0]*sin(aux_s1_2)*cos(aux_s2_2) + 0.036999999999999998*u[0] + 0.91600000000000004*u[1] - 0.014*u[2] + 0.16600000000000001*x[0] - 0.25800000000000001*x[1] - 1.0529999999999999*x[2] + 0.070000000000000007*x[3] + 0.45100000000000001*x[4] - 0.13400000000000001*x[5] - 0.072999999999999995*x[6] - 0.36299999999999999*x[7] - 0.28100000000000003*x[8] + 0.088999999999999996*x[9]) + x[2];
    if(isNANorINF(x_next[2])) { PRNT("    @k %d: x_next[2] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[2]); return 0; }
    x_next[3]= p[1][0]*(p[0][0]*sin(aux_s1_3)*cos(aux_s2_3) - 0.41599999999999998*u[0] + 0.039*u[1] - 0.64600000000000002*u[2] + 0.47799999999999998*x[0] + 0.099000000000000005*x[1] - 0.153*x[2] - 0.72999999999999998*x[3] - 0.26500000000000001*x[4] + 0.23599999999999999*x[5] - 0.53900000000000003*x[6] + 0.217*x[7] - 0.16700000000000001*x[8] + 0.063*x[9]) + x[3];
    if(isNANorINF(x_next[3])) { PRNT("    @k %d: x_next[3] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[3]); return 0; }
    x_next[4]= p[1][0]*(p[0][0]*sin(aux_s1_4)*cos(aux_s2_4) + 0.64600000000000002*u[0] + 0.14899999999999999*u[1] + 0.69899999999999995*u[2] - 0.23999999999999999*x[0] + 0.129*x[1] + 0.029000000000000001*x[2] - 0.17799999999999999*x[3] - 1.095*x[4] - 0.049000000000000002*x[5] - 0.153*x[6] + 0.28299999999999997*x[7] - 0.188*x[8] + 0.23100000000000001*x[9]) + x[4];
    if(isNANorINF(x_next[4])) { PRNT("    @k %d: x_next[4] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[4]); return 0; }
    x_next[5]= p[1][0]*(p[0][0]*sin(aux_s1_5)*cos(aux_s2_5) + 0.35699999999999998*u[0] - 0.51200000000000001*u[1] - 0.379*u[2] - 0.014999999999999999*x[0] - 0.17299999999999999*x[1] - 0.16900000000000001*x[2] + 0.032000000000000001*x[3] - 0.22800000000000001*x[4] - 1.1479999999999999*x[5] - 0.20999999999999999*x[6] - 0.104*x[7] - 0.252*x[8] - 0.031*x[9]) + x[5];
    if(isNANorINF(x_next[5])) { PRNT("    @k %d: x_next[5] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[5]); return 0; }
    x_next[6]= p[1][0]*(p[0][0]*sin(aux_s1_6)*cos(aux_s2_6) + 0.76100000000000001*u[0] + 0.064000000000000001*u[1] + 0.086999999999999994*u[2] - 0.113*x[0] + 0.083000000000000004*x[1] + 0.016*x[2] + 0.222*x[3] + 0.099000000000000005*x[4] + 0.10299999999999999*x[5] - 1.161*x[6] + 0.33900000000000002*x[7] + 0.065000000000000002*x[8] - 0.25700000000000001*x[9]) + x[6];
    if(isNANorINF(x_next[6])) { PRNT("    @k %d: x_next[6] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[6]); return 0; }
    x_next[7]= p[1][0]*(p[0][0]*sin(aux_s1_7)*cos(aux_s2_7) - 0.48099999999999998*u[0] - 0.159*u[1] - 0.245*u[2] + 0.307*x[0] - 0.002*x[1] + 0.159*x[2] - 0.125*x[3] - 0.17699999999999999*x[4] + 0.070999999999999994*x[5] + 0.11*x[6] - 1.0600000000000001*x[7] + 0.039*x[8] + 0.129*x[9]) + x[7];
    if(isNANorINF(x_next[7])) { PRNT("    @k %d: x_next[7] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[7]); return 0; }
    x_next[8]= p[1][0]*(p[0][0]*sin(aux_s1_8)*cos(aux_s2_8) + 0.215*u[0] + 0.86899999999999999*u[1] + 1.629*u[2] - 0.014*x[0] - 0.033000000000000002*x[1] + 0.17299999999999999*x[2] - 0.042000000000000003*x[3] - 0.184*x[4] - 0.021000000000000001*x[5] - 0.014999999999999999*x[6] + 0.436*x[7] - 1.2010000000000001*x[8] + 0.122*x[9]) + x[8];
    if(isNANorINF(x_next[8])) { PRNT("    @k %d: x_next[8] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[8]); return 0; }
    x_next[9]= p[1][0]*(p[0][0]*sin(aux_s1_9)*cos(aux_s2_9) - 0.105*u[0] - 1.7509999999999999*u[1] + 0.24099999999999999*u[2] + 0.096000000000000002*x[0] + 0.037999999999999999*x[1] + 0.025999999999999999*x[2] - 0.13200000000000001*x[3] + 0.23599999999999999*x[4] - 0.032000000000000001*x[5] + 0.222*x[6] - 0.151*x[7] - 0.14699999999999999*x[8] - 1.343*x[9]) + x[9];
    if(isNANorINF(x_next[9])) { PRNT("    @k %d: x_next[9] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[9]); return 0; }
    return 1;
}

void clampU(double *u, trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    double bound;

    /* h[1]= -lim + u0 - x1/2 */
    bound= p[2][0] + (1.0/2.0)*x[1];
    if(u[0]>bound) u[0]= bound;
    /* h[2]= -lim - u0 */
    bound= -p[2][0];
    if(u[0]<bound) u[0]= bound;
    /* h[3]= -lim + u1 */
    bound= p[2][0];
    if(u[1]>bound) u[1]= bound;
    /* h[4]= -lim - u1 */
    bound= -p[2][0];
    if(u[1]<bound) u[1]= bound;
    /* h[5]= -lim + u2 */
    bound= p[2][0];
    if(u[2]>bound) u[2]= bound;
    /* h[6]= -lim - u2 - x0**2/5 */
    bound= -p[2][0] - 1.0/5.0*(x[0]*x[0]);
    if(u[2]<bound) u[2]= bound;
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

    /* h[1]= -lim + u0 - x1/2 */
    bound= p[2][0] + (1.0/2.0)*x[1];
    if(t->upper[0]>bound) { t->upper[0]= bound; active[1][0]= 0; }
    /* h[2]= -lim - u0 */
    bound= -p[2][0];
    if(t->lower[0]<bound) { t->lower[0]= bound; active[0][0]= 1; }
    /* h[3]= -lim + u1 */
    bound= p[2][0];
    if(t->upper[1]>bound) { t->upper[1]= bound; active[1][1]= 2; }
    /* h[4]= -lim - u1 */
    bound= -p[2][0];
    if(t->lower[1]<bound) { t->lower[1]= bound; active[0][1]= 3; }
    /* h[5]= -lim + u2 */
    bound= p[2][0];
    if(t->upper[2]>bound) { t->upper[2]= bound; active[1][2]= 4; }
    /* h[6]= -lim - u2 - x0**2/5 */
    bound= -p[2][0] - 1.0/5.0*(x[0]*x[0]);
    if(t->lower[2]<bound) { t->lower[2]= bound; active[0][2]= 5; }

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
                    hx_[1]= -1.0/2.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    hx_[4]= 0.0;
                    hx_[5]= 0.0;
                    hx_[6]= 0.0;
                    hx_[7]= 0.0;
                    hx_[8]= 0.0;
                    hx_[9]= 0.0;
                    sign[iu]= 1.0;
                    break;
                case 1:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    hx_[4]= 0.0;
                    hx_[5]= 0.0;
                    hx_[6]= 0.0;
                    hx_[7]= 0.0;
                    hx_[8]= 0.0;
                    hx_[9]= 0.0;
                    sign[iu]= -1.0;
                    break;
                case 2:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    hx_[4]= 0.0;
                    hx_[5]= 0.0;
                    hx_[6]= 0.0;
                    hx_[7]= 0.0;
                    hx_[8]= 0.0;
                    hx_[9]= 0.0;
                    sign[iu]= 1.0;
                    break;
                case 3:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    hx_[4]= 0.0;
                    hx_[5]= 0.0;
                    hx_[6]= 0.0;
                    hx_[7]= 0.0;
                    hx_[8]= 0.0;
                    hx_[9]= 0.0;
                    sign[iu]= -1.0;
                    break;
                case 4:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    hx_[4]= 0.0;
                    hx_[5]= 0.0;
                    hx_[6]= 0.0;
                    hx_[7]= 0.0;
                    hx_[8]= 0.0;
                    hx_[9]= 0.0;
                    sign[iu]= 1.0;
                    break;
                case 5:
                    hx_[0]= -2.0/5.0*x[0];
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
                    hx_[3]= 0.0;
                    hx_[4]= 0.0;
                    hx_[5]= 0.0;
                    hx_[6]= 0.0;
                    hx_[7]= 0.0;
                    hx_[8]= 0.0;
                    hx_[9]= 0.0;
                    sign[iu]= -1.0;
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

    aux_s1_0= -0.045999999999999999*x[0] - 0.059999999999999998*x[1] - 0.59199999999999997*x[2] - 0.33500000000000002*x[3] - 0.60999999999999999*x[4] - 0.128*x[5] - 0.54900000000000004*x[6] + 0.11799999999999999*x[7] - 0.60399999999999998*x[8] - 0.63600000000000001*x[9];
    if(isNANorINF(aux_s1_0)) { PRNT("    @k %d: aux_s1_0 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_0); return 0; }
    aux_s1_1= -0.84499999999999997*x[0] + 0.42099999999999999*x[1] + 0.34799999999999998*x[2] + 0.378*x[3] - 0.071999999999999995*x[4] + 0.629*x[5] - 0.51400000000000001*x[6] - 0.11600000000000001*x[7] + 0.19500000000000001*x[8] - 0.871*x[9];
    if(isNANorINF(aux_s1_1)) { PRNT("    @k %d: aux_s1_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_1); return 0; }
    aux_s1_2= 0.754*x[0] + 0.25700000000000001*x[1] - 0.066000000000000003*x[2] - 0.66200000000000003*x[3] - 1.0740000000000001*x[4] - 0.51300000000000001*x[5] + 0.25700000000000001*x[6] + 0.504*x[7] + 0.34999999999999998*x[8] - 0.222*x[9];
    if(isNANorINF(aux_s1_2)) { PRNT("    @k %d: aux_s1_2 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_2); return 0; }
    aux_s1_3= -0.079000000000000001*x[0] - 0.41199999999999998*x[1] + 1.103*x[2] + 0.042000000000000003*x[3] + 0.050000000000000003*x[4] - 0.34999999999999998*x[5] - 0.57299999999999995*x[6] + 0.253*x[7] + 1.1739999999999999*x[8] - 0.53300000000000003*x[9];
    if(isNANorINF(aux_s1_3)) { PRNT("    @k %d: aux_s1_3 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_3); return 0; }
    aux_s1_4= -0.019*x[0] + 0.42999999999999999*x[1] - 0.59499999999999997*x[2] - 0.37*x[3] + 1.1950000000000001*x[4] + 0.34899999999999998*x[5] - 0.32200000000000001*x[6] + 0.0089999999999999993*x[7] + 0.36199999999999999*x[8] - 0.084000000000000005*x[9];
    if(isNANorINF(aux_s1_4)) { PRNT("    @k %d: aux_s1_4 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_4); return 0; }
    aux_s1_5= 0.29299999999999998*x[0] + 0.22*x[1] + 0.63*x[2] + 0.53600000000000003*x[3] - 0.002*x[4] + 0.35799999999999998*x[5] - 0.047*x[6] - 0.33500000000000002*x[7] + 1.2450000000000001*x[8] + 0.042999999999999997*x[9];
    if(isNANorINF(aux_s1_5)) { PRNT("    @k %d: aux_s1_5 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_5); return 0; }
    aux_s1_6= 0.67300000000000004*x[0] - 0.14499999999999999*x[1] - 0.33200000000000002*x[2] - 1.0189999999999999*x[3] + 0.79100000000000004*x[4] - 0.55900000000000005*x[5] - 0.55000000000000004*x[6] + 0.70099999999999996*x[7] + 0.39800000000000002*x[8] + 0.22*x[9];
    if(isNANorINF(aux_s1_6)) { PRNT("    @k %d: aux_s1_6 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_6); return 0; }
    aux_s1_7= -0.54600000000000004*x[0] + 0.086999999999999994*x[1] - 0.222*x[2] - 0.52300000000000002*x[3] - 0.216*x[4] + 0.17199999999999999*x[5] - 0.45700000000000002*x[6] - 0.17499999999999999*x[7] - 0.122*x[8] + 0.46999999999999997*x[9];
    if(isNANorINF(aux_s1_7)) { PRNT("    @k %d: aux_s1_7 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_7); return 0; }
    aux_s1_8= 0.53000000000000003*x[0] + 0.010999999999999999*x[1] + 0.29699999999999999*x[2] - 0.33400000000000002*x[3] + 1.6259999999999999*x[4] + 0.38900000000000001*x[5] - 0.35199999999999998*x[6] + 0.042999999999999997*x[7] + 0.34999999999999998*x[8] - 0.73699999999999999*x[9];
    if(isNANorINF(aux_s1_8)) { PRNT("    @k %d: aux_s1_8 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_8); return 0; }
    aux_s1_9= 1.03*x[0] + 0.318*x[1] + 0.38200000000000001*x[2] + 0.70099999999999996*x[3] + 0.33700000000000002*x[4] + 0.30499999999999999*x[5] + 0.437*x[6] + 0.746*x[7] - 0.498*x[8] - 0.498*x[9];
    if(isNANorINF(aux_s1_9)) { PRNT("    @k %d: aux_s1_9 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s1_9); return 0; }
    return 1;
}

static int calcXUVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    double **const p= o->p;
    const double w_pen= o->w_pen_l;

    aux_s2_0= 0.91800000000000004*u[0] - 0.40300000000000002*u[1] - 0.20100000000000001*u[2];
    if(isNANorINF(aux_s2_0)) { PRNT("    @k %d: aux_s2_0 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_0); return 0; }
    aux_s2_1= 0.97999999999999998*u[0] + 0.63500000000000001*u[1] + 0.68999999999999995*u[2];
    if(isNANorINF(aux_s2_1)) { PRNT("    @k %d: aux_s2_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_1); return 0; }
    aux_s2_2= -1.3280000000000001*u[0] + 1.016*u[1] - 0.35799999999999998*u[2];
    if(isNANorINF(aux_s2_2)) { PRNT("    @k %d: aux_s2_2 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_2); return 0; }
    aux_s2_3= 0.73699999999999999*u[0] + 0.249*u[1] + 2.125*u[2];
    if(isNANorINF(aux_s2_3)) { PRNT("    @k %d: aux_s2_3 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_3); return 0; }
    aux_s2_4= 0.24299999999999999*u[0] - 0.13900000000000001*u[1] + 0.089999999999999997*u[2];
    if(isNANorINF(aux_s2_4)) { PRNT("    @k %d: aux_s2_4 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_4); return 0; }
    aux_s2_5= -0.33000000000000002*u[0] + 1.036*u[1] + 1.105*u[2];
    if(isNANorINF(aux_s2_5)) { PRNT("    @k %d: aux_s2_5 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_5); return 0; }
    aux_s2_6= 0.89400000000000002*u[0] + 0.031*u[1] - 0.042000000000000003*u[2];
    if(isNANorINF(aux_s2_6)) { PRNT("    @k %d: aux_s2_6 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_6); return 0; }
    aux_s2_7= 0.035999999999999997*u[0] - 0.90500000000000003*u[1] - 0.42099999999999999*u[2];
    if(isNANorINF(aux_s2_7)) { PRNT("    @k %d: aux_s2_7 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_7); return 0; }
    aux_s2_8= -0.69199999999999995*u[0] - 0.13700000000000001*u[1] + 0.058999999999999997*u[2];
    if(isNANorINF(aux_s2_8)) { PRNT("    @k %d: aux_s2_8 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_8); return 0; }
    aux_s2_9= 0.14299999999999999*u[0] + 1.9039999999999999*u[1] + 0.25*u[2];
    if(isNANorINF(aux_s2_9)) { PRNT("    @k %d: aux_s2_9 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_s2_9); return 0; }
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

#if FULL_DDP
#endif
    return 1;
}

#ifndef ILQG_REC  /* a back-end may define these two before including this file */
#define ILQG_REC(member, index) t->member[index]
#define ILQG_REC_DONE(member, first, count)  /* entries first .. first+count-1 have been assigned */
#endif
/* the entries bp_derivsL_first assigns outside the runs, as X(member, index) ... */
#define ILQG_REC_DIRECT(X) X(cxx, 2) X(cxx, 5) X(cxx, 9) X(cxx, 14) X(cxx, 20) X(cxx, 27) X(cxx, 35) X(cxx, 44) X(cxx, 54) X(cu, 0) X(cu, 1) X(cu, 2)
static int bp_derivsL_first(trajEl_t *t, int k, double **p) {
    const double *const x= t->x;
    const double *const u= t->u;

    /* auxiliaries read here, taken once */
    const double v_aux_s1_0= aux_s1_0, v_aux_s2_0= aux_s2_0, v_aux_s1_1= aux_s1_1, v_aux_s2_1= aux_s2_1;
    const double v_aux_s1_2= aux_s1_2, v_aux_s2_2= aux_s2_2, v_aux_s1_3= aux_s1_3, v_aux_s2_3= aux_s2_3;
    const double v_aux_s1_4= aux_s1_4, v_aux_s2_4= aux_s2_4, v_aux_s1_5= aux_s1_5, v_aux_s2_5= aux_s2_5;
    const double v_aux_s1_6= aux_s1_6, v_aux_s2_6= aux_s2_6, v_aux_s1_7= aux_s1_7, v_aux_s2_7= aux_s2_7;
    const double v_aux_s1_8= aux_s1_8, v_aux_s2_8= aux_s2_8, v_aux_s1_9= aux_s1_9, v_aux_s2_9= aux_s2_9;
    /* ... and their sines and cosines */
    const double cos_v_aux_s1_0= cos(v_aux_s1_0), cos_v_aux_s2_0= cos(v_aux_s2_0), cos_v_aux_s1_1= cos(v_aux_s1_1), cos_v_aux_s2_1= cos(v_aux_s2_1);
    const double cos_v_aux_s1_2= cos(v_aux_s1_2), cos_v_aux_s2_2= cos(v_aux_s2_2), cos_v_aux_s1_3= cos(v_aux_s1_3), cos_v_aux_s2_3= cos(v_aux_s2_3);
    const double cos_v_aux_s1_4= cos(v_aux_s1_4), cos_v_aux_s2_4= cos(v_aux_s2_4), cos_v_aux_s1_5= cos(v_aux_s1_5), cos_v_aux_s2_5= cos(v_aux_s2_5);
    const double cos_v_aux_s1_6= cos(v_aux_s1_6), cos_v_aux_s2_6= cos(v_aux_s2_6), cos_v_aux_s1_7= cos(v_aux_s1_7), cos_v_aux_s2_7= cos(v_aux_s2_7);
    const double cos_v_aux_s1_8= cos(v_aux_s1_8), cos_v_aux_s2_8= cos(v_aux_s2_8), cos_v_aux_s1_9= cos(v_aux_s1_9), cos_v_aux_s2_9= cos(v_aux_s2_9);
    const double sin_v_aux_s1_0= sin(v_aux_s1_0), sin_v_aux_s2_0= sin(v_aux_s2_0), sin_v_aux_s1_1= sin(v_aux_s1_1), sin_v_aux_s2_1= sin(v_aux_s2_1);
    const double sin_v_aux_s1_2= sin(v_aux_s1_2), sin_v_aux_s2_2= sin(v_aux_s2_2), sin_v_aux_s1_3= sin(v_aux_s1_3), sin_v_aux_s2_3= sin(v_aux_s2_3);
    const double sin_v_aux_s1_4= sin(v_aux_s1_4), sin_v_aux_s2_4= sin(v_aux_s2_4), sin_v_aux_s1_5= sin(v_aux_s1_5), sin_v_aux_s2_5= sin(v_aux_s2_5);
    const double sin_v_aux_s1_6= sin(v_aux_s1_6), sin_v_aux_s2_6= sin(v_aux_s2_6), sin_v_aux_s1_7= sin(v_aux_s1_7), sin_v_aux_s2_7= sin(v_aux_s2_7);
    const double sin_v_aux_s1_8= sin(v_aux_s1_8), sin_v_aux_s2_8= sin(v_aux_s2_8), sin_v_aux_s1_9= sin(v_aux_s1_9), sin_v_aux_s2_9= sin(v_aux_s2_9);

    /* products shared by several entries */
    const double cs0= p[1][0];
    if(isNANorINF(cs0)) { PRNT("    @k %d: cs0 in line %d is nan or inf: %g\n", k, __LINE__-1, cs0); return 0; }
    const double cs1= p[0][0]*p[1][0]*cos_v_aux_s1_0*cos_v_aux_s2_0;
    if(isNANorINF(cs1)) { PRNT("    @k %d: cs1 in line %d is nan or inf: %g\n", k, __LINE__-1, cs1); return 0; }
    const double cs2= p[0][0]*p[1][0]*cos_v_aux_s1_1*cos_v_aux_s2_1;
    if(isNANorINF(cs2)) { PRNT("    @k %d: cs2 in line %d is nan or inf: %g\n", k, __LINE__-1, cs2); return 0; }
    const double cs3= p[0][0]*p[1][0]*cos_v_aux_s1_2*cos_v_aux_s2_2;
    if(isNANorINF(cs3)) { PRNT("    @k %d: cs3 in line %d is nan or inf: %g\n", k, __LINE__-1, cs3); return 0; }
    const double cs4= p[0][0]*p[1][0]*cos_v_aux_s1_3*cos_v_aux_s2_3;
    if(isNANorINF(cs4)) { PRNT("    @k %d: cs4 in line %d is nan or inf: %g\n", k, __LINE__-1, cs4); return 0; }
    const double cs5= p[0][0]*p[1][0]*cos_v_aux_s1_4*cos_v_aux_s2_4;
    if(isNANorINF(cs5)) { PRNT("    @k %d: cs5 in line %d is nan or inf: %g\n", k, __LINE__-1, cs5); return 0; }
    const double cs6= p[0][0]*p[1][0]*cos_v_aux_s1_5*cos_v_aux_s2_5;
    if(isNANorINF(cs6)) { PRNT("    @k %d: cs6 in line %d is nan or inf: %g\n", k, __LINE__-1, cs6); return 0; }
    const double cs7= p[0][0]*p[1][0]*cos_v_aux_s1_6*cos_v_aux_s2_6;
    if(isNANorINF(cs7)) { PRNT("    @k %d: cs7 in line %d is nan or inf: %g\n", k, __LINE__-1, cs7); return 0; }
    const double cs8= p[0][0]*p[1][0]*cos_v_aux_s1_7*cos_v_aux_s2_7;
    if(isNANorINF(cs8)) { PRNT("    @k %d: cs8 in line %d is nan or inf: %g\n", k, __LINE__-1, cs8); return 0; }
    const double cs9= p[0][0]*p[1][0]*cos_v_aux_s1_8*cos_v_aux_s2_8;
    if(isNANorINF(cs9)) { PRNT("    @k %d: cs9 in line %d is nan or inf: %g\n", k, __LINE__-1, cs9); return 0; }
    const double cs10= p[0][0]*p[1][0]*cos_v_aux_s1_9*cos_v_aux_s2_9;
    if(isNANorINF(cs10)) { PRNT("    @k %d: cs10 in line %d is nan or inf: %g\n", k, __LINE__-1, cs10); return 0; }
    const double cs11= p[0][0]*p[1][0]*sin_v_aux_s1_0*sin_v_aux_s2_0;
    if(isNANorINF(cs11)) { PRNT("    @k %d: cs11 in line %d is nan or inf: %g\n", k, __LINE__-1, cs11); return 0; }
    const double cs12= p[0][0]*p[1][0]*sin_v_aux_s1_1*sin_v_aux_s2_1;
    if(isNANorINF(cs12)) { PRNT("    @k %d: cs12 in line %d is nan or inf: %g\n", k, __LINE__-1, cs12); return 0; }
    const double cs13= p[0][0]*p[1][0]*sin_v_aux_s1_2*sin_v_aux_s2_2;
    if(isNANorINF(cs13)) { PRNT("    @k %d: cs13 in line %d is nan or inf: %g\n", k, __LINE__-1, cs13); return 0; }
    const double cs14= p[0][0]*p[1][0]*sin_v_aux_s1_3*sin_v_aux_s2_3;
    if(isNANorINF(cs14)) { PRNT("    @k %d: cs14 in line %d is nan or inf: %g\n", k, __LINE__-1, cs14); return 0; }
    const double cs15= p[0][0]*p[1][0]*sin_v_aux_s1_4*sin_v_aux_s2_4;
    if(isNANorINF(cs15)) { PRNT("    @k %d: cs15 in line %d is nan or inf: %g\n", k, __LINE__-1, cs15); return 0; }
    const double cs16= p[0][0]*p[1][0]*sin_v_aux_s1_5*sin_v_aux_s2_5;
    if(isNANorINF(cs16)) { PRNT("    @k %d: cs16 in line %d is nan or inf: %g\n", k, __LINE__-1, cs16); return 0; }
    const double cs17= p[0][0]*p[1][0]*sin_v_aux_s1_6*sin_v_aux_s2_6;
    if(isNANorINF(cs17)) { PRNT("    @k %d: cs17 in line %d is nan or inf: %g\n", k, __LINE__-1, cs17); return 0; }
    const double cs18= p[0][0]*p[1][0]*sin_v_aux_s1_7*sin_v_aux_s2_7;
    if(isNANorINF(cs18)) { PRNT("    @k %d: cs18 in line %d is nan or inf: %g\n", k, __LINE__-1, cs18); return 0; }
    const double cs19= p[0][0]*p[1][0]*sin_v_aux_s1_8*sin_v_aux_s2_8;
    if(isNANorINF(cs19)) { PRNT("    @k %d: cs19 in line %d is nan or inf: %g\n", k, __LINE__-1, cs19); return 0; }
    const double cs20= p[0][0]*p[1][0]*sin_v_aux_s1_9*sin_v_aux_s2_9;
    if(isNANorINF(cs20)) { PRNT("    @k %d: cs20 in line %d is nan or inf: %g\n", k, __LINE__-1, cs20); return 0; }
    const double cs21= p[5][0]*x[0]/sqrt((p[3][0]*p[3][0]) + (x[0]*x[0]));
    if(isNANorINF(cs21)) { PRNT("    @k %d: cs21 in line %d is nan or inf: %g\n", k, __LINE__-1, cs21); return 0; }
    const double cs22= p[5][1]*x[1]/sqrt((p[3][0]*p[3][0]) + (x[1]*x[1]));
    if(isNANorINF(cs22)) { PRNT("    @k %d: cs22 in line %d is nan or inf: %g\n", k, __LINE__-1, cs22); return 0; }
    const double cs23= p[5][2]*x[2]/sqrt((p[3][0]*p[3][0]) + (x[2]*x[2]));
    if(isNANorINF(cs23)) { PRNT("    @k %d: cs23 in line %d is nan or inf: %g\n", k, __LINE__-1, cs23); return 0; }
    const double cs24= p[5][3]*x[3]/sqrt((p[3][0]*p[3][0]) + (x[3]*x[3]));
    if(isNANorINF(cs24)) { PRNT("    @k %d: cs24 in line %d is nan or inf: %g\n", k, __LINE__-1, cs24); return 0; }
    const double cs25= p[5][4]*x[4]/sqrt((p[3][0]*p[3][0]) + (x[4]*x[4]));
    if(isNANorINF(cs25)) { PRNT("    @k %d: cs25 in line %d is nan or inf: %g\n", k, __LINE__-1, cs25); return 0; }
    const double cs26= p[5][5]*x[5]/sqrt((p[3][0]*p[3][0]) + (x[5]*x[5]));
    if(isNANorINF(cs26)) { PRNT("    @k %d: cs26 in line %d is nan or inf: %g\n", k, __LINE__-1, cs26); return 0; }
    const double cs27= p[5][6]*x[6]/sqrt((p[3][0]*p[3][0]) + (x[6]*x[6]));
    if(isNANorINF(cs27)) { PRNT("    @k %d: cs27 in line %d is nan or inf: %g\n", k, __LINE__-1, cs27); return 0; }
    const double cs28= p[5][7]*x[7]/sqrt((p[3][0]*p[3][0]) + (x[7]*x[7]));
    if(isNANorINF(cs28)) { PRNT("    @k %d: cs28 in line %d is nan or inf: %g\n", k, __LINE__-1, cs28); return 0; }
    const double cs29= p[5][8]*x[8]/sqrt((p[3][0]*p[3][0]) + (x[8]*x[8]));
    if(isNANorINF(cs29)) { PRNT("    @k %d: cs29 in line %d is nan or inf: %g\n", k, __LINE__-1, cs29); return 0; }
    const double cs30= p[5][9]*x[9]/sqrt((p[3][0]*p[3][0]) + (x[9]*x[9]));
    if(isNANorINF(cs30)) { PRNT("    @k %d: cs30 in line %d is nan or inf: %g\n", k, __LINE__-1, cs30); return 0; }
    const double cs31= p[5][0]/sqrt((p[3][0]*p[3][0]) + (x[0]*x[0]));
    if(isNANorINF(cs31)) { PRNT("    @k %d: cs31 in line %d is nan or inf: %g\n", k, __LINE__-1, cs31); return 0; }
    const double cs32= p[5][0]*(x[0]*x[0])/(((p[3][0]*p[3][0]) + (x[0]*x[0]))*sqrt((p[3][0]*p[3][0]) + (x[0]*x[0])));
    if(isNANorINF(cs32)) { PRNT("    @k %d: cs32 in line %d is nan or inf: %g\n", k, __LINE__-1, cs32); return 0; }
    const double cs33= p[5][1]/sqrt((p[3][0]*p[3][0]) + (x[1]*x[1]));
    if(isNANorINF(cs33)) { PRNT("    @k %d: cs33 in line %d is nan or inf: %g\n", k, __LINE__-1, cs33); return 0; }
    const double cs34= p[5][1]*(x[1]*x[1])/(((p[3][0]*p[3][0]) + (x[1]*x[1]))*sqrt((p[3][0]*p[3][0]) + (x[1]*x[1])));
    if(isNANorINF(cs34)) { PRNT("    @k %d: cs34 in line %d is nan or inf: %g\n", k, __LINE__-1, cs34); return 0; }
    const double cs35= p[5][2]/sqrt((p[3][0]*p[3][0]) + (x[2]*x[2]));
    if(isNANorINF(cs35)) { PRNT("    @k %d: cs35 in line %d is nan or inf: %g\n", k, __LINE__-1, cs35); return 0; }
    const double cs36= p[5][2]*(x[2]*x[2])/(((p[3][0]*p[3][0]) + (x[2]*x[2]))*sqrt((p[3][0]*p[3][0]) + (x[2]*x[2])));
    if(isNANorINF(cs36)) { PRNT("    @k %d: cs36 in line %d is nan or inf: %g\n", k, __LINE__-1, cs36); return 0; }
    const double cs37= p[5][3]/sqrt((p[3][0]*p[3][0]) + (x[3]*x[3]));
    if(isNANorINF(cs37)) { PRNT("    @k %d: cs37 in line %d is nan or inf: %g\n", k, __LINE__-1, cs37); return 0; }
    const double cs38= p[5][3]*(x[3]*x[3])/(((p[3][0]*p[3][0]) + (x[3]*x[3]))*sqrt((p[3][0]*p[3][0]) + (x[3]*x[3])));
    if(isNANorINF(cs38)) { PRNT("    @k %d: cs38 in line %d is nan or inf: %g\n", k, __LINE__-1, cs38); return 0; }
    const double cs39= p[5][4]/sqrt((p[3][0]*p[3][0]) + (x[4]*x[4]));
    if(isNANorINF(cs39)) { PRNT("    @k %d: cs39 in line %d is nan or inf: %g\n", k, __LINE__-1, cs39); return 0; }
    const double cs40= p[5][4]*(x[4]*x[4])/(((p[3][0]*p[3][0]) + (x[4]*x[4]))*sqrt((p[3][0]*p[3][0]) + (x[4]*x[4])));
    if(isNANorINF(cs40)) { PRNT("    @k %d: cs40 in line %d is nan or inf: %g\n", k, __LINE__-1, cs40); return 0; }
    const double cs41= p[5][5]/sqrt((p[3][0]*p[3][0]) + (x[5]*x[5]));
    if(isNANorINF(cs41)) { PRNT("    @k %d: cs41 in line %d is nan or inf: %g\n", k, __LINE__-1, cs41); return 0; }
    const double cs42= p[5][5]*(x[5]*x[5])/(((p[3][0]*p[3][0]) + (x[5]*x[5]))*sqrt((p[3][0]*p[3][0]) + (x[5]*x[5])));
    if(isNANorINF(cs42)) { PRNT("    @k %d: cs42 in line %d is nan or inf: %g\n", k, __LINE__-1, cs42); return 0; }
    const double cs43= p[5][6]/sqrt((p[3][0]*p[3][0]) + (x[6]*x[6]));
    if(isNANorINF(cs43)) { PRNT("    @k %d: cs43 in line %d is nan or inf: %g\n", k, __LINE__-1, cs43); return 0; }
    const double cs44= p[5][6]*(x[6]*x[6])/(((p[3][0]*p[3][0]) + (x[6]*x[6]))*sqrt((p[3][0]*p[3][0]) + (x[6]*x[6])));
    if(isNANorINF(cs44)) { PRNT("    @k %d: cs44 in line %d is nan or inf: %g\n", k, __LINE__-1, cs44); return 0; }
    const double cs45= p[5][7]/sqrt((p[3][0]*p[3][0]) + (x[7]*x[7]));
    if(isNANorINF(cs45)) { PRNT("    @k %d: cs45 in line %d is nan or inf: %g\n", k, __LINE__-1, cs45); return 0; }
    const double cs46= p[5][7]*(x[7]*x[7])/(((p[3][0]*p[3][0]) + (x[7]*x[7]))*sqrt((p[3][0]*p[3][0]) + (x[7]*x[7])));
    if(isNANorINF(cs46)) { PRNT("    @k %d: cs46 in line %d is nan or inf: %g\n", k, __LINE__-1, cs46); return 0; }
    const double cs47= p[5][8]/sqrt((p[3][0]*p[3][0]) + (x[8]*x[8]));
    if(isNANorINF(cs47)) { PRNT("    @k %d: cs47 in line %d is nan or inf: %g\n", k, __LINE__-1, cs47); return 0; }
    const double cs48= p[5][8]*(x[8]*x[8])/(((p[3][0]*p[3][0]) + (x[8]*x[8]))*sqrt((p[3][0]*p[3][0]) + (x[8]*x[8])));
    if(isNANorINF(cs48)) { PRNT("    @k %d: cs48 in line %d is nan or inf: %g\n", k, __LINE__-1, cs48); return 0; }
    const double cs49= p[5][9]/sqrt((p[3][0]*p[3][0]) + (x[9]*x[9]));
    if(isNANorINF(cs49)) { PRNT("    @k %d: cs49 in line %d is nan or inf: %g\n", k, __LINE__-1, cs49); return 0; }
    const double cs50= p[5][9]*(x[9]*x[9])/(((p[3][0]*p[3][0]) + (x[9]*x[9]))*sqrt((p[3][0]*p[3][0]) + (x[9]*x[9])));
    if(isNANorINF(cs50)) { PRNT("    @k %d: cs50 in line %d is nan or inf: %g\n", k, __LINE__-1, cs50); return 0; }
    const double cs51= p[6][0]*u[0];
    if(isNANorINF(cs51)) { PRNT("    @k %d: cs51 in line %d is nan or inf: %g\n", k, __LINE__-1, cs51); return 0; }
    const double cs52= p[6][1]*u[1];
    if(isNANorINF(cs52)) { PRNT("    @k %d: cs52 in line %d is nan or inf: %g\n", k, __LINE__-1, cs52); return 0; }
    const double cs53= p[6][2]*u[2];
    if(isNANorINF(cs53)) { PRNT("    @k %d: cs53 in line %d is nan or inf: %g\n", k, __LINE__-1, cs53); return 0; }

    /* dynamics */
    ILQG_REC(fx, 0)= -1.417*cs0 - 0.045999999999999999*cs1 + 1.0;
    ILQG_REC(fx, 1)= -0.20000000000000001*cs0 - 0.84499999999999997*cs2;
    if(isNANorINF(ILQG_REC(fx, 0))) { PRNT("    @k %d: t->fx[0] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 0)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 1))) { PRNT("    @k %d: t->fx[1] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 1)); return 0; }
    ILQG_REC(fx, 2)= 0.16600000000000001*cs0 + 0.754*cs3;
    ILQG_REC(fx, 3)= 0.47799999999999998*cs0 - 0.079000000000000001*cs4;
    if(isNANorINF(ILQG_REC(fx, 2))) { PRNT("    @k %d: t->fx[2] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 2)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 3))) { PRNT("    @k %d: t->fx[3] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 3)); return 0; }
    ILQG_REC(fx, 4)= -0.23999999999999999*cs0 - 0.019*cs5;
    ILQG_REC(fx, 5)= -0.014999999999999999*cs0 + 0.29299999999999998*cs6;
    if(isNANorINF(ILQG_REC(fx, 4))) { PRNT("    @k %d: t->fx[4] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 4)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 5))) { PRNT("    @k %d: t->fx[5] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 5)); return 0; }
    ILQG_REC(fx, 6)= -0.113*cs0 + 0.67300000000000004*cs7;
    ILQG_REC(fx, 7)= 0.307*cs0 - 0.54600000000000004*cs8;
    if(isNANorINF(ILQG_REC(fx, 6))) { PRNT("    @k %d: t->fx[6] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 6)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 7))) { PRNT("    @k %d: t->fx[7] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 7)); return 0; }
    ILQG_REC(fx, 8)= -0.014*cs0 + 0.53000000000000003*cs9;
    ILQG_REC(fx, 9)= 0.096000000000000002*cs0 + 1.03*cs10;
    if(isNANorINF(ILQG_REC(fx, 8))) { PRNT("    @k %d: t->fx[8] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 8)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 9))) { PRNT("    @k %d: t->fx[9] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 9)); return 0; }
    ILQG_REC(fx, 10)= 0.089999999999999997*cs0 - 0.059999999999999998*cs1;
    ILQG_REC(fx, 11)= -0.95899999999999996*cs0 + 0.42099999999999999*cs2 + 1.0;
    if(isNANorINF(ILQG_REC(fx, 10))) { PRNT("    @k %d: t->fx[10] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 10)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 11))) { PRNT("    @k %d: t->fx[11] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 11)); return 0; }
    ILQG_REC(fx, 12)= -0.25800000000000001*cs0 + 0.25700000000000001*cs3;
    ILQG_REC(fx, 13)= 0.099000000000000005*cs0 - 0.41199999999999998*cs4;
    if(isNANorINF(ILQG_REC(fx, 12))) { PRNT("    @k %d: t->fx[12] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 12)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 13))) { PRNT("    @k %d: t->fx[13] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 13)); return 0; }
    ILQG_REC(fx, 14)= 0.129*cs0 + 0.42999999999999999*cs5;
    ILQG_REC(fx, 15)= -0.17299999999999999*cs0 + 0.22*cs6;
    if(isNANorINF(ILQG_REC(fx, 14))) { PRNT("    @k %d: t->fx[14] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 14)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 15))) { PRNT("    @k %d: t->fx[15] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 15)); return 0; }
    ILQG_REC(fx, 16)= 0.083000000000000004*cs0 - 0.14499999999999999*cs7;
    ILQG_REC(fx, 17)= -0.002*cs0 + 0.086999999999999994*cs8;
    if(isNANorINF(ILQG_REC(fx, 16))) { PRNT("    @k %d: t->fx[16] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 16)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 17))) { PRNT("    @k %d: t->fx[17] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 17)); return 0; }
    ILQG_REC(fx, 18)= -0.033000000000000002*cs0 + 0.010999999999999999*cs9;
    ILQG_REC(fx, 19)= 0.037999999999999999*cs0 + 0.318*cs10;
    if(isNANorINF(ILQG_REC(fx, 18))) { PRNT("    @k %d: t->fx[18] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 18)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 19))) { PRNT("    @k %d: t->fx[19] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 19)); return 0; }
    ILQG_REC(fx, 20)= -0.094*cs0 - 0.59199999999999997*cs1;
    ILQG_REC(fx, 21)= -0.10000000000000001*cs0 + 0.34799999999999998*cs2;
    if(isNANorINF(ILQG_REC(fx, 20))) { PRNT("    @k %d: t->fx[20] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 20)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 21))) { PRNT("    @k %d: t->fx[21] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 21)); return 0; }
    ILQG_REC(fx, 22)= -1.0529999999999999*cs0 - 0.066000000000000003*cs3 + 1.0;
    ILQG_REC(fx, 23)= -0.153*cs0 + 1.103*cs4;
    if(isNANorINF(ILQG_REC(fx, 22))) { PRNT("    @k %d: t->fx[22] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 22)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 23))) { PRNT("    @k %d: t->fx[23] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 23)); return 0; }
    ILQG_REC(fx, 24)= 0.029000000000000001*cs0 - 0.59499999999999997*cs5;
    ILQG_REC(fx, 25)= -0.16900000000000001*cs0 + 0.63*cs6;
    if(isNANorINF(ILQG_REC(fx, 24))) { PRNT("    @k %d: t->fx[24] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 24)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 25))) { PRNT("    @k %d: t->fx[25] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 25)); return 0; }
    ILQG_REC(fx, 26)= 0.016*cs0 - 0.33200000000000002*cs7;
    ILQG_REC(fx, 27)= 0.159*cs0 - 0.222*cs8;
    if(isNANorINF(ILQG_REC(fx, 26))) { PRNT("    @k %d: t->fx[26] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 26)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 27))) { PRNT("    @k %d: t->fx[27] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 27)); return 0; }
    ILQG_REC(fx, 28)= 0.17299999999999999*cs0 + 0.29699999999999999*cs9;
    ILQG_REC(fx, 29)= 0.025999999999999999*cs0 + 0.38200000000000001*cs10;
    if(isNANorINF(ILQG_REC(fx, 28))) { PRNT("    @k %d: t->fx[28] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 28)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 29))) { PRNT("    @k %d: t->fx[29] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 29)); return 0; }
    ILQG_REC(fx, 30)= 0.096000000000000002*cs0 - 0.33500000000000002*cs1;
    ILQG_REC(fx, 31)= -0.029000000000000001*cs0 + 0.378*cs2;
    if(isNANorINF(ILQG_REC(fx, 30))) { PRNT("    @k %d: t->fx[30] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 30)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 31))) { PRNT("    @k %d: t->fx[31] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 31)); return 0; }
    ILQG_REC(fx, 32)= 0.070000000000000007*cs0 - 0.66200000000000003*cs3;
    if(isNANorINF(ILQG_REC(fx, 32))) { PRNT("    @k %d: t->fx[32] in line %d is nan or inf: %g\n", k, __LINE__-1, ILQG_REC(fx, 32)); return 0; }
    ILQG_REC_DONE(fx, 0, 33)
    ILQG_REC(fx, 33)= -0.72999999999999998*cs0 + 0.042000000000000003*cs4 + 1.0;
    ILQG_REC(fx, 34)= -0.17799999999999999*cs0 - 0.37*cs5;
    if(isNANorINF(ILQG_REC(fx, 33))) { PRNT("    @k %d: t->fx[33] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 33)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 34))) { PRNT("    @k %d: t->fx[34] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 34)); return 0; }
    ILQG_REC(fx, 35)= 0.032000000000000001*cs0 + 0.53600000000000003*cs6;
    ILQG_REC(fx, 36)= 0.222*cs0 - 1.0189999999999999*cs7;
    if(isNANorINF(ILQG_REC(fx, 35))) { PRNT("    @k %d: t->fx[35] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 35)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 36))) { PRNT("    @k %d: t->fx[36] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 36)); return 0; }
    ILQG_REC(fx, 37)= -0.125*cs0 - 0.52300000000000002*cs8;
    ILQG_REC(fx, 38)= -0.042000000000000003*cs0 - 0.33400000000000002*cs9;
    if(isNANorINF(ILQG_REC(fx, 37))) { PRNT("    @k %d: t->fx[37] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 37)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 38))) { PRNT("    @k %d: t->fx[38] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 38)); return 0; }
    ILQG_REC(fx, 39)= -0.13200000000000001*cs0 + 0.70099999999999996*cs10;
    ILQG_REC(fx, 40)= 0.29999999999999999*cs0 - 0.60999999999999999*cs1;
    if(isNANorINF(ILQG_REC(fx, 39))) { PRNT("    @k %d: t->fx[39] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 39)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 40))) { PRNT("    @k %d: t->fx[40] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 40)); return 0; }
    ILQG_REC(fx, 41)= -0.081000000000000003*cs0 - 0.071999999999999995*cs2;
    ILQG_REC(fx, 42)= 0.45100000000000001*cs0 - 1.0740000000000001*cs3;
    if(isNANorINF(ILQG_REC(fx, 41))) { PRNT("    @k %d: t->fx[41] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 41)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 42))) { PRNT("    @k %d: t->fx[42] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 42)); return 0; }
    ILQG_REC(fx, 43)= -0.26500000000000001*cs0 + 0.050000000000000003*cs4;
    ILQG_REC(fx, 44)= -1.095*cs0 + 1.1950000000000001*cs5 + 1.0;
    if(isNANorINF(ILQG_REC(fx, 43))) { PRNT("    @k %d: t->fx[43] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 43)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 44))) { PRNT("    @k %d: t->fx[44] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 44)); return 0; }
    ILQG_REC(fx, 45)= -0.22800000000000001*cs0 - 0.002*cs6;
    ILQG_REC(fx, 46)= 0.099000000000000005*cs0 + 0.79100000000000004*cs7;
    if(isNANorINF(ILQG_REC(fx, 45))) { PRNT("    @k %d: t->fx[45] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 45)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 46))) { PRNT("    @k %d: t->fx[46] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 46)); return 0; }
    ILQG_REC(fx, 47)= -0.17699999999999999*cs0 - 0.216*cs8;
    ILQG_REC(fx, 48)= -0.184*cs0 + 1.6259999999999999*cs9;
    if(isNANorINF(ILQG_REC(fx, 47))) { PRNT("    @k %d: t->fx[47] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 47)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 48))) { PRNT("    @k %d: t->fx[48] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 48)); return 0; }
    ILQG_REC(fx, 49)= 0.23599999999999999*cs0 + 0.33700000000000002*cs10;
    ILQG_REC(fx, 50)= 0.309*cs0 - 0.128*cs1;
    if(isNANorINF(ILQG_REC(fx, 49))) { PRNT("    @k %d: t->fx[49] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 49)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 50))) { PRNT("    @k %d: t->fx[50] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 50)); return 0; }
    ILQG_REC(fx, 51)= -0.22600000000000001*cs0 + 0.629*cs2;
    ILQG_REC(fx, 52)= -0.13400000000000001*cs0 - 0.51300000000000001*cs3;
    if(isNANorINF(ILQG_REC(fx, 51))) { PRNT("    @k %d: t->fx[51] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 51)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 52))) { PRNT("    @k %d: t->fx[52] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 52)); return 0; }
    ILQG_REC(fx, 53)= 0.23599999999999999*cs0 - 0.34999999999999998*cs4;
    ILQG_REC(fx, 54)= -0.049000000000000002*cs0 + 0.34899999999999998*cs5;
    if(isNANorINF(ILQG_REC(fx, 53))) { PRNT("    @k %d: t->fx[53] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 53)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 54))) { PRNT("    @k %d: t->fx[54] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 54)); return 0; }
    ILQG_REC(fx, 55)= -1.1479999999999999*cs0 + 0.35799999999999998*cs6 + 1.0;
    ILQG_REC(fx, 56)= 0.10299999999999999*cs0 - 0.55900000000000005*cs7;
    if(isNANorINF(ILQG_REC(fx, 55))) { PRNT("    @k %d: t->fx[55] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 55)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 56))) { PRNT("    @k %d: t->fx[56] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 56)); return 0; }
    ILQG_REC(fx, 57)= 0.070999999999999994*cs0 + 0.17199999999999999*cs8;
    ILQG_REC(fx, 58)= -0.021000000000000001*cs0 + 0.38900000000000001*cs9;
    if(isNANorINF(ILQG_REC(fx, 57))) { PRNT("    @k %d: t->fx[57] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 57)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 58))) { PRNT("    @k %d: t->fx[58] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 58)); return 0; }
    ILQG_REC(fx, 59)= -0.032000000000000001*cs0 + 0.30499999999999999*cs10;
    ILQG_REC(fx, 60)= -0.041000000000000002*cs0 - 0.54900000000000004*cs1;
    if(isNANorINF(ILQG_REC(fx, 59))) { PRNT("    @k %d: t->fx[59] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 59)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 60))) { PRNT("    @k %d: t->fx[60] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 60)); return 0; }
    ILQG_REC(fx, 61)= -0.059999999999999998*cs0 - 0.51400000000000001*cs2;
    ILQG_REC(fx, 62)= -0.072999999999999995*cs0 + 0.25700000000000001*cs3;
    if(isNANorINF(ILQG_REC(fx, 61))) { PRNT("    @k %d: t->fx[61] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 61)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 62))) { PRNT("    @k %d: t->fx[62] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 62)); return 0; }
    ILQG_REC(fx, 63)= -0.53900000000000003*cs0 - 0.57299999999999995*cs4;
    ILQG_REC(fx, 64)= -0.153*cs0 - 0.32200000000000001*cs5;
    if(isNANorINF(ILQG_REC(fx, 63))) { PRNT("    @k %d: t->fx[63] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 63)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 64))) { PRNT("    @k %d: t->fx[64] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 64)); return 0; }
    ILQG_REC(fx, 65)= -0.20999999999999999*cs0 - 0.047*cs6;
    ILQG_REC(fx, 66)= -1.161*cs0 - 0.55000000000000004*cs7 + 1.0;
    if(isNANorINF(ILQG_REC(fx, 65))) { PRNT("    @k %d: t->fx[65] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 65)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 66))) { PRNT("    @k %d: t->fx[66] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 66)); return 0; }
    ILQG_REC(fx, 67)= 0.11*cs0 - 0.45700000000000002*cs8;
    ILQG_REC(fx, 68)= -0.014999999999999999*cs0 - 0.35199999999999998*cs9;
    if(isNANorINF(ILQG_REC(fx, 67))) { PRNT("    @k %d: t->fx[67] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 67)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 68))) { PRNT("    @k %d: t->fx[68] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 68)); return 0; }
    ILQG_REC(fx, 69)= 0.222*cs0 + 0.437*cs10;
    ILQG_REC(fx, 70)= 0.17199999999999999*cs0 + 0.11799999999999999*cs1;
    if(isNANorINF(ILQG_REC(fx, 69))) { PRNT("    @k %d: t->fx[69] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 69)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 70))) { PRNT("    @k %d: t->fx[70] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 70)); return 0; }
    ILQG_REC(fx, 71)= -0.248*cs0 - 0.11600000000000001*cs2;
    ILQG_REC(fx, 72)= -0.36299999999999999*cs0 + 0.504*cs3;
    if(isNANorINF(ILQG_REC(fx, 71))) { PRNT("    @k %d: t->fx[71] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 71)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 72))) { PRNT("    @k %d: t->fx[72] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 72)); return 0; }
    ILQG_REC(fx, 73)= 0.217*cs0 + 0.253*cs4;
    ILQG_REC(fx, 74)= 0.28299999999999997*cs0 + 0.0089999999999999993*cs5;
    if(isNANorINF(ILQG_REC(fx, 73))) { PRNT("    @k %d: t->fx[73] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 73)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 74))) { PRNT("    @k %d: t->fx[74] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 74)); return 0; }
    ILQG_REC(fx, 75)= -0.104*cs0 - 0.33500000000000002*cs6;
    ILQG_REC(fx, 76)= 0.33900000000000002*cs0 + 0.70099999999999996*cs7;
    if(isNANorINF(ILQG_REC(fx, 75))) { PRNT("    @k %d: t->fx[75] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 75)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 76))) { PRNT("    @k %d: t->fx[76] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 76)); return 0; }
    ILQG_REC(fx, 77)= -1.0600000000000001*cs0 - 0.17499999999999999*cs8 + 1.0;
    ILQG_REC(fx, 78)= 0.436*cs0 + 0.042999999999999997*cs9;
    if(isNANorINF(ILQG_REC(fx, 77))) { PRNT("    @k %d: t->fx[77] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 77)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 78))) { PRNT("    @k %d: t->fx[78] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 78)); return 0; }
    ILQG_REC(fx, 79)= -0.151*cs0 + 0.746*cs10;
    ILQG_REC(fx, 80)= -0.34599999999999997*cs0 - 0.60399999999999998*cs1;
    if(isNANorINF(ILQG_REC(fx, 79))) { PRNT("    @k %d: t->fx[79] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 79)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 80))) { PRNT("    @k %d: t->fx[80] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 80)); return 0; }
    ILQG_REC(fx, 81)= -0.095000000000000001*cs0 + 0.19500000000000001*cs2;
    ILQG_REC(fx, 82)= -0.28100000000000003*cs0 + 0.34999999999999998*cs3;
    if(isNANorINF(ILQG_REC(fx, 81))) { PRNT("    @k %d: t->fx[81] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 81)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 82))) { PRNT("    @k %d: t->fx[82] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 82)); return 0; }
    ILQG_REC(fx, 83)= -0.16700000000000001*cs0 + 1.1739999999999999*cs4;
    ILQG_REC(fx, 84)= -0.188*cs0 + 0.36199999999999999*cs5;
    if(isNANorINF(ILQG_REC(fx, 83))) { PRNT("    @k %d: t->fx[83] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 83)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 84))) { PRNT("    @k %d: t->fx[84] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 84)); return 0; }
    ILQG_REC(fx, 85)= -0.252*cs0 + 1.2450000000000001*cs6;
    ILQG_REC(fx, 86)= 0.065000000000000002*cs0 + 0.39800000000000002*cs7;
    if(isNANorINF(ILQG_REC(fx, 85))) { PRNT("    @k %d: t->fx[85] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 85)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 86))) { PRNT("    @k %d: t->fx[86] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 86)); return 0; }
    ILQG_REC(fx, 87)= 0.039*cs0 - 0.122*cs8;
    ILQG_REC(fx, 88)= -1.2010000000000001*cs0 + 0.34999999999999998*cs9 + 1.0;
    if(isNANorINF(ILQG_REC(fx, 87))) { PRNT("    @k %d: t->fx[87] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 87)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 88))) { PRNT("    @k %d: t->fx[88] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 88)); return 0; }
    ILQG_REC(fx, 89)= -0.14699999999999999*cs0 - 0.498*cs10;
    ILQG_REC(fx, 90)= 0.025999999999999999*cs0 - 0.63600000000000001*cs1;
    if(isNANorINF(ILQG_REC(fx, 89))) { PRNT("    @k %d: t->fx[89] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 89)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 90))) { PRNT("    @k %d: t->fx[90] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 90)); return 0; }
    ILQG_REC(fx, 91)= -0.097000000000000003*cs0 - 0.871*cs2;
    ILQG_REC(fx, 92)= 0.088999999999999996*cs0 - 0.222*cs3;
    if(isNANorINF(ILQG_REC(fx, 91))) { PRNT("    @k %d: t->fx[91] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 91)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 92))) { PRNT("    @k %d: t->fx[92] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 92)); return 0; }
    ILQG_REC(fx, 93)= 0.063*cs0 - 0.53300000000000003*cs4;
    ILQG_REC(fx, 94)= 0.23100000000000001*cs0 - 0.084000000000000005*cs5;
    if(isNANorINF(ILQG_REC(fx, 93))) { PRNT("    @k %d: t->fx[93] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 93)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 94))) { PRNT("    @k %d: t->fx[94] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 94)); return 0; }
    ILQG_REC(fx, 95)= -0.031*cs0 + 0.042999999999999997*cs6;
    ILQG_REC(fx, 96)= -0.25700000000000001*cs0 + 0.22*cs7;
    if(isNANorINF(ILQG_REC(fx, 95))) { PRNT("    @k %d: t->fx[95] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 95)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 96))) { PRNT("    @k %d: t->fx[96] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 96)); return 0; }
    ILQG_REC_DONE(fx, 33, 64)
    ILQG_REC(fx, 97)= 0.129*cs0 + 0.46999999999999997*cs8;
    ILQG_REC(fx, 98)= 0.122*cs0 - 0.73699999999999999*cs9;
    if(isNANorINF(ILQG_REC(fx, 97))) { PRNT("    @k %d: t->fx[97] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 97)); return 0; }
    if(isNANorINF(ILQG_REC(fx, 98))) { PRNT("    @k %d: t->fx[98] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fx, 98)); return 0; }
    ILQG_REC(fx, 99)= -1.343*cs0 - 0.498*cs10 + 1.0;
    if(isNANorINF(ILQG_REC(fx, 99))) { PRNT("    @k %d: t->fx[99] in line %d is nan or inf: %g\n", k, __LINE__-1, ILQG_REC(fx, 99)); return 0; }

    ILQG_REC(fu, 0)= -0.69999999999999996*cs0 - 0.91800000000000004*cs11;
    ILQG_REC(fu, 1)= 0.51000000000000001*cs0 - 0.97999999999999998*cs12;
    if(isNANorINF(ILQG_REC(fu, 0))) { PRNT("    @k %d: t->fu[0] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 0)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 1))) { PRNT("    @k %d: t->fu[1] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 1)); return 0; }
    ILQG_REC(fu, 2)= 0.036999999999999998*cs0 + 1.3280000000000001*cs13;
    ILQG_REC(fu, 3)= -0.41599999999999998*cs0 - 0.73699999999999999*cs14;
    if(isNANorINF(ILQG_REC(fu, 2))) { PRNT("    @k %d: t->fu[2] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 2)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 3))) { PRNT("    @k %d: t->fu[3] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 3)); return 0; }
    ILQG_REC(fu, 4)= 0.64600000000000002*cs0 - 0.24299999999999999*cs15;
    ILQG_REC(fu, 5)= 0.35699999999999998*cs0 + 0.33000000000000002*cs16;
    if(isNANorINF(ILQG_REC(fu, 4))) { PRNT("    @k %d: t->fu[4] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 4)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 5))) { PRNT("    @k %d: t->fu[5] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 5)); return 0; }
    ILQG_REC(fu, 6)= 0.76100000000000001*cs0 - 0.89400000000000002*cs17;
    ILQG_REC(fu, 7)= -0.48099999999999998*cs0 - 0.035999999999999997*cs18;
    if(isNANorINF(ILQG_REC(fu, 6))) { PRNT("    @k %d: t->fu[6] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 6)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 7))) { PRNT("    @k %d: t->fu[7] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 7)); return 0; }
    ILQG_REC(fu, 8)= 0.215*cs0 + 0.69199999999999995*cs19;
    ILQG_REC(fu, 9)= -0.105*cs0 - 0.14299999999999999*cs20;
    if(isNANorINF(ILQG_REC(fu, 8))) { PRNT("    @k %d: t->fu[8] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 8)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 9))) { PRNT("    @k %d: t->fu[9] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 9)); return 0; }
    ILQG_REC(fu, 10)= 0.30599999999999999*cs0 + 0.40300000000000002*cs11;
    ILQG_REC(fu, 11)= -0.027*cs0 - 0.63500000000000001*cs12;
    if(isNANorINF(ILQG_REC(fu, 10))) { PRNT("    @k %d: t->fu[10] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 10)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 11))) { PRNT("    @k %d: t->fu[11] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 11)); return 0; }
    ILQG_REC(fu, 12)= 0.91600000000000004*cs0 - 1.016*cs13;
    ILQG_REC(fu, 13)= 0.039*cs0 - 0.249*cs14;
    if(isNANorINF(ILQG_REC(fu, 12))) { PRNT("    @k %d: t->fu[12] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 12)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 13))) { PRNT("    @k %d: t->fu[13] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 13)); return 0; }
    ILQG_REC(fu, 14)= 0.14899999999999999*cs0 + 0.13900000000000001*cs15;
    ILQG_REC(fu, 15)= -0.51200000000000001*cs0 - 1.036*cs16;
    if(isNANorINF(ILQG_REC(fu, 14))) { PRNT("    @k %d: t->fu[14] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 14)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 15))) { PRNT("    @k %d: t->fu[15] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 15)); return 0; }
    ILQG_REC(fu, 16)= 0.064000000000000001*cs0 - 0.031*cs17;
    ILQG_REC(fu, 17)= -0.159*cs0 + 0.90500000000000003*cs18;
    if(isNANorINF(ILQG_REC(fu, 16))) { PRNT("    @k %d: t->fu[16] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 16)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 17))) { PRNT("    @k %d: t->fu[17] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 17)); return 0; }
    ILQG_REC(fu, 18)= 0.86899999999999999*cs0 + 0.13700000000000001*cs19;
    ILQG_REC(fu, 19)= -1.7509999999999999*cs0 - 1.9039999999999999*cs20;
    if(isNANorINF(ILQG_REC(fu, 18))) { PRNT("    @k %d: t->fu[18] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 18)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 19))) { PRNT("    @k %d: t->fu[19] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 19)); return 0; }
    ILQG_REC(fu, 20)= 0.151*cs0 + 0.20100000000000001*cs11;
    ILQG_REC(fu, 21)= 0.122*cs0 - 0.68999999999999995*cs12;
    if(isNANorINF(ILQG_REC(fu, 20))) { PRNT("    @k %d: t->fu[20] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 20)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 21))) { PRNT("    @k %d: t->fu[21] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 21)); return 0; }
    ILQG_REC(fu, 22)= -0.014*cs0 + 0.35799999999999998*cs13;
    ILQG_REC(fu, 23)= -0.64600000000000002*cs0 - 2.125*cs14;
    if(isNANorINF(ILQG_REC(fu, 22))) { PRNT("    @k %d: t->fu[22] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 22)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 23))) { PRNT("    @k %d: t->fu[23] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 23)); return 0; }
    ILQG_REC(fu, 24)= 0.69899999999999995*cs0 - 0.089999999999999997*cs15;
    ILQG_REC(fu, 25)= -0.379*cs0 - 1.105*cs16;
    if(isNANorINF(ILQG_REC(fu, 24))) { PRNT("    @k %d: t->fu[24] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 24)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 25))) { PRNT("    @k %d: t->fu[25] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 25)); return 0; }
    ILQG_REC(fu, 26)= 0.086999999999999994*cs0 + 0.042000000000000003*cs17;
    ILQG_REC(fu, 27)= -0.245*cs0 + 0.42099999999999999*cs18;
    if(isNANorINF(ILQG_REC(fu, 26))) { PRNT("    @k %d: t->fu[26] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 26)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 27))) { PRNT("    @k %d: t->fu[27] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 27)); return 0; }
    ILQG_REC(fu, 28)= 1.629*cs0 - 0.058999999999999997*cs19;
    ILQG_REC(fu, 29)= 0.24099999999999999*cs0 - 0.25*cs20;
    if(isNANorINF(ILQG_REC(fu, 28))) { PRNT("    @k %d: t->fu[28] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 28)); return 0; }
    if(isNANorINF(ILQG_REC(fu, 29))) { PRNT("    @k %d: t->fu[29] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(fu, 29)); return 0; }
    ILQG_REC_DONE(fx, 97, 33)

    /* cost */
    ILQG_REC(cx, 0)= cs21;
    ILQG_REC(cx, 1)= cs22;
    if(isNANorINF(ILQG_REC(cx, 0))) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 0)); return 0; }
    if(isNANorINF(ILQG_REC(cx, 1))) { PRNT("    @k %d: t->cx[1] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 1)); return 0; }
    ILQG_REC(cx, 2)= cs23;
    ILQG_REC(cx, 3)= cs24;
    if(isNANorINF(ILQG_REC(cx, 2))) { PRNT("    @k %d: t->cx[2] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 2)); return 0; }
    if(isNANorINF(ILQG_REC(cx, 3))) { PRNT("    @k %d: t->cx[3] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 3)); return 0; }
    ILQG_REC(cx, 4)= cs25;
    ILQG_REC(cx, 5)= cs26;
    if(isNANorINF(ILQG_REC(cx, 4))) { PRNT("    @k %d: t->cx[4] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 4)); return 0; }
    if(isNANorINF(ILQG_REC(cx, 5))) { PRNT("    @k %d: t->cx[5] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 5)); return 0; }
    ILQG_REC(cx, 6)= cs27;
    ILQG_REC(cx, 7)= cs28;
    if(isNANorINF(ILQG_REC(cx, 6))) { PRNT("    @k %d: t->cx[6] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 6)); return 0; }
    if(isNANorINF(ILQG_REC(cx, 7))) { PRNT("    @k %d: t->cx[7] in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_REC(cx, 7)); return 0; }
    ILQG_REC(cx, 8)= cs29;
    if(isNANorINF(ILQG_REC(cx, 8))) { PRNT("    @k %d: t->cx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, ILQG_REC(cx, 8)); return 0; }
    ILQG_REC_DONE(cx, 0, 9)
    ILQG_REC(cx, 9)= cs30;
    if(isNANorINF(ILQG_REC(cx, 9))) { PRNT("    @k %d: t->cx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, ILQG_REC(cx, 9)); return 0; }

    ILQG_REC(cxx, 0)= cs31 - cs32;
    if(isNANorINF(ILQG_REC(cxx, 0))) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, ILQG_REC(cxx, 0)); return 0; }
    ILQG_REC_DONE(cx, 9, 2)
    t->cxx[2]= cs33 - cs34;
    if(isNANorINF(t->cxx[2])) { PRNT("    @k %d: t->cxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[2]); return 0; }
    t->cxx[5]= cs35 - cs36;
    if(isNANorINF(t->cxx[5])) { PRNT("    @k %d: t->cxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[5]); return 0; }
    t->cxx[9]= cs37 - cs38;
    if(isNANorINF(t->cxx[9])) { PRNT("    @k %d: t->cxx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[9]); return 0; }
    t->cxx[14]= cs39 - cs40;
    if(isNANorINF(t->cxx[14])) { PRNT("    @k %d: t->cxx[14] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[14]); return 0; }
    t->cxx[20]= cs41 - cs42;
    if(isNANorINF(t->cxx[20])) { PRNT("    @k %d: t->cxx[20] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[20]); return 0; }
    t->cxx[27]= cs43 - cs44;
    if(isNANorINF(t->cxx[27])) { PRNT("    @k %d: t->cxx[27] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[27]); return 0; }
    t->cxx[35]= cs45 - cs46;
    if(isNANorINF(t->cxx[35])) { PRNT("    @k %d: t->cxx[35] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[35]); return 0; }
    t->cxx[44]= cs47 - cs48;
    if(isNANorINF(t->cxx[44])) { PRNT("    @k %d: t->cxx[44] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[44]); return 0; }
    t->cxx[54]= cs49 - cs50;
    if(isNANorINF(t->cxx[54])) { PRNT("    @k %d: t->cxx[54] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[54]); return 0; }

    t->cu[0]= 2.0*cs51;
    t->cu[1]= 2.0*cs52;
    if(isNANorINF(t->cu[0])) { PRNT("    @k %d: t->cu[0] in line %d is nan or inf: %g\n", k, __LINE__-2, t->cu[0]); return 0; }
    if(isNANorINF(t->cu[1])) { PRNT("    @k %d: t->cu[1] in line %d is nan or inf: %g\n", k, __LINE__-2, t->cu[1]); return 0; }
    t->cu[2]= 2.0*cs53;
    if(isNANorINF(t->cu[2])) { PRNT("    @k %d: t->cu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cu[2]); return 0; }



    return 1;
}

#if FULL_DDP
static int bp_derivsL_second(trajEl_t *t, int k, double **p) {
    const double *const x= t->x;
    const double *const u= t->u;

    /* products shared by the entries of the tensors */
    const double ct0= p[0][0]*p[1][0]*sin(aux_s1_0)*cos(aux_s2_0);
    if(isNANorINF(ct0)) { PRNT("    @k %d: ct0 in line %d is nan or inf: %g\n", k, __LINE__-1, ct0); return 0; }
    const double ct1= p[0][0]*p[1][0]*sin(aux_s1_1)*cos(aux_s2_1);
    if(isNANorINF(ct1)) { PRNT("    @k %d: ct1 in line %d is nan or inf: %g\n", k, __LINE__-1, ct1); return 0; }
    const double ct2= p[0][0]*p[1][0]*sin(aux_s1_2)*cos(aux_s2_2);
    if(isNANorINF(ct2)) { PRNT("    @k %d: ct2 in line %d is nan or inf: %g\n", k, __LINE__-1, ct2); return 0; }
    const double ct3= p[0][0]*p[1][0]*sin(aux_s1_3)*cos(aux_s2_3);
    if(isNANorINF(ct3)) { PRNT("    @k %d: ct3 in line %d is nan or inf: %g\n", k, __LINE__-1, ct3); return 0; }
    const double ct4= p[0][0]*p[1][0]*sin(aux_s1_4)*cos(aux_s2_4);
    if(isNANorINF(ct4)) { PRNT("    @k %d: ct4 in line %d is nan or inf: %g\n", k, __LINE__-1, ct4); return 0; }
    const double ct5= p[0][0]*p[1][0]*sin(aux_s1_5)*cos(aux_s2_5);
    if(isNANorINF(ct5)) { PRNT("    @k %d: ct5 in line %d is nan or inf: %g\n", k, __LINE__-1, ct5); return 0; }
    const double ct6= p[0][0]*p[1][0]*sin(aux_s1_6)*cos(aux_s2_6);
    if(isNANorINF(ct6)) { PRNT("    @k %d: ct6 in line %d is nan or inf: %g\n", k, __LINE__-1, ct6); return 0; }
    const double ct7= p[0][0]*p[1][0]*sin(aux_s1_7)*cos(aux_s2_7);
    if(isNANorINF(ct7)) { PRNT("    @k %d: ct7 in line %d is nan or inf: %g\n", k, __LINE__-1, ct7); return 0; }
    const double ct8= p[0][0]*p[1][0]*sin(aux_s1_8)*cos(aux_s2_8);
    if(isNANorINF(ct8)) { PRNT("    @k %d: ct8 in line %d is nan or inf: %g\n", k, __LINE__-1, ct8); return 0; }
    const double ct9= p[0][0]*p[1][0]*sin(aux_s1_9)*cos(aux_s2_9);
    if(isNANorINF(ct9)) { PRNT("    @k %d: ct9 in line %d is nan or inf: %g\n", k, __LINE__-1, ct9); return 0; }
    const double ct10= p[0][0]*p[1][0]*sin(aux_s2_0)*cos(aux_s1_0);
    if(isNANorINF(ct10)) { PRNT("    @k %d: ct10 in line %d is nan or inf: %g\n", k, __LINE__-1, ct10); return 0; }
    const double ct11= p[0][0]*p[1][0]*sin(aux_s2_1)*cos(aux_s1_1);
    if(isNANorINF(ct11)) { PRNT("    @k %d: ct11 in line %d is nan or inf: %g\n", k, __LINE__-1, ct11); return 0; }
    const double ct12= p[0][0]*p[1][0]*sin(aux_s2_2)*cos(aux_s1_2);
    if(isNANorINF(ct12)) { PRNT("    @k %d: ct12 in line %d is nan or inf: %g\n", k, __LINE__-1, ct12); return 0; }
    const double ct13= p[0][0]*p[1][0]*sin(aux_s2_3)*cos(aux_s1_3);
    if(isNANorINF(ct13)) { PRNT("    @k %d: ct13 in line %d is nan or inf: %g\n", k, __LINE__-1, ct13); return 0; }
    const double ct14= p[0][0]*p[1][0]*sin(aux_s2_4)*cos(aux_s1_4);
    if(isNANorINF(ct14)) { PRNT("    @k %d: ct14 in line %d is nan or inf: %g\n", k, __LINE__-1, ct14); return 0; }
    const double ct15= p[0][0]*p[1][0]*sin(aux_s2_5)*cos(aux_s1_5);
    if(isNANorINF(ct15)) { PRNT("    @k %d: ct15 in line %d is nan or inf: %g\n", k, __LINE__-1, ct15); return 0; }
    const double ct16= p[0][0]*p[1][0]*sin(aux_s2_6)*cos(aux_s1_6);
    if(isNANorINF(ct16)) { PRNT("    @k %d: ct16 in line %d is nan or inf: %g\n", k, __LINE__-1, ct16); return 0; }
    const double ct17= p[0][0]*p[1][0]*sin(aux_s2_7)*cos(aux_s1_7);
    if(isNANorINF(ct17)) { PRNT("    @k %d: ct17 in line %d is nan or inf: %g\n", k, __LINE__-1, ct17); return 0; }
    const double ct18= p[0][0]*p[1][0]*sin(aux_s2_8)*cos(aux_s1_8);
    if(isNANorINF(ct18)) { PRNT("    @k %d: ct18 in line %d is nan or inf: %g\n", k, __LINE__-1, ct18); return 0; }
    const double ct19= p[0][0]*p[1][0]*sin(aux_s2_9)*cos(aux_s1_9);
    if(isNANorINF(ct19)) { PRNT("    @k %d: ct19 in line %d is nan or inf: %g\n", k, __LINE__-1, ct19); return 0; }

    t->fxx[0]= -0.0021159999999999998*ct0;
    if(isNANorINF(t->fxx[0])) { PRNT("    @k %d: t->fxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[0]); return 0; }
    t->fxx[1]= -0.0027599999999999999*ct0;
    if(isNANorINF(t->fxx[1])) { PRNT("    @k %d: t->fxx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[1]); return 0; }
    t->fxx[2]= -0.0035999999999999999*ct0;
    if(isNANorINF(t->fxx[2])) { PRNT("    @k %d: t->fxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[2]); return 0; }
    t->fxx[3]= -0.027231999999999999*ct0;
    if(isNANorINF(t->fxx[3])) { PRNT("    @k %d: t->fxx[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[3]); return 0; }
    t->fxx[4]= -0.035519999999999996*ct0;
    if(isNANorINF(t->fxx[4])) { PRNT("    @k %d: t->fxx[4] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[4]); return 0; }
    t->fxx[5]= -0.35046399999999994*ct0;
    if(isNANorINF(t->fxx[5])) { PRNT("    @k %d: t->fxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[5]); return 0; }
    t->fxx[6]= -0.01541*ct0;
    if(isNANorINF(t->fxx[6])) { PRNT("    @k %d: t->fxx[6] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[6]); return 0; }
    t->fxx[7]= -0.0201*ct0;
    if(isNANorINF(t->fxx[7])) { PRNT("    @k %d: t->fxx[7] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[7]); return 0; }
    t->fxx[8]= -0.19832*ct0;
    if(isNANorINF(t->fxx[8])) { PRNT("    @k %d: t->fxx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[8]); return 0; }
    t->fxx[9]= -0.11222500000000002*ct0;
    if(isNANorINF(t->fxx[9])) { PRNT("    @k %d: t->fxx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[9]); return 0; }
    t->fxx[10]= -0.028059999999999998*ct0;
    if(isNANorINF(t->fxx[10])) { PRNT("    @k %d: t->fxx[10] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[10]); return 0; }
    t->fxx[11]= -0.036600000000000001*ct0;
    if(isNANorINF(t->fxx[11])) { PRNT("    @k %d: t->fxx[11] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[11]); return 0; }
    t->fxx[12]= -0.36112*ct0;
    if(isNANorINF(t->fxx[12])) { PRNT("    @k %d: t->fxx[12] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[12]); return 0; }
    t->fxx[13]= -0.20435*ct0;
    if(isNANorINF(t->fxx[13])) { PRNT("    @k %d: t->fxx[13] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[13]); return 0; }
    t->fxx[14]= -0.37209999999999999*ct0;
    if(isNANorINF(t->fxx[14])) { PRNT("    @k %d: t->fxx[14] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[14]); return 0; }
    t->fxx[15]= -0.005888*ct0;
    if(isNANorINF(t->fxx[15])) { PRNT("    @k %d: t->fxx[15] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[15]); return 0; }
    t->fxx[16]= -0.0076800000000000002*ct0;
    if(isNANorINF(t->fxx[16])) { PRNT("    @k %d: t->fxx[16] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[16]); return 0; }
    t->fxx[17]= -0.075775999999999996*ct0;
    if(isNANorINF(t->fxx[17])) { PRNT("    @k %d: t->fxx[17] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[17]); return 0; }
    t->fxx[18]= -0.042880000000000001*ct0;
    if(isNANorINF(t->fxx[18])) { PRNT("    @k %d: t->fxx[18] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[18]); return 0; }
    t->fxx[19]= -0.078079999999999997*ct0;
    if(isNANorINF(t->fxx[19])) { PRNT("    @k %d: t->fxx[19] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[19]); return 0; }
    t->fxx[20]= -0.016383999999999999*ct0;
    if(isNANorINF(t->fxx[20])) { PRNT("    @k %d: t->fxx[20] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[20]); return 0; }
    t->fxx[21]= -0.025254000000000002*ct0;
    if(isNANorINF(t->fxx[21])) { PRNT("    @k %d: t->fxx[21] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[21]); return 0; }
    t->fxx[22]= -0.032940000000000004*ct0;
    if(isNANorINF(t->fxx[22])) { PRNT("    @k %d: t->fxx[22] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[22]); return 0; }
    t->fxx[23]= -0.32500800000000002*ct0;
    if(isNANorINF(t->fxx[23])) { PRNT("    @k %d: t->fxx[23] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[23]); return 0; }
    t->fxx[24]= -0.18391500000000002*ct0;
    if(isNANorINF(t->fxx[24])) { PRNT("    @k %d: t->fxx[24] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[24]); return 0; }
    t->fxx[25]= -0.33489000000000002*ct0;
    if(isNANorINF(t->fxx[25])) { PRNT("    @k %d: t->fxx[25] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[25]); return 0; }
    t->fxx[26]= -0.070272000000000001*ct0;
    if(isNANorINF(t->fxx[26])) { PRNT("    @k %d: t->fxx[26] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[26]); return 0; }
    t->fxx[27]= -0.30140100000000003*ct0;
    if(isNANorINF(t->fxx[27])) { PRNT("    @k %d: t->fxx[27] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[27]); return 0; }
    t->fxx[28]= 0.0054279999999999997*ct0;
    if(isNANorINF(t->fxx[28])) { PRNT("    @k %d: t->fxx[28] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[28]); return 0; }
    t->fxx[29]= 0.0070799999999999995*ct0;
    if(isNANorINF(t->fxx[29])) { PRNT("    @k %d: t->fxx[29] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[29]); return 0; }
    t->fxx[30]= 0.069855999999999988*ct0;
    if(isNANorINF(t->fxx[30])) { PRNT("    @k %d: t->fxx[30] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[30]); return 0; }
    t->fxx[31]= 0.039530000000000003*ct0;
    if(isNANorINF(t->fxx[31])) { PRNT("    @k %d: t->fxx[31] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[31]); return 0; }
    t->fxx[32]= 0.071979999999999988*ct0;
    if(isNANorINF(t->fxx[32])) { PRNT("    @k %d: t->fxx[32] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[32]); return 0; }
    t->fxx[33]= 0.015103999999999999*ct0;
    if(isNANorINF(t->fxx[33])) { PRNT("    @k %d: t->fxx[33] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[33]); return 0; }
    t->fxx[34]= 0.064782000000000006*ct0;
    if(isNANorINF(t->fxx[34])) { PRNT("    @k %d: t->fxx[34] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[34]); return 0; }
    t->fxx[35]= -0.013923999999999999*ct0;
    if(isNANorINF(t->fxx[35])) { PRNT("    @k %d: t->fxx[35] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[35]); return 0; }
    t->fxx[36]= -0.027784*ct0;
    if(isNANorINF(t->fxx[36])) { PRNT("    @k %d: t->fxx[36] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[36]); return 0; }
    t->fxx[37]= -0.036239999999999994*ct0;
    if(isNANorINF(t->fxx[37])) { PRNT("    @k %d: t->fxx[37] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[37]); return 0; }
    t->fxx[38]= -0.357568*ct0;
    if(isNANorINF(t->fxx[38])) { PRNT("    @k %d: t->fxx[38] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[38]); return 0; }
    t->fxx[39]= -0.20233999999999999*ct0;
    if(isNANorINF(t->fxx[39])) { PRNT("    @k %d: t->fxx[39] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[39]); return 0; }
    t->fxx[40]= -0.36843999999999999*ct0;
    if(isNANorINF(t->fxx[40])) { PRNT("    @k %d: t->fxx[40] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[40]); return 0; }
    t->fxx[41]= -0.077312000000000006*ct0;
    if(isNANorINF(t->fxx[41])) { PRNT("    @k %d: t->fxx[41] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[41]); return 0; }
    t->fxx[42]= -0.331596*ct0;
    if(isNANorINF(t->fxx[42])) { PRNT("    @k %d: t->fxx[42] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[42]); return 0; }
    t->fxx[43]= 0.071271999999999988*ct0;
    if(isNANorINF(t->fxx[43])) { PRNT("    @k %d: t->fxx[43] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[43]); return 0; }
    t->fxx[44]= -0.36481599999999997*ct0;
    if(isNANorINF(t->fxx[44])) { PRNT("    @k %d: t->fxx[44] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[44]); return 0; }
    t->fxx[45]= -0.029256000000000001*ct0;
    if(isNANorINF(t->fxx[45])) { PRNT("    @k %d: t->fxx[45] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[45]); return 0; }
    t->fxx[46]= -0.038159999999999999*ct0;
    if(isNANorINF(t->fxx[46])) { PRNT("    @k %d: t->fxx[46] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[46]); return 0; }
    t->fxx[47]= -0.37651200000000001*ct0;
    if(isNANorINF(t->fxx[47])) { PRNT("    @k %d: t->fxx[47] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[47]); return 0; }
    t->fxx[48]= -0.21306000000000003*ct0;
    if(isNANorINF(t->fxx[48])) { PRNT("    @k %d: t->fxx[48] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[48]); return 0; }
    t->fxx[49]= -0.38795999999999997*ct0;
    if(isNANorINF(t->fxx[49])) { PRNT("    @k %d: t->fxx[49] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[49]); return 0; }
    t->fxx[50]= -0.081408000000000008*ct0;
    if(isNANorINF(t->fxx[50])) { PRNT("    @k %d: t->fxx[50] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[50]); return 0; }
    t->fxx[51]= -0.34916400000000003*ct0;
    if(isNANorINF(t->fxx[51])) { PRNT("    @k %d: t->fxx[51] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[51]); return 0; }
    t->fxx[52]= 0.075048000000000004*ct0;
    if(isNANorINF(t->fxx[52])) { PRNT("    @k %d: t->fxx[52] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[52]); return 0; }
    t->fxx[53]= -0.38414399999999999*ct0;
    if(isNANorINF(t->fxx[53])) { PRNT("    @k %d: t->fxx[53] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[53]); return 0; }
    t->fxx[54]= -0.40449600000000002*ct0;
    if(isNANorINF(t->fxx[54])) { PRNT("    @k %d: t->fxx[54] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[54]); return 0; }
    t->fxx[55]= -0.71402499999999991*ct1;
    if(isNANorINF(t->fxx[55])) { PRNT("    @k %d: t->fxx[55] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[55]); return 0; }
    t->fxx[56]= 0.35574499999999998*ct1;
    if(isNANorINF(t->fxx[56])) { PRNT("    @k %d: t->fxx[56] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[56]); return 0; }
    t->fxx[57]= -0.17724099999999998*ct1;
    if(isNANorINF(t->fxx[57])) { PRNT("    @k %d: t->fxx[57] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[57]); return 0; }
    t->fxx[58]= 0.29405999999999999*ct1;
    if(isNANorINF(t->fxx[58])) { PRNT("    @k %d: t->fxx[58] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[58]); return 0; }
    t->fxx[59]= -0.14650799999999997*ct1;
    if(isNANorINF(t->fxx[59])) { PRNT("    @k %d: t->fxx[59] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[59]); return 0; }
    t->fxx[60]= -0.12110399999999999*ct1;
    if(isNANorINF(t->fxx[60])) { PRNT("    @k %d: t->fxx[60] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[60]); return 0; }
    t->fxx[61]= 0.31940999999999997*ct1;
    if(isNANorINF(t->fxx[61])) { PRNT("    @k %d: t->fxx[61] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[61]); return 0; }
    t->fxx[62]= -0.159138*ct1;
    if(isNANorINF(t->fxx[62])) { PRNT("    @k %d: t->fxx[62] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[62]); return 0; }
    t->fxx[63]= -0.13154399999999999*ct1;
    if(isNANorINF(t->fxx[63])) { PRNT("    @k %d: t->fxx[63] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[63]); return 0; }
    t->fxx[64]= -0.14288400000000001*ct1;
    if(isNANorINF(t->fxx[64])) { PRNT("    @k %d: t->fxx[64] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[64]); return 0; }
    t->fxx[65]= -0.060839999999999991*ct1;
    if(isNANorINF(t->fxx[65])) { PRNT("    @k %d: t->fxx[65] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[65]); return 0; }
    t->fxx[66]= 0.030311999999999995*ct1;
    if(isNANorINF(t->fxx[66])) { PRNT("    @k %d: t->fxx[66] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[66]); return 0; }
    t->fxx[67]= 0.025055999999999995*ct1;
    if(isNANorINF(t->fxx[67])) { PRNT("    @k %d: t->fxx[67] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[67]); return 0; }
    t->fxx[68]= 0.027215999999999997*ct1;
    if(isNANorINF(t->fxx[68])) { PRNT("    @k %d: t->fxx[68] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[68]); return 0; }
    t->fxx[69]= -0.0051839999999999994*ct1;
    if(isNANorINF(t->fxx[69])) { PRNT("    @k %d: t->fxx[69] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[69]); return 0; }
    t->fxx[70]= 0.53150500000000001*ct1;
    if(isNANorINF(t->fxx[70])) { PRNT("    @k %d: t->fxx[70] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[70]); return 0; }
    t->fxx[71]= -0.26480900000000002*ct1;
    if(isNANorINF(t->fxx[71])) { PRNT("    @k %d: t->fxx[71] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[71]); return 0; }
    t->fxx[72]= -0.21889199999999998*ct1;
    if(isNANorINF(t->fxx[72])) { PRNT("    @k %d: t->fxx[72] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[72]); return 0; }
    t->fxx[73]= -0.237762*ct1;
    if(isNANorINF(t->fxx[73])) { PRNT("    @k %d: t->fxx[73] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[73]); return 0; }
    t->fxx[74]= 0.045287999999999995*ct1;
    if(isNANorINF(t->fxx[74])) { PRNT("    @k %d: t->fxx[74] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[74]); return 0; }
    t->fxx[75]= -0.39564100000000002*ct1;
    if(isNANorINF(t->fxx[75])) { PRNT("    @k %d: t->fxx[75] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[75]); return 0; }
    t->fxx[76]= -0.43432999999999999*ct1;
    if(isNANorINF(t->fxx[76])) { PRNT("    @k %d: t->fxx[76] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[76]); return 0; }
    t->fxx[77]= 0.216394*ct1;
    if(isNANorINF(t->fxx[77])) { PRNT("    @k %d: t->fxx[77] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[77]); return 0; }
    t->fxx[78]= 0.178872*ct1;
    if(isNANorINF(t->fxx[78])) { PRNT("    @k %d: t->fxx[78] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[78]); return 0; }
    t->fxx[79]= 0.19429199999999999*ct1;
    if(isNANorINF(t->fxx[79])) { PRNT("    @k %d: t->fxx[79] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[79]); return 0; }
    t->fxx[80]= -0.037007999999999999*ct1;
    if(isNANorINF(t->fxx[80])) { PRNT("    @k %d: t->fxx[80] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[80]); return 0; }
    t->fxx[81]= 0.32330599999999998*ct1;
    if(isNANorINF(t->fxx[81])) { PRNT("    @k %d: t->fxx[81] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[81]); return 0; }
    t->fxx[82]= -0.26419599999999999*ct1;
    if(isNANorINF(t->fxx[82])) { PRNT("    @k %d: t->fxx[82] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[82]); return 0; }
    t->fxx[83]= -0.098019999999999996*ct1;
    if(isNANorINF(t->fxx[83])) { PRNT("    @k %d: t->fxx[83] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[83]); return 0; }
    t->fxx[84]= 0.048835999999999997*ct1;
    if(isNANorINF(t->fxx[84])) { PRNT("    @k %d: t->fxx[84] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[84]); return 0; }
    t->fxx[85]= 0.040368000000000001*ct1;
    if(isNANorINF(t->fxx[85])) { PRNT("    @k %d: t->fxx[85] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[85]); return 0; }
    t->fxx[86]= 0.043848000000000005*ct1;
    if(isNANorINF(t->fxx[86])) { PRNT("    @k %d: t->fxx[86] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[86]); return 0; }
    t->fxx[87]= -0.008352*ct1;
    if(isNANorINF(t->fxx[87])) { PRNT("    @k %d: t->fxx[87] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[87]); return 0; }
    t->fxx[88]= 0.072964000000000001*ct1;
    if(isNANorINF(t->fxx[88])) { PRNT("    @k %d: t->fxx[88] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[88]); return 0; }
    t->fxx[89]= -0.059624000000000003*ct1;
    if(isNANorINF(t->fxx[89])) { PRNT("    @k %d: t->fxx[89] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[89]); return 0; }
    t->fxx[90]= -0.013456000000000001*ct1;
    if(isNANorINF(t->fxx[90])) { PRNT("    @k %d: t->fxx[90] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[90]); return 0; }
    t->fxx[91]= 0.164775*ct1;
    if(isNANorINF(t->fxx[91])) { PRNT("    @k %d: t->fxx[91] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[91]); return 0; }
    t->fxx[92]= -0.082095000000000001*ct1;
    if(isNANorINF(t->fxx[92])) { PRNT("    @k %d: t->fxx[92] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[92]); return 0; }
    t->fxx[93]= -0.067860000000000004*ct1;
    if(isNANorINF(t->fxx[93])) { PRNT("    @k %d: t->fxx[93] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[93]); return 0; }
    t->fxx[94]= -0.073709999999999998*ct1;
    if(isNANorINF(t->fxx[94])) { PRNT("    @k %d: t->fxx[94] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[94]); return 0; }
    t->fxx[95]= 0.014039999999999999*ct1;
    if(isNANorINF(t->fxx[95])) { PRNT("    @k %d: t->fxx[95] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[95]); return 0; }
    t->fxx[96]= -0.122655*ct1;
    if(isNANorINF(t->fxx[96])) { PRNT("    @k %d: t->fxx[96] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[96]); return 0; }
    t->fxx[97]= 0.10023*ct1;
    if(isNANorINF(t->fxx[97])) { PRNT("    @k %d: t->fxx[97] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[97]); return 0; }
    t->fxx[98]= 0.022620000000000001*ct1;
    if(isNANorINF(t->fxx[98])) { PRNT("    @k %d: t->fxx[98] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[98]); return 0; }
    t->fxx[99]= -0.038025000000000003*ct1;
    if(isNANorINF(t->fxx[99])) { PRNT("    @k %d: t->fxx[99] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[99]); return 0; }
    t->fxx[100]= -0.73599499999999995*ct1;
    if(isNANorINF(t->fxx[100])) { PRNT("    @k %d: t->fxx[100] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[100]); return 0; }
    t->fxx[101]= 0.36669099999999999*ct1;
    if(isNANorINF(t->fxx[101])) { PRNT("    @k %d: t->fxx[101] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[101]); return 0; }
    t->fxx[102]= 0.30310799999999999*ct1;
    if(isNANorINF(t->fxx[102])) { PRNT("    @k %d: t->fxx[102] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[102]); return 0; }
    t->fxx[103]= 0.32923799999999998*ct1;
    if(isNANorINF(t->fxx[103])) { PRNT("    @k %d: t->fxx[103] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[103]); return 0; }
    t->fxx[104]= -0.06271199999999999*ct1;
    if(isNANorINF(t->fxx[104])) { PRNT("    @k %d: t->fxx[104] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[104]); return 0; }
    t->fxx[105]= 0.54785899999999998*ct1;
    if(isNANorINF(t->fxx[105])) { PRNT("    @k %d: t->fxx[105] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[105]); return 0; }
    t->fxx[106]= -0.44769400000000004*ct1;
    if(isNANorINF(t->fxx[106])) { PRNT("    @k %d: t->fxx[106] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[106]); return 0; }
    t->fxx[107]= -0.101036*ct1;
    if(isNANorINF(t->fxx[107])) { PRNT("    @k %d: t->fxx[107] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[107]); return 0; }
    t->fxx[108]= 0.169845*ct1;
    if(isNANorINF(t->fxx[108])) { PRNT("    @k %d: t->fxx[108] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[108]); return 0; }
    t->fxx[109]= -0.75864100000000001*ct1;
    if(isNANorINF(t->fxx[109])) { PRNT("    @k %d: t->fxx[109] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[109]); return 0; }
    t->fxx[110]= -0.56851600000000002*ct2;
    if(isNANorINF(t->fxx[110])) { PRNT("    @k %d: t->fxx[110] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[110]); return 0; }
    t->fxx[111]= -0.19377800000000001*ct2;
    if(isNANorINF(t->fxx[111])) { PRNT("    @k %d: t->fxx[111] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[111]); return 0; }
    t->fxx[112]= -0.066048999999999997*ct2;
    if(isNANorINF(t->fxx[112])) { PRNT("    @k %d: t->fxx[112] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[112]); return 0; }
    t->fxx[113]= 0.049764000000000003*ct2;
    if(isNANorINF(t->fxx[113])) { PRNT("    @k %d: t->fxx[113] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[113]); return 0; }
    t->fxx[114]= 0.016962000000000001*ct2;
    if(isNANorINF(t->fxx[114])) { PRNT("    @k %d: t->fxx[114] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[114]); return 0; }
    t->fxx[115]= -0.0043560000000000005*ct2;
    if(isNANorINF(t->fxx[115])) { PRNT("    @k %d: t->fxx[115] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[115]); return 0; }
    t->fxx[116]= 0.49914800000000004*ct2;
    if(isNANorINF(t->fxx[116])) { PRNT("    @k %d: t->fxx[116] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[116]); return 0; }
    t->fxx[117]= 0.17013400000000001*ct2;
    if(isNANorINF(t->fxx[117])) { PRNT("    @k %d: t->fxx[117] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[117]); return 0; }
    t->fxx[118]= -0.043692000000000002*ct2;
    if(isNANorINF(t->fxx[118])) { PRNT("    @k %d: t->fxx[118] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[118]); return 0; }
    t->fxx[119]= -0.43824400000000002*ct2;
    if(isNANorINF(t->fxx[119])) { PRNT("    @k %d: t->fxx[119] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[119]); return 0; }
    t->fxx[120]= 0.80979600000000007*ct2;
    if(isNANorINF(t->fxx[120])) { PRNT("    @k %d: t->fxx[120] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[120]); return 0; }
    t->fxx[121]= 0.27601800000000004*ct2;
    if(isNANorINF(t->fxx[121])) { PRNT("    @k %d: t->fxx[121] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[121]); return 0; }
    t->fxx[122]= -0.070884000000000003*ct2;
    if(isNANorINF(t->fxx[122])) { PRNT("    @k %d: t->fxx[122] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[122]); return 0; }
    t->fxx[123]= -0.71098800000000006*ct2;
    if(isNANorINF(t->fxx[123])) { PRNT("    @k %d: t->fxx[123] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[123]); return 0; }
    t->fxx[124]= -1.1534760000000002*ct2;
    if(isNANorINF(t->fxx[124])) { PRNT("    @k %d: t->fxx[124] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[124]); return 0; }
    t->fxx[125]= 0.38680200000000003*ct2;
    if(isNANorINF(t->fxx[125])) { PRNT("    @k %d: t->fxx[125] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[125]); return 0; }
    t->fxx[126]= 0.13184100000000001*ct2;
    if(isNANorINF(t->fxx[126])) { PRNT("    @k %d: t->fxx[126] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[126]); return 0; }
    t->fxx[127]= -0.033857999999999999*ct2;
    if(isNANorINF(t->fxx[127])) { PRNT("    @k %d: t->fxx[127] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[127]); return 0; }
    t->fxx[128]= -0.33960600000000002*ct2;
    if(isNANorINF(t->fxx[128])) { PRNT("    @k %d: t->fxx[128] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[128]); return 0; }
    t->fxx[129]= -0.55096200000000006*ct2;
    if(isNANorINF(t->fxx[129])) { PRNT("    @k %d: t->fxx[129] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[129]); return 0; }
    t->fxx[130]= -0.26316899999999999*ct2;
    if(isNANorINF(t->fxx[130])) { PRNT("    @k %d: t->fxx[130] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[130]); return 0; }
    t->fxx[131]= -0.19377800000000001*ct2;
    if(isNANorINF(t->fxx[131])) { PRNT("    @k %d: t->fxx[131] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[131]); return 0; }
    t->fxx[132]= -0.066048999999999997*ct2;
    if(isNANorINF(t->fxx[132])) { PRNT("    @k %d: t->fxx[132] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[132]); return 0; }
    t->fxx[133]= 0.016962000000000001*ct2;
    if(isNANorINF(t->fxx[133])) { PRNT("    @k %d: t->fxx[133] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[133]); return 0; }
    t->fxx[134]= 0.17013400000000001*ct2;
    if(isNANorINF(t->fxx[134])) { PRNT("    @k %d: t->fxx[134] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[134]); return 0; }
    t->fxx[135]= 0.27601800000000004*ct2;
    if(isNANorINF(t->fxx[135])) { PRNT("    @k %d: t->fxx[135] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[135]); return 0; }
    t->fxx[136]= 0.13184100000000001*ct2;
    if(isNANorINF(t->fxx[136])) { PRNT("    @k %d: t->fxx[136] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[136]); return 0; }
    t->fxx[137]= -0.066048999999999997*ct2;
    if(isNANorINF(t->fxx[137])) { PRNT("    @k %d: t->fxx[137] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[137]); return 0; }
    t->fxx[138]= -0.38001600000000002*ct2;
    if(isNANorINF(t->fxx[138])) { PRNT("    @k %d: t->fxx[138] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[138]); return 0; }
    t->fxx[139]= -0.129528*ct2;
    if(isNANorINF(t->fxx[139])) { PRNT("    @k %d: t->fxx[139] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[139]); return 0; }
    t->fxx[140]= 0.033264000000000002*ct2;
    if(isNANorINF(t->fxx[140])) { PRNT("    @k %d: t->fxx[140] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[140]); return 0; }
    t->fxx[141]= 0.333648*ct2;
    if(isNANorINF(t->fxx[141])) { PRNT("    @k %d: t->fxx[141] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[141]); return 0; }
    t->fxx[142]= 0.541296*ct2;
    if(isNANorINF(t->fxx[142])) { PRNT("    @k %d: t->fxx[142] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[142]); return 0; }
    t->fxx[143]= 0.258552*ct2;
    if(isNANorINF(t->fxx[143])) { PRNT("    @k %d: t->fxx[143] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[143]); return 0; }
    t->fxx[144]= -0.129528*ct2;
    if(isNANorINF(t->fxx[144])) { PRNT("    @k %d: t->fxx[144] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[144]); return 0; }
    t->fxx[145]= -0.25401600000000002*ct2;
    if(isNANorINF(t->fxx[145])) { PRNT("    @k %d: t->fxx[145] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[145]); return 0; }
    t->fxx[146]= -0.26389999999999997*ct2;
    if(isNANorINF(t->fxx[146])) { PRNT("    @k %d: t->fxx[146] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[146]); return 0; }
    t->fxx[147]= -0.089950000000000002*ct2;
    if(isNANorINF(t->fxx[147])) { PRNT("    @k %d: t->fxx[147] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[147]); return 0; }
    t->fxx[148]= 0.023099999999999999*ct2;
    if(isNANorINF(t->fxx[148])) { PRNT("    @k %d: t->fxx[148] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[148]); return 0; }
    t->fxx[149]= 0.23169999999999999*ct2;
    if(isNANorINF(t->fxx[149])) { PRNT("    @k %d: t->fxx[149] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[149]); return 0; }
    t->fxx[150]= 0.37590000000000001*ct2;
    if(isNANorINF(t->fxx[150])) { PRNT("    @k %d: t->fxx[150] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[150]); return 0; }
    t->fxx[151]= 0.17954999999999999*ct2;
    if(isNANorINF(t->fxx[151])) { PRNT("    @k %d: t->fxx[151] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[151]); return 0; }
    t->fxx[152]= -0.089950000000000002*ct2;
    if(isNANorINF(t->fxx[152])) { PRNT("    @k %d: t->fxx[152] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[152]); return 0; }
    t->fxx[153]= -0.1764*ct2;
    if(isNANorINF(t->fxx[153])) { PRNT("    @k %d: t->fxx[153] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[153]); return 0; }
    t->fxx[154]= -0.12249999999999998*ct2;
    if(isNANorINF(t->fxx[154])) { PRNT("    @k %d: t->fxx[154] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[154]); return 0; }
    t->fxx[155]= 0.16738800000000001*ct2;
    if(isNANorINF(t->fxx[155])) { PRNT("    @k %d: t->fxx[155] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[155]); return 0; }
    t->fxx[156]= 0.057054000000000001*ct2;
    if(isNANorINF(t->fxx[156])) { PRNT("    @k %d: t->fxx[156] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[156]); return 0; }
    t->fxx[157]= -0.014652*ct2;
    if(isNANorINF(t->fxx[157])) { PRNT("    @k %d: t->fxx[157] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[157]); return 0; }
    t->fxx[158]= -0.14696400000000001*ct2;
    if(isNANorINF(t->fxx[158])) { PRNT("    @k %d: t->fxx[158] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[158]); return 0; }
    t->fxx[159]= -0.23842800000000003*ct2;
    if(isNANorINF(t->fxx[159])) { PRNT("    @k %d: t->fxx[159] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[159]); return 0; }
    t->fxx[160]= -0.113886*ct2;
    if(isNANorINF(t->fxx[160])) { PRNT("    @k %d: t->fxx[160] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[160]); return 0; }
    t->fxx[161]= 0.057054000000000001*ct2;
    if(isNANorINF(t->fxx[161])) { PRNT("    @k %d: t->fxx[161] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[161]); return 0; }
    t->fxx[162]= 0.111888*ct2;
    if(isNANorINF(t->fxx[162])) { PRNT("    @k %d: t->fxx[162] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[162]); return 0; }
    t->fxx[163]= 0.077699999999999991*ct2;
    if(isNANorINF(t->fxx[163])) { PRNT("    @k %d: t->fxx[163] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[163]); return 0; }
    t->fxx[164]= -0.049284000000000001*ct2;
    if(isNANorINF(t->fxx[164])) { PRNT("    @k %d: t->fxx[164] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[164]); return 0; }
    t->fxx[165]= -0.006241*ct3;
    if(isNANorINF(t->fxx[165])) { PRNT("    @k %d: t->fxx[165] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[165]); return 0; }
    t->fxx[166]= -0.032548000000000001*ct3;
    if(isNANorINF(t->fxx[166])) { PRNT("    @k %d: t->fxx[166] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[166]); return 0; }
    t->fxx[167]= -0.16974399999999998*ct3;
    if(isNANorINF(t->fxx[167])) { PRNT("    @k %d: t->fxx[167] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[167]); return 0; }
    t->fxx[168]= 0.087137000000000006*ct3;
    if(isNANorINF(t->fxx[168])) { PRNT("    @k %d: t->fxx[168] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[168]); return 0; }
    t->fxx[169]= 0.45443599999999995*ct3;
    if(isNANorINF(t->fxx[169])) { PRNT("    @k %d: t->fxx[169] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[169]); return 0; }
    t->fxx[170]= -1.2166090000000001*ct3;
    if(isNANorINF(t->fxx[170])) { PRNT("    @k %d: t->fxx[170] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[170]); return 0; }
    t->fxx[171]= 0.0033180000000000002*ct3;
    if(isNANorINF(t->fxx[171])) { PRNT("    @k %d: t->fxx[171] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[171]); return 0; }
    t->fxx[172]= 0.017304*ct3;
    if(isNANorINF(t->fxx[172])) { PRNT("    @k %d: t->fxx[172] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[172]); return 0; }
    t->fxx[173]= -0.046325999999999999*ct3;
    if(isNANorINF(t->fxx[173])) { PRNT("    @k %d: t->fxx[173] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[173]); return 0; }
    t->fxx[174]= -0.0017640000000000002*ct3;
    if(isNANorINF(t->fxx[174])) { PRNT("    @k %d: t->fxx[174] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[174]); return 0; }
    t->fxx[175]= 0.0039500000000000004*ct3;
    if(isNANorINF(t->fxx[175])) { PRNT("    @k %d: t->fxx[175] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[175]); return 0; }
    t->fxx[176]= 0.0206*ct3;
    if(isNANorINF(t->fxx[176])) { PRNT("    @k %d: t->fxx[176] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[176]); return 0; }
    t->fxx[177]= -0.055150000000000005*ct3;
    if(isNANorINF(t->fxx[177])) { PRNT("    @k %d: t->fxx[177] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[177]); return 0; }
    t->fxx[178]= -0.0021000000000000003*ct3;
    if(isNANorINF(t->fxx[178])) { PRNT("    @k %d: t->fxx[178] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[178]); return 0; }
    t->fxx[179]= -0.0025000000000000005*ct3;
    if(isNANorINF(t->fxx[179])) { PRNT("    @k %d: t->fxx[179] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[179]); return 0; }
    t->fxx[180]= -0.027649999999999997*ct3;
    if(isNANorINF(t->fxx[180])) { PRNT("    @k %d: t->fxx[180] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[180]); return 0; }
    t->fxx[181]= -0.14419999999999999*ct3;
    if(isNANorINF(t->fxx[181])) { PRNT("    @k %d: t->fxx[181] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[181]); return 0; }
    t->fxx[182]= 0.38604999999999995*ct3;
    if(isNANorINF(t->fxx[182])) { PRNT("    @k %d: t->fxx[182] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[182]); return 0; }
    t->fxx[183]= 0.0147*ct3;
    if(isNANorINF(t->fxx[183])) { PRNT("    @k %d: t->fxx[183] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[183]); return 0; }
    t->fxx[184]= 0.017499999999999998*ct3;
    if(isNANorINF(t->fxx[184])) { PRNT("    @k %d: t->fxx[184] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[184]); return 0; }
    t->fxx[185]= -0.12249999999999998*ct3;
    if(isNANorINF(t->fxx[185])) { PRNT("    @k %d: t->fxx[185] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[185]); return 0; }
    t->fxx[186]= -0.045266999999999995*ct3;
    if(isNANorINF(t->fxx[186])) { PRNT("    @k %d: t->fxx[186] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[186]); return 0; }
    t->fxx[187]= -0.23607599999999998*ct3;
    if(isNANorINF(t->fxx[187])) { PRNT("    @k %d: t->fxx[187] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[187]); return 0; }
    t->fxx[188]= 0.63201899999999989*ct3;
    if(isNANorINF(t->fxx[188])) { PRNT("    @k %d: t->fxx[188] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[188]); return 0; }
    t->fxx[189]= 0.024066000000000001*ct3;
    if(isNANorINF(t->fxx[189])) { PRNT("    @k %d: t->fxx[189] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[189]); return 0; }
    t->fxx[190]= 0.028649999999999998*ct3;
    if(isNANorINF(t->fxx[190])) { PRNT("    @k %d: t->fxx[190] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[190]); return 0; }
    t->fxx[191]= -0.20054999999999998*ct3;
    if(isNANorINF(t->fxx[191])) { PRNT("    @k %d: t->fxx[191] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[191]); return 0; }
    t->fxx[192]= -0.32832899999999993*ct3;
    if(isNANorINF(t->fxx[192])) { PRNT("    @k %d: t->fxx[192] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[192]); return 0; }
    t->fxx[193]= 0.019987000000000001*ct3;
    if(isNANorINF(t->fxx[193])) { PRNT("    @k %d: t->fxx[193] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[193]); return 0; }
    t->fxx[194]= 0.104236*ct3;
    if(isNANorINF(t->fxx[194])) { PRNT("    @k %d: t->fxx[194] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[194]); return 0; }
    t->fxx[195]= -0.279059*ct3;
    if(isNANorINF(t->fxx[195])) { PRNT("    @k %d: t->fxx[195] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[195]); return 0; }
    t->fxx[196]= -0.010626*ct3;
    if(isNANorINF(t->fxx[196])) { PRNT("    @k %d: t->fxx[196] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[196]); return 0; }
    t->fxx[197]= -0.012650000000000002*ct3;
    if(isNANorINF(t->fxx[197])) { PRNT("    @k %d: t->fxx[197] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[197]); return 0; }
    t->fxx[198]= 0.08854999999999999*ct3;
    if(isNANorINF(t->fxx[198])) { PRNT("    @k %d: t->fxx[198] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[198]); return 0; }
    t->fxx[199]= 0.14496899999999999*ct3;
    if(isNANorINF(t->fxx[199])) { PRNT("    @k %d: t->fxx[199] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[199]); return 0; }
    t->fxx[200]= -0.064008999999999996*ct3;
    if(isNANorINF(t->fxx[200])) { PRNT("    @k %d: t->fxx[200] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[200]); return 0; }
    t->fxx[201]= 0.092745999999999995*ct3;
    if(isNANorINF(t->fxx[201])) { PRNT("    @k %d: t->fxx[201] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[201]); return 0; }
    t->fxx[202]= 0.48368799999999995*ct3;
    if(isNANorINF(t->fxx[202])) { PRNT("    @k %d: t->fxx[202] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[202]); return 0; }
    t->fxx[203]= -1.2949219999999999*ct3;
    if(isNANorINF(t->fxx[203])) { PRNT("    @k %d: t->fxx[203] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[203]); return 0; }
    t->fxx[204]= -0.049307999999999998*ct3;
    if(isNANorINF(t->fxx[204])) { PRNT("    @k %d: t->fxx[204] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[204]); return 0; }
    t->fxx[205]= -0.058700000000000002*ct3;
    if(isNANorINF(t->fxx[205])) { PRNT("    @k %d: t->fxx[205] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[205]); return 0; }
    t->fxx[206]= 0.41089999999999993*ct3;
    if(isNANorINF(t->fxx[206])) { PRNT("    @k %d: t->fxx[206] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[206]); return 0; }
    t->fxx[207]= 0.67270199999999991*ct3;
    if(isNANorINF(t->fxx[207])) { PRNT("    @k %d: t->fxx[207] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[207]); return 0; }
    t->fxx[208]= -0.29702200000000001*ct3;
    if(isNANorINF(t->fxx[208])) { PRNT("    @k %d: t->fxx[208] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[208]); return 0; }
    t->fxx[209]= -1.3782759999999998*ct3;
    if(isNANorINF(t->fxx[209])) { PRNT("    @k %d: t->fxx[209] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[209]); return 0; }
    t->fxx[210]= -0.042107000000000006*ct3;
    if(isNANorINF(t->fxx[210])) { PRNT("    @k %d: t->fxx[210] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[210]); return 0; }
    t->fxx[211]= -0.21959600000000001*ct3;
    if(isNANorINF(t->fxx[211])) { PRNT("    @k %d: t->fxx[211] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[211]); return 0; }
    t->fxx[212]= 0.58789900000000006*ct3;
    if(isNANorINF(t->fxx[212])) { PRNT("    @k %d: t->fxx[212] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[212]); return 0; }
    t->fxx[213]= 0.022386000000000003*ct3;
    if(isNANorINF(t->fxx[213])) { PRNT("    @k %d: t->fxx[213] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[213]); return 0; }
    t->fxx[214]= 0.026650000000000004*ct3;
    if(isNANorINF(t->fxx[214])) { PRNT("    @k %d: t->fxx[214] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[214]); return 0; }
    t->fxx[215]= -0.18654999999999999*ct3;
    if(isNANorINF(t->fxx[215])) { PRNT("    @k %d: t->fxx[215] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[215]); return 0; }
    t->fxx[216]= -0.30540899999999999*ct3;
    if(isNANorINF(t->fxx[216])) { PRNT("    @k %d: t->fxx[216] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[216]); return 0; }
    t->fxx[217]= 0.134849*ct3;
    if(isNANorINF(t->fxx[217])) { PRNT("    @k %d: t->fxx[217] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[217]); return 0; }
    t->fxx[218]= 0.62574200000000002*ct3;
    if(isNANorINF(t->fxx[218])) { PRNT("    @k %d: t->fxx[218] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[218]); return 0; }
    t->fxx[219]= -0.28408900000000004*ct3;
    if(isNANorINF(t->fxx[219])) { PRNT("    @k %d: t->fxx[219] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[219]); return 0; }
    t->fxx[220]= -0.00036099999999999999*ct4;
    if(isNANorINF(t->fxx[220])) { PRNT("    @k %d: t->fxx[220] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[220]); return 0; }
    t->fxx[221]= 0.0081700000000000002*ct4;
    if(isNANorINF(t->fxx[221])) { PRNT("    @k %d: t->fxx[221] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[221]); return 0; }
    t->fxx[222]= -0.18489999999999998*ct4;
    if(isNANorINF(t->fxx[222])) { PRNT("    @k %d: t->fxx[222] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[222]); return 0; }
    t->fxx[223]= -0.011304999999999999*ct4;
    if(isNANorINF(t->fxx[223])) { PRNT("    @k %d: t->fxx[223] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[223]); return 0; }
    t->fxx[224]= 0.25584999999999997*ct4;
    if(isNANorINF(t->fxx[224])) { PRNT("    @k %d: t->fxx[224] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[224]); return 0; }
    t->fxx[225]= -0.35402499999999998*ct4;
    if(isNANorINF(t->fxx[225])) { PRNT("    @k %d: t->fxx[225] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[225]); return 0; }
    t->fxx[226]= -0.0070299999999999998*ct4;
    if(isNANorINF(t->fxx[226])) { PRNT("    @k %d: t->fxx[226] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[226]); return 0; }
    t->fxx[227]= 0.15909999999999999*ct4;
    if(isNANorINF(t->fxx[227])) { PRNT("    @k %d: t->fxx[227] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[227]); return 0; }
    t->fxx[228]= -0.22014999999999998*ct4;
    if(isNANorINF(t->fxx[228])) { PRNT("    @k %d: t->fxx[228] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[228]); return 0; }
    t->fxx[229]= -0.13689999999999999*ct4;
    if(isNANorINF(t->fxx[229])) { PRNT("    @k %d: t->fxx[229] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[229]); return 0; }
    t->fxx[230]= 0.022704999999999999*ct4;
    if(isNANorINF(t->fxx[230])) { PRNT("    @k %d: t->fxx[230] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[230]); return 0; }
    t->fxx[231]= -0.51385000000000003*ct4;
    if(isNANorINF(t->fxx[231])) { PRNT("    @k %d: t->fxx[231] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[231]); return 0; }
    t->fxx[232]= 0.71102500000000002*ct4;
    if(isNANorINF(t->fxx[232])) { PRNT("    @k %d: t->fxx[232] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[232]); return 0; }
    t->fxx[233]= 0.44215000000000004*ct4;
    if(isNANorINF(t->fxx[233])) { PRNT("    @k %d: t->fxx[233] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[233]); return 0; }
    t->fxx[234]= -1.4280250000000001*ct4;
    if(isNANorINF(t->fxx[234])) { PRNT("    @k %d: t->fxx[234] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[234]); return 0; }
    t->fxx[235]= 0.0066309999999999997*ct4;
    if(isNANorINF(t->fxx[235])) { PRNT("    @k %d: t->fxx[235] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[235]); return 0; }
    t->fxx[236]= -0.15006999999999998*ct4;
    if(isNANorINF(t->fxx[236])) { PRNT("    @k %d: t->fxx[236] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[236]); return 0; }
    t->fxx[237]= 0.20765499999999998*ct4;
    if(isNANorINF(t->fxx[237])) { PRNT("    @k %d: t->fxx[237] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[237]); return 0; }
    t->fxx[238]= 0.12912999999999999*ct4;
    if(isNANorINF(t->fxx[238])) { PRNT("    @k %d: t->fxx[238] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[238]); return 0; }
    t->fxx[239]= -0.41705500000000001*ct4;
    if(isNANorINF(t->fxx[239])) { PRNT("    @k %d: t->fxx[239] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[239]); return 0; }
    t->fxx[240]= -0.12180099999999998*ct4;
    if(isNANorINF(t->fxx[240])) { PRNT("    @k %d: t->fxx[240] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[240]); return 0; }
    t->fxx[241]= -0.0061180000000000002*ct4;
    if(isNANorINF(t->fxx[241])) { PRNT("    @k %d: t->fxx[241] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[241]); return 0; }
    t->fxx[242]= 0.13846*ct4;
    if(isNANorINF(t->fxx[242])) { PRNT("    @k %d: t->fxx[242] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[242]); return 0; }
    t->fxx[243]= -0.19159000000000001*ct4;
    if(isNANorINF(t->fxx[243])) { PRNT("    @k %d: t->fxx[243] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[243]); return 0; }
    t->fxx[244]= -0.11914*ct4;
    if(isNANorINF(t->fxx[244])) { PRNT("    @k %d: t->fxx[244] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[244]); return 0; }
    t->fxx[245]= 0.38479000000000002*ct4;
    if(isNANorINF(t->fxx[245])) { PRNT("    @k %d: t->fxx[245] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[245]); return 0; }
    t->fxx[246]= 0.11237799999999999*ct4;
    if(isNANorINF(t->fxx[246])) { PRNT("    @k %d: t->fxx[246] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[246]); return 0; }
    t->fxx[247]= -0.10368400000000001*ct4;
    if(isNANorINF(t->fxx[247])) { PRNT("    @k %d: t->fxx[247] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[247]); return 0; }
    t->fxx[248]= 0.00017099999999999998*ct4;
    if(isNANorINF(t->fxx[248])) { PRNT("    @k %d: t->fxx[248] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[248]); return 0; }
    t->fxx[249]= -0.0038699999999999997*ct4;
    if(isNANorINF(t->fxx[249])) { PRNT("    @k %d: t->fxx[249] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[249]); return 0; }
    t->fxx[250]= 0.0053549999999999995*ct4;
    if(isNANorINF(t->fxx[250])) { PRNT("    @k %d: t->fxx[250] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[250]); return 0; }
    t->fxx[251]= 0.0033299999999999996*ct4;
    if(isNANorINF(t->fxx[251])) { PRNT("    @k %d: t->fxx[251] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[251]); return 0; }
    t->fxx[252]= -0.010754999999999999*ct4;
    if(isNANorINF(t->fxx[252])) { PRNT("    @k %d: t->fxx[252] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[252]); return 0; }
    t->fxx[253]= -0.0031409999999999997*ct4;
    if(isNANorINF(t->fxx[253])) { PRNT("    @k %d: t->fxx[253] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[253]); return 0; }
    t->fxx[254]= 0.002898*ct4;
    if(isNANorINF(t->fxx[254])) { PRNT("    @k %d: t->fxx[254] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[254]); return 0; }
    t->fxx[255]= -8.099999999999999e-5*ct4;
    if(isNANorINF(t->fxx[255])) { PRNT("    @k %d: t->fxx[255] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[255]); return 0; }
    t->fxx[256]= 0.0068779999999999996*ct4;
    if(isNANorINF(t->fxx[256])) { PRNT("    @k %d: t->fxx[256] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[256]); return 0; }
    t->fxx[257]= -0.15565999999999999*ct4;
    if(isNANorINF(t->fxx[257])) { PRNT("    @k %d: t->fxx[257] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[257]); return 0; }
    t->fxx[258]= 0.21538999999999997*ct4;
    if(isNANorINF(t->fxx[258])) { PRNT("    @k %d: t->fxx[258] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[258]); return 0; }
    t->fxx[259]= 0.13394*ct4;
    if(isNANorINF(t->fxx[259])) { PRNT("    @k %d: t->fxx[259] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[259]); return 0; }
    t->fxx[260]= -0.43259000000000003*ct4;
    if(isNANorINF(t->fxx[260])) { PRNT("    @k %d: t->fxx[260] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[260]); return 0; }
    t->fxx[261]= -0.12633799999999998*ct4;
    if(isNANorINF(t->fxx[261])) { PRNT("    @k %d: t->fxx[261] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[261]); return 0; }
    t->fxx[262]= 0.116564*ct4;
    if(isNANorINF(t->fxx[262])) { PRNT("    @k %d: t->fxx[262] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[262]); return 0; }
    t->fxx[263]= -0.0032579999999999996*ct4;
    if(isNANorINF(t->fxx[263])) { PRNT("    @k %d: t->fxx[263] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[263]); return 0; }
    t->fxx[264]= -0.13104399999999999*ct4;
    if(isNANorINF(t->fxx[264])) { PRNT("    @k %d: t->fxx[264] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[264]); return 0; }
    t->fxx[265]= -0.001596*ct4;
    if(isNANorINF(t->fxx[265])) { PRNT("    @k %d: t->fxx[265] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[265]); return 0; }
    t->fxx[266]= 0.036119999999999999*ct4;
    if(isNANorINF(t->fxx[266])) { PRNT("    @k %d: t->fxx[266] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[266]); return 0; }
    t->fxx[267]= -0.049980000000000004*ct4;
    if(isNANorINF(t->fxx[267])) { PRNT("    @k %d: t->fxx[267] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[267]); return 0; }
    t->fxx[268]= -0.03108*ct4;
    if(isNANorINF(t->fxx[268])) { PRNT("    @k %d: t->fxx[268] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[268]); return 0; }
    t->fxx[269]= 0.10038000000000001*ct4;
    if(isNANorINF(t->fxx[269])) { PRNT("    @k %d: t->fxx[269] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[269]); return 0; }
    t->fxx[270]= 0.029315999999999998*ct4;
    if(isNANorINF(t->fxx[270])) { PRNT("    @k %d: t->fxx[270] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[270]); return 0; }
    t->fxx[271]= -0.027048000000000003*ct4;
    if(isNANorINF(t->fxx[271])) { PRNT("    @k %d: t->fxx[271] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[271]); return 0; }
    t->fxx[272]= 0.00075599999999999994*ct4;
    if(isNANorINF(t->fxx[272])) { PRNT("    @k %d: t->fxx[272] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[272]); return 0; }
    t->fxx[273]= 0.030408000000000001*ct4;
    if(isNANorINF(t->fxx[273])) { PRNT("    @k %d: t->fxx[273] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[273]); return 0; }
    t->fxx[274]= -0.0070560000000000006*ct4;
    if(isNANorINF(t->fxx[274])) { PRNT("    @k %d: t->fxx[274] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[274]); return 0; }
    t->fxx[275]= -0.085848999999999995*ct5;
    if(isNANorINF(t->fxx[275])) { PRNT("    @k %d: t->fxx[275] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[275]); return 0; }
    t->fxx[276]= -0.064460000000000003*ct5;
    if(isNANorINF(t->fxx[276])) { PRNT("    @k %d: t->fxx[276] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[276]); return 0; }
    t->fxx[277]= -0.048399999999999999*ct5;
    if(isNANorINF(t->fxx[277])) { PRNT("    @k %d: t->fxx[277] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[277]); return 0; }
    t->fxx[278]= -0.18459*ct5;
    if(isNANorINF(t->fxx[278])) { PRNT("    @k %d: t->fxx[278] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[278]); return 0; }
    t->fxx[279]= -0.1386*ct5;
    if(isNANorINF(t->fxx[279])) { PRNT("    @k %d: t->fxx[279] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[279]); return 0; }
    t->fxx[280]= -0.39690000000000003*ct5;
    if(isNANorINF(t->fxx[280])) { PRNT("    @k %d: t->fxx[280] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[280]); return 0; }
    t->fxx[281]= -0.15704799999999999*ct5;
    if(isNANorINF(t->fxx[281])) { PRNT("    @k %d: t->fxx[281] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[281]); return 0; }
    t->fxx[282]= -0.11792000000000001*ct5;
    if(isNANorINF(t->fxx[282])) { PRNT("    @k %d: t->fxx[282] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[282]); return 0; }
    t->fxx[283]= -0.33768000000000004*ct5;
    if(isNANorINF(t->fxx[283])) { PRNT("    @k %d: t->fxx[283] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[283]); return 0; }
    t->fxx[284]= -0.28729600000000005*ct5;
    if(isNANorINF(t->fxx[284])) { PRNT("    @k %d: t->fxx[284] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[284]); return 0; }
    t->fxx[285]= 0.00058599999999999993*ct5;
    if(isNANorINF(t->fxx[285])) { PRNT("    @k %d: t->fxx[285] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[285]); return 0; }
    t->fxx[286]= 0.00044000000000000002*ct5;
    if(isNANorINF(t->fxx[286])) { PRNT("    @k %d: t->fxx[286] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[286]); return 0; }
    t->fxx[287]= 0.0012600000000000001*ct5;
    if(isNANorINF(t->fxx[287])) { PRNT("    @k %d: t->fxx[287] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[287]); return 0; }
    t->fxx[288]= 0.001072*ct5;
    if(isNANorINF(t->fxx[288])) { PRNT("    @k %d: t->fxx[288] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[288]); return 0; }
    t->fxx[289]= -3.9999999999999998e-6*ct5;
    if(isNANorINF(t->fxx[289])) { PRNT("    @k %d: t->fxx[289] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[289]); return 0; }
    t->fxx[290]= -0.10489399999999999*ct5;
    if(isNANorINF(t->fxx[290])) { PRNT("    @k %d: t->fxx[290] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[290]); return 0; }
    t->fxx[291]= -0.078759999999999997*ct5;
    if(isNANorINF(t->fxx[291])) { PRNT("    @k %d: t->fxx[291] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[291]); return 0; }
    t->fxx[292]= -0.22553999999999999*ct5;
    if(isNANorINF(t->fxx[292])) { PRNT("    @k %d: t->fxx[292] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[292]); return 0; }
    t->fxx[293]= -0.191888*ct5;
    if(isNANorINF(t->fxx[293])) { PRNT("    @k %d: t->fxx[293] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[293]); return 0; }
    t->fxx[294]= 0.00071599999999999995*ct5;
    if(isNANorINF(t->fxx[294])) { PRNT("    @k %d: t->fxx[294] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[294]); return 0; }
    t->fxx[295]= -0.128164*ct5;
    if(isNANorINF(t->fxx[295])) { PRNT("    @k %d: t->fxx[295] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[295]); return 0; }
    t->fxx[296]= 0.013770999999999999*ct5;
    if(isNANorINF(t->fxx[296])) { PRNT("    @k %d: t->fxx[296] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[296]); return 0; }
    t->fxx[297]= 0.01034*ct5;
    if(isNANorINF(t->fxx[297])) { PRNT("    @k %d: t->fxx[297] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[297]); return 0; }
    t->fxx[298]= 0.029610000000000001*ct5;
    if(isNANorINF(t->fxx[298])) { PRNT("    @k %d: t->fxx[298] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[298]); return 0; }
    t->fxx[299]= 0.025192000000000003*ct5;
    if(isNANorINF(t->fxx[299])) { PRNT("    @k %d: t->fxx[299] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[299]); return 0; }
    t->fxx[300]= -9.4000000000000008e-5*ct5;
    if(isNANorINF(t->fxx[300])) { PRNT("    @k %d: t->fxx[300] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[300]); return 0; }
    t->fxx[301]= 0.016826000000000001*ct5;
    if(isNANorINF(t->fxx[301])) { PRNT("    @k %d: t->fxx[301] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[301]); return 0; }
    t->fxx[302]= -0.002209*ct5;
    if(isNANorINF(t->fxx[302])) { PRNT("    @k %d: t->fxx[302] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[302]); return 0; }
    t->fxx[303]= 0.098155000000000006*ct5;
    if(isNANorINF(t->fxx[303])) { PRNT("    @k %d: t->fxx[303] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[303]); return 0; }
    t->fxx[304]= 0.073700000000000002*ct5;
    if(isNANorINF(t->fxx[304])) { PRNT("    @k %d: t->fxx[304] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[304]); return 0; }
    t->fxx[305]= 0.21105000000000002*ct5;
    if(isNANorINF(t->fxx[305])) { PRNT("    @k %d: t->fxx[305] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[305]); return 0; }
    t->fxx[306]= 0.17956000000000003*ct5;
    if(isNANorINF(t->fxx[306])) { PRNT("    @k %d: t->fxx[306] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[306]); return 0; }
    t->fxx[307]= -0.00067000000000000002*ct5;
    if(isNANorINF(t->fxx[307])) { PRNT("    @k %d: t->fxx[307] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[307]); return 0; }
    t->fxx[308]= 0.11993000000000001*ct5;
    if(isNANorINF(t->fxx[308])) { PRNT("    @k %d: t->fxx[308] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[308]); return 0; }
    t->fxx[309]= -0.015745000000000002*ct5;
    if(isNANorINF(t->fxx[309])) { PRNT("    @k %d: t->fxx[309] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[309]); return 0; }
    t->fxx[310]= -0.11222500000000002*ct5;
    if(isNANorINF(t->fxx[310])) { PRNT("    @k %d: t->fxx[310] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[310]); return 0; }
    t->fxx[311]= -0.36478500000000003*ct5;
    if(isNANorINF(t->fxx[311])) { PRNT("    @k %d: t->fxx[311] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[311]); return 0; }
    t->fxx[312]= -0.27390000000000003*ct5;
    if(isNANorINF(t->fxx[312])) { PRNT("    @k %d: t->fxx[312] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[312]); return 0; }
    t->fxx[313]= -0.7843500000000001*ct5;
    if(isNANorINF(t->fxx[313])) { PRNT("    @k %d: t->fxx[313] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[313]); return 0; }
    t->fxx[314]= -0.66732000000000014*ct5;
    if(isNANorINF(t->fxx[314])) { PRNT("    @k %d: t->fxx[314] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[314]); return 0; }
    t->fxx[315]= 0.0024900000000000005*ct5;
    if(isNANorINF(t->fxx[315])) { PRNT("    @k %d: t->fxx[315] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[315]); return 0; }
    t->fxx[316]= -0.44571*ct5;
    if(isNANorINF(t->fxx[316])) { PRNT("    @k %d: t->fxx[316] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[316]); return 0; }
    t->fxx[317]= 0.058515000000000005*ct5;
    if(isNANorINF(t->fxx[317])) { PRNT("    @k %d: t->fxx[317] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[317]); return 0; }
    t->fxx[318]= 0.41707500000000008*ct5;
    if(isNANorINF(t->fxx[318])) { PRNT("    @k %d: t->fxx[318] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[318]); return 0; }
    t->fxx[319]= -1.5500250000000002*ct5;
    if(isNANorINF(t->fxx[319])) { PRNT("    @k %d: t->fxx[319] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[319]); return 0; }
    t->fxx[320]= -0.012598999999999999*ct5;
    if(isNANorINF(t->fxx[320])) { PRNT("    @k %d: t->fxx[320] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[320]); return 0; }
    t->fxx[321]= -0.0094599999999999997*ct5;
    if(isNANorINF(t->fxx[321])) { PRNT("    @k %d: t->fxx[321] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[321]); return 0; }
    t->fxx[322]= -0.027089999999999999*ct5;
    if(isNANorINF(t->fxx[322])) { PRNT("    @k %d: t->fxx[322] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[322]); return 0; }
    t->fxx[323]= -0.023047999999999999*ct5;
    if(isNANorINF(t->fxx[323])) { PRNT("    @k %d: t->fxx[323] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[323]); return 0; }
    t->fxx[324]= 8.599999999999999e-5*ct5;
    if(isNANorINF(t->fxx[324])) { PRNT("    @k %d: t->fxx[324] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[324]); return 0; }
    t->fxx[325]= -0.015393999999999998*ct5;
    if(isNANorINF(t->fxx[325])) { PRNT("    @k %d: t->fxx[325] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[325]); return 0; }
    t->fxx[326]= 0.0020209999999999998*ct5;
    if(isNANorINF(t->fxx[326])) { PRNT("    @k %d: t->fxx[326] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[326]); return 0; }
    t->fxx[327]= 0.014404999999999999*ct5;
    if(isNANorINF(t->fxx[327])) { PRNT("    @k %d: t->fxx[327] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[327]); return 0; }
    t->fxx[328]= -0.053534999999999999*ct5;
    if(isNANorINF(t->fxx[328])) { PRNT("    @k %d: t->fxx[328] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[328]); return 0; }
    t->fxx[329]= -0.0018489999999999997*ct5;
    if(isNANorINF(t->fxx[329])) { PRNT("    @k %d: t->fxx[329] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[329]); return 0; }
    t->fxx[330]= -0.45292900000000008*ct6;
    if(isNANorINF(t->fxx[330])) { PRNT("    @k %d: t->fxx[330] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[330]); return 0; }
    t->fxx[331]= 0.097585000000000005*ct6;
    if(isNANorINF(t->fxx[331])) { PRNT("    @k %d: t->fxx[331] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[331]); return 0; }
    t->fxx[332]= -0.021024999999999999*ct6;
    if(isNANorINF(t->fxx[332])) { PRNT("    @k %d: t->fxx[332] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[332]); return 0; }
    t->fxx[333]= 0.22343600000000002*ct6;
    if(isNANorINF(t->fxx[333])) { PRNT("    @k %d: t->fxx[333] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[333]); return 0; }
    t->fxx[334]= -0.048140000000000002*ct6;
    if(isNANorINF(t->fxx[334])) { PRNT("    @k %d: t->fxx[334] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[334]); return 0; }
    t->fxx[335]= -0.11022400000000002*ct6;
    if(isNANorINF(t->fxx[335])) { PRNT("    @k %d: t->fxx[335] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[335]); return 0; }
    t->fxx[336]= 0.68578700000000004*ct6;
    if(isNANorINF(t->fxx[336])) { PRNT("    @k %d: t->fxx[336] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[336]); return 0; }
    t->fxx[337]= -0.14775499999999997*ct6;
    if(isNANorINF(t->fxx[337])) { PRNT("    @k %d: t->fxx[337] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[337]); return 0; }
    t->fxx[338]= -0.338308*ct6;
    if(isNANorINF(t->fxx[338])) { PRNT("    @k %d: t->fxx[338] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[338]); return 0; }
    t->fxx[339]= -1.0383609999999999*ct6;
    if(isNANorINF(t->fxx[339])) { PRNT("    @k %d: t->fxx[339] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[339]); return 0; }
    t->fxx[340]= -0.53234300000000001*ct6;
    if(isNANorINF(t->fxx[340])) { PRNT("    @k %d: t->fxx[340] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[340]); return 0; }
    t->fxx[341]= 0.11469499999999999*ct6;
    if(isNANorINF(t->fxx[341])) { PRNT("    @k %d: t->fxx[341] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[341]); return 0; }
    t->fxx[342]= 0.26261200000000001*ct6;
    if(isNANorINF(t->fxx[342])) { PRNT("    @k %d: t->fxx[342] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[342]); return 0; }
    t->fxx[343]= 0.806029*ct6;
    if(isNANorINF(t->fxx[343])) { PRNT("    @k %d: t->fxx[343] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[343]); return 0; }
    t->fxx[344]= -0.62568100000000004*ct6;
    if(isNANorINF(t->fxx[344])) { PRNT("    @k %d: t->fxx[344] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[344]); return 0; }
    t->fxx[345]= 0.37620700000000007*ct6;
    if(isNANorINF(t->fxx[345])) { PRNT("    @k %d: t->fxx[345] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[345]); return 0; }
    t->fxx[346]= -0.081055000000000002*ct6;
    if(isNANorINF(t->fxx[346])) { PRNT("    @k %d: t->fxx[346] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[346]); return 0; }
    t->fxx[347]= -0.18558800000000003*ct6;
    if(isNANorINF(t->fxx[347])) { PRNT("    @k %d: t->fxx[347] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[347]); return 0; }
    t->fxx[348]= -0.56962100000000004*ct6;
    if(isNANorINF(t->fxx[348])) { PRNT("    @k %d: t->fxx[348] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[348]); return 0; }
    t->fxx[349]= 0.44216900000000003*ct6;
    if(isNANorINF(t->fxx[349])) { PRNT("    @k %d: t->fxx[349] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[349]); return 0; }
    t->fxx[350]= -0.31248100000000006*ct6;
    if(isNANorINF(t->fxx[350])) { PRNT("    @k %d: t->fxx[350] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[350]); return 0; }
    t->fxx[351]= 0.37015000000000003*ct6;
    if(isNANorINF(t->fxx[351])) { PRNT("    @k %d: t->fxx[351] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[351]); return 0; }
    t->fxx[352]= -0.079750000000000001*ct6;
    if(isNANorINF(t->fxx[352])) { PRNT("    @k %d: t->fxx[352] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[352]); return 0; }
    t->fxx[353]= -0.18260000000000001*ct6;
    if(isNANorINF(t->fxx[353])) { PRNT("    @k %d: t->fxx[353] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[353]); return 0; }
    t->fxx[354]= -0.56045*ct6;
    if(isNANorINF(t->fxx[354])) { PRNT("    @k %d: t->fxx[354] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[354]); return 0; }
    t->fxx[355]= 0.43505000000000005*ct6;
    if(isNANorINF(t->fxx[355])) { PRNT("    @k %d: t->fxx[355] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[355]); return 0; }
    t->fxx[356]= -0.30745000000000006*ct6;
    if(isNANorINF(t->fxx[356])) { PRNT("    @k %d: t->fxx[356] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[356]); return 0; }
    t->fxx[357]= -0.30250000000000005*ct6;
    if(isNANorINF(t->fxx[357])) { PRNT("    @k %d: t->fxx[357] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[357]); return 0; }
    t->fxx[358]= -0.471773*ct6;
    if(isNANorINF(t->fxx[358])) { PRNT("    @k %d: t->fxx[358] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[358]); return 0; }
    t->fxx[359]= 0.10164499999999999*ct6;
    if(isNANorINF(t->fxx[359])) { PRNT("    @k %d: t->fxx[359] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[359]); return 0; }
    t->fxx[360]= 0.23273199999999999*ct6;
    if(isNANorINF(t->fxx[360])) { PRNT("    @k %d: t->fxx[360] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[360]); return 0; }
    t->fxx[361]= 0.71431899999999993*ct6;
    if(isNANorINF(t->fxx[361])) { PRNT("    @k %d: t->fxx[361] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[361]); return 0; }
    t->fxx[362]= -0.55449099999999996*ct6;
    if(isNANorINF(t->fxx[362])) { PRNT("    @k %d: t->fxx[362] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[362]); return 0; }
    t->fxx[363]= 0.39185900000000001*ct6;
    if(isNANorINF(t->fxx[363])) { PRNT("    @k %d: t->fxx[363] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[363]); return 0; }
    t->fxx[364]= 0.38555*ct6;
    if(isNANorINF(t->fxx[364])) { PRNT("    @k %d: t->fxx[364] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[364]); return 0; }
    t->fxx[365]= -0.49140099999999992*ct6;
    if(isNANorINF(t->fxx[365])) { PRNT("    @k %d: t->fxx[365] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[365]); return 0; }
    t->fxx[366]= -0.26785400000000004*ct6;
    if(isNANorINF(t->fxx[366])) { PRNT("    @k %d: t->fxx[366] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[366]); return 0; }
    t->fxx[367]= 0.057709999999999997*ct6;
    if(isNANorINF(t->fxx[367])) { PRNT("    @k %d: t->fxx[367] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[367]); return 0; }
    t->fxx[368]= 0.132136*ct6;
    if(isNANorINF(t->fxx[368])) { PRNT("    @k %d: t->fxx[368] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[368]); return 0; }
    t->fxx[369]= 0.40556199999999998*ct6;
    if(isNANorINF(t->fxx[369])) { PRNT("    @k %d: t->fxx[369] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[369]); return 0; }
    t->fxx[370]= -0.31481800000000004*ct6;
    if(isNANorINF(t->fxx[370])) { PRNT("    @k %d: t->fxx[370] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[370]); return 0; }
    t->fxx[371]= 0.22248200000000004*ct6;
    if(isNANorINF(t->fxx[371])) { PRNT("    @k %d: t->fxx[371] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[371]); return 0; }
    t->fxx[372]= 0.21890000000000004*ct6;
    if(isNANorINF(t->fxx[372])) { PRNT("    @k %d: t->fxx[372] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[372]); return 0; }
    t->fxx[373]= -0.27899800000000002*ct6;
    if(isNANorINF(t->fxx[373])) { PRNT("    @k %d: t->fxx[373] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[373]); return 0; }
    t->fxx[374]= -0.15840400000000002*ct6;
    if(isNANorINF(t->fxx[374])) { PRNT("    @k %d: t->fxx[374] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[374]); return 0; }
    t->fxx[375]= -0.14806*ct6;
    if(isNANorINF(t->fxx[375])) { PRNT("    @k %d: t->fxx[375] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[375]); return 0; }
    t->fxx[376]= 0.031899999999999998*ct6;
    if(isNANorINF(t->fxx[376])) { PRNT("    @k %d: t->fxx[376] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[376]); return 0; }
    t->fxx[377]= 0.073040000000000008*ct6;
    if(isNANorINF(t->fxx[377])) { PRNT("    @k %d: t->fxx[377] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[377]); return 0; }
    t->fxx[378]= 0.22417999999999999*ct6;
    if(isNANorINF(t->fxx[378])) { PRNT("    @k %d: t->fxx[378] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[378]); return 0; }
    t->fxx[379]= -0.17402000000000001*ct6;
    if(isNANorINF(t->fxx[379])) { PRNT("    @k %d: t->fxx[379] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[379]); return 0; }
    t->fxx[380]= 0.12298000000000001*ct6;
    if(isNANorINF(t->fxx[380])) { PRNT("    @k %d: t->fxx[380] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[380]); return 0; }
    t->fxx[381]= 0.12100000000000001*ct6;
    if(isNANorINF(t->fxx[381])) { PRNT("    @k %d: t->fxx[381] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[381]); return 0; }
    t->fxx[382]= -0.15422*ct6;
    if(isNANorINF(t->fxx[382])) { PRNT("    @k %d: t->fxx[382] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[382]); return 0; }
    t->fxx[383]= -0.087559999999999999*ct6;
    if(isNANorINF(t->fxx[383])) { PRNT("    @k %d: t->fxx[383] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[383]); return 0; }
    t->fxx[384]= -0.048399999999999999*ct6;
    if(isNANorINF(t->fxx[384])) { PRNT("    @k %d: t->fxx[384] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[384]); return 0; }
    t->fxx[385]= -0.29811600000000005*ct7;
    if(isNANorINF(t->fxx[385])) { PRNT("    @k %d: t->fxx[385] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[385]); return 0; }
    t->fxx[386]= 0.047502000000000003*ct7;
    if(isNANorINF(t->fxx[386])) { PRNT("    @k %d: t->fxx[386] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[386]); return 0; }
    t->fxx[387]= -0.0075689999999999993*ct7;
    if(isNANorINF(t->fxx[387])) { PRNT("    @k %d: t->fxx[387] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[387]); return 0; }
    t->fxx[388]= -0.12121200000000001*ct7;
    if(isNANorINF(t->fxx[388])) { PRNT("    @k %d: t->fxx[388] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[388]); return 0; }
    t->fxx[389]= 0.019313999999999998*ct7;
    if(isNANorINF(t->fxx[389])) { PRNT("    @k %d: t->fxx[389] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[389]); return 0; }
    t->fxx[390]= -0.049284000000000001*ct7;
    if(isNANorINF(t->fxx[390])) { PRNT("    @k %d: t->fxx[390] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[390]); return 0; }
    t->fxx[391]= -0.28555800000000003*ct7;
    if(isNANorINF(t->fxx[391])) { PRNT("    @k %d: t->fxx[391] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[391]); return 0; }
    t->fxx[392]= 0.045501*ct7;
    if(isNANorINF(t->fxx[392])) { PRNT("    @k %d: t->fxx[392] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[392]); return 0; }
    t->fxx[393]= -0.116106*ct7;
    if(isNANorINF(t->fxx[393])) { PRNT("    @k %d: t->fxx[393] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[393]); return 0; }
    t->fxx[394]= -0.27352900000000002*ct7;
    if(isNANorINF(t->fxx[394])) { PRNT("    @k %d: t->fxx[394] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[394]); return 0; }
    t->fxx[395]= -0.11793600000000001*ct7;
    if(isNANorINF(t->fxx[395])) { PRNT("    @k %d: t->fxx[395] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[395]); return 0; }
    t->fxx[396]= 0.018792*ct7;
    if(isNANorINF(t->fxx[396])) { PRNT("    @k %d: t->fxx[396] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[396]); return 0; }
    t->fxx[397]= -0.047952000000000002*ct7;
    if(isNANorINF(t->fxx[397])) { PRNT("    @k %d: t->fxx[397] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[397]); return 0; }
    t->fxx[398]= -0.112968*ct7;
    if(isNANorINF(t->fxx[398])) { PRNT("    @k %d: t->fxx[398] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[398]); return 0; }
    t->fxx[399]= -0.046655999999999996*ct7;
    if(isNANorINF(t->fxx[399])) { PRNT("    @k %d: t->fxx[399] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[399]); return 0; }
    t->fxx[400]= 0.093911999999999995*ct7;
    if(isNANorINF(t->fxx[400])) { PRNT("    @k %d: t->fxx[400] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[400]); return 0; }
    t->fxx[401]= -0.014963999999999998*ct7;
    if(isNANorINF(t->fxx[401])) { PRNT("    @k %d: t->fxx[401] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[401]); return 0; }
    t->fxx[402]= 0.038183999999999996*ct7;
    if(isNANorINF(t->fxx[402])) { PRNT("    @k %d: t->fxx[402] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[402]); return 0; }
    t->fxx[403]= 0.089955999999999994*ct7;
    if(isNANorINF(t->fxx[403])) { PRNT("    @k %d: t->fxx[403] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[403]); return 0; }
    t->fxx[404]= 0.037151999999999998*ct7;
    if(isNANorINF(t->fxx[404])) { PRNT("    @k %d: t->fxx[404] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[404]); return 0; }
    t->fxx[405]= -0.029583999999999996*ct7;
    if(isNANorINF(t->fxx[405])) { PRNT("    @k %d: t->fxx[405] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[405]); return 0; }
    t->fxx[406]= -0.24952200000000002*ct7;
    if(isNANorINF(t->fxx[406])) { PRNT("    @k %d: t->fxx[406] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[406]); return 0; }
    t->fxx[407]= 0.039758999999999996*ct7;
    if(isNANorINF(t->fxx[407])) { PRNT("    @k %d: t->fxx[407] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[407]); return 0; }
    t->fxx[408]= -0.101454*ct7;
    if(isNANorINF(t->fxx[408])) { PRNT("    @k %d: t->fxx[408] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[408]); return 0; }
    t->fxx[409]= -0.23901100000000003*ct7;
    if(isNANorINF(t->fxx[409])) { PRNT("    @k %d: t->fxx[409] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[409]); return 0; }
    t->fxx[410]= -0.098712000000000008*ct7;
    if(isNANorINF(t->fxx[410])) { PRNT("    @k %d: t->fxx[410] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[410]); return 0; }
    t->fxx[411]= 0.078603999999999993*ct7;
    if(isNANorINF(t->fxx[411])) { PRNT("    @k %d: t->fxx[411] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[411]); return 0; }
    t->fxx[412]= -0.20884900000000001*ct7;
    if(isNANorINF(t->fxx[412])) { PRNT("    @k %d: t->fxx[412] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[412]); return 0; }
    t->fxx[413]= -0.095549999999999996*ct7;
    if(isNANorINF(t->fxx[413])) { PRNT("    @k %d: t->fxx[413] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[413]); return 0; }
    t->fxx[414]= 0.015224999999999997*ct7;
    if(isNANorINF(t->fxx[414])) { PRNT("    @k %d: t->fxx[414] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[414]); return 0; }
    t->fxx[415]= -0.038849999999999996*ct7;
    if(isNANorINF(t->fxx[415])) { PRNT("    @k %d: t->fxx[415] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[415]); return 0; }
    t->fxx[416]= -0.091524999999999995*ct7;
    if(isNANorINF(t->fxx[416])) { PRNT("    @k %d: t->fxx[416] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[416]); return 0; }
    t->fxx[417]= -0.0378*ct7;
    if(isNANorINF(t->fxx[417])) { PRNT("    @k %d: t->fxx[417] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[417]); return 0; }
    t->fxx[418]= 0.030099999999999995*ct7;
    if(isNANorINF(t->fxx[418])) { PRNT("    @k %d: t->fxx[418] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[418]); return 0; }
    t->fxx[419]= -0.079975000000000004*ct7;
    if(isNANorINF(t->fxx[419])) { PRNT("    @k %d: t->fxx[419] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[419]); return 0; }
    t->fxx[420]= -0.030624999999999996*ct7;
    if(isNANorINF(t->fxx[420])) { PRNT("    @k %d: t->fxx[420] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[420]); return 0; }
    t->fxx[421]= -0.066612000000000005*ct7;
    if(isNANorINF(t->fxx[421])) { PRNT("    @k %d: t->fxx[421] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[421]); return 0; }
    t->fxx[422]= 0.010613999999999998*ct7;
    if(isNANorINF(t->fxx[422])) { PRNT("    @k %d: t->fxx[422] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[422]); return 0; }
    t->fxx[423]= -0.027084*ct7;
    if(isNANorINF(t->fxx[423])) { PRNT("    @k %d: t->fxx[423] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[423]); return 0; }
    t->fxx[424]= -0.063806000000000002*ct7;
    if(isNANorINF(t->fxx[424])) { PRNT("    @k %d: t->fxx[424] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[424]); return 0; }
    t->fxx[425]= -0.026352*ct7;
    if(isNANorINF(t->fxx[425])) { PRNT("    @k %d: t->fxx[425] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[425]); return 0; }
    t->fxx[426]= 0.020983999999999999*ct7;
    if(isNANorINF(t->fxx[426])) { PRNT("    @k %d: t->fxx[426] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[426]); return 0; }
    t->fxx[427]= -0.055753999999999998*ct7;
    if(isNANorINF(t->fxx[427])) { PRNT("    @k %d: t->fxx[427] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[427]); return 0; }
    t->fxx[428]= -0.021349999999999997*ct7;
    if(isNANorINF(t->fxx[428])) { PRNT("    @k %d: t->fxx[428] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[428]); return 0; }
    t->fxx[429]= -0.014884*ct7;
    if(isNANorINF(t->fxx[429])) { PRNT("    @k %d: t->fxx[429] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[429]); return 0; }
    t->fxx[430]= 0.25662000000000001*ct7;
    if(isNANorINF(t->fxx[430])) { PRNT("    @k %d: t->fxx[430] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[430]); return 0; }
    t->fxx[431]= -0.040889999999999996*ct7;
    if(isNANorINF(t->fxx[431])) { PRNT("    @k %d: t->fxx[431] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[431]); return 0; }
    t->fxx[432]= 0.10434*ct7;
    if(isNANorINF(t->fxx[432])) { PRNT("    @k %d: t->fxx[432] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[432]); return 0; }
    t->fxx[433]= 0.24581*ct7;
    if(isNANorINF(t->fxx[433])) { PRNT("    @k %d: t->fxx[433] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[433]); return 0; }
    t->fxx[434]= 0.10152*ct7;
    if(isNANorINF(t->fxx[434])) { PRNT("    @k %d: t->fxx[434] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[434]); return 0; }
    t->fxx[435]= -0.080839999999999995*ct7;
    if(isNANorINF(t->fxx[435])) { PRNT("    @k %d: t->fxx[435] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[435]); return 0; }
    t->fxx[436]= 0.21479000000000001*ct7;
    if(isNANorINF(t->fxx[436])) { PRNT("    @k %d: t->fxx[436] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[436]); return 0; }
    t->fxx[437]= 0.08224999999999999*ct7;
    if(isNANorINF(t->fxx[437])) { PRNT("    @k %d: t->fxx[437] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[437]); return 0; }
    t->fxx[438]= 0.057339999999999995*ct7;
    if(isNANorINF(t->fxx[438])) { PRNT("    @k %d: t->fxx[438] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[438]); return 0; }
    t->fxx[439]= -0.22089999999999999*ct7;
    if(isNANorINF(t->fxx[439])) { PRNT("    @k %d: t->fxx[439] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[439]); return 0; }
    t->fxx[440]= -0.28090000000000004*ct8;
    if(isNANorINF(t->fxx[440])) { PRNT("    @k %d: t->fxx[440] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[440]); return 0; }
    t->fxx[441]= -0.0058300000000000001*ct8;
    if(isNANorINF(t->fxx[441])) { PRNT("    @k %d: t->fxx[441] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[441]); return 0; }
    t->fxx[442]= -0.00012099999999999999*ct8;
    if(isNANorINF(t->fxx[442])) { PRNT("    @k %d: t->fxx[442] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[442]); return 0; }
    t->fxx[443]= -0.15740999999999999*ct8;
    if(isNANorINF(t->fxx[443])) { PRNT("    @k %d: t->fxx[443] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[443]); return 0; }
    t->fxx[444]= -0.0032669999999999995*ct8;
    if(isNANorINF(t->fxx[444])) { PRNT("    @k %d: t->fxx[444] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[444]); return 0; }
    t->fxx[445]= -0.088208999999999996*ct8;
    if(isNANorINF(t->fxx[445])) { PRNT("    @k %d: t->fxx[445] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[445]); return 0; }
    t->fxx[446]= 0.17702000000000001*ct8;
    if(isNANorINF(t->fxx[446])) { PRNT("    @k %d: t->fxx[446] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[446]); return 0; }
    t->fxx[447]= 0.0036740000000000002*ct8;
    if(isNANorINF(t->fxx[447])) { PRNT("    @k %d: t->fxx[447] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[447]); return 0; }
    t->fxx[448]= 0.099197999999999995*ct8;
    if(isNANorINF(t->fxx[448])) { PRNT("    @k %d: t->fxx[448] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[448]); return 0; }
    t->fxx[449]= -0.11155600000000002*ct8;
    if(isNANorINF(t->fxx[449])) { PRNT("    @k %d: t->fxx[449] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[449]); return 0; }
    t->fxx[450]= -0.86177999999999999*ct8;
    if(isNANorINF(t->fxx[450])) { PRNT("    @k %d: t->fxx[450] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[450]); return 0; }
    t->fxx[451]= -0.017885999999999999*ct8;
    if(isNANorINF(t->fxx[451])) { PRNT("    @k %d: t->fxx[451] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[451]); return 0; }
    t->fxx[452]= -0.48292199999999996*ct8;
    if(isNANorINF(t->fxx[452])) { PRNT("    @k %d: t->fxx[452] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[452]); return 0; }
    t->fxx[453]= 0.54308400000000001*ct8;
    if(isNANorINF(t->fxx[453])) { PRNT("    @k %d: t->fxx[453] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[453]); return 0; }
    t->fxx[454]= -2.6438759999999997*ct8;
    if(isNANorINF(t->fxx[454])) { PRNT("    @k %d: t->fxx[454] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[454]); return 0; }
    t->fxx[455]= -0.20617000000000002*ct8;
    if(isNANorINF(t->fxx[455])) { PRNT("    @k %d: t->fxx[455] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[455]); return 0; }
    t->fxx[456]= -0.0042789999999999998*ct8;
    if(isNANorINF(t->fxx[456])) { PRNT("    @k %d: t->fxx[456] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[456]); return 0; }
    t->fxx[457]= -0.115533*ct8;
    if(isNANorINF(t->fxx[457])) { PRNT("    @k %d: t->fxx[457] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[457]); return 0; }
    t->fxx[458]= 0.12992600000000001*ct8;
    if(isNANorINF(t->fxx[458])) { PRNT("    @k %d: t->fxx[458] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[458]); return 0; }
    t->fxx[459]= -0.63251400000000002*ct8;
    if(isNANorINF(t->fxx[459])) { PRNT("    @k %d: t->fxx[459] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[459]); return 0; }
    t->fxx[460]= -0.15132100000000001*ct8;
    if(isNANorINF(t->fxx[460])) { PRNT("    @k %d: t->fxx[460] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[460]); return 0; }
    t->fxx[461]= 0.18656*ct8;
    if(isNANorINF(t->fxx[461])) { PRNT("    @k %d: t->fxx[461] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[461]); return 0; }
    t->fxx[462]= 0.0038719999999999996*ct8;
    if(isNANorINF(t->fxx[462])) { PRNT("    @k %d: t->fxx[462] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[462]); return 0; }
    t->fxx[463]= 0.10454399999999998*ct8;
    if(isNANorINF(t->fxx[463])) { PRNT("    @k %d: t->fxx[463] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[463]); return 0; }
    t->fxx[464]= -0.11756800000000001*ct8;
    if(isNANorINF(t->fxx[464])) { PRNT("    @k %d: t->fxx[464] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[464]); return 0; }
    t->fxx[465]= 0.57235199999999997*ct8;
    if(isNANorINF(t->fxx[465])) { PRNT("    @k %d: t->fxx[465] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[465]); return 0; }
    t->fxx[466]= 0.13692799999999999*ct8;
    if(isNANorINF(t->fxx[466])) { PRNT("    @k %d: t->fxx[466] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[466]); return 0; }
    t->fxx[467]= -0.12390399999999999*ct8;
    if(isNANorINF(t->fxx[467])) { PRNT("    @k %d: t->fxx[467] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[467]); return 0; }
    t->fxx[468]= -0.022789999999999998*ct8;
    if(isNANorINF(t->fxx[468])) { PRNT("    @k %d: t->fxx[468] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[468]); return 0; }
    t->fxx[469]= -0.00047299999999999995*ct8;
    if(isNANorINF(t->fxx[469])) { PRNT("    @k %d: t->fxx[469] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[469]); return 0; }
    t->fxx[470]= -0.012770999999999998*ct8;
    if(isNANorINF(t->fxx[470])) { PRNT("    @k %d: t->fxx[470] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[470]); return 0; }
    t->fxx[471]= 0.014362*ct8;
    if(isNANorINF(t->fxx[471])) { PRNT("    @k %d: t->fxx[471] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[471]); return 0; }
    t->fxx[472]= -0.069917999999999994*ct8;
    if(isNANorINF(t->fxx[472])) { PRNT("    @k %d: t->fxx[472] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[472]); return 0; }
    t->fxx[473]= -0.016726999999999999*ct8;
    if(isNANorINF(t->fxx[473])) { PRNT("    @k %d: t->fxx[473] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[473]); return 0; }
    t->fxx[474]= 0.015135999999999998*ct8;
    if(isNANorINF(t->fxx[474])) { PRNT("    @k %d: t->fxx[474] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[474]); return 0; }
    t->fxx[475]= -0.0018489999999999997*ct8;
    if(isNANorINF(t->fxx[475])) { PRNT("    @k %d: t->fxx[475] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[475]); return 0; }
    t->fxx[476]= -0.1855*ct8;
    if(isNANorINF(t->fxx[476])) { PRNT("    @k %d: t->fxx[476] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[476]); return 0; }
    t->fxx[477]= -0.0038499999999999997*ct8;
    if(isNANorINF(t->fxx[477])) { PRNT("    @k %d: t->fxx[477] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[477]); return 0; }
    t->fxx[478]= -0.10394999999999999*ct8;
    if(isNANorINF(t->fxx[478])) { PRNT("    @k %d: t->fxx[478] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[478]); return 0; }
    t->fxx[479]= 0.1169*ct8;
    if(isNANorINF(t->fxx[479])) { PRNT("    @k %d: t->fxx[479] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[479]); return 0; }
    t->fxx[480]= -0.56909999999999994*ct8;
    if(isNANorINF(t->fxx[480])) { PRNT("    @k %d: t->fxx[480] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[480]); return 0; }
    t->fxx[481]= -0.13614999999999999*ct8;
    if(isNANorINF(t->fxx[481])) { PRNT("    @k %d: t->fxx[481] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[481]); return 0; }
    t->fxx[482]= 0.12319999999999999*ct8;
    if(isNANorINF(t->fxx[482])) { PRNT("    @k %d: t->fxx[482] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[482]); return 0; }
    t->fxx[483]= -0.015049999999999997*ct8;
    if(isNANorINF(t->fxx[483])) { PRNT("    @k %d: t->fxx[483] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[483]); return 0; }
    t->fxx[484]= -0.12249999999999998*ct8;
    if(isNANorINF(t->fxx[484])) { PRNT("    @k %d: t->fxx[484] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[484]); return 0; }
    t->fxx[485]= 0.39061000000000001*ct8;
    if(isNANorINF(t->fxx[485])) { PRNT("    @k %d: t->fxx[485] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[485]); return 0; }
    t->fxx[486]= 0.0081069999999999996*ct8;
    if(isNANorINF(t->fxx[486])) { PRNT("    @k %d: t->fxx[486] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[486]); return 0; }
    t->fxx[487]= 0.218889*ct8;
    if(isNANorINF(t->fxx[487])) { PRNT("    @k %d: t->fxx[487] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[487]); return 0; }
    t->fxx[488]= -0.24615800000000002*ct8;
    if(isNANorINF(t->fxx[488])) { PRNT("    @k %d: t->fxx[488] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[488]); return 0; }
    t->fxx[489]= 1.1983619999999999*ct8;
    if(isNANorINF(t->fxx[489])) { PRNT("    @k %d: t->fxx[489] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[489]); return 0; }
    t->fxx[490]= 0.28669300000000003*ct8;
    if(isNANorINF(t->fxx[490])) { PRNT("    @k %d: t->fxx[490] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[490]); return 0; }
    t->fxx[491]= -0.25942399999999999*ct8;
    if(isNANorINF(t->fxx[491])) { PRNT("    @k %d: t->fxx[491] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[491]); return 0; }
    t->fxx[492]= 0.031690999999999997*ct8;
    if(isNANorINF(t->fxx[492])) { PRNT("    @k %d: t->fxx[492] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[492]); return 0; }
    t->fxx[493]= 0.25794999999999996*ct8;
    if(isNANorINF(t->fxx[493])) { PRNT("    @k %d: t->fxx[493] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[493]); return 0; }
    t->fxx[494]= -0.54316900000000001*ct8;
    if(isNANorINF(t->fxx[494])) { PRNT("    @k %d: t->fxx[494] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[494]); return 0; }
    t->fxx[495]= -1.0609*ct9;
    if(isNANorINF(t->fxx[495])) { PRNT("    @k %d: t->fxx[495] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[495]); return 0; }
    t->fxx[496]= -0.32754*ct9;
    if(isNANorINF(t->fxx[496])) { PRNT("    @k %d: t->fxx[496] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[496]); return 0; }
    t->fxx[497]= -0.10112400000000001*ct9;
    if(isNANorINF(t->fxx[497])) { PRNT("    @k %d: t->fxx[497] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[497]); return 0; }
    t->fxx[498]= -0.39346000000000003*ct9;
    if(isNANorINF(t->fxx[498])) { PRNT("    @k %d: t->fxx[498] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[498]); return 0; }
    t->fxx[499]= -0.121476*ct9;
    if(isNANorINF(t->fxx[499])) { PRNT("    @k %d: t->fxx[499] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[499]); return 0; }
    t->fxx[500]= -0.145924*ct9;
    if(isNANorINF(t->fxx[500])) { PRNT("    @k %d: t->fxx[500] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[500]); return 0; }
    t->fxx[501]= -0.72202999999999995*ct9;
    if(isNANorINF(t->fxx[501])) { PRNT("    @k %d: t->fxx[501] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[501]); return 0; }
    t->fxx[502]= -0.22291799999999998*ct9;
    if(isNANorINF(t->fxx[502])) { PRNT("    @k %d: t->fxx[502] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[502]); return 0; }
    t->fxx[503]= -0.26778199999999996*ct9;
    if(isNANorINF(t->fxx[503])) { PRNT("    @k %d: t->fxx[503] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[503]); return 0; }
    t->fxx[504]= -0.49140099999999992*ct9;
    if(isNANorINF(t->fxx[504])) { PRNT("    @k %d: t->fxx[504] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[504]); return 0; }
    t->fxx[505]= -0.34711000000000003*ct9;
    if(isNANorINF(t->fxx[505])) { PRNT("    @k %d: t->fxx[505] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[505]); return 0; }
    t->fxx[506]= -0.10716600000000001*ct9;
    if(isNANorINF(t->fxx[506])) { PRNT("    @k %d: t->fxx[506] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[506]); return 0; }
    t->fxx[507]= -0.12873400000000002*ct9;
    if(isNANorINF(t->fxx[507])) { PRNT("    @k %d: t->fxx[507] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[507]); return 0; }
    t->fxx[508]= -0.236237*ct9;
    if(isNANorINF(t->fxx[508])) { PRNT("    @k %d: t->fxx[508] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[508]); return 0; }
    t->fxx[509]= -0.11356900000000002*ct9;
    if(isNANorINF(t->fxx[509])) { PRNT("    @k %d: t->fxx[509] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[509]); return 0; }
    t->fxx[510]= -0.31414999999999998*ct9;
    if(isNANorINF(t->fxx[510])) { PRNT("    @k %d: t->fxx[510] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[510]); return 0; }
    t->fxx[511]= -0.096989999999999993*ct9;
    if(isNANorINF(t->fxx[511])) { PRNT("    @k %d: t->fxx[511] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[511]); return 0; }
    t->fxx[512]= -0.11651*ct9;
    if(isNANorINF(t->fxx[512])) { PRNT("    @k %d: t->fxx[512] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[512]); return 0; }
    t->fxx[513]= -0.213805*ct9;
    if(isNANorINF(t->fxx[513])) { PRNT("    @k %d: t->fxx[513] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[513]); return 0; }
    t->fxx[514]= -0.102785*ct9;
    if(isNANorINF(t->fxx[514])) { PRNT("    @k %d: t->fxx[514] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[514]); return 0; }
    t->fxx[515]= -0.093024999999999997*ct9;
    if(isNANorINF(t->fxx[515])) { PRNT("    @k %d: t->fxx[515] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[515]); return 0; }
    t->fxx[516]= -0.45011000000000001*ct9;
    if(isNANorINF(t->fxx[516])) { PRNT("    @k %d: t->fxx[516] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[516]); return 0; }
    t->fxx[517]= -0.13896600000000001*ct9;
    if(isNANorINF(t->fxx[517])) { PRNT("    @k %d: t->fxx[517] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[517]); return 0; }
    t->fxx[518]= -0.166934*ct9;
    if(isNANorINF(t->fxx[518])) { PRNT("    @k %d: t->fxx[518] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[518]); return 0; }
    t->fxx[519]= -0.30633699999999997*ct9;
    if(isNANorINF(t->fxx[519])) { PRNT("    @k %d: t->fxx[519] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[519]); return 0; }
    t->fxx[520]= -0.14726900000000001*ct9;
    if(isNANorINF(t->fxx[520])) { PRNT("    @k %d: t->fxx[520] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[520]); return 0; }
    t->fxx[521]= -0.13328499999999999*ct9;
    if(isNANorINF(t->fxx[521])) { PRNT("    @k %d: t->fxx[521] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[521]); return 0; }
    t->fxx[522]= -0.190969*ct9;
    if(isNANorINF(t->fxx[522])) { PRNT("    @k %d: t->fxx[522] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[522]); return 0; }
    t->fxx[523]= -0.76838000000000006*ct9;
    if(isNANorINF(t->fxx[523])) { PRNT("    @k %d: t->fxx[523] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[523]); return 0; }
    t->fxx[524]= -0.23722799999999999*ct9;
    if(isNANorINF(t->fxx[524])) { PRNT("    @k %d: t->fxx[524] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[524]); return 0; }
    t->fxx[525]= -0.284972*ct9;
    if(isNANorINF(t->fxx[525])) { PRNT("    @k %d: t->fxx[525] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[525]); return 0; }
    t->fxx[526]= -0.52294599999999991*ct9;
    if(isNANorINF(t->fxx[526])) { PRNT("    @k %d: t->fxx[526] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[526]); return 0; }
    t->fxx[527]= -0.25140200000000001*ct9;
    if(isNANorINF(t->fxx[527])) { PRNT("    @k %d: t->fxx[527] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[527]); return 0; }
    t->fxx[528]= -0.22752999999999998*ct9;
    if(isNANorINF(t->fxx[528])) { PRNT("    @k %d: t->fxx[528] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[528]); return 0; }
    t->fxx[529]= -0.32600200000000001*ct9;
    if(isNANorINF(t->fxx[529])) { PRNT("    @k %d: t->fxx[529] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[529]); return 0; }
    t->fxx[530]= -0.55651600000000001*ct9;
    if(isNANorINF(t->fxx[530])) { PRNT("    @k %d: t->fxx[530] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[530]); return 0; }
    t->fxx[531]= 0.51294000000000006*ct9;
    if(isNANorINF(t->fxx[531])) { PRNT("    @k %d: t->fxx[531] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[531]); return 0; }
    t->fxx[532]= 0.158364*ct9;
    if(isNANorINF(t->fxx[532])) { PRNT("    @k %d: t->fxx[532] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[532]); return 0; }
    t->fxx[533]= 0.19023600000000002*ct9;
    if(isNANorINF(t->fxx[533])) { PRNT("    @k %d: t->fxx[533] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[533]); return 0; }
    t->fxx[534]= 0.34909799999999996*ct9;
    if(isNANorINF(t->fxx[534])) { PRNT("    @k %d: t->fxx[534] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[534]); return 0; }
    t->fxx[535]= 0.167826*ct9;
    if(isNANorINF(t->fxx[535])) { PRNT("    @k %d: t->fxx[535] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[535]); return 0; }
    t->fxx[536]= 0.15189*ct9;
    if(isNANorINF(t->fxx[536])) { PRNT("    @k %d: t->fxx[536] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[536]); return 0; }
    t->fxx[537]= 0.21762599999999999*ct9;
    if(isNANorINF(t->fxx[537])) { PRNT("    @k %d: t->fxx[537] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[537]); return 0; }
    t->fxx[538]= 0.371508*ct9;
    if(isNANorINF(t->fxx[538])) { PRNT("    @k %d: t->fxx[538] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[538]); return 0; }
    t->fxx[539]= -0.248004*ct9;
    if(isNANorINF(t->fxx[539])) { PRNT("    @k %d: t->fxx[539] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[539]); return 0; }
    t->fxx[540]= 0.51294000000000006*ct9;
    if(isNANorINF(t->fxx[540])) { PRNT("    @k %d: t->fxx[540] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[540]); return 0; }
    t->fxx[541]= 0.158364*ct9;
    if(isNANorINF(t->fxx[541])) { PRNT("    @k %d: t->fxx[541] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[541]); return 0; }
    t->fxx[542]= 0.19023600000000002*ct9;
    if(isNANorINF(t->fxx[542])) { PRNT("    @k %d: t->fxx[542] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[542]); return 0; }
    t->fxx[543]= 0.34909799999999996*ct9;
    if(isNANorINF(t->fxx[543])) { PRNT("    @k %d: t->fxx[543] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[543]); return 0; }
    t->fxx[544]= 0.167826*ct9;
    if(isNANorINF(t->fxx[544])) { PRNT("    @k %d: t->fxx[544] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[544]); return 0; }
    t->fxx[545]= 0.15189*ct9;
    if(isNANorINF(t->fxx[545])) { PRNT("    @k %d: t->fxx[545] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[545]); return 0; }
    t->fxx[546]= 0.21762599999999999*ct9;
    if(isNANorINF(t->fxx[546])) { PRNT("    @k %d: t->fxx[546] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[546]); return 0; }
    t->fxx[547]= 0.371508*ct9;
    if(isNANorINF(t->fxx[547])) { PRNT("    @k %d: t->fxx[547] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[547]); return 0; }
    t->fxx[548]= -0.248004*ct9;
    if(isNANorINF(t->fxx[548])) { PRNT("    @k %d: t->fxx[548] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[548]); return 0; }
    t->fxx[549]= -0.248004*ct9;
    if(isNANorINF(t->fxx[549])) { PRNT("    @k %d: t->fxx[549] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[549]); return 0; }

    t->fuu[0]= -0.84272400000000003*ct0;
    if(isNANorINF(t->fuu[0])) { PRNT("    @k %d: t->fuu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[0]); return 0; }
    t->fuu[1]= 0.36995400000000006*ct0;
    if(isNANorINF(t->fuu[1])) { PRNT("    @k %d: t->fuu[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[1]); return 0; }
    t->fuu[2]= -0.16240900000000003*ct0;
    if(isNANorINF(t->fuu[2])) { PRNT("    @k %d: t->fuu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[2]); return 0; }
    t->fuu[3]= 0.18451800000000002*ct0;
    if(isNANorINF(t->fuu[3])) { PRNT("    @k %d: t->fuu[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[3]); return 0; }
    t->fuu[4]= -0.081003000000000006*ct0;
    if(isNANorINF(t->fuu[4])) { PRNT("    @k %d: t->fuu[4] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[4]); return 0; }
    t->fuu[5]= -0.040401000000000006*ct0;
    if(isNANorINF(t->fuu[5])) { PRNT("    @k %d: t->fuu[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[5]); return 0; }
    t->fuu[6]= -0.96039999999999992*ct1;
    if(isNANorINF(t->fuu[6])) { PRNT("    @k %d: t->fuu[6] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[6]); return 0; }
    t->fuu[7]= -0.62229999999999996*ct1;
    if(isNANorINF(t->fuu[7])) { PRNT("    @k %d: t->fuu[7] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[7]); return 0; }
    t->fuu[8]= -0.403225*ct1;
    if(isNANorINF(t->fuu[8])) { PRNT("    @k %d: t->fuu[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[8]); return 0; }
    t->fuu[9]= -0.67619999999999991*ct1;
    if(isNANorINF(t->fuu[9])) { PRNT("    @k %d: t->fuu[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[9]); return 0; }
    t->fuu[10]= -0.43814999999999998*ct1;
    if(isNANorINF(t->fuu[10])) { PRNT("    @k %d: t->fuu[10] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[10]); return 0; }
    t->fuu[11]= -0.47609999999999991*ct1;
    if(isNANorINF(t->fuu[11])) { PRNT("    @k %d: t->fuu[11] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[11]); return 0; }
    t->fuu[12]= -1.7635840000000003*ct2;
    if(isNANorINF(t->fuu[12])) { PRNT("    @k %d: t->fuu[12] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[12]); return 0; }
    t->fuu[13]= 1.349248*ct2;
    if(isNANorINF(t->fuu[13])) { PRNT("    @k %d: t->fuu[13] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[13]); return 0; }
    t->fuu[14]= -1.0322560000000001*ct2;
    if(isNANorINF(t->fuu[14])) { PRNT("    @k %d: t->fuu[14] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[14]); return 0; }
    t->fuu[15]= -0.47542400000000001*ct2;
    if(isNANorINF(t->fuu[15])) { PRNT("    @k %d: t->fuu[15] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[15]); return 0; }
    t->fuu[16]= 0.363728*ct2;
    if(isNANorINF(t->fuu[16])) { PRNT("    @k %d: t->fuu[16] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[16]); return 0; }
    t->fuu[17]= -0.128164*ct2;
    if(isNANorINF(t->fuu[17])) { PRNT("    @k %d: t->fuu[17] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[17]); return 0; }
    t->fuu[18]= -0.54316900000000001*ct3;
    if(isNANorINF(t->fuu[18])) { PRNT("    @k %d: t->fuu[18] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[18]); return 0; }
    t->fuu[19]= -0.18351300000000001*ct3;
    if(isNANorINF(t->fuu[19])) { PRNT("    @k %d: t->fuu[19] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[19]); return 0; }
    t->fuu[20]= -0.062001000000000001*ct3;
    if(isNANorINF(t->fuu[20])) { PRNT("    @k %d: t->fuu[20] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[20]); return 0; }
    t->fuu[21]= -1.566125*ct3;
    if(isNANorINF(t->fuu[21])) { PRNT("    @k %d: t->fuu[21] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[21]); return 0; }
    t->fuu[22]= -0.52912499999999996*ct3;
    if(isNANorINF(t->fuu[22])) { PRNT("    @k %d: t->fuu[22] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[22]); return 0; }
    t->fuu[23]= -4.515625*ct3;
    if(isNANorINF(t->fuu[23])) { PRNT("    @k %d: t->fuu[23] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[23]); return 0; }
    t->fuu[24]= -0.059048999999999997*ct4;
    if(isNANorINF(t->fuu[24])) { PRNT("    @k %d: t->fuu[24] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[24]); return 0; }
    t->fuu[25]= 0.033777000000000001*ct4;
    if(isNANorINF(t->fuu[25])) { PRNT("    @k %d: t->fuu[25] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[25]); return 0; }
    t->fuu[26]= -0.019321000000000005*ct4;
    if(isNANorINF(t->fuu[26])) { PRNT("    @k %d: t->fuu[26] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[26]); return 0; }
    t->fuu[27]= -0.021869999999999997*ct4;
    if(isNANorINF(t->fuu[27])) { PRNT("    @k %d: t->fuu[27] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[27]); return 0; }
    t->fuu[28]= 0.01251*ct4;
    if(isNANorINF(t->fuu[28])) { PRNT("    @k %d: t->fuu[28] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[28]); return 0; }
    t->fuu[29]= -0.0080999999999999996*ct4;
    if(isNANorINF(t->fuu[29])) { PRNT("    @k %d: t->fuu[29] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[29]); return 0; }
    t->fuu[30]= -0.10890000000000001*ct5;
    if(isNANorINF(t->fuu[30])) { PRNT("    @k %d: t->fuu[30] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[30]); return 0; }
    t->fuu[31]= 0.34188000000000002*ct5;
    if(isNANorINF(t->fuu[31])) { PRNT("    @k %d: t->fuu[31] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[31]); return 0; }
    t->fuu[32]= -1.073296*ct5;
    if(isNANorINF(t->fuu[32])) { PRNT("    @k %d: t->fuu[32] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[32]); return 0; }
    t->fuu[33]= 0.36465000000000003*ct5;
    if(isNANorINF(t->fuu[33])) { PRNT("    @k %d: t->fuu[33] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[33]); return 0; }
    t->fuu[34]= -1.1447799999999999*ct5;
    if(isNANorINF(t->fuu[34])) { PRNT("    @k %d: t->fuu[34] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[34]); return 0; }
    t->fuu[35]= -1.221025*ct5;
    if(isNANorINF(t->fuu[35])) { PRNT("    @k %d: t->fuu[35] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[35]); return 0; }
    t->fuu[36]= -0.79923600000000006*ct6;
    if(isNANorINF(t->fuu[36])) { PRNT("    @k %d: t->fuu[36] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[36]); return 0; }
    t->fuu[37]= -0.027713999999999999*ct6;
    if(isNANorINF(t->fuu[37])) { PRNT("    @k %d: t->fuu[37] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[37]); return 0; }
    t->fuu[38]= -0.00096099999999999994*ct6;
    if(isNANorINF(t->fuu[38])) { PRNT("    @k %d: t->fuu[38] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[38]); return 0; }
    t->fuu[39]= 0.037548000000000005*ct6;
    if(isNANorINF(t->fuu[39])) { PRNT("    @k %d: t->fuu[39] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[39]); return 0; }
    t->fuu[40]= 0.001302*ct6;
    if(isNANorINF(t->fuu[40])) { PRNT("    @k %d: t->fuu[40] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[40]); return 0; }
    t->fuu[41]= -0.0017640000000000002*ct6;
    if(isNANorINF(t->fuu[41])) { PRNT("    @k %d: t->fuu[41] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[41]); return 0; }
    t->fuu[42]= -0.0012959999999999998*ct7;
    if(isNANorINF(t->fuu[42])) { PRNT("    @k %d: t->fuu[42] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[42]); return 0; }
    t->fuu[43]= 0.032579999999999998*ct7;
    if(isNANorINF(t->fuu[43])) { PRNT("    @k %d: t->fuu[43] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[43]); return 0; }
    t->fuu[44]= -0.819025*ct7;
    if(isNANorINF(t->fuu[44])) { PRNT("    @k %d: t->fuu[44] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[44]); return 0; }
    t->fuu[45]= 0.015155999999999998*ct7;
    if(isNANorINF(t->fuu[45])) { PRNT("    @k %d: t->fuu[45] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[45]); return 0; }
    t->fuu[46]= -0.38100499999999998*ct7;
    if(isNANorINF(t->fuu[46])) { PRNT("    @k %d: t->fuu[46] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[46]); return 0; }
    t->fuu[47]= -0.17724099999999998*ct7;
    if(isNANorINF(t->fuu[47])) { PRNT("    @k %d: t->fuu[47] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[47]); return 0; }
    t->fuu[48]= -0.47886399999999996*ct8;
    if(isNANorINF(t->fuu[48])) { PRNT("    @k %d: t->fuu[48] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[48]); return 0; }
    t->fuu[49]= -0.094803999999999999*ct8;
    if(isNANorINF(t->fuu[49])) { PRNT("    @k %d: t->fuu[49] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[49]); return 0; }
    t->fuu[50]= -0.018769000000000004*ct8;
    if(isNANorINF(t->fuu[50])) { PRNT("    @k %d: t->fuu[50] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[50]); return 0; }
    t->fuu[51]= 0.040827999999999996*ct8;
    if(isNANorINF(t->fuu[51])) { PRNT("    @k %d: t->fuu[51] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[51]); return 0; }
    t->fuu[52]= 0.0080829999999999999*ct8;
    if(isNANorINF(t->fuu[52])) { PRNT("    @k %d: t->fuu[52] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[52]); return 0; }
    t->fuu[53]= -0.0034809999999999997*ct8;
    if(isNANorINF(t->fuu[53])) { PRNT("    @k %d: t->fuu[53] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[53]); return 0; }
    t->fuu[54]= -0.020448999999999995*ct9;
    if(isNANorINF(t->fuu[54])) { PRNT("    @k %d: t->fuu[54] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[54]); return 0; }
    t->fuu[55]= -0.27227199999999996*ct9;
    if(isNANorINF(t->fuu[55])) { PRNT("    @k %d: t->fuu[55] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[55]); return 0; }
    t->fuu[56]= -3.6252159999999995*ct9;
    if(isNANorINF(t->fuu[56])) { PRNT("    @k %d: t->fuu[56] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[56]); return 0; }
    t->fuu[57]= -0.035749999999999997*ct9;
    if(isNANorINF(t->fuu[57])) { PRNT("    @k %d: t->fuu[57] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[57]); return 0; }
    t->fuu[58]= -0.47599999999999998*ct9;
    if(isNANorINF(t->fuu[58])) { PRNT("    @k %d: t->fuu[58] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[58]); return 0; }
    t->fuu[59]= -0.0625*ct9;
    if(isNANorINF(t->fuu[59])) { PRNT("    @k %d: t->fuu[59] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fuu[59]); return 0; }

    t->fxu[0]= 0.042228000000000002*ct10;
    if(isNANorINF(t->fxu[0])) { PRNT("    @k %d: t->fxu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[0]); return 0; }
    t->fxu[1]= 0.055079999999999997*ct10;
    if(isNANorINF(t->fxu[1])) { PRNT("    @k %d: t->fxu[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[1]); return 0; }
    t->fxu[2]= 0.54345600000000005*ct10;
    if(isNANorINF(t->fxu[2])) { PRNT("    @k %d: t->fxu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[2]); return 0; }
    t->fxu[3]= 0.30753000000000003*ct10;
    if(isNANorINF(t->fxu[3])) { PRNT("    @k %d: t->fxu[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[3]); return 0; }
    t->fxu[4]= 0.55998000000000003*ct10;
    if(isNANorINF(t->fxu[4])) { PRNT("    @k %d: t->fxu[4] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[4]); return 0; }
    t->fxu[5]= 0.11750400000000001*ct10;
    if(isNANorINF(t->fxu[5])) { PRNT("    @k %d: t->fxu[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[5]); return 0; }
    t->fxu[6]= 0.50398200000000004*ct10;
    if(isNANorINF(t->fxu[6])) { PRNT("    @k %d: t->fxu[6] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[6]); return 0; }
    t->fxu[7]= -0.108324*ct10;
    if(isNANorINF(t->fxu[7])) { PRNT("    @k %d: t->fxu[7] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[7]); return 0; }
    t->fxu[8]= 0.55447199999999996*ct10;
    if(isNANorINF(t->fxu[8])) { PRNT("    @k %d: t->fxu[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[8]); return 0; }
    t->fxu[9]= 0.58384800000000003*ct10;
    if(isNANorINF(t->fxu[9])) { PRNT("    @k %d: t->fxu[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[9]); return 0; }
    t->fxu[10]= -0.018538000000000002*ct10;
    if(isNANorINF(t->fxu[10])) { PRNT("    @k %d: t->fxu[10] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[10]); return 0; }
    t->fxu[11]= -0.02418*ct10;
    if(isNANorINF(t->fxu[11])) { PRNT("    @k %d: t->fxu[11] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[11]); return 0; }
    t->fxu[12]= -0.23857600000000001*ct10;
    if(isNANorINF(t->fxu[12])) { PRNT("    @k %d: t->fxu[12] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[12]); return 0; }
    t->fxu[13]= -0.13500500000000001*ct10;
    if(isNANorINF(t->fxu[13])) { PRNT("    @k %d: t->fxu[13] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[13]); return 0; }
    t->fxu[14]= -0.24583000000000002*ct10;
    if(isNANorINF(t->fxu[14])) { PRNT("    @k %d: t->fxu[14] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[14]); return 0; }
    t->fxu[15]= -0.051584000000000005*ct10;
    if(isNANorINF(t->fxu[15])) { PRNT("    @k %d: t->fxu[15] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[15]); return 0; }
    t->fxu[16]= -0.22124700000000003*ct10;
    if(isNANorINF(t->fxu[16])) { PRNT("    @k %d: t->fxu[16] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[16]); return 0; }
    t->fxu[17]= 0.047553999999999999*ct10;
    if(isNANorINF(t->fxu[17])) { PRNT("    @k %d: t->fxu[17] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[17]); return 0; }
    t->fxu[18]= -0.24341200000000002*ct10;
    if(isNANorINF(t->fxu[18])) { PRNT("    @k %d: t->fxu[18] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[18]); return 0; }
    t->fxu[19]= -0.25630800000000004*ct10;
    if(isNANorINF(t->fxu[19])) { PRNT("    @k %d: t->fxu[19] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[19]); return 0; }
    t->fxu[20]= -0.0092460000000000007*ct10;
    if(isNANorINF(t->fxu[20])) { PRNT("    @k %d: t->fxu[20] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[20]); return 0; }
    t->fxu[21]= -0.01206*ct10;
    if(isNANorINF(t->fxu[21])) { PRNT("    @k %d: t->fxu[21] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[21]); return 0; }
    t->fxu[22]= -0.118992*ct10;
    if(isNANorINF(t->fxu[22])) { PRNT("    @k %d: t->fxu[22] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[22]); return 0; }
    t->fxu[23]= -0.067335000000000006*ct10;
    if(isNANorINF(t->fxu[23])) { PRNT("    @k %d: t->fxu[23] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[23]); return 0; }
    t->fxu[24]= -0.12261000000000001*ct10;
    if(isNANorINF(t->fxu[24])) { PRNT("    @k %d: t->fxu[24] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[24]); return 0; }
    t->fxu[25]= -0.025728000000000001*ct10;
    if(isNANorINF(t->fxu[25])) { PRNT("    @k %d: t->fxu[25] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[25]); return 0; }
    t->fxu[26]= -0.11034900000000002*ct10;
    if(isNANorINF(t->fxu[26])) { PRNT("    @k %d: t->fxu[26] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[26]); return 0; }
    t->fxu[27]= 0.023717999999999999*ct10;
    if(isNANorINF(t->fxu[27])) { PRNT("    @k %d: t->fxu[27] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[27]); return 0; }
    t->fxu[28]= -0.121404*ct10;
    if(isNANorINF(t->fxu[28])) { PRNT("    @k %d: t->fxu[28] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[28]); return 0; }
    t->fxu[29]= -0.12783600000000001*ct10;
    if(isNANorINF(t->fxu[29])) { PRNT("    @k %d: t->fxu[29] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[29]); return 0; }
    t->fxu[30]= 0.82809999999999995*ct11;
    if(isNANorINF(t->fxu[30])) { PRNT("    @k %d: t->fxu[30] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[30]); return 0; }
    t->fxu[31]= -0.41258*ct11;
    if(isNANorINF(t->fxu[31])) { PRNT("    @k %d: t->fxu[31] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[31]); return 0; }
    t->fxu[32]= -0.34103999999999995*ct11;
    if(isNANorINF(t->fxu[32])) { PRNT("    @k %d: t->fxu[32] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[32]); return 0; }
    t->fxu[33]= -0.37043999999999999*ct11;
    if(isNANorINF(t->fxu[33])) { PRNT("    @k %d: t->fxu[33] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[33]); return 0; }
    t->fxu[34]= 0.070559999999999998*ct11;
    if(isNANorINF(t->fxu[34])) { PRNT("    @k %d: t->fxu[34] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[34]); return 0; }
    t->fxu[35]= -0.61641999999999997*ct11;
    if(isNANorINF(t->fxu[35])) { PRNT("    @k %d: t->fxu[35] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[35]); return 0; }
    t->fxu[36]= 0.50372000000000006*ct11;
    if(isNANorINF(t->fxu[36])) { PRNT("    @k %d: t->fxu[36] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[36]); return 0; }
    t->fxu[37]= 0.11368*ct11;
    if(isNANorINF(t->fxu[37])) { PRNT("    @k %d: t->fxu[37] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[37]); return 0; }
    t->fxu[38]= -0.19109999999999999*ct11;
    if(isNANorINF(t->fxu[38])) { PRNT("    @k %d: t->fxu[38] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[38]); return 0; }
    t->fxu[39]= 0.85358000000000001*ct11;
    if(isNANorINF(t->fxu[39])) { PRNT("    @k %d: t->fxu[39] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[39]); return 0; }
    t->fxu[40]= 0.53657500000000002*ct11;
    if(isNANorINF(t->fxu[40])) { PRNT("    @k %d: t->fxu[40] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[40]); return 0; }
    t->fxu[41]= -0.26733499999999999*ct11;
    if(isNANorINF(t->fxu[41])) { PRNT("    @k %d: t->fxu[41] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[41]); return 0; }
    t->fxu[42]= -0.22097999999999998*ct11;
    if(isNANorINF(t->fxu[42])) { PRNT("    @k %d: t->fxu[42] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[42]); return 0; }
    t->fxu[43]= -0.24002999999999999*ct11;
    if(isNANorINF(t->fxu[43])) { PRNT("    @k %d: t->fxu[43] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[43]); return 0; }
    t->fxu[44]= 0.045719999999999997*ct11;
    if(isNANorINF(t->fxu[44])) { PRNT("    @k %d: t->fxu[44] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[44]); return 0; }
    t->fxu[45]= -0.39941500000000002*ct11;
    if(isNANorINF(t->fxu[45])) { PRNT("    @k %d: t->fxu[45] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[45]); return 0; }
    t->fxu[46]= 0.32639000000000001*ct11;
    if(isNANorINF(t->fxu[46])) { PRNT("    @k %d: t->fxu[46] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[46]); return 0; }
    t->fxu[47]= 0.073660000000000003*ct11;
    if(isNANorINF(t->fxu[47])) { PRNT("    @k %d: t->fxu[47] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[47]); return 0; }
    t->fxu[48]= -0.123825*ct11;
    if(isNANorINF(t->fxu[48])) { PRNT("    @k %d: t->fxu[48] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[48]); return 0; }
    t->fxu[49]= 0.55308500000000005*ct11;
    if(isNANorINF(t->fxu[49])) { PRNT("    @k %d: t->fxu[49] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[49]); return 0; }
    t->fxu[50]= 0.58304999999999996*ct11;
    if(isNANorINF(t->fxu[50])) { PRNT("    @k %d: t->fxu[50] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[50]); return 0; }
    t->fxu[51]= -0.29048999999999997*ct11;
    if(isNANorINF(t->fxu[51])) { PRNT("    @k %d: t->fxu[51] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[51]); return 0; }
    t->fxu[52]= -0.24011999999999997*ct11;
    if(isNANorINF(t->fxu[52])) { PRNT("    @k %d: t->fxu[52] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[52]); return 0; }
    t->fxu[53]= -0.26082*ct11;
    if(isNANorINF(t->fxu[53])) { PRNT("    @k %d: t->fxu[53] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[53]); return 0; }
    t->fxu[54]= 0.049679999999999995*ct11;
    if(isNANorINF(t->fxu[54])) { PRNT("    @k %d: t->fxu[54] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[54]); return 0; }
    t->fxu[55]= -0.43400999999999995*ct11;
    if(isNANorINF(t->fxu[55])) { PRNT("    @k %d: t->fxu[55] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[55]); return 0; }
    t->fxu[56]= 0.35465999999999998*ct11;
    if(isNANorINF(t->fxu[56])) { PRNT("    @k %d: t->fxu[56] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[56]); return 0; }
    t->fxu[57]= 0.08004*ct11;
    if(isNANorINF(t->fxu[57])) { PRNT("    @k %d: t->fxu[57] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[57]); return 0; }
    t->fxu[58]= -0.13455*ct11;
    if(isNANorINF(t->fxu[58])) { PRNT("    @k %d: t->fxu[58] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[58]); return 0; }
    t->fxu[59]= 0.60098999999999991*ct11;
    if(isNANorINF(t->fxu[59])) { PRNT("    @k %d: t->fxu[59] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[59]); return 0; }
    t->fxu[60]= 1.001312*ct12;
    if(isNANorINF(t->fxu[60])) { PRNT("    @k %d: t->fxu[60] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[60]); return 0; }
    t->fxu[61]= 0.34129600000000004*ct12;
    if(isNANorINF(t->fxu[61])) { PRNT("    @k %d: t->fxu[61] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[61]); return 0; }
    t->fxu[62]= -0.087648000000000004*ct12;
    if(isNANorINF(t->fxu[62])) { PRNT("    @k %d: t->fxu[62] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[62]); return 0; }
    t->fxu[63]= -0.87913600000000014*ct12;
    if(isNANorINF(t->fxu[63])) { PRNT("    @k %d: t->fxu[63] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[63]); return 0; }
    t->fxu[64]= -1.4262720000000002*ct12;
    if(isNANorINF(t->fxu[64])) { PRNT("    @k %d: t->fxu[64] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[64]); return 0; }
    t->fxu[65]= -0.68126400000000009*ct12;
    if(isNANorINF(t->fxu[65])) { PRNT("    @k %d: t->fxu[65] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[65]); return 0; }
    t->fxu[66]= 0.34129600000000004*ct12;
    if(isNANorINF(t->fxu[66])) { PRNT("    @k %d: t->fxu[66] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[66]); return 0; }
    t->fxu[67]= 0.66931200000000002*ct12;
    if(isNANorINF(t->fxu[67])) { PRNT("    @k %d: t->fxu[67] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[67]); return 0; }
    t->fxu[68]= 0.46479999999999999*ct12;
    if(isNANorINF(t->fxu[68])) { PRNT("    @k %d: t->fxu[68] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[68]); return 0; }
    t->fxu[69]= -0.29481600000000002*ct12;
    if(isNANorINF(t->fxu[69])) { PRNT("    @k %d: t->fxu[69] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[69]); return 0; }
    t->fxu[70]= -0.76606399999999997*ct12;
    if(isNANorINF(t->fxu[70])) { PRNT("    @k %d: t->fxu[70] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[70]); return 0; }
    t->fxu[71]= -0.26111200000000001*ct12;
    if(isNANorINF(t->fxu[71])) { PRNT("    @k %d: t->fxu[71] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[71]); return 0; }
    t->fxu[72]= 0.067056000000000004*ct12;
    if(isNANorINF(t->fxu[72])) { PRNT("    @k %d: t->fxu[72] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[72]); return 0; }
    t->fxu[73]= 0.67259200000000008*ct12;
    if(isNANorINF(t->fxu[73])) { PRNT("    @k %d: t->fxu[73] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[73]); return 0; }
    t->fxu[74]= 1.0911840000000002*ct12;
    if(isNANorINF(t->fxu[74])) { PRNT("    @k %d: t->fxu[74] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[74]); return 0; }
    t->fxu[75]= 0.521208*ct12;
    if(isNANorINF(t->fxu[75])) { PRNT("    @k %d: t->fxu[75] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[75]); return 0; }
    t->fxu[76]= -0.26111200000000001*ct12;
    if(isNANorINF(t->fxu[76])) { PRNT("    @k %d: t->fxu[76] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[76]); return 0; }
    t->fxu[77]= -0.51206399999999996*ct12;
    if(isNANorINF(t->fxu[77])) { PRNT("    @k %d: t->fxu[77] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[77]); return 0; }
    t->fxu[78]= -0.35559999999999997*ct12;
    if(isNANorINF(t->fxu[78])) { PRNT("    @k %d: t->fxu[78] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[78]); return 0; }
    t->fxu[79]= 0.225552*ct12;
    if(isNANorINF(t->fxu[79])) { PRNT("    @k %d: t->fxu[79] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[79]); return 0; }
    t->fxu[80]= 0.26993200000000001*ct12;
    if(isNANorINF(t->fxu[80])) { PRNT("    @k %d: t->fxu[80] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[80]); return 0; }
    t->fxu[81]= 0.092006000000000004*ct12;
    if(isNANorINF(t->fxu[81])) { PRNT("    @k %d: t->fxu[81] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[81]); return 0; }
    t->fxu[82]= -0.023628*ct12;
    if(isNANorINF(t->fxu[82])) { PRNT("    @k %d: t->fxu[82] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[82]); return 0; }
    t->fxu[83]= -0.23699600000000001*ct12;
    if(isNANorINF(t->fxu[83])) { PRNT("    @k %d: t->fxu[83] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[83]); return 0; }
    t->fxu[84]= -0.384492*ct12;
    if(isNANorINF(t->fxu[84])) { PRNT("    @k %d: t->fxu[84] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[84]); return 0; }
    t->fxu[85]= -0.18365399999999998*ct12;
    if(isNANorINF(t->fxu[85])) { PRNT("    @k %d: t->fxu[85] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[85]); return 0; }
    t->fxu[86]= 0.092006000000000004*ct12;
    if(isNANorINF(t->fxu[86])) { PRNT("    @k %d: t->fxu[86] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[86]); return 0; }
    t->fxu[87]= 0.18043199999999998*ct12;
    if(isNANorINF(t->fxu[87])) { PRNT("    @k %d: t->fxu[87] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[87]); return 0; }
    t->fxu[88]= 0.12529999999999999*ct12;
    if(isNANorINF(t->fxu[88])) { PRNT("    @k %d: t->fxu[88] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[88]); return 0; }
    t->fxu[89]= -0.079475999999999991*ct12;
    if(isNANorINF(t->fxu[89])) { PRNT("    @k %d: t->fxu[89] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[89]); return 0; }
    t->fxu[90]= 0.058222999999999997*ct13;
    if(isNANorINF(t->fxu[90])) { PRNT("    @k %d: t->fxu[90] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[90]); return 0; }
    t->fxu[91]= 0.30364399999999997*ct13;
    if(isNANorINF(t->fxu[91])) { PRNT("    @k %d: t->fxu[91] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[91]); return 0; }
    t->fxu[92]= -0.81291099999999994*ct13;
    if(isNANorINF(t->fxu[92])) { PRNT("    @k %d: t->fxu[92] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[92]); return 0; }
    t->fxu[93]= -0.030954000000000002*ct13;
    if(isNANorINF(t->fxu[93])) { PRNT("    @k %d: t->fxu[93] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[93]); return 0; }
    t->fxu[94]= -0.036850000000000001*ct13;
    if(isNANorINF(t->fxu[94])) { PRNT("    @k %d: t->fxu[94] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[94]); return 0; }
    t->fxu[95]= 0.25794999999999996*ct13;
    if(isNANorINF(t->fxu[95])) { PRNT("    @k %d: t->fxu[95] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[95]); return 0; }
    t->fxu[96]= 0.42230099999999998*ct13;
    if(isNANorINF(t->fxu[96])) { PRNT("    @k %d: t->fxu[96] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[96]); return 0; }
    t->fxu[97]= -0.18646099999999999*ct13;
    if(isNANorINF(t->fxu[97])) { PRNT("    @k %d: t->fxu[97] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[97]); return 0; }
    t->fxu[98]= -0.86523799999999995*ct13;
    if(isNANorINF(t->fxu[98])) { PRNT("    @k %d: t->fxu[98] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[98]); return 0; }
    t->fxu[99]= 0.39282100000000003*ct13;
    if(isNANorINF(t->fxu[99])) { PRNT("    @k %d: t->fxu[99] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[99]); return 0; }
    t->fxu[100]= 0.019671000000000001*ct13;
    if(isNANorINF(t->fxu[100])) { PRNT("    @k %d: t->fxu[100] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[100]); return 0; }
    t->fxu[101]= 0.102588*ct13;
    if(isNANorINF(t->fxu[101])) { PRNT("    @k %d: t->fxu[101] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[101]); return 0; }
    t->fxu[102]= -0.27464699999999997*ct13;
    if(isNANorINF(t->fxu[102])) { PRNT("    @k %d: t->fxu[102] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[102]); return 0; }
    t->fxu[103]= -0.010458*ct13;
    if(isNANorINF(t->fxu[103])) { PRNT("    @k %d: t->fxu[103] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[103]); return 0; }
    t->fxu[104]= -0.012450000000000001*ct13;
    if(isNANorINF(t->fxu[104])) { PRNT("    @k %d: t->fxu[104] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[104]); return 0; }
    t->fxu[105]= 0.087149999999999991*ct13;
    if(isNANorINF(t->fxu[105])) { PRNT("    @k %d: t->fxu[105] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[105]); return 0; }
    t->fxu[106]= 0.142677*ct13;
    if(isNANorINF(t->fxu[106])) { PRNT("    @k %d: t->fxu[106] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[106]); return 0; }
    t->fxu[107]= -0.062996999999999997*ct13;
    if(isNANorINF(t->fxu[107])) { PRNT("    @k %d: t->fxu[107] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[107]); return 0; }
    t->fxu[108]= -0.29232599999999997*ct13;
    if(isNANorINF(t->fxu[108])) { PRNT("    @k %d: t->fxu[108] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[108]); return 0; }
    t->fxu[109]= 0.132717*ct13;
    if(isNANorINF(t->fxu[109])) { PRNT("    @k %d: t->fxu[109] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[109]); return 0; }
    t->fxu[110]= 0.167875*ct13;
    if(isNANorINF(t->fxu[110])) { PRNT("    @k %d: t->fxu[110] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[110]); return 0; }
    t->fxu[111]= 0.87549999999999994*ct13;
    if(isNANorINF(t->fxu[111])) { PRNT("    @k %d: t->fxu[111] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[111]); return 0; }
    t->fxu[112]= -2.3438750000000002*ct13;
    if(isNANorINF(t->fxu[112])) { PRNT("    @k %d: t->fxu[112] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[112]); return 0; }
    t->fxu[113]= -0.08925000000000001*ct13;
    if(isNANorINF(t->fxu[113])) { PRNT("    @k %d: t->fxu[113] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[113]); return 0; }
    t->fxu[114]= -0.10625000000000001*ct13;
    if(isNANorINF(t->fxu[114])) { PRNT("    @k %d: t->fxu[114] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[114]); return 0; }
    t->fxu[115]= 0.74374999999999991*ct13;
    if(isNANorINF(t->fxu[115])) { PRNT("    @k %d: t->fxu[115] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[115]); return 0; }
    t->fxu[116]= 1.217625*ct13;
    if(isNANorINF(t->fxu[116])) { PRNT("    @k %d: t->fxu[116] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[116]); return 0; }
    t->fxu[117]= -0.53762500000000002*ct13;
    if(isNANorINF(t->fxu[117])) { PRNT("    @k %d: t->fxu[117] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[117]); return 0; }
    t->fxu[118]= -2.4947499999999998*ct13;
    if(isNANorINF(t->fxu[118])) { PRNT("    @k %d: t->fxu[118] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[118]); return 0; }
    t->fxu[119]= 1.132625*ct13;
    if(isNANorINF(t->fxu[119])) { PRNT("    @k %d: t->fxu[119] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[119]); return 0; }
    t->fxu[120]= 0.0046169999999999996*ct14;
    if(isNANorINF(t->fxu[120])) { PRNT("    @k %d: t->fxu[120] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[120]); return 0; }
    t->fxu[121]= -0.10449*ct14;
    if(isNANorINF(t->fxu[121])) { PRNT("    @k %d: t->fxu[121] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[121]); return 0; }
    t->fxu[122]= 0.14458499999999999*ct14;
    if(isNANorINF(t->fxu[122])) { PRNT("    @k %d: t->fxu[122] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[122]); return 0; }
    t->fxu[123]= 0.08990999999999999*ct14;
    if(isNANorINF(t->fxu[123])) { PRNT("    @k %d: t->fxu[123] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[123]); return 0; }
    t->fxu[124]= -0.290385*ct14;
    if(isNANorINF(t->fxu[124])) { PRNT("    @k %d: t->fxu[124] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[124]); return 0; }
    t->fxu[125]= -0.084806999999999994*ct14;
    if(isNANorINF(t->fxu[125])) { PRNT("    @k %d: t->fxu[125] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[125]); return 0; }
    t->fxu[126]= 0.078245999999999996*ct14;
    if(isNANorINF(t->fxu[126])) { PRNT("    @k %d: t->fxu[126] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[126]); return 0; }
    t->fxu[127]= -0.0021869999999999997*ct14;
    if(isNANorINF(t->fxu[127])) { PRNT("    @k %d: t->fxu[127] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[127]); return 0; }
    t->fxu[128]= -0.087965999999999989*ct14;
    if(isNANorINF(t->fxu[128])) { PRNT("    @k %d: t->fxu[128] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[128]); return 0; }
    t->fxu[129]= 0.020412*ct14;
    if(isNANorINF(t->fxu[129])) { PRNT("    @k %d: t->fxu[129] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[129]); return 0; }
    t->fxu[130]= -0.0026410000000000001*ct14;
    if(isNANorINF(t->fxu[130])) { PRNT("    @k %d: t->fxu[130] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[130]); return 0; }
    t->fxu[131]= 0.059770000000000004*ct14;
    if(isNANorINF(t->fxu[131])) { PRNT("    @k %d: t->fxu[131] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[131]); return 0; }
    t->fxu[132]= -0.082705000000000001*ct14;
    if(isNANorINF(t->fxu[132])) { PRNT("    @k %d: t->fxu[132] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[132]); return 0; }
    t->fxu[133]= -0.051430000000000003*ct14;
    if(isNANorINF(t->fxu[133])) { PRNT("    @k %d: t->fxu[133] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[133]); return 0; }
    t->fxu[134]= 0.16610500000000003*ct14;
    if(isNANorINF(t->fxu[134])) { PRNT("    @k %d: t->fxu[134] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[134]); return 0; }
    t->fxu[135]= 0.048510999999999999*ct14;
    if(isNANorINF(t->fxu[135])) { PRNT("    @k %d: t->fxu[135] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[135]); return 0; }
    t->fxu[136]= -0.044758000000000006*ct14;
    if(isNANorINF(t->fxu[136])) { PRNT("    @k %d: t->fxu[136] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[136]); return 0; }
    t->fxu[137]= 0.0012509999999999999*ct14;
    if(isNANorINF(t->fxu[137])) { PRNT("    @k %d: t->fxu[137] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[137]); return 0; }
    t->fxu[138]= 0.050318000000000002*ct14;
    if(isNANorINF(t->fxu[138])) { PRNT("    @k %d: t->fxu[138] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[138]); return 0; }
    t->fxu[139]= -0.011676000000000002*ct14;
    if(isNANorINF(t->fxu[139])) { PRNT("    @k %d: t->fxu[139] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[139]); return 0; }
    t->fxu[140]= 0.0017099999999999999*ct14;
    if(isNANorINF(t->fxu[140])) { PRNT("    @k %d: t->fxu[140] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[140]); return 0; }
    t->fxu[141]= -0.038699999999999998*ct14;
    if(isNANorINF(t->fxu[141])) { PRNT("    @k %d: t->fxu[141] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[141]); return 0; }
    t->fxu[142]= 0.053549999999999993*ct14;
    if(isNANorINF(t->fxu[142])) { PRNT("    @k %d: t->fxu[142] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[142]); return 0; }
    t->fxu[143]= 0.033299999999999996*ct14;
    if(isNANorINF(t->fxu[143])) { PRNT("    @k %d: t->fxu[143] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[143]); return 0; }
    t->fxu[144]= -0.10755000000000001*ct14;
    if(isNANorINF(t->fxu[144])) { PRNT("    @k %d: t->fxu[144] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[144]); return 0; }
    t->fxu[145]= -0.031409999999999993*ct14;
    if(isNANorINF(t->fxu[145])) { PRNT("    @k %d: t->fxu[145] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[145]); return 0; }
    t->fxu[146]= 0.028979999999999999*ct14;
    if(isNANorINF(t->fxu[146])) { PRNT("    @k %d: t->fxu[146] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[146]); return 0; }
    t->fxu[147]= -0.00080999999999999996*ct14;
    if(isNANorINF(t->fxu[147])) { PRNT("    @k %d: t->fxu[147] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[147]); return 0; }
    t->fxu[148]= -0.032579999999999998*ct14;
    if(isNANorINF(t->fxu[148])) { PRNT("    @k %d: t->fxu[148] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[148]); return 0; }
    t->fxu[149]= 0.0075599999999999999*ct14;
    if(isNANorINF(t->fxu[149])) { PRNT("    @k %d: t->fxu[149] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[149]); return 0; }
    t->fxu[150]= 0.096689999999999998*ct15;
    if(isNANorINF(t->fxu[150])) { PRNT("    @k %d: t->fxu[150] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[150]); return 0; }
    t->fxu[151]= 0.072599999999999998*ct15;
    if(isNANorINF(t->fxu[151])) { PRNT("    @k %d: t->fxu[151] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[151]); return 0; }
    t->fxu[152]= 0.2079*ct15;
    if(isNANorINF(t->fxu[152])) { PRNT("    @k %d: t->fxu[152] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[152]); return 0; }
    t->fxu[153]= 0.17688000000000001*ct15;
    if(isNANorINF(t->fxu[153])) { PRNT("    @k %d: t->fxu[153] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[153]); return 0; }
    t->fxu[154]= -0.00066*ct15;
    if(isNANorINF(t->fxu[154])) { PRNT("    @k %d: t->fxu[154] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[154]); return 0; }
    t->fxu[155]= 0.11814*ct15;
    if(isNANorINF(t->fxu[155])) { PRNT("    @k %d: t->fxu[155] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[155]); return 0; }
    t->fxu[156]= -0.015510000000000001*ct15;
    if(isNANorINF(t->fxu[156])) { PRNT("    @k %d: t->fxu[156] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[156]); return 0; }
    t->fxu[157]= -0.11055000000000001*ct15;
    if(isNANorINF(t->fxu[157])) { PRNT("    @k %d: t->fxu[157] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[157]); return 0; }
    t->fxu[158]= 0.41085000000000005*ct15;
    if(isNANorINF(t->fxu[158])) { PRNT("    @k %d: t->fxu[158] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[158]); return 0; }
    t->fxu[159]= 0.014189999999999999*ct15;
    if(isNANorINF(t->fxu[159])) { PRNT("    @k %d: t->fxu[159] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[159]); return 0; }
    t->fxu[160]= -0.30354799999999998*ct15;
    if(isNANorINF(t->fxu[160])) { PRNT("    @k %d: t->fxu[160] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[160]); return 0; }
    t->fxu[161]= -0.22792000000000001*ct15;
    if(isNANorINF(t->fxu[161])) { PRNT("    @k %d: t->fxu[161] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[161]); return 0; }
    t->fxu[162]= -0.65268000000000004*ct15;
    if(isNANorINF(t->fxu[162])) { PRNT("    @k %d: t->fxu[162] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[162]); return 0; }
    t->fxu[163]= -0.55529600000000001*ct15;
    if(isNANorINF(t->fxu[163])) { PRNT("    @k %d: t->fxu[163] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[163]); return 0; }
    t->fxu[164]= 0.0020720000000000001*ct15;
    if(isNANorINF(t->fxu[164])) { PRNT("    @k %d: t->fxu[164] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[164]); return 0; }
    t->fxu[165]= -0.370888*ct15;
    if(isNANorINF(t->fxu[165])) { PRNT("    @k %d: t->fxu[165] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[165]); return 0; }
    t->fxu[166]= 0.048691999999999999*ct15;
    if(isNANorINF(t->fxu[166])) { PRNT("    @k %d: t->fxu[166] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[166]); return 0; }
    t->fxu[167]= 0.34706000000000004*ct15;
    if(isNANorINF(t->fxu[167])) { PRNT("    @k %d: t->fxu[167] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[167]); return 0; }
    t->fxu[168]= -1.2898200000000002*ct15;
    if(isNANorINF(t->fxu[168])) { PRNT("    @k %d: t->fxu[168] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[168]); return 0; }
    t->fxu[169]= -0.044547999999999997*ct15;
    if(isNANorINF(t->fxu[169])) { PRNT("    @k %d: t->fxu[169] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[169]); return 0; }
    t->fxu[170]= -0.32376499999999997*ct15;
    if(isNANorINF(t->fxu[170])) { PRNT("    @k %d: t->fxu[170] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[170]); return 0; }
    t->fxu[171]= -0.24310000000000001*ct15;
    if(isNANorINF(t->fxu[171])) { PRNT("    @k %d: t->fxu[171] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[171]); return 0; }
    t->fxu[172]= -0.69615000000000005*ct15;
    if(isNANorINF(t->fxu[172])) { PRNT("    @k %d: t->fxu[172] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[172]); return 0; }
    t->fxu[173]= -0.59228000000000003*ct15;
    if(isNANorINF(t->fxu[173])) { PRNT("    @k %d: t->fxu[173] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[173]); return 0; }
    t->fxu[174]= 0.0022100000000000002*ct15;
    if(isNANorINF(t->fxu[174])) { PRNT("    @k %d: t->fxu[174] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[174]); return 0; }
    t->fxu[175]= -0.39559*ct15;
    if(isNANorINF(t->fxu[175])) { PRNT("    @k %d: t->fxu[175] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[175]); return 0; }
    t->fxu[176]= 0.051935000000000002*ct15;
    if(isNANorINF(t->fxu[176])) { PRNT("    @k %d: t->fxu[176] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[176]); return 0; }
    t->fxu[177]= 0.37017500000000003*ct15;
    if(isNANorINF(t->fxu[177])) { PRNT("    @k %d: t->fxu[177] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[177]); return 0; }
    t->fxu[178]= -1.3757250000000001*ct15;
    if(isNANorINF(t->fxu[178])) { PRNT("    @k %d: t->fxu[178] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[178]); return 0; }
    t->fxu[179]= -0.047514999999999995*ct15;
    if(isNANorINF(t->fxu[179])) { PRNT("    @k %d: t->fxu[179] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[179]); return 0; }
    t->fxu[180]= -0.60166200000000003*ct16;
    if(isNANorINF(t->fxu[180])) { PRNT("    @k %d: t->fxu[180] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[180]); return 0; }
    t->fxu[181]= 0.12963*ct16;
    if(isNANorINF(t->fxu[181])) { PRNT("    @k %d: t->fxu[181] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[181]); return 0; }
    t->fxu[182]= 0.29680800000000002*ct16;
    if(isNANorINF(t->fxu[182])) { PRNT("    @k %d: t->fxu[182] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[182]); return 0; }
    t->fxu[183]= 0.91098599999999996*ct16;
    if(isNANorINF(t->fxu[183])) { PRNT("    @k %d: t->fxu[183] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[183]); return 0; }
    t->fxu[184]= -0.70715400000000006*ct16;
    if(isNANorINF(t->fxu[184])) { PRNT("    @k %d: t->fxu[184] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[184]); return 0; }
    t->fxu[185]= 0.49974600000000008*ct16;
    if(isNANorINF(t->fxu[185])) { PRNT("    @k %d: t->fxu[185] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[185]); return 0; }
    t->fxu[186]= 0.49170000000000003*ct16;
    if(isNANorINF(t->fxu[186])) { PRNT("    @k %d: t->fxu[186] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[186]); return 0; }
    t->fxu[187]= -0.62669399999999997*ct16;
    if(isNANorINF(t->fxu[187])) { PRNT("    @k %d: t->fxu[187] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[187]); return 0; }
    t->fxu[188]= -0.35581200000000002*ct16;
    if(isNANorINF(t->fxu[188])) { PRNT("    @k %d: t->fxu[188] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[188]); return 0; }
    t->fxu[189]= -0.19667999999999999*ct16;
    if(isNANorINF(t->fxu[189])) { PRNT("    @k %d: t->fxu[189] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[189]); return 0; }
    t->fxu[190]= -0.020863*ct16;
    if(isNANorINF(t->fxu[190])) { PRNT("    @k %d: t->fxu[190] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[190]); return 0; }
    t->fxu[191]= 0.0044949999999999999*ct16;
    if(isNANorINF(t->fxu[191])) { PRNT("    @k %d: t->fxu[191] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[191]); return 0; }
    t->fxu[192]= 0.010292000000000001*ct16;
    if(isNANorINF(t->fxu[192])) { PRNT("    @k %d: t->fxu[192] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[192]); return 0; }
    t->fxu[193]= 0.031588999999999999*ct16;
    if(isNANorINF(t->fxu[193])) { PRNT("    @k %d: t->fxu[193] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[193]); return 0; }
    t->fxu[194]= -0.024521000000000001*ct16;
    if(isNANorINF(t->fxu[194])) { PRNT("    @k %d: t->fxu[194] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[194]); return 0; }
    t->fxu[195]= 0.017329000000000001*ct16;
    if(isNANorINF(t->fxu[195])) { PRNT("    @k %d: t->fxu[195] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[195]); return 0; }
    t->fxu[196]= 0.017050000000000003*ct16;
    if(isNANorINF(t->fxu[196])) { PRNT("    @k %d: t->fxu[196] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[196]); return 0; }
    t->fxu[197]= -0.021730999999999997*ct16;
    if(isNANorINF(t->fxu[197])) { PRNT("    @k %d: t->fxu[197] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[197]); return 0; }
    t->fxu[198]= -0.012338*ct16;
    if(isNANorINF(t->fxu[198])) { PRNT("    @k %d: t->fxu[198] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[198]); return 0; }
    t->fxu[199]= -0.0068199999999999997*ct16;
    if(isNANorINF(t->fxu[199])) { PRNT("    @k %d: t->fxu[199] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[199]); return 0; }
    t->fxu[200]= 0.028266000000000003*ct16;
    if(isNANorINF(t->fxu[200])) { PRNT("    @k %d: t->fxu[200] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[200]); return 0; }
    t->fxu[201]= -0.0060899999999999999*ct16;
    if(isNANorINF(t->fxu[201])) { PRNT("    @k %d: t->fxu[201] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[201]); return 0; }
    t->fxu[202]= -0.013944000000000002*ct16;
    if(isNANorINF(t->fxu[202])) { PRNT("    @k %d: t->fxu[202] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[202]); return 0; }
    t->fxu[203]= -0.042797999999999996*ct16;
    if(isNANorINF(t->fxu[203])) { PRNT("    @k %d: t->fxu[203] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[203]); return 0; }
    t->fxu[204]= 0.033222000000000002*ct16;
    if(isNANorINF(t->fxu[204])) { PRNT("    @k %d: t->fxu[204] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[204]); return 0; }
    t->fxu[205]= -0.023478000000000002*ct16;
    if(isNANorINF(t->fxu[205])) { PRNT("    @k %d: t->fxu[205] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[205]); return 0; }
    t->fxu[206]= -0.023100000000000002*ct16;
    if(isNANorINF(t->fxu[206])) { PRNT("    @k %d: t->fxu[206] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[206]); return 0; }
    t->fxu[207]= 0.029441999999999999*ct16;
    if(isNANorINF(t->fxu[207])) { PRNT("    @k %d: t->fxu[207] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[207]); return 0; }
    t->fxu[208]= 0.016716000000000002*ct16;
    if(isNANorINF(t->fxu[208])) { PRNT("    @k %d: t->fxu[208] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[208]); return 0; }
    t->fxu[209]= 0.0092399999999999999*ct16;
    if(isNANorINF(t->fxu[209])) { PRNT("    @k %d: t->fxu[209] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[209]); return 0; }
    t->fxu[210]= 0.019656*ct17;
    if(isNANorINF(t->fxu[210])) { PRNT("    @k %d: t->fxu[210] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[210]); return 0; }
    t->fxu[211]= -0.0031319999999999994*ct17;
    if(isNANorINF(t->fxu[211])) { PRNT("    @k %d: t->fxu[211] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[211]); return 0; }
    t->fxu[212]= 0.0079919999999999991*ct17;
    if(isNANorINF(t->fxu[212])) { PRNT("    @k %d: t->fxu[212] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[212]); return 0; }
    t->fxu[213]= 0.018828000000000001*ct17;
    if(isNANorINF(t->fxu[213])) { PRNT("    @k %d: t->fxu[213] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[213]); return 0; }
    t->fxu[214]= 0.0077759999999999991*ct17;
    if(isNANorINF(t->fxu[214])) { PRNT("    @k %d: t->fxu[214] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[214]); return 0; }
    t->fxu[215]= -0.0061919999999999987*ct17;
    if(isNANorINF(t->fxu[215])) { PRNT("    @k %d: t->fxu[215] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[215]); return 0; }
    t->fxu[216]= 0.016451999999999998*ct17;
    if(isNANorINF(t->fxu[216])) { PRNT("    @k %d: t->fxu[216] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[216]); return 0; }
    t->fxu[217]= 0.0062999999999999992*ct17;
    if(isNANorINF(t->fxu[217])) { PRNT("    @k %d: t->fxu[217] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[217]); return 0; }
    t->fxu[218]= 0.0043919999999999992*ct17;
    if(isNANorINF(t->fxu[218])) { PRNT("    @k %d: t->fxu[218] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[218]); return 0; }
    t->fxu[219]= -0.016919999999999998*ct17;
    if(isNANorINF(t->fxu[219])) { PRNT("    @k %d: t->fxu[219] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[219]); return 0; }
    t->fxu[220]= -0.49413000000000007*ct17;
    if(isNANorINF(t->fxu[220])) { PRNT("    @k %d: t->fxu[220] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[220]); return 0; }
    t->fxu[221]= 0.078734999999999999*ct17;
    if(isNANorINF(t->fxu[221])) { PRNT("    @k %d: t->fxu[221] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[221]); return 0; }
    t->fxu[222]= -0.20091000000000001*ct17;
    if(isNANorINF(t->fxu[222])) { PRNT("    @k %d: t->fxu[222] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[222]); return 0; }
    t->fxu[223]= -0.47331500000000004*ct17;
    if(isNANorINF(t->fxu[223])) { PRNT("    @k %d: t->fxu[223] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[223]); return 0; }
    t->fxu[224]= -0.19548000000000001*ct17;
    if(isNANorINF(t->fxu[224])) { PRNT("    @k %d: t->fxu[224] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[224]); return 0; }
    t->fxu[225]= 0.15565999999999999*ct17;
    if(isNANorINF(t->fxu[225])) { PRNT("    @k %d: t->fxu[225] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[225]); return 0; }
    t->fxu[226]= -0.41358500000000004*ct17;
    if(isNANorINF(t->fxu[226])) { PRNT("    @k %d: t->fxu[226] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[226]); return 0; }
    t->fxu[227]= -0.15837499999999999*ct17;
    if(isNANorINF(t->fxu[227])) { PRNT("    @k %d: t->fxu[227] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[227]); return 0; }
    t->fxu[228]= -0.11040999999999999*ct17;
    if(isNANorINF(t->fxu[228])) { PRNT("    @k %d: t->fxu[228] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[228]); return 0; }
    t->fxu[229]= 0.42535000000000001*ct17;
    if(isNANorINF(t->fxu[229])) { PRNT("    @k %d: t->fxu[229] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[229]); return 0; }
    t->fxu[230]= -0.22986600000000001*ct17;
    if(isNANorINF(t->fxu[230])) { PRNT("    @k %d: t->fxu[230] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[230]); return 0; }
    t->fxu[231]= 0.036626999999999993*ct17;
    if(isNANorINF(t->fxu[231])) { PRNT("    @k %d: t->fxu[231] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[231]); return 0; }
    t->fxu[232]= -0.093462000000000003*ct17;
    if(isNANorINF(t->fxu[232])) { PRNT("    @k %d: t->fxu[232] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[232]); return 0; }
    t->fxu[233]= -0.22018299999999999*ct17;
    if(isNANorINF(t->fxu[233])) { PRNT("    @k %d: t->fxu[233] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[233]); return 0; }
    t->fxu[234]= -0.090935999999999989*ct17;
    if(isNANorINF(t->fxu[234])) { PRNT("    @k %d: t->fxu[234] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[234]); return 0; }
    t->fxu[235]= 0.07241199999999999*ct17;
    if(isNANorINF(t->fxu[235])) { PRNT("    @k %d: t->fxu[235] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[235]); return 0; }
    t->fxu[236]= -0.19239700000000001*ct17;
    if(isNANorINF(t->fxu[236])) { PRNT("    @k %d: t->fxu[236] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[236]); return 0; }
    t->fxu[237]= -0.07367499999999999*ct17;
    if(isNANorINF(t->fxu[237])) { PRNT("    @k %d: t->fxu[237] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[237]); return 0; }
    t->fxu[238]= -0.051361999999999998*ct17;
    if(isNANorINF(t->fxu[238])) { PRNT("    @k %d: t->fxu[238] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[238]); return 0; }
    t->fxu[239]= 0.19786999999999999*ct17;
    if(isNANorINF(t->fxu[239])) { PRNT("    @k %d: t->fxu[239] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[239]); return 0; }
    t->fxu[240]= 0.36675999999999997*ct18;
    if(isNANorINF(t->fxu[240])) { PRNT("    @k %d: t->fxu[240] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[240]); return 0; }
    t->fxu[241]= 0.007611999999999999*ct18;
    if(isNANorINF(t->fxu[241])) { PRNT("    @k %d: t->fxu[241] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[241]); return 0; }
    t->fxu[242]= 0.20552399999999998*ct18;
    if(isNANorINF(t->fxu[242])) { PRNT("    @k %d: t->fxu[242] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[242]); return 0; }
    t->fxu[243]= -0.231128*ct18;
    if(isNANorINF(t->fxu[243])) { PRNT("    @k %d: t->fxu[243] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[243]); return 0; }
    t->fxu[244]= 1.1251919999999997*ct18;
    if(isNANorINF(t->fxu[244])) { PRNT("    @k %d: t->fxu[244] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[244]); return 0; }
    t->fxu[245]= 0.26918799999999998*ct18;
    if(isNANorINF(t->fxu[245])) { PRNT("    @k %d: t->fxu[245] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[245]); return 0; }
    t->fxu[246]= -0.24358399999999997*ct18;
    if(isNANorINF(t->fxu[246])) { PRNT("    @k %d: t->fxu[246] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[246]); return 0; }
    t->fxu[247]= 0.029755999999999994*ct18;
    if(isNANorINF(t->fxu[247])) { PRNT("    @k %d: t->fxu[247] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[247]); return 0; }
    t->fxu[248]= 0.24219999999999997*ct18;
    if(isNANorINF(t->fxu[248])) { PRNT("    @k %d: t->fxu[248] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[248]); return 0; }
    t->fxu[249]= -0.5100039999999999*ct18;
    if(isNANorINF(t->fxu[249])) { PRNT("    @k %d: t->fxu[249] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[249]); return 0; }
    t->fxu[250]= 0.072610000000000008*ct18;
    if(isNANorINF(t->fxu[250])) { PRNT("    @k %d: t->fxu[250] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[250]); return 0; }
    t->fxu[251]= 0.0015070000000000001*ct18;
    if(isNANorINF(t->fxu[251])) { PRNT("    @k %d: t->fxu[251] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[251]); return 0; }
    t->fxu[252]= 0.040689000000000003*ct18;
    if(isNANorINF(t->fxu[252])) { PRNT("    @k %d: t->fxu[252] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[252]); return 0; }
    t->fxu[253]= -0.045758000000000007*ct18;
    if(isNANorINF(t->fxu[253])) { PRNT("    @k %d: t->fxu[253] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[253]); return 0; }
    t->fxu[254]= 0.22276200000000002*ct18;
    if(isNANorINF(t->fxu[254])) { PRNT("    @k %d: t->fxu[254] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[254]); return 0; }
    t->fxu[255]= 0.053293000000000007*ct18;
    if(isNANorINF(t->fxu[255])) { PRNT("    @k %d: t->fxu[255] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[255]); return 0; }
    t->fxu[256]= -0.048224000000000003*ct18;
    if(isNANorINF(t->fxu[256])) { PRNT("    @k %d: t->fxu[256] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[256]); return 0; }
    t->fxu[257]= 0.0058910000000000004*ct18;
    if(isNANorINF(t->fxu[257])) { PRNT("    @k %d: t->fxu[257] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[257]); return 0; }
    t->fxu[258]= 0.04795*ct18;
    if(isNANorINF(t->fxu[258])) { PRNT("    @k %d: t->fxu[258] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[258]); return 0; }
    t->fxu[259]= -0.100969*ct18;
    if(isNANorINF(t->fxu[259])) { PRNT("    @k %d: t->fxu[259] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[259]); return 0; }
    t->fxu[260]= -0.031269999999999999*ct18;
    if(isNANorINF(t->fxu[260])) { PRNT("    @k %d: t->fxu[260] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[260]); return 0; }
    t->fxu[261]= -0.00064899999999999995*ct18;
    if(isNANorINF(t->fxu[261])) { PRNT("    @k %d: t->fxu[261] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[261]); return 0; }
    t->fxu[262]= -0.017522999999999997*ct18;
    if(isNANorINF(t->fxu[262])) { PRNT("    @k %d: t->fxu[262] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[262]); return 0; }
    t->fxu[263]= 0.019706000000000001*ct18;
    if(isNANorINF(t->fxu[263])) { PRNT("    @k %d: t->fxu[263] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[263]); return 0; }
    t->fxu[264]= -0.095933999999999992*ct18;
    if(isNANorINF(t->fxu[264])) { PRNT("    @k %d: t->fxu[264] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[264]); return 0; }
    t->fxu[265]= -0.022950999999999999*ct18;
    if(isNANorINF(t->fxu[265])) { PRNT("    @k %d: t->fxu[265] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[265]); return 0; }
    t->fxu[266]= 0.020767999999999998*ct18;
    if(isNANorINF(t->fxu[266])) { PRNT("    @k %d: t->fxu[266] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[266]); return 0; }
    t->fxu[267]= -0.0025369999999999998*ct18;
    if(isNANorINF(t->fxu[267])) { PRNT("    @k %d: t->fxu[267] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[267]); return 0; }
    t->fxu[268]= -0.020649999999999998*ct18;
    if(isNANorINF(t->fxu[268])) { PRNT("    @k %d: t->fxu[268] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[268]); return 0; }
    t->fxu[269]= 0.043482999999999994*ct18;
    if(isNANorINF(t->fxu[269])) { PRNT("    @k %d: t->fxu[269] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[269]); return 0; }
    t->fxu[270]= -0.14729*ct19;
    if(isNANorINF(t->fxu[270])) { PRNT("    @k %d: t->fxu[270] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[270]); return 0; }
    t->fxu[271]= -0.045473999999999994*ct19;
    if(isNANorINF(t->fxu[271])) { PRNT("    @k %d: t->fxu[271] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[271]); return 0; }
    t->fxu[272]= -0.054625999999999994*ct19;
    if(isNANorINF(t->fxu[272])) { PRNT("    @k %d: t->fxu[272] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[272]); return 0; }
    t->fxu[273]= -0.10024299999999998*ct19;
    if(isNANorINF(t->fxu[273])) { PRNT("    @k %d: t->fxu[273] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[273]); return 0; }
    t->fxu[274]= -0.048190999999999998*ct19;
    if(isNANorINF(t->fxu[274])) { PRNT("    @k %d: t->fxu[274] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[274]); return 0; }
    t->fxu[275]= -0.043614999999999994*ct19;
    if(isNANorINF(t->fxu[275])) { PRNT("    @k %d: t->fxu[275] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[275]); return 0; }
    t->fxu[276]= -0.062490999999999998*ct19;
    if(isNANorINF(t->fxu[276])) { PRNT("    @k %d: t->fxu[276] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[276]); return 0; }
    t->fxu[277]= -0.106678*ct19;
    if(isNANorINF(t->fxu[277])) { PRNT("    @k %d: t->fxu[277] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[277]); return 0; }
    t->fxu[278]= 0.071214*ct19;
    if(isNANorINF(t->fxu[278])) { PRNT("    @k %d: t->fxu[278] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[278]); return 0; }
    t->fxu[279]= 0.071214*ct19;
    if(isNANorINF(t->fxu[279])) { PRNT("    @k %d: t->fxu[279] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[279]); return 0; }
    t->fxu[280]= -1.96112*ct19;
    if(isNANorINF(t->fxu[280])) { PRNT("    @k %d: t->fxu[280] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[280]); return 0; }
    t->fxu[281]= -0.60547200000000001*ct19;
    if(isNANorINF(t->fxu[281])) { PRNT("    @k %d: t->fxu[281] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[281]); return 0; }
    t->fxu[282]= -0.72732799999999997*ct19;
    if(isNANorINF(t->fxu[282])) { PRNT("    @k %d: t->fxu[282] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[282]); return 0; }
    t->fxu[283]= -1.3347039999999999*ct19;
    if(isNANorINF(t->fxu[283])) { PRNT("    @k %d: t->fxu[283] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[283]); return 0; }
    t->fxu[284]= -0.641648*ct19;
    if(isNANorINF(t->fxu[284])) { PRNT("    @k %d: t->fxu[284] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[284]); return 0; }
    t->fxu[285]= -0.58072000000000001*ct19;
    if(isNANorINF(t->fxu[285])) { PRNT("    @k %d: t->fxu[285] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[285]); return 0; }
    t->fxu[286]= -0.83204800000000001*ct19;
    if(isNANorINF(t->fxu[286])) { PRNT("    @k %d: t->fxu[286] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[286]); return 0; }
    t->fxu[287]= -1.4203839999999999*ct19;
    if(isNANorINF(t->fxu[287])) { PRNT("    @k %d: t->fxu[287] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[287]); return 0; }
    t->fxu[288]= 0.94819199999999992*ct19;
    if(isNANorINF(t->fxu[288])) { PRNT("    @k %d: t->fxu[288] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[288]); return 0; }
    t->fxu[289]= 0.94819199999999992*ct19;
    if(isNANorINF(t->fxu[289])) { PRNT("    @k %d: t->fxu[289] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[289]); return 0; }
    t->fxu[290]= -0.25750000000000001*ct19;
    if(isNANorINF(t->fxu[290])) { PRNT("    @k %d: t->fxu[290] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[290]); return 0; }
    t->fxu[291]= -0.079500000000000001*ct19;
    if(isNANorINF(t->fxu[291])) { PRNT("    @k %d: t->fxu[291] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[291]); return 0; }
    t->fxu[292]= -0.095500000000000002*ct19;
    if(isNANorINF(t->fxu[292])) { PRNT("    @k %d: t->fxu[292] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[292]); return 0; }
    t->fxu[293]= -0.17524999999999999*ct19;
    if(isNANorINF(t->fxu[293])) { PRNT("    @k %d: t->fxu[293] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[293]); return 0; }
    t->fxu[294]= -0.084250000000000005*ct19;
    if(isNANorINF(t->fxu[294])) { PRNT("    @k %d: t->fxu[294] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[294]); return 0; }
    t->fxu[295]= -0.076249999999999998*ct19;
    if(isNANorINF(t->fxu[295])) { PRNT("    @k %d: t->fxu[295] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[295]); return 0; }
    t->fxu[296]= -0.10925*ct19;
    if(isNANorINF(t->fxu[296])) { PRNT("    @k %d: t->fxu[296] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[296]); return 0; }
    t->fxu[297]= -0.1865*ct19;
    if(isNANorINF(t->fxu[297])) { PRNT("    @k %d: t->fxu[297] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[297]); return 0; }
    t->fxu[298]= 0.1245*ct19;
    if(isNANorINF(t->fxu[298])) { PRNT("    @k %d: t->fxu[298] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[298]); return 0; }
    t->fxu[299]= 0.1245*ct19;
    if(isNANorINF(t->fxu[299])) { PRNT("    @k %d: t->fxu[299] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxu[299]); return 0; }

    return 1;
}
#endif

static int bp_derivsL(trajEl_t *t, int k, double **p) {
    if(!bp_derivsL_first(t, k, p)) return 0;
#if FULL_DDP
    if(!bp_derivsL_second(t, k, p)) return 0;
#endif
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

    t->cx[0]= 2.0*p[4][0]*x[0];
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }
    t->cx[1]= 2.0*p[4][1]*x[1];
    if(isNANorINF(t->cx[1])) { PRNT("    @k %d: t->cx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[1]); return 0; }
    t->cx[2]= 2.0*p[4][2]*x[2];
    if(isNANorINF(t->cx[2])) { PRNT("    @k %d: t->cx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[2]); return 0; }
    t->cx[3]= 2.0*p[4][3]*x[3];
    if(isNANorINF(t->cx[3])) { PRNT("    @k %d: t->cx[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[3]); return 0; }
    t->cx[4]= 2.0*p[4][4]*x[4];
    if(isNANorINF(t->cx[4])) { PRNT("    @k %d: t->cx[4] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[4]); return 0; }
    t->cx[5]= 2.0*p[4][5]*x[5];
    if(isNANorINF(t->cx[5])) { PRNT("    @k %d: t->cx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[5]); return 0; }
    t->cx[6]= 2.0*p[4][6]*x[6];
    if(isNANorINF(t->cx[6])) { PRNT("    @k %d: t->cx[6] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[6]); return 0; }
    t->cx[7]= 2.0*p[4][7]*x[7];
    if(isNANorINF(t->cx[7])) { PRNT("    @k %d: t->cx[7] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[7]); return 0; }
    t->cx[8]= 2.0*p[4][8]*x[8];
    if(isNANorINF(t->cx[8])) { PRNT("    @k %d: t->cx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[8]); return 0; }
    t->cx[9]= 2.0*p[4][9]*x[9];
    if(isNANorINF(t->cx[9])) { PRNT("    @k %d: t->cx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[9]); return 0; }

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

        t->cxx[1]= 0.0;
        t->cxx[3]= 0.0;
        t->cxx[4]= 0.0;
        t->cxx[6]= 0.0;
        t->cxx[7]= 0.0;
        t->cxx[8]= 0.0;
        t->cxx[10]= 0.0;
        t->cxx[11]= 0.0;
        t->cxx[12]= 0.0;
        t->cxx[13]= 0.0;
        t->cxx[15]= 0.0;
        t->cxx[16]= 0.0;
        t->cxx[17]= 0.0;
        t->cxx[18]= 0.0;
        t->cxx[19]= 0.0;
        t->cxx[21]= 0.0;
        t->cxx[22]= 0.0;
        t->cxx[23]= 0.0;
        t->cxx[24]= 0.0;
        t->cxx[25]= 0.0;
        t->cxx[26]= 0.0;
        t->cxx[28]= 0.0;
        t->cxx[29]= 0.0;
        t->cxx[30]= 0.0;
        t->cxx[31]= 0.0;
        t->cxx[32]= 0.0;
        t->cxx[33]= 0.0;
        t->cxx[34]= 0.0;
        t->cxx[36]= 0.0;
        t->cxx[37]= 0.0;
        t->cxx[38]= 0.0;
        t->cxx[39]= 0.0;
        t->cxx[40]= 0.0;
        t->cxx[41]= 0.0;
        t->cxx[42]= 0.0;
        t->cxx[43]= 0.0;
        t->cxx[45]= 0.0;
        t->cxx[46]= 0.0;
        t->cxx[47]= 0.0;
        t->cxx[48]= 0.0;
        t->cxx[49]= 0.0;
        t->cxx[50]= 0.0;
        t->cxx[51]= 0.0;
        t->cxx[52]= 0.0;
        t->cxx[53]= 0.0;


        t->cuu[0]= 2.0*p[6][0];
        if(isNANorINF(t->cuu[0])) { PRNT("    @k %d: t->cuu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[0]); return 0; }
        t->cuu[1]= 0.0;
        t->cuu[2]= 2.0*p[6][1];
        if(isNANorINF(t->cuu[2])) { PRNT("    @k %d: t->cuu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[2]); return 0; }
        t->cuu[3]= 0.0;
        t->cuu[4]= 0.0;
        t->cuu[5]= 2.0*p[6][2];
        if(isNANorINF(t->cuu[5])) { PRNT("    @k %d: t->cuu[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[5]); return 0; }

        t->cxu[0]= 0.0;
        t->cxu[1]= 0.0;
        t->cxu[2]= 0.0;
        t->cxu[3]= 0.0;
        t->cxu[4]= 0.0;
        t->cxu[5]= 0.0;
        t->cxu[6]= 0.0;
        t->cxu[7]= 0.0;
        t->cxu[8]= 0.0;
        t->cxu[9]= 0.0;
        t->cxu[10]= 0.0;
        t->cxu[11]= 0.0;
        t->cxu[12]= 0.0;
        t->cxu[13]= 0.0;
        t->cxu[14]= 0.0;
        t->cxu[15]= 0.0;
        t->cxu[16]= 0.0;
        t->cxu[17]= 0.0;
        t->cxu[18]= 0.0;
        t->cxu[19]= 0.0;
        t->cxu[20]= 0.0;
        t->cxu[21]= 0.0;
        t->cxu[22]= 0.0;
        t->cxu[23]= 0.0;
        t->cxu[24]= 0.0;
        t->cxu[25]= 0.0;
        t->cxu[26]= 0.0;
        t->cxu[27]= 0.0;
        t->cxu[28]= 0.0;
        t->cxu[29]= 0.0;

        /* dynamics */


#if FULL_DDP



#endif
    }
    return 1;
}

static int init_final(trajFin_t *t, tOptSet *o) {
    double **const p= o->p;
    const int k= o->n_hor;


    t->cxx[0]= 2.0*p[4][0];
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }
    t->cxx[1]= 0.0;
    t->cxx[2]= 2.0*p[4][1];
    if(isNANorINF(t->cxx[2])) { PRNT("    @k %d: t->cxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[2]); return 0; }
    t->cxx[3]= 0.0;
    t->cxx[4]= 0.0;
    t->cxx[5]= 2.0*p[4][2];
    if(isNANorINF(t->cxx[5])) { PRNT("    @k %d: t->cxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[5]); return 0; }
    t->cxx[6]= 0.0;
    t->cxx[7]= 0.0;
    t->cxx[8]= 0.0;
    t->cxx[9]= 2.0*p[4][3];
    if(isNANorINF(t->cxx[9])) { PRNT("    @k %d: t->cxx[9] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[9]); return 0; }
    t->cxx[10]= 0.0;
    t->cxx[11]= 0.0;
    t->cxx[12]= 0.0;
    t->cxx[13]= 0.0;
    t->cxx[14]= 2.0*p[4][4];
    if(isNANorINF(t->cxx[14])) { PRNT("    @k %d: t->cxx[14] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[14]); return 0; }
    t->cxx[15]= 0.0;
    t->cxx[16]= 0.0;
    t->cxx[17]= 0.0;
    t->cxx[18]= 0.0;
    t->cxx[19]= 0.0;
    t->cxx[20]= 2.0*p[4][5];
    if(isNANorINF(t->cxx[20])) { PRNT("    @k %d: t->cxx[20] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[20]); return 0; }
    t->cxx[21]= 0.0;
    t->cxx[22]= 0.0;
    t->cxx[23]= 0.0;
    t->cxx[24]= 0.0;
    t->cxx[25]= 0.0;
    t->cxx[26]= 0.0;
    t->cxx[27]= 2.0*p[4][6];
    if(isNANorINF(t->cxx[27])) { PRNT("    @k %d: t->cxx[27] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[27]); return 0; }
    t->cxx[28]= 0.0;
    t->cxx[29]= 0.0;
    t->cxx[30]= 0.0;
    t->cxx[31]= 0.0;
    t->cxx[32]= 0.0;
    t->cxx[33]= 0.0;
    t->cxx[34]= 0.0;
    t->cxx[35]= 2.0*p[4][7];
    if(isNANorINF(t->cxx[35])) { PRNT("    @k %d: t->cxx[35] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[35]); return 0; }
    t->cxx[36]= 0.0;
    t->cxx[37]= 0.0;
    t->cxx[38]= 0.0;
    t->cxx[39]= 0.0;
    t->cxx[40]= 0.0;
    t->cxx[41]= 0.0;
    t->cxx[42]= 0.0;
    t->cxx[43]= 0.0;
    t->cxx[44]= 2.0*p[4][8];
    if(isNANorINF(t->cxx[44])) { PRNT("    @k %d: t->cxx[44] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[44]); return 0; }
    t->cxx[45]= 0.0;
    t->cxx[46]= 0.0;
    t->cxx[47]= 0.0;
    t->cxx[48]= 0.0;
    t->cxx[49]= 0.0;
    t->cxx[50]= 0.0;
    t->cxx[51]= 0.0;
    t->cxx[52]= 0.0;
    t->cxx[53]= 0.0;
    t->cxx[54]= 2.0*p[4][9];
    if(isNANorINF(t->cxx[54])) { PRNT("    @k %d: t->cxx[54] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[54]); return 0; }
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

#if FULL_DDP
/* ---- additive: the second derivatives of the dynamics in factored form (batched back-ends; the
 * reference's solver never reads this).  Every entry of slice i (the second derivatives of f_i) is a number
 * times ONE product shared by the slice,
 *     t->fxx[i*sizeofQxx + e] == ilqg_tensor_coef_xx[i*sizeofQxx + e] * basis[ilqg_tensor_slice_xx[i]]   (likewise fuu, fxu)
 * and bp_tensor_basis() evaluates the ILQG_TENSOR_NBASIS products of one step exactly as bp_derivsL does. */
#ifndef ILQG_BASIS  /* a back-end may define these two before including this file */
#define ILQG_BASIS(index) basis[index]
#define ILQG_BASIS_DONE(count)  /* all products have been assigned */
#endif
static int bp_tensor_basis(double *basis, trajEl_t *t, int k, double **p) {
    const double *const x= t->x;
    const double *const u= t->u;

    /* auxiliaries read here, taken once */
    const double v_aux_s1_0= aux_s1_0, v_aux_s2_0= aux_s2_0, v_aux_s1_1= aux_s1_1, v_aux_s2_1= aux_s2_1;
    const double v_aux_s1_2= aux_s1_2, v_aux_s2_2= aux_s2_2, v_aux_s1_3= aux_s1_3, v_aux_s2_3= aux_s2_3;
    const double v_aux_s1_4= aux_s1_4, v_aux_s2_4= aux_s2_4, v_aux_s1_5= aux_s1_5, v_aux_s2_5= aux_s2_5;
    const double v_aux_s1_6= aux_s1_6, v_aux_s2_6= aux_s2_6, v_aux_s1_7= aux_s1_7, v_aux_s2_7= aux_s2_7;
    const double v_aux_s1_8= aux_s1_8, v_aux_s2_8= aux_s2_8, v_aux_s1_9= aux_s1_9, v_aux_s2_9= aux_s2_9;
    /* ... and their sines and cosines */
    const double sin_v_aux_s1_0= sin(v_aux_s1_0), cos_v_aux_s2_0= cos(v_aux_s2_0), sin_v_aux_s1_1= sin(v_aux_s1_1), cos_v_aux_s2_1= cos(v_aux_s2_1);
    const double sin_v_aux_s1_2= sin(v_aux_s1_2), cos_v_aux_s2_2= cos(v_aux_s2_2), sin_v_aux_s1_3= sin(v_aux_s1_3), cos_v_aux_s2_3= cos(v_aux_s2_3);
    const double sin_v_aux_s1_4= sin(v_aux_s1_4), cos_v_aux_s2_4= cos(v_aux_s2_4), sin_v_aux_s1_5= sin(v_aux_s1_5), cos_v_aux_s2_5= cos(v_aux_s2_5);
    const double sin_v_aux_s1_6= sin(v_aux_s1_6), cos_v_aux_s2_6= cos(v_aux_s2_6), sin_v_aux_s1_7= sin(v_aux_s1_7), cos_v_aux_s2_7= cos(v_aux_s2_7);
    const double sin_v_aux_s1_8= sin(v_aux_s1_8), cos_v_aux_s2_8= cos(v_aux_s2_8), sin_v_aux_s1_9= sin(v_aux_s1_9), cos_v_aux_s2_9= cos(v_aux_s2_9);
    const double sin_v_aux_s2_0= sin(v_aux_s2_0), cos_v_aux_s1_0= cos(v_aux_s1_0), sin_v_aux_s2_1= sin(v_aux_s2_1), cos_v_aux_s1_1= cos(v_aux_s1_1);
    const double sin_v_aux_s2_2= sin(v_aux_s2_2), cos_v_aux_s1_2= cos(v_aux_s1_2), sin_v_aux_s2_3= sin(v_aux_s2_3), cos_v_aux_s1_3= cos(v_aux_s1_3);
    const double sin_v_aux_s2_4= sin(v_aux_s2_4), cos_v_aux_s1_4= cos(v_aux_s1_4), sin_v_aux_s2_5= sin(v_aux_s2_5), cos_v_aux_s1_5= cos(v_aux_s1_5);
    const double sin_v_aux_s2_6= sin(v_aux_s2_6), cos_v_aux_s1_6= cos(v_aux_s1_6), sin_v_aux_s2_7= sin(v_aux_s2_7), cos_v_aux_s1_7= cos(v_aux_s1_7);
    const double sin_v_aux_s2_8= sin(v_aux_s2_8), cos_v_aux_s1_8= cos(v_aux_s1_8), sin_v_aux_s2_9= sin(v_aux_s2_9), cos_v_aux_s1_9= cos(v_aux_s1_9);

    ILQG_BASIS(0)= p[0][0]*p[1][0]*sin_v_aux_s1_0*cos_v_aux_s2_0;
    ILQG_BASIS(1)= p[0][0]*p[1][0]*sin_v_aux_s1_1*cos_v_aux_s2_1;
    if(isNANorINF(ILQG_BASIS(0))) { PRNT("    @k %d: ILQG_BASIS(0) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(0)); return 0; }
    if(isNANorINF(ILQG_BASIS(1))) { PRNT("    @k %d: ILQG_BASIS(1) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(1)); return 0; }
    ILQG_BASIS(2)= p[0][0]*p[1][0]*sin_v_aux_s1_2*cos_v_aux_s2_2;
    ILQG_BASIS(3)= p[0][0]*p[1][0]*sin_v_aux_s1_3*cos_v_aux_s2_3;
    if(isNANorINF(ILQG_BASIS(2))) { PRNT("    @k %d: ILQG_BASIS(2) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(2)); return 0; }
    if(isNANorINF(ILQG_BASIS(3))) { PRNT("    @k %d: ILQG_BASIS(3) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(3)); return 0; }
    ILQG_BASIS(4)= p[0][0]*p[1][0]*sin_v_aux_s1_4*cos_v_aux_s2_4;
    ILQG_BASIS(5)= p[0][0]*p[1][0]*sin_v_aux_s1_5*cos_v_aux_s2_5;
    if(isNANorINF(ILQG_BASIS(4))) { PRNT("    @k %d: ILQG_BASIS(4) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(4)); return 0; }
    if(isNANorINF(ILQG_BASIS(5))) { PRNT("    @k %d: ILQG_BASIS(5) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(5)); return 0; }
    ILQG_BASIS(6)= p[0][0]*p[1][0]*sin_v_aux_s1_6*cos_v_aux_s2_6;
    ILQG_BASIS(7)= p[0][0]*p[1][0]*sin_v_aux_s1_7*cos_v_aux_s2_7;
    if(isNANorINF(ILQG_BASIS(6))) { PRNT("    @k %d: ILQG_BASIS(6) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(6)); return 0; }
    if(isNANorINF(ILQG_BASIS(7))) { PRNT("    @k %d: ILQG_BASIS(7) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(7)); return 0; }
    ILQG_BASIS(8)= p[0][0]*p[1][0]*sin_v_aux_s1_8*cos_v_aux_s2_8;
    ILQG_BASIS(9)= p[0][0]*p[1][0]*sin_v_aux_s1_9*cos_v_aux_s2_9;
    if(isNANorINF(ILQG_BASIS(8))) { PRNT("    @k %d: ILQG_BASIS(8) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(8)); return 0; }
    if(isNANorINF(ILQG_BASIS(9))) { PRNT("    @k %d: ILQG_BASIS(9) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(9)); return 0; }
    ILQG_BASIS(10)= p[0][0]*p[1][0]*sin_v_aux_s2_0*cos_v_aux_s1_0;
    ILQG_BASIS(11)= p[0][0]*p[1][0]*sin_v_aux_s2_1*cos_v_aux_s1_1;
    if(isNANorINF(ILQG_BASIS(10))) { PRNT("    @k %d: ILQG_BASIS(10) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(10)); return 0; }
    if(isNANorINF(ILQG_BASIS(11))) { PRNT("    @k %d: ILQG_BASIS(11) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(11)); return 0; }
    ILQG_BASIS(12)= p[0][0]*p[1][0]*sin_v_aux_s2_2*cos_v_aux_s1_2;
    ILQG_BASIS(13)= p[0][0]*p[1][0]*sin_v_aux_s2_3*cos_v_aux_s1_3;
    if(isNANorINF(ILQG_BASIS(12))) { PRNT("    @k %d: ILQG_BASIS(12) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(12)); return 0; }
    if(isNANorINF(ILQG_BASIS(13))) { PRNT("    @k %d: ILQG_BASIS(13) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(13)); return 0; }
    ILQG_BASIS(14)= p[0][0]*p[1][0]*sin_v_aux_s2_4*cos_v_aux_s1_4;
    ILQG_BASIS(15)= p[0][0]*p[1][0]*sin_v_aux_s2_5*cos_v_aux_s1_5;
    if(isNANorINF(ILQG_BASIS(14))) { PRNT("    @k %d: ILQG_BASIS(14) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(14)); return 0; }
    if(isNANorINF(ILQG_BASIS(15))) { PRNT("    @k %d: ILQG_BASIS(15) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(15)); return 0; }
    ILQG_BASIS(16)= p[0][0]*p[1][0]*sin_v_aux_s2_6*cos_v_aux_s1_6;
    ILQG_BASIS(17)= p[0][0]*p[1][0]*sin_v_aux_s2_7*cos_v_aux_s1_7;
    if(isNANorINF(ILQG_BASIS(16))) { PRNT("    @k %d: ILQG_BASIS(16) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(16)); return 0; }
    if(isNANorINF(ILQG_BASIS(17))) { PRNT("    @k %d: ILQG_BASIS(17) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(17)); return 0; }
    ILQG_BASIS(18)= p[0][0]*p[1][0]*sin_v_aux_s2_8*cos_v_aux_s1_8;
    ILQG_BASIS(19)= p[0][0]*p[1][0]*sin_v_aux_s2_9*cos_v_aux_s1_9;
    if(isNANorINF(ILQG_BASIS(18))) { PRNT("    @k %d: ILQG_BASIS(18) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(18)); return 0; }
    if(isNANorINF(ILQG_BASIS(19))) { PRNT("    @k %d: ILQG_BASIS(19) in line %d is nan or inf: %g\n", k, __LINE__-2, ILQG_BASIS(19)); return 0; }
    ILQG_BASIS_DONE(20)
    return 1;
}

static const double ilqg_tensor_coef_xx[550]= {
    -0.0021159999999999998, -0.0027599999999999999, -0.0035999999999999999, -0.027231999999999999, -0.035519999999999996, -0.35046399999999994,
    -0.01541, -0.0201, -0.19832, -0.11222500000000002, -0.028059999999999998, -0.036600000000000001,
    -0.36112, -0.20435, -0.37209999999999999, -0.005888, -0.0076800000000000002, -0.075775999999999996,
    -0.042880000000000001, -0.078079999999999997, -0.016383999999999999, -0.025254000000000002, -0.032940000000000004, -0.32500800000000002,
    -0.18391500000000002, -0.33489000000000002, -0.070272000000000001, -0.30140100000000003, 0.0054279999999999997, 0.0070799999999999995,
    0.069855999999999988, 0.039530000000000003, 0.071979999999999988, 0.015103999999999999, 0.064782000000000006, -0.013923999999999999,
    -0.027784, -0.036239999999999994, -0.357568, -0.20233999999999999, -0.36843999999999999, -0.077312000000000006,
    -0.331596, 0.071271999999999988, -0.36481599999999997, -0.029256000000000001, -0.038159999999999999, -0.37651200000000001,
    -0.21306000000000003, -0.38795999999999997, -0.081408000000000008, -0.34916400000000003, 0.075048000000000004, -0.38414399999999999,
    -0.40449600000000002, -0.71402499999999991, 0.35574499999999998, -0.17724099999999998, 0.29405999999999999, -0.14650799999999997,
    -0.12110399999999999, 0.31940999999999997, -0.159138, -0.13154399999999999, -0.14288400000000001, -0.060839999999999991,
    0.030311999999999995, 0.025055999999999995, 0.027215999999999997, -0.0051839999999999994, 0.53150500000000001, -0.26480900000000002,
    -0.21889199999999998, -0.237762, 0.045287999999999995, -0.39564100000000002, -0.43432999999999999, 0.216394,
    0.178872, 0.19429199999999999, -0.037007999999999999, 0.32330599999999998, -0.26419599999999999, -0.098019999999999996,
    0.048835999999999997, 0.040368000000000001, 0.043848000000000005, -0.008352, 0.072964000000000001, -0.059624000000000003,
    -0.013456000000000001, 0.164775, -0.082095000000000001, -0.067860000000000004, -0.073709999999999998, 0.014039999999999999,
    -0.122655, 0.10023, 0.022620000000000001, -0.038025000000000003, -0.73599499999999995, 0.36669099999999999,
    0.30310799999999999, 0.32923799999999998, -0.06271199999999999, 0.54785899999999998, -0.44769400000000004, -0.101036,
    0.169845, -0.75864100000000001, -0.56851600000000002, -0.19377800000000001, -0.066048999999999997, 0.049764000000000003,
    0.016962000000000001, -0.0043560000000000005, 0.49914800000000004, 0.17013400000000001, -0.043692000000000002, -0.43824400000000002,
    0.80979600000000007, 0.27601800000000004, -0.070884000000000003, -0.71098800000000006, -1.1534760000000002, 0.38680200000000003,
    0.13184100000000001, -0.033857999999999999, -0.33960600000000002, -0.55096200000000006, -0.26316899999999999, -0.19377800000000001,
    -0.066048999999999997, 0.016962000000000001, 0.17013400000000001, 0.27601800000000004, 0.13184100000000001, -0.066048999999999997,
    -0.38001600000000002, -0.129528, 0.033264000000000002, 0.333648, 0.541296, 0.258552,
    -0.129528, -0.25401600000000002, -0.26389999999999997, -0.089950000000000002, 0.023099999999999999, 0.23169999999999999,
    0.37590000000000001, 0.17954999999999999, -0.089950000000000002, -0.1764, -0.12249999999999998, 0.16738800000000001,
    0.057054000000000001, -0.014652, -0.14696400000000001, -0.23842800000000003, -0.113886, 0.057054000000000001,
    0.111888, 0.077699999999999991, -0.049284000000000001, -0.006241, -0.032548000000000001, -0.16974399999999998,
    0.087137000000000006, 0.45443599999999995, -1.2166090000000001, 0.0033180000000000002, 0.017304, -0.046325999999999999,
    -0.0017640000000000002, 0.0039500000000000004, 0.0206, -0.055150000000000005, -0.0021000000000000003, -0.0025000000000000005,
    -0.027649999999999997, -0.14419999999999999, 0.38604999999999995, 0.0147, 0.017499999999999998, -0.12249999999999998,
    -0.045266999999999995, -0.23607599999999998, 0.63201899999999989, 0.024066000000000001, 0.028649999999999998, -0.20054999999999998,
    -0.32832899999999993, 0.019987000000000001, 0.104236, -0.279059, -0.010626, -0.012650000000000002,
    0.08854999999999999, 0.14496899999999999, -0.064008999999999996, 0.092745999999999995, 0.48368799999999995, -1.2949219999999999,
    -0.049307999999999998, -0.058700000000000002, 0.41089999999999993, 0.67270199999999991, -0.29702200000000001, -1.3782759999999998,
    -0.042107000000000006, -0.21959600000000001, 0.58789900000000006, 0.022386000000000003, 0.026650000000000004, -0.18654999999999999,
    -0.30540899999999999, 0.134849, 0.62574200000000002, -0.28408900000000004, -0.00036099999999999999, 0.0081700000000000002,
    -0.18489999999999998, -0.011304999999999999, 0.25584999999999997, -0.35402499999999998, -0.0070299999999999998, 0.15909999999999999,
    -0.22014999999999998, -0.13689999999999999, 0.022704999999999999, -0.51385000000000003, 0.71102500000000002, 0.44215000000000004,
    -1.4280250000000001, 0.0066309999999999997, -0.15006999999999998, 0.20765499999999998, 0.12912999999999999, -0.41705500000000001,
    -0.12180099999999998, -0.0061180000000000002, 0.13846, -0.19159000000000001, -0.11914, 0.38479000000000002,
    0.11237799999999999, -0.10368400000000001, 0.00017099999999999998, -0.0038699999999999997, 0.0053549999999999995, 0.0033299999999999996,
    -0.010754999999999999, -0.0031409999999999997, 0.002898, -8.099999999999999e-5, 0.0068779999999999996, -0.15565999999999999,
    0.21538999999999997, 0.13394, -0.43259000000000003, -0.12633799999999998, 0.116564, -0.0032579999999999996,
    -0.13104399999999999, -0.001596, 0.036119999999999999, -0.049980000000000004, -0.03108, 0.10038000000000001,
    0.029315999999999998, -0.027048000000000003, 0.00075599999999999994, 0.030408000000000001, -0.0070560000000000006, -0.085848999999999995,
    -0.064460000000000003, -0.048399999999999999, -0.18459, -0.1386, -0.39690000000000003, -0.15704799999999999,
    -0.11792000000000001, -0.33768000000000004, -0.28729600000000005, 0.00058599999999999993, 0.00044000000000000002, 0.0012600000000000001,
    0.001072, -3.9999999999999998e-6, -0.10489399999999999, -0.078759999999999997, -0.22553999999999999, -0.191888,
    0.00071599999999999995, -0.128164, 0.013770999999999999, 0.01034, 0.029610000000000001, 0.025192000000000003,
    -9.4000000000000008e-5, 0.016826000000000001, -0.002209, 0.098155000000000006, 0.073700000000000002, 0.21105000000000002,
    0.17956000000000003, -0.00067000000000000002, 0.11993000000000001, -0.015745000000000002, -0.11222500000000002, -0.36478500000000003,
    -0.27390000000000003, -0.7843500000000001, -0.66732000000000014, 0.0024900000000000005, -0.44571, 0.058515000000000005,
    0.41707500000000008, -1.5500250000000002, -0.012598999999999999, -0.0094599999999999997, -0.027089999999999999, -0.023047999999999999,
    8.599999999999999e-5, -0.015393999999999998, 0.0020209999999999998, 0.014404999999999999, -0.053534999999999999, -0.0018489999999999997,
    -0.45292900000000008, 0.097585000000000005, -0.021024999999999999, 0.22343600000000002, -0.048140000000000002, -0.11022400000000002,
    0.68578700000000004, -0.14775499999999997, -0.338308, -1.0383609999999999, -0.53234300000000001, 0.11469499999999999,
    0.26261200000000001, 0.806029, -0.62568100000000004, 0.37620700000000007, -0.081055000000000002, -0.18558800000000003,
    -0.56962100000000004, 0.44216900000000003, -0.31248100000000006, 0.37015000000000003, -0.079750000000000001, -0.18260000000000001,
    -0.56045, 0.43505000000000005, -0.30745000000000006, -0.30250000000000005, -0.471773, 0.10164499999999999,
    0.23273199999999999, 0.71431899999999993, -0.55449099999999996, 0.39185900000000001, 0.38555, -0.49140099999999992,
    -0.26785400000000004, 0.057709999999999997, 0.132136, 0.40556199999999998, -0.31481800000000004, 0.22248200000000004,
    0.21890000000000004, -0.27899800000000002, -0.15840400000000002, -0.14806, 0.031899999999999998, 0.073040000000000008,
    0.22417999999999999, -0.17402000000000001, 0.12298000000000001, 0.12100000000000001, -0.15422, -0.087559999999999999,
    -0.048399999999999999, -0.29811600000000005, 0.047502000000000003, -0.0075689999999999993, -0.12121200000000001, 0.019313999999999998,
    -0.049284000000000001, -0.28555800000000003, 0.045501, -0.116106, -0.27352900000000002, -0.11793600000000001,
    0.018792, -0.047952000000000002, -0.112968, -0.046655999999999996, 0.093911999999999995, -0.014963999999999998,
    0.038183999999999996, 0.089955999999999994, 0.037151999999999998, -0.029583999999999996, -0.24952200000000002, 0.039758999999999996,
    -0.101454, -0.23901100000000003, -0.098712000000000008, 0.078603999999999993, -0.20884900000000001, -0.095549999999999996,
    0.015224999999999997, -0.038849999999999996, -0.091524999999999995, -0.0378, 0.030099999999999995, -0.079975000000000004,
    -0.030624999999999996, -0.066612000000000005, 0.010613999999999998, -0.027084, -0.063806000000000002, -0.026352,
    0.020983999999999999, -0.055753999999999998, -0.021349999999999997, -0.014884, 0.25662000000000001, -0.040889999999999996,
    0.10434, 0.24581, 0.10152, -0.080839999999999995, 0.21479000000000001, 0.08224999999999999,
    0.057339999999999995, -0.22089999999999999, -0.28090000000000004, -0.0058300000000000001, -0.00012099999999999999, -0.15740999999999999,
    -0.0032669999999999995, -0.088208999999999996, 0.17702000000000001, 0.0036740000000000002, 0.099197999999999995, -0.11155600000000002,
    -0.86177999999999999, -0.017885999999999999, -0.48292199999999996, 0.54308400000000001, -2.6438759999999997, -0.20617000000000002,
    -0.0042789999999999998, -0.115533, 0.12992600000000001, -0.63251400000000002, -0.15132100000000001, 0.18656,
    0.0038719999999999996, 0.10454399999999998, -0.11756800000000001, 0.57235199999999997, 0.13692799999999999, -0.12390399999999999,
    -0.022789999999999998, -0.00047299999999999995, -0.012770999999999998, 0.014362, -0.069917999999999994, -0.016726999999999999,
    0.015135999999999998, -0.0018489999999999997, -0.1855, -0.0038499999999999997, -0.10394999999999999, 0.1169,
    -0.56909999999999994, -0.13614999999999999, 0.12319999999999999, -0.015049999999999997, -0.12249999999999998, 0.39061000000000001,
    0.0081069999999999996, 0.218889, -0.24615800000000002, 1.1983619999999999, 0.28669300000000003, -0.25942399999999999,
    0.031690999999999997, 0.25794999999999996, -0.54316900000000001, -1.0609, -0.32754, -0.10112400000000001,
    -0.39346000000000003, -0.121476, -0.145924, -0.72202999999999995, -0.22291799999999998, -0.26778199999999996,
    -0.49140099999999992, -0.34711000000000003, -0.10716600000000001, -0.12873400000000002, -0.236237, -0.11356900000000002,
    -0.31414999999999998, -0.096989999999999993, -0.11651, -0.213805, -0.102785, -0.093024999999999997,
    -0.45011000000000001, -0.13896600000000001, -0.166934, -0.30633699999999997, -0.14726900000000001, -0.13328499999999999,
    -0.190969, -0.76838000000000006, -0.23722799999999999, -0.284972, -0.52294599999999991, -0.25140200000000001,
    -0.22752999999999998, -0.32600200000000001, -0.55651600000000001, 0.51294000000000006, 0.158364, 0.19023600000000002,
    0.34909799999999996, 0.167826, 0.15189, 0.21762599999999999, 0.371508, -0.248004,
    0.51294000000000006, 0.158364, 0.19023600000000002, 0.34909799999999996, 0.167826, 0.15189,
    0.21762599999999999, 0.371508, -0.248004, -0.248004,
};
static const int ilqg_tensor_slice_xx[N_X]= {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
static const double ilqg_tensor_coef_uu[60]= {
    -0.84272400000000003, 0.36995400000000006, -0.16240900000000003, 0.18451800000000002, -0.081003000000000006, -0.040401000000000006,
    -0.96039999999999992, -0.62229999999999996, -0.403225, -0.67619999999999991, -0.43814999999999998, -0.47609999999999991,
    -1.7635840000000003, 1.349248, -1.0322560000000001, -0.47542400000000001, 0.363728, -0.128164,
    -0.54316900000000001, -0.18351300000000001, -0.062001000000000001, -1.566125, -0.52912499999999996, -4.515625,
    -0.059048999999999997, 0.033777000000000001, -0.019321000000000005, -0.021869999999999997, 0.01251, -0.0080999999999999996,
    -0.10890000000000001, 0.34188000000000002, -1.073296, 0.36465000000000003, -1.1447799999999999, -1.221025,
    -0.79923600000000006, -0.027713999999999999, -0.00096099999999999994, 0.037548000000000005, 0.001302, -0.0017640000000000002,
    -0.0012959999999999998, 0.032579999999999998, -0.819025, 0.015155999999999998, -0.38100499999999998, -0.17724099999999998,
    -0.47886399999999996, -0.094803999999999999, -0.018769000000000004, 0.040827999999999996, 0.0080829999999999999, -0.0034809999999999997,
    -0.020448999999999995, -0.27227199999999996, -3.6252159999999995, -0.035749999999999997, -0.47599999999999998, -0.0625,
};
static const int ilqg_tensor_slice_uu[N_X]= {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
static const double ilqg_tensor_coef_xu[300]= {
    0.042228000000000002, 0.055079999999999997, 0.54345600000000005, 0.30753000000000003, 0.55998000000000003, 0.11750400000000001,
    0.50398200000000004, -0.108324, 0.55447199999999996, 0.58384800000000003, -0.018538000000000002, -0.02418,
    -0.23857600000000001, -0.13500500000000001, -0.24583000000000002, -0.051584000000000005, -0.22124700000000003, 0.047553999999999999,
    -0.24341200000000002, -0.25630800000000004, -0.0092460000000000007, -0.01206, -0.118992, -0.067335000000000006,
    -0.12261000000000001, -0.025728000000000001, -0.11034900000000002, 0.023717999999999999, -0.121404, -0.12783600000000001,
    0.82809999999999995, -0.41258, -0.34103999999999995, -0.37043999999999999, 0.070559999999999998, -0.61641999999999997,
    0.50372000000000006, 0.11368, -0.19109999999999999, 0.85358000000000001, 0.53657500000000002, -0.26733499999999999,
    -0.22097999999999998, -0.24002999999999999, 0.045719999999999997, -0.39941500000000002, 0.32639000000000001, 0.073660000000000003,
    -0.123825, 0.55308500000000005, 0.58304999999999996, -0.29048999999999997, -0.24011999999999997, -0.26082,
    0.049679999999999995, -0.43400999999999995, 0.35465999999999998, 0.08004, -0.13455, 0.60098999999999991,
    1.001312, 0.34129600000000004, -0.087648000000000004, -0.87913600000000014, -1.4262720000000002, -0.68126400000000009,
    0.34129600000000004, 0.66931200000000002, 0.46479999999999999, -0.29481600000000002, -0.76606399999999997, -0.26111200000000001,
    0.067056000000000004, 0.67259200000000008, 1.0911840000000002, 0.521208, -0.26111200000000001, -0.51206399999999996,
    -0.35559999999999997, 0.225552, 0.26993200000000001, 0.092006000000000004, -0.023628, -0.23699600000000001,
    -0.384492, -0.18365399999999998, 0.092006000000000004, 0.18043199999999998, 0.12529999999999999, -0.079475999999999991,
    0.058222999999999997, 0.30364399999999997, -0.81291099999999994, -0.030954000000000002, -0.036850000000000001, 0.25794999999999996,
    0.42230099999999998, -0.18646099999999999, -0.86523799999999995, 0.39282100000000003, 0.019671000000000001, 0.102588,
    -0.27464699999999997, -0.010458, -0.012450000000000001, 0.087149999999999991, 0.142677, -0.062996999999999997,
    -0.29232599999999997, 0.132717, 0.167875, 0.87549999999999994, -2.3438750000000002, -0.08925000000000001,
    -0.10625000000000001, 0.74374999999999991, 1.217625, -0.53762500000000002, -2.4947499999999998, 1.132625,
    0.0046169999999999996, -0.10449, 0.14458499999999999, 0.08990999999999999, -0.290385, -0.084806999999999994,
    0.078245999999999996, -0.0021869999999999997, -0.087965999999999989, 0.020412, -0.0026410000000000001, 0.059770000000000004,
    -0.082705000000000001, -0.051430000000000003, 0.16610500000000003, 0.048510999999999999, -0.044758000000000006, 0.0012509999999999999,
    0.050318000000000002, -0.011676000000000002, 0.0017099999999999999, -0.038699999999999998, 0.053549999999999993, 0.033299999999999996,
    -0.10755000000000001, -0.031409999999999993, 0.028979999999999999, -0.00080999999999999996, -0.032579999999999998, 0.0075599999999999999,
    0.096689999999999998, 0.072599999999999998, 0.2079, 0.17688000000000001, -0.00066, 0.11814,
    -0.015510000000000001, -0.11055000000000001, 0.41085000000000005, 0.014189999999999999, -0.30354799999999998, -0.22792000000000001,
    -0.65268000000000004, -0.55529600000000001, 0.0020720000000000001, -0.370888, 0.048691999999999999, 0.34706000000000004,
    -1.2898200000000002, -0.044547999999999997, -0.32376499999999997, -0.24310000000000001, -0.69615000000000005, -0.59228000000000003,
    0.0022100000000000002, -0.39559, 0.051935000000000002, 0.37017500000000003, -1.3757250000000001, -0.047514999999999995,
    -0.60166200000000003, 0.12963, 0.29680800000000002, 0.91098599999999996, -0.70715400000000006, 0.49974600000000008,
    0.49170000000000003, -0.62669399999999997, -0.35581200000000002, -0.19667999999999999, -0.020863, 0.0044949999999999999,
    0.010292000000000001, 0.031588999999999999, -0.024521000000000001, 0.017329000000000001, 0.017050000000000003, -0.021730999999999997,
    -0.012338, -0.0068199999999999997, 0.028266000000000003, -0.0060899999999999999, -0.013944000000000002, -0.042797999999999996,
    0.033222000000000002, -0.023478000000000002, -0.023100000000000002, 0.029441999999999999, 0.016716000000000002, 0.0092399999999999999,
    0.019656, -0.0031319999999999994, 0.0079919999999999991, 0.018828000000000001, 0.0077759999999999991, -0.0061919999999999987,
    0.016451999999999998, 0.0062999999999999992, 0.0043919999999999992, -0.016919999999999998, -0.49413000000000007, 0.078734999999999999,
    -0.20091000000000001, -0.47331500000000004, -0.19548000000000001, 0.15565999999999999, -0.41358500000000004, -0.15837499999999999,
    -0.11040999999999999, 0.42535000000000001, -0.22986600000000001, 0.036626999999999993, -0.093462000000000003, -0.22018299999999999,
    -0.090935999999999989, 0.07241199999999999, -0.19239700000000001, -0.07367499999999999, -0.051361999999999998, 0.19786999999999999,
    0.36675999999999997, 0.007611999999999999, 0.20552399999999998, -0.231128, 1.1251919999999997, 0.26918799999999998,
    -0.24358399999999997, 0.029755999999999994, 0.24219999999999997, -0.5100039999999999, 0.072610000000000008, 0.0015070000000000001,
    0.040689000000000003, -0.045758000000000007, 0.22276200000000002, 0.053293000000000007, -0.048224000000000003, 0.0058910000000000004,
    0.04795, -0.100969, -0.031269999999999999, -0.00064899999999999995, -0.017522999999999997, 0.019706000000000001,
    -0.095933999999999992, -0.022950999999999999, 0.020767999999999998, -0.0025369999999999998, -0.020649999999999998, 0.043482999999999994,
    -0.14729, -0.045473999999999994, -0.054625999999999994, -0.10024299999999998, -0.048190999999999998, -0.043614999999999994,
    -0.062490999999999998, -0.106678, 0.071214, 0.071214, -1.96112, -0.60547200000000001,
    -0.72732799999999997, -1.3347039999999999, -0.641648, -0.58072000000000001, -0.83204800000000001, -1.4203839999999999,
    0.94819199999999992, 0.94819199999999992, -0.25750000000000001, -0.079500000000000001, -0.095500000000000002, -0.17524999999999999,
    -0.084250000000000005, -0.076249999999999998, -0.10925, -0.1865, 0.1245, 0.1245,
};
static const int ilqg_tensor_slice_xu[N_X]= {10, 11, 12, 13, 14, 15, 16, 17, 18, 19};
#endif

/* ---- additive: one step of forward_pass in ILQG_ROLLOUT_PARTS independent parts (batched back-ends that put
 * several wavefronts on a trajectory's step; the reference's solver never calls this).  Part r: component r of the
 * dynamics and the summands r, r + N_X, ... of the running cost, term[] indexed by their place in ddpL's sum:
 * t->c == ((term[0] + term[1]) + term[2]) + ...  A NaN or Inf in a guarded value sets bad[0]. */
#define ILQG_ROLLOUT_PARTS 10
#define ILQG_ROLLOUT_TERMS 13
#ifndef ILQG_PART_SIN  /* a back-end may define these two before including this file */
#define ILQG_PART_SIN(v) sin(v)
#define ILQG_PART_COS(v) cos(v)
#endif
#ifndef ILQG_PART_FN  /* ... and the function's storage class / attributes */
#define ILQG_PART_FN static
#endif
typedef struct {
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
} ilqg_step_aux_t;
ILQG_PART_FN void ilqg_step_part(int part, double x_next[], double term[], int bad[], const double *x, const double *u, int k, double **p, int N) {
    ilqg_step_aux_t aux_, *const t= &aux_;

    switch(part) {
    case 0:
        aux_s1_0= -0.045999999999999999*x[0] - 0.059999999999999998*x[1] - 0.59199999999999997*x[2] - 0.33500000000000002*x[3] - 0.60999999999999999*x[4] - 0.128*x[5] - 0.54900000000000004*x[6] + 0.11799999999999999*x[7] - 0.60399999999999998*x[8] - 0.63600000000000001*x[9];
        if(!(fabs(aux_s1_0) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_0= 0.91800000000000004*u[0] - 0.40300000000000002*u[1] - 0.20100000000000001*u[2];
        if(!(fabs(aux_s2_0) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[0]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_0)*ILQG_PART_COS(aux_s2_0) - 0.69999999999999996*u[0] + 0.30599999999999999*u[1] + 0.151*u[2] - 1.417*x[0] + 0.089999999999999997*x[1] - 0.094*x[2] + 0.096000000000000002*x[3] + 0.29999999999999999*x[4] + 0.309*x[5] - 0.041000000000000002*x[6] + 0.17199999999999999*x[7] - 0.34599999999999997*x[8] + 0.025999999999999999*x[9]) + x[0];
        if(!(fabs(x_next[0]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[0]= p[5][0]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[0]*x[0])));
        term[10]= p[6][0]*(u[0]*u[0]);
        break;
    case 1:
        aux_s1_1= -0.84499999999999997*x[0] + 0.42099999999999999*x[1] + 0.34799999999999998*x[2] + 0.378*x[3] - 0.071999999999999995*x[4] + 0.629*x[5] - 0.51400000000000001*x[6] - 0.11600000000000001*x[7] + 0.19500000000000001*x[8] - 0.871*x[9];
        if(!(fabs(aux_s1_1) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_1= 0.97999999999999998*u[0] + 0.63500000000000001*u[1] + 0.68999999999999995*u[2];
        if(!(fabs(aux_s2_1) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[1]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_1)*ILQG_PART_COS(aux_s2_1) + 0.51000000000000001*u[0] - 0.027*u[1] + 0.122*u[2] - 0.20000000000000001*x[0] - 0.95899999999999996*x[1] - 0.10000000000000001*x[2] - 0.029000000000000001*x[3] - 0.081000000000000003*x[4] - 0.22600000000000001*x[5] - 0.059999999999999998*x[6] - 0.248*x[7] - 0.095000000000000001*x[8] - 0.097000000000000003*x[9]) + x[1];
        if(!(fabs(x_next[1]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[1]= p[5][1]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[1]*x[1])));
        term[11]= p[6][1]*(u[1]*u[1]);
        break;
    case 2:
        aux_s1_2= 0.754*x[0] + 0.25700000000000001*x[1] - 0.066000000000000003*x[2] - 0.66200000000000003*x[3] - 1.0740000000000001*x[4] - 0.51300000000000001*x[5] + 0.25700000000000001*x[6] + 0.504*x[7] + 0.34999999999999998*x[8] - 0.222*x[9];
        if(!(fabs(aux_s1_2) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_2= -1.3280000000000001*u[0] + 1.016*u[1] - 0.35799999999999998*u[2];
        if(!(fabs(aux_s2_2) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[2]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_2)*ILQG_PART_COS(aux_s2_2) + 0.036999999999999998*u[0] + 0.91600000000000004*u[1] - 0.014*u[2] + 0.16600000000000001*x[0] - 0.25800000000000001*x[1] - 1.0529999999999999*x[2] + 0.070000000000000007*x[3] + 0.45100000000000001*x[4] - 0.13400000000000001*x[5] - 0.072999999999999995*x[6] - 0.36299999999999999*x[7] - 0.28100000000000003*x[8] + 0.088999999999999996*x[9]) + x[2];
        if(!(fabs(x_next[2]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[2]= p[5][2]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[2]*x[2])));
        term[12]= p[6][2]*(u[2]*u[2]);
        break;
    case 3:
        aux_s1_3= -0.079000000000000001*x[0] - 0.41199999999999998*x[1] + 1.103*x[2] + 0.042000000000000003*x[3] + 0.050000000000000003*x[4] - 0.34999999999999998*x[5] - 0.57299999999999995*x[6] + 0.253*x[7] + 1.1739999999999999*x[8] - 0.53300000000000003*x[9];
        if(!(fabs(aux_s1_3) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_3= 0.73699999999999999*u[0] + 0.249*u[1] + 2.125*u[2];
        if(!(fabs(aux_s2_3) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[3]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_3)*ILQG_PART_COS(aux_s2_3) - 0.41599999999999998*u[0] + 0.039*u[1] - 0.64600000000000002*u[2] + 0.47799999999999998*x[0] + 0.099000000000000005*x[1] - 0.153*x[2] - 0.72999999999999998*x[3] - 0.26500000000000001*x[4] + 0.23599999999999999*x[5] - 0.53900000000000003*x[6] + 0.217*x[7] - 0.16700000000000001*x[8] + 0.063*x[9]) + x[3];
        if(!(fabs(x_next[3]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[3]= p[5][3]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[3]*x[3])));
        break;
    case 4:
        aux_s1_4= -0.019*x[0] + 0.42999999999999999*x[1] - 0.59499999999999997*x[2] - 0.37*x[3] + 1.1950000000000001*x[4] + 0.34899999999999998*x[5] - 0.32200000000000001*x[6] + 0.0089999999999999993*x[7] + 0.36199999999999999*x[8] - 0.084000000000000005*x[9];
        if(!(fabs(aux_s1_4) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_4= 0.24299999999999999*u[0] - 0.13900000000000001*u[1] + 0.089999999999999997*u[2];
        if(!(fabs(aux_s2_4) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[4]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_4)*ILQG_PART_COS(aux_s2_4) + 0.64600000000000002*u[0] + 0.14899999999999999*u[1] + 0.69899999999999995*u[2] - 0.23999999999999999*x[0] + 0.129*x[1] + 0.029000000000000001*x[2] - 0.17799999999999999*x[3] - 1.095*x[4] - 0.049000000000000002*x[5] - 0.153*x[6] + 0.28299999999999997*x[7] - 0.188*x[8] + 0.23100000000000001*x[9]) + x[4];
        if(!(fabs(x_next[4]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[4]= p[5][4]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[4]*x[4])));
        break;
    case 5:
        aux_s1_5= 0.29299999999999998*x[0] + 0.22*x[1] + 0.63*x[2] + 0.53600000000000003*x[3] - 0.002*x[4] + 0.35799999999999998*x[5] - 0.047*x[6] - 0.33500000000000002*x[7] + 1.2450000000000001*x[8] + 0.042999999999999997*x[9];
        if(!(fabs(aux_s1_5) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_5= -0.33000000000000002*u[0] + 1.036*u[1] + 1.105*u[2];
        if(!(fabs(aux_s2_5) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[5]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_5)*ILQG_PART_COS(aux_s2_5) + 0.35699999999999998*u[0] - 0.51200000000000001*u[1] - 0.379*u[2] - 0.014999999999999999*x[0] - 0.17299999999999999*x[1] - 0.16900000000000001*x[2] + 0.032000000000000001*x[3] - 0.22800000000000001*x[4] - 1.1479999999999999*x[5] - 0.20999999999999999*x[6] - 0.104*x[7] - 0.252*x[8] - 0.031*x[9]) + x[5];
        if(!(fabs(x_next[5]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[5]= p[5][5]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[5]*x[5])));
        break;
    case 6:
        aux_s1_6= 0.67300000000000004*x[0] - 0.14499999999999999*x[1] - 0.33200000000000002*x[2] - 1.0189999999999999*x[3] + 0.79100000000000004*x[4] - 0.55900000000000005*x[5] - 0.55000000000000004*x[6] + 0.70099999999999996*x[7] + 0.39800000000000002*x[8] + 0.22*x[9];
        if(!(fabs(aux_s1_6) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_6= 0.89400000000000002*u[0] + 0.031*u[1] - 0.042000000000000003*u[2];
        if(!(fabs(aux_s2_6) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[6]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_6)*ILQG_PART_COS(aux_s2_6) + 0.76100000000000001*u[0] + 0.064000000000000001*u[1] + 0.086999999999999994*u[2] - 0.113*x[0] + 0.083000000000000004*x[1] + 0.016*x[2] + 0.222*x[3] + 0.099000000000000005*x[4] + 0.10299999999999999*x[5] - 1.161*x[6] + 0.33900000000000002*x[7] + 0.065000000000000002*x[8] - 0.25700000000000001*x[9]) + x[6];
        if(!(fabs(x_next[6]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[6]= p[5][6]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[6]*x[6])));
        break;
    case 7:
        aux_s1_7= -0.54600000000000004*x[0] + 0.086999999999999994*x[1] - 0.222*x[2] - 0.52300000000000002*x[3] - 0.216*x[4] + 0.17199999999999999*x[5] - 0.45700000000000002*x[6] - 0.17499999999999999*x[7] - 0.122*x[8] + 0.46999999999999997*x[9];
        if(!(fabs(aux_s1_7) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_7= 0.035999999999999997*u[0] - 0.90500000000000003*u[1] - 0.42099999999999999*u[2];
        if(!(fabs(aux_s2_7) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[7]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_7)*ILQG_PART_COS(aux_s2_7) - 0.48099999999999998*u[0] - 0.159*u[1] - 0.245*u[2] + 0.307*x[0] - 0.002*x[1] + 0.159*x[2] - 0.125*x[3] - 0.17699999999999999*x[4] + 0.070999999999999994*x[5] + 0.11*x[6] - 1.0600000000000001*x[7] + 0.039*x[8] + 0.129*x[9]) + x[7];
        if(!(fabs(x_next[7]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[7]= p[5][7]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[7]*x[7])));
        break;
    case 8:
        aux_s1_8= 0.53000000000000003*x[0] + 0.010999999999999999*x[1] + 0.29699999999999999*x[2] - 0.33400000000000002*x[3] + 1.6259999999999999*x[4] + 0.38900000000000001*x[5] - 0.35199999999999998*x[6] + 0.042999999999999997*x[7] + 0.34999999999999998*x[8] - 0.73699999999999999*x[9];
        if(!(fabs(aux_s1_8) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_8= -0.69199999999999995*u[0] - 0.13700000000000001*u[1] + 0.058999999999999997*u[2];
        if(!(fabs(aux_s2_8) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[8]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_8)*ILQG_PART_COS(aux_s2_8) + 0.215*u[0] + 0.86899999999999999*u[1] + 1.629*u[2] - 0.014*x[0] - 0.033000000000000002*x[1] + 0.17299999999999999*x[2] - 0.042000000000000003*x[3] - 0.184*x[4] - 0.021000000000000001*x[5] - 0.014999999999999999*x[6] + 0.436*x[7] - 1.2010000000000001*x[8] + 0.122*x[9]) + x[8];
        if(!(fabs(x_next[8]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[8]= p[5][8]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[8]*x[8])));
        break;
    case 9:
        aux_s1_9= 1.03*x[0] + 0.318*x[1] + 0.38200000000000001*x[2] + 0.70099999999999996*x[3] + 0.33700000000000002*x[4] + 0.30499999999999999*x[5] + 0.437*x[6] + 0.746*x[7] - 0.498*x[8] - 0.498*x[9];
        if(!(fabs(aux_s1_9) <= 1.7976931348623157e308)) bad[0]= 1;
        aux_s2_9= 0.14299999999999999*u[0] + 1.9039999999999999*u[1] + 0.25*u[2];
        if(!(fabs(aux_s2_9) <= 1.7976931348623157e308)) bad[0]= 1;
        x_next[9]= p[1][0]*(p[0][0]*ILQG_PART_SIN(aux_s1_9)*ILQG_PART_COS(aux_s2_9) - 0.105*u[0] - 1.7509999999999999*u[1] + 0.24099999999999999*u[2] + 0.096000000000000002*x[0] + 0.037999999999999999*x[1] + 0.025999999999999999*x[2] - 0.13200000000000001*x[3] + 0.23599999999999999*x[4] - 0.032000000000000001*x[5] + 0.222*x[6] - 0.151*x[7] - 0.14699999999999999*x[8] - 1.343*x[9]) + x[9];
        if(!(fabs(x_next[9]) <= 1.7976931348623157e308)) bad[0]= 1;
        term[9]= p[5][9]*(-p[3][0] + sqrt((p[3][0]*p[3][0]) + (x[9]*x[9])));
        break;
    default: break;
    }
}
