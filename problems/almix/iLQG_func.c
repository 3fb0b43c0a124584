/* Problem functions for 'AlMix' emitted by tools/gen_problem.py. Do not edit.
 * Function set, signatures and evaluation order: reference iLQG_func.tem:40-521. */
#include "iLQG.h"
#include "matMult.h"

#define mcond(cond, a, dummy, b) ((cond)? a: b)
#define sec(x) (1.0/cos(x))
#define csc(x) (1.0/sin(x))

int n_params= 7;

tParamDesc p_name1= {"cf", 3, 0};
tParamDesc p_name2= {"cu", 2, 0};
tParamDesc p_name3= {"cx", 3, 0};
tParamDesc p_name4= {"h", 1, 0};
tParamDesc p_name5= {"lim", 2, 0};
tParamDesc p_name6= {"tgt", 3, 0};
tParamDesc p_name7= {"vref", -1, 0};
int n_vars= 0;

tParamDesc *paramdesc[]= {&p_name1, &p_name2, &p_name3, &p_name4, &p_name5, &p_name6, &p_name7};

#define aux_gap t->gap
#define aux_hle_1 t->hle_1
#define aux_ple_1 t->ple_1
#define aux_hli_1 t->hli_1
#define aux_pli_1 t->pli_1
#define aux_hli_2 t->hli_2
#define aux_pli_2 t->pli_2
#define aux_hfe_1 t->hfe_1
#define aux_pfe_1 t->pfe_1
#define aux_hfe_2 t->hfe_2
#define aux_pfe_2 t->pfe_2
#define aux_hfi_1 t->hfi_1
#define aux_pfi_1 t->pfi_1
#define daux_dhle_1_x1 t->dhle_1_x1
#define daux_dple_1_x1 t->dple_1_x1
#define daux_dpli_1_x1 t->dpli_1_x1
#define daux_dpli_2_x1 t->dpli_2_x1
#define daux_dple_1_u1 t->dple_1_u1
#define daux_dple_1_x1x1 t->dple_1_x1x1
#define daux_dpli_1_x1x1 t->dpli_1_x1x1
#define daux_dpli_2_x1x1 t->dpli_2_x1x1
#define daux_dple_1_u1u1 t->dple_1_u1u1
#define daux_dple_1_u1x1 t->dple_1_u1x1
#define daux_dpfe_2_x0 t->dpfe_2_x0
#define daux_dpfi_1_x0 t->dpfi_1_x0
#define daux_dpfe_1_x1 t->dpfe_1_x1
#define daux_dpfe_2_x2 t->dpfe_2_x2
#define daux_dpfe_2_x0x0 t->dpfe_2_x0x0
#define daux_dpfi_1_x0x0 t->dpfi_1_x0x0
#define daux_dpfe_2_x0x2 t->dpfe_2_x0x2
#define daux_dpfe_1_x1x1 t->dpfe_1_x1x1
#define daux_dpfe_2_x2x2 t->dpfe_2_x2x2

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

    t->c= aux_ple_1 + aux_pli_1 + aux_pli_2 + p[1][0]*(u[0]*u[0]) + p[1][1]*(u[1]*u[1]) + p[2][0]*(sqrt((aux_gap*aux_gap) + 1.0) - 1.0) + p[2][1]*(x[1]*x[1]) + p[2][2]*(x[2]*x[2]);
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpF(trajFin_t *t, tOptSet *o) {
    const double *const x= t->x;
    const int k= o->n_hor;
    double **const p= o->p;

    t->c= aux_pfe_1 + aux_pfe_2 + aux_pfi_1 + p[0][0]*((-p[5][0] + x[0])*(-p[5][0] + x[0])) + p[0][1]*(x[1]*x[1]) + p[0][2]*((-p[5][0] + x[2])*(-p[5][0] + x[2]));
    if(isNANorINF(t->c)) { PRNT("    @k %d: t->c in line %d is nan or inf: %g\n", k, __LINE__-1, t->c); return 0; }
    return 1;
}

static int ddpf(double x_next[], trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    const double *const u= t->u;

    x_next[0]= p[3][0]*x[1] + x[0];
    if(isNANorINF(x_next[0])) { PRNT("    @k %d: x_next[0] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[0]); return 0; }
    x_next[1]= p[3][0]*(-1.0/4.0*aux_gap + u[0] - 1.0/2.0*(x[1]*x[1]*x[1])) + x[1];
    if(isNANorINF(x_next[1])) { PRNT("    @k %d: x_next[1] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[1]); return 0; }
    x_next[2]= p[3][0]*((1.0/2.0)*aux_gap + u[1]) + x[2];
    if(isNANorINF(x_next[2])) { PRNT("    @k %d: x_next[2] in line %d is nan or inf: %g\n", k, __LINE__-1, x_next[2]); return 0; }
    return 1;
}

void clampU(double *u, trajEl_t *t, int k, double **p, int N) {
    const double *const x= t->x;
    double bound;

    /* h[1]= a - lim[1] */
    bound= p[4][1];
    if(u[0]>bound) u[0]= bound;
    /* h[2]= -a + lim[0] */
    bound= p[4][0];
    if(u[0]<bound) u[0]= bound;
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

    /* h[1]= a - lim[1] */
    bound= p[4][1];
    if(t->upper[0]>bound) { t->upper[0]= bound; active[1][0]= 0; }
    /* h[2]= -a + lim[0] */
    bound= p[4][0];
    if(t->lower[0]<bound) { t->lower[0]= bound; active[0][0]= 1; }

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
                    sign[iu]= 1.0;
                    break;
                case 1:
                    hx_[0]= 0.0;
                    hx_[1]= 0.0;
                    hx_[2]= 0.0;
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

    aux_gap= x[0] - x[2];
    if(isNANorINF(aux_gap)) { PRNT("    @k %d: aux_gap in line %d is nan or inf: %g\n", k, __LINE__-1, aux_gap); return 0; }
    aux_hli_1= -p[5][2] + x[1];
    if(isNANorINF(aux_hli_1)) { PRNT("    @k %d: aux_hli_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hli_1); return 0; }
    aux_pli_1= ((aux_hli_1 >= 0.0) ? (
   aux_hli_1*m->mu_li[0]*(aux_hli_1*w_pen + 1.0)
)
: (
   aux_hli_1*m->mu_li[0]/(-aux_hli_1*w_pen + 1.0)
));
    if(isNANorINF(aux_pli_1)) { PRNT("    @k %d: aux_pli_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_pli_1); return 0; }
    aux_hli_2= -p[5][2] - x[1];
    if(isNANorINF(aux_hli_2)) { PRNT("    @k %d: aux_hli_2 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hli_2); return 0; }
    aux_pli_2= ((aux_hli_2 >= 0.0) ? (
   aux_hli_2*m->mu_li[1]*(aux_hli_2*w_pen + 1.0)
)
: (
   aux_hli_2*m->mu_li[1]/(-aux_hli_2*w_pen + 1.0)
));
    if(isNANorINF(aux_pli_2)) { PRNT("    @k %d: aux_pli_2 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_pli_2); return 0; }
    return 1;
}

static int calcXUVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    double **const p= o->p;
    const double w_pen= o->w_pen_l;

    aux_hle_1= -1.0/2.0*p[6][k]*x[1] + u[1];
    if(isNANorINF(aux_hle_1)) { PRNT("    @k %d: aux_hle_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hle_1); return 0; }
    aux_ple_1= 0.5*(aux_hle_1*aux_hle_1)*w_pen + aux_hle_1*m->mu_le[0];
    if(isNANorINF(aux_ple_1)) { PRNT("    @k %d: aux_ple_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_ple_1); return 0; }
    return 1;
}

static int calcFVariableAux(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {
    const double *const x= t->x;
    double **const p= o->p;
    const double w_pen= o->w_pen_f;
    const int k= o->n_hor;

    aux_gap= x[0] - x[2];
    if(isNANorINF(aux_gap)) { PRNT("    @k %d: aux_gap in line %d is nan or inf: %g\n", k, __LINE__-1, aux_gap); return 0; }
    aux_hfe_1= x[1];
    if(isNANorINF(aux_hfe_1)) { PRNT("    @k %d: aux_hfe_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hfe_1); return 0; }
    aux_pfe_1= 0.5*(aux_hfe_1*aux_hfe_1)*w_pen + aux_hfe_1*m->mu_fe[0];
    if(isNANorINF(aux_pfe_1)) { PRNT("    @k %d: aux_pfe_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_pfe_1); return 0; }
    aux_hfe_2= aux_gap;
    if(isNANorINF(aux_hfe_2)) { PRNT("    @k %d: aux_hfe_2 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hfe_2); return 0; }
    aux_pfe_2= 0.5*(aux_hfe_2*aux_hfe_2)*w_pen + aux_hfe_2*m->mu_fe[1];
    if(isNANorINF(aux_pfe_2)) { PRNT("    @k %d: aux_pfe_2 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_pfe_2); return 0; }
    aux_hfi_1= -p[5][1] + x[0];
    if(isNANorINF(aux_hfi_1)) { PRNT("    @k %d: aux_hfi_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_hfi_1); return 0; }
    aux_pfi_1= ((aux_hfi_1 >= 0.0) ? (
   aux_hfi_1*m->mu_fi[0]*(aux_hfi_1*w_pen + 1.0)
)
: (
   aux_hfi_1*m->mu_fi[0]/(-aux_hfi_1*w_pen + 1.0)
));
    if(isNANorINF(aux_pfi_1)) { PRNT("    @k %d: aux_pfi_1 in line %d is nan or inf: %g\n", k, __LINE__-1, aux_pfi_1); return 0; }
    return 1;
}

static int calcLAuxDeriv(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {
    const double *const x= t->x;
    const double *const u= t->u;
    const double w_pen= o->w_pen_l;
    double **const p= o->p;

    daux_dhle_1_x1= -1.0/2.0*p[6][k];
    if(isNANorINF(daux_dhle_1_x1)) { PRNT("    @k %d: daux_dhle_1_x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dhle_1_x1); return 0; }
    daux_dple_1_x1= daux_dhle_1_x1*(1.0*aux_hle_1*w_pen + m->mu_le[0]);
    if(isNANorINF(daux_dple_1_x1)) { PRNT("    @k %d: daux_dple_1_x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dple_1_x1); return 0; }
    daux_dpli_1_x1= ((aux_hli_1 >= 0.0) ? (
   aux_hli_1*m->mu_li[0]*w_pen + m->mu_li[0]*(aux_hli_1*w_pen + 1.0)
)
: (
   aux_hli_1*m->mu_li[0]*w_pen/((-aux_hli_1*w_pen + 1.0)*(-aux_hli_1*w_pen + 1.0)) + m->mu_li[0]/(-aux_hli_1*w_pen + 1.0)
));
    if(isNANorINF(daux_dpli_1_x1)) { PRNT("    @k %d: daux_dpli_1_x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpli_1_x1); return 0; }
    daux_dpli_2_x1= -((aux_hli_2 >= 0.0) ? (
   aux_hli_2*m->mu_li[1]*w_pen + m->mu_li[1]*(aux_hli_2*w_pen + 1.0)
)
: (
   aux_hli_2*m->mu_li[1]*w_pen/((-aux_hli_2*w_pen + 1.0)*(-aux_hli_2*w_pen + 1.0)) + m->mu_li[1]/(-aux_hli_2*w_pen + 1.0)
));
    if(isNANorINF(daux_dpli_2_x1)) { PRNT("    @k %d: daux_dpli_2_x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpli_2_x1); return 0; }
    daux_dple_1_u1= 1.0*aux_hle_1*w_pen + m->mu_le[0];
    if(isNANorINF(daux_dple_1_u1)) { PRNT("    @k %d: daux_dple_1_u1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dple_1_u1); return 0; }
    daux_dple_1_x1x1= 1.0*(daux_dhle_1_x1*daux_dhle_1_x1)*w_pen;
    if(isNANorINF(daux_dple_1_x1x1)) { PRNT("    @k %d: daux_dple_1_x1x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dple_1_x1x1); return 0; }
    daux_dpli_1_x1x1= ((aux_hli_1 >= 0.0) ? (
   2.0*m->mu_li[0]*w_pen
)
: (
   2.0*aux_hli_1*m->mu_li[0]*(w_pen*w_pen)/((-aux_hli_1*w_pen + 1.0)*(-aux_hli_1*w_pen + 1.0)*(-aux_hli_1*w_pen + 1.0)) + 2.0*m->mu_li[0]*w_pen/((-aux_hli_1*w_pen + 1.0)*(-aux_hli_1*w_pen + 1.0))
));
    if(isNANorINF(daux_dpli_1_x1x1)) { PRNT("    @k %d: daux_dpli_1_x1x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpli_1_x1x1); return 0; }
    daux_dpli_2_x1x1= ((aux_hli_2 >= 0.0) ? (
   2.0*m->mu_li[1]*w_pen
)
: (
   2.0*aux_hli_2*m->mu_li[1]*(w_pen*w_pen)/((-aux_hli_2*w_pen + 1.0)*(-aux_hli_2*w_pen + 1.0)*(-aux_hli_2*w_pen + 1.0)) + 2.0*m->mu_li[1]*w_pen/((-aux_hli_2*w_pen + 1.0)*(-aux_hli_2*w_pen + 1.0))
));
    if(isNANorINF(daux_dpli_2_x1x1)) { PRNT("    @k %d: daux_dpli_2_x1x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpli_2_x1x1); return 0; }
    daux_dple_1_u1u1= 1.0*w_pen;
    if(isNANorINF(daux_dple_1_u1u1)) { PRNT("    @k %d: daux_dple_1_u1u1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dple_1_u1u1); return 0; }
    daux_dple_1_u1x1= 1.0*daux_dhle_1_x1*w_pen;
    if(isNANorINF(daux_dple_1_u1x1)) { PRNT("    @k %d: daux_dple_1_u1x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dple_1_u1x1); return 0; }
#if FULL_DDP
#endif
    return 1;
}

static int bp_derivsL(trajEl_t *t, int k, double **p) {
    const double *const x= t->x;
    const double *const u= t->u;

    /* dynamics */
    t->fx[4]= -3.0/2.0*p[3][0]*(x[1]*x[1]) + 1.0;
    if(isNANorINF(t->fx[4])) { PRNT("    @k %d: t->fx[4] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[4]); return 0; }


#if FULL_DDP
    t->fxx[8]= -3.0*p[3][0]*x[1];
    if(isNANorINF(t->fxx[8])) { PRNT("    @k %d: t->fxx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fxx[8]); return 0; }

#endif
    /* cost */
    t->cx[0]= aux_gap*p[2][0]/sqrt((aux_gap*aux_gap) + 1.0);
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }
    t->cx[1]= daux_dple_1_x1 + daux_dpli_1_x1 + daux_dpli_2_x1 + 2.0*p[2][1]*x[1];
    if(isNANorINF(t->cx[1])) { PRNT("    @k %d: t->cx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[1]); return 0; }
    t->cx[2]= -aux_gap*p[2][0]/sqrt((aux_gap*aux_gap) + 1.0) + 2.0*p[2][2]*x[2];
    if(isNANorINF(t->cx[2])) { PRNT("    @k %d: t->cx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[2]); return 0; }

    t->cxx[0]= -(aux_gap*aux_gap)*p[2][0]/(((aux_gap*aux_gap) + 1.0)*sqrt((aux_gap*aux_gap) + 1.0)) + p[2][0]/sqrt((aux_gap*aux_gap) + 1.0);
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }
    t->cxx[2]= daux_dple_1_x1x1 + daux_dpli_1_x1x1 + daux_dpli_2_x1x1 + 2.0*p[2][1];
    if(isNANorINF(t->cxx[2])) { PRNT("    @k %d: t->cxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[2]); return 0; }
    t->cxx[3]= (aux_gap*aux_gap)*p[2][0]/(((aux_gap*aux_gap) + 1.0)*sqrt((aux_gap*aux_gap) + 1.0)) - p[2][0]/sqrt((aux_gap*aux_gap) + 1.0);
    if(isNANorINF(t->cxx[3])) { PRNT("    @k %d: t->cxx[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[3]); return 0; }
    t->cxx[5]= -(aux_gap*aux_gap)*p[2][0]/(((aux_gap*aux_gap) + 1.0)*sqrt((aux_gap*aux_gap) + 1.0)) + p[2][0]/sqrt((aux_gap*aux_gap) + 1.0) + 2.0*p[2][2];
    if(isNANorINF(t->cxx[5])) { PRNT("    @k %d: t->cxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[5]); return 0; }

    t->cu[0]= 2.0*p[1][0]*u[0];
    if(isNANorINF(t->cu[0])) { PRNT("    @k %d: t->cu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cu[0]); return 0; }
    t->cu[1]= daux_dple_1_u1 + 2.0*p[1][1]*u[1];
    if(isNANorINF(t->cu[1])) { PRNT("    @k %d: t->cu[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cu[1]); return 0; }

    t->cuu[2]= daux_dple_1_u1u1 + 2.0*p[1][1];
    if(isNANorINF(t->cuu[2])) { PRNT("    @k %d: t->cuu[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[2]); return 0; }

    t->cxu[4]= daux_dple_1_u1x1;
    if(isNANorINF(t->cxu[4])) { PRNT("    @k %d: t->cxu[4] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxu[4]); return 0; }

    return 1;
}

static int calcFAuxDeriv(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {
    const double *const x= t->x;
    const double w_pen= o->w_pen_f;
    double **const p= o->p;
    const int k= o->n_hor;

    daux_dpfe_2_x0= 1.0*aux_hfe_2*w_pen + m->mu_fe[1];
    if(isNANorINF(daux_dpfe_2_x0)) { PRNT("    @k %d: daux_dpfe_2_x0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_2_x0); return 0; }
    daux_dpfi_1_x0= ((aux_hfi_1 >= 0.0) ? (
   aux_hfi_1*m->mu_fi[0]*w_pen + m->mu_fi[0]*(aux_hfi_1*w_pen + 1.0)
)
: (
   aux_hfi_1*m->mu_fi[0]*w_pen/((-aux_hfi_1*w_pen + 1.0)*(-aux_hfi_1*w_pen + 1.0)) + m->mu_fi[0]/(-aux_hfi_1*w_pen + 1.0)
));
    if(isNANorINF(daux_dpfi_1_x0)) { PRNT("    @k %d: daux_dpfi_1_x0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfi_1_x0); return 0; }
    daux_dpfe_1_x1= 1.0*aux_hfe_1*w_pen + m->mu_fe[0];
    if(isNANorINF(daux_dpfe_1_x1)) { PRNT("    @k %d: daux_dpfe_1_x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_1_x1); return 0; }
    daux_dpfe_2_x2= -1.0*aux_hfe_2*w_pen - m->mu_fe[1];
    if(isNANorINF(daux_dpfe_2_x2)) { PRNT("    @k %d: daux_dpfe_2_x2 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_2_x2); return 0; }
    daux_dpfe_2_x0x0= 1.0*w_pen;
    if(isNANorINF(daux_dpfe_2_x0x0)) { PRNT("    @k %d: daux_dpfe_2_x0x0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_2_x0x0); return 0; }
    daux_dpfi_1_x0x0= ((aux_hfi_1 >= 0.0) ? (
   2.0*m->mu_fi[0]*w_pen
)
: (
   2.0*aux_hfi_1*m->mu_fi[0]*(w_pen*w_pen)/((-aux_hfi_1*w_pen + 1.0)*(-aux_hfi_1*w_pen + 1.0)*(-aux_hfi_1*w_pen + 1.0)) + 2.0*m->mu_fi[0]*w_pen/((-aux_hfi_1*w_pen + 1.0)*(-aux_hfi_1*w_pen + 1.0))
));
    if(isNANorINF(daux_dpfi_1_x0x0)) { PRNT("    @k %d: daux_dpfi_1_x0x0 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfi_1_x0x0); return 0; }
    daux_dpfe_2_x0x2= -1.0*w_pen;
    if(isNANorINF(daux_dpfe_2_x0x2)) { PRNT("    @k %d: daux_dpfe_2_x0x2 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_2_x0x2); return 0; }
    daux_dpfe_1_x1x1= 1.0*w_pen;
    if(isNANorINF(daux_dpfe_1_x1x1)) { PRNT("    @k %d: daux_dpfe_1_x1x1 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_1_x1x1); return 0; }
    daux_dpfe_2_x2x2= 1.0*w_pen;
    if(isNANorINF(daux_dpfe_2_x2x2)) { PRNT("    @k %d: daux_dpfe_2_x2x2 in line %d is nan or inf: %g\n", k, __LINE__-1, daux_dpfe_2_x2x2); return 0; }
    return 1;
}

static int bp_derivsF(trajFin_t *t, int k, double **p) {
    const double *const x= t->x;

    t->cx[0]= daux_dpfe_2_x0 + daux_dpfi_1_x0 + p[0][0]*(-2.0*p[5][0] + 2.0*x[0]);
    if(isNANorINF(t->cx[0])) { PRNT("    @k %d: t->cx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[0]); return 0; }
    t->cx[1]= daux_dpfe_1_x1 + 2.0*p[0][1]*x[1];
    if(isNANorINF(t->cx[1])) { PRNT("    @k %d: t->cx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[1]); return 0; }
    t->cx[2]= daux_dpfe_2_x2 + p[0][2]*(-2.0*p[5][0] + 2.0*x[2]);
    if(isNANorINF(t->cx[2])) { PRNT("    @k %d: t->cx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cx[2]); return 0; }

    t->cxx[0]= daux_dpfe_2_x0x0 + daux_dpfi_1_x0x0 + 2.0*p[0][0];
    if(isNANorINF(t->cxx[0])) { PRNT("    @k %d: t->cxx[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[0]); return 0; }
    t->cxx[2]= daux_dpfe_1_x1x1 + 2.0*p[0][1];
    if(isNANorINF(t->cxx[2])) { PRNT("    @k %d: t->cxx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[2]); return 0; }
    t->cxx[3]= daux_dpfe_2_x0x2;
    if(isNANorINF(t->cxx[3])) { PRNT("    @k %d: t->cxx[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[3]); return 0; }
    t->cxx[5]= daux_dpfe_2_x2x2 + 2.0*p[0][2];
    if(isNANorINF(t->cxx[5])) { PRNT("    @k %d: t->cxx[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cxx[5]); return 0; }
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
        t->cxx[4]= 0.0;


        t->cuu[0]= 2.0*p[1][0];
        if(isNANorINF(t->cuu[0])) { PRNT("    @k %d: t->cuu[0] in line %d is nan or inf: %g\n", k, __LINE__-1, t->cuu[0]); return 0; }
        t->cuu[1]= 0.0;

        t->cxu[0]= 0.0;
        t->cxu[1]= 0.0;
        t->cxu[2]= 0.0;
        t->cxu[3]= 0.0;
        t->cxu[5]= 0.0;

        /* dynamics */
        t->fx[0]= 1.0;
        t->fx[1]= -1.0/4.0*p[3][0];
        if(isNANorINF(t->fx[1])) { PRNT("    @k %d: t->fx[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[1]); return 0; }
        t->fx[2]= (1.0/2.0)*p[3][0];
        if(isNANorINF(t->fx[2])) { PRNT("    @k %d: t->fx[2] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[2]); return 0; }
        t->fx[3]= p[3][0];
        if(isNANorINF(t->fx[3])) { PRNT("    @k %d: t->fx[3] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[3]); return 0; }
        t->fx[5]= 0.0;
        t->fx[6]= 0.0;
        t->fx[7]= (1.0/4.0)*p[3][0];
        if(isNANorINF(t->fx[7])) { PRNT("    @k %d: t->fx[7] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[7]); return 0; }
        t->fx[8]= 1.0 - 1.0/2.0*p[3][0];
        if(isNANorINF(t->fx[8])) { PRNT("    @k %d: t->fx[8] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fx[8]); return 0; }

        t->fu[0]= 0.0;
        t->fu[1]= p[3][0];
        if(isNANorINF(t->fu[1])) { PRNT("    @k %d: t->fu[1] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[1]); return 0; }
        t->fu[2]= 0.0;
        t->fu[3]= 0.0;
        t->fu[4]= 0.0;
        t->fu[5]= p[3][0];
        if(isNANorINF(t->fu[5])) { PRNT("    @k %d: t->fu[5] in line %d is nan or inf: %g\n", k, __LINE__-1, t->fu[5]); return 0; }

#if FULL_DDP
        t->fxx[0]= 0.0;
        t->fxx[1]= 0.0;
        t->fxx[2]= 0.0;
        t->fxx[3]= 0.0;
        t->fxx[4]= 0.0;
        t->fxx[5]= 0.0;
        t->fxx[6]= 0.0;
        t->fxx[7]= 0.0;
        t->fxx[9]= 0.0;
        t->fxx[10]= 0.0;
        t->fxx[11]= 0.0;
        t->fxx[12]= 0.0;
        t->fxx[13]= 0.0;
        t->fxx[14]= 0.0;
        t->fxx[15]= 0.0;
        t->fxx[16]= 0.0;
        t->fxx[17]= 0.0;

        { int e_; for(e_= 0; e_<N_X*sizeofQuu; e_++) t->fuu[e_]= 0.0; }

        { int e_; for(e_= 0; e_<N_X*sizeofQxu; e_++) t->fxu[e_]= 0.0; }

#endif
    }
    return 1;
}

static int init_final(trajFin_t *t, tOptSet *o) {
    double **const p= o->p;
    const int k= o->n_hor;


    t->cxx[1]= 0.0;
    t->cxx[4]= 0.0;
    return 1;
}

int init_trajectory(traj_t *t, tOptSet *o) {
    return init_running(t->t, o) && init_final(&t->f, o);
}

static int init_multipliers_running(tOptSet *o) {
    multipliersEl_t *m= o->multipliers.t;
    multipliersEl_t *const end= m + o->n_hor;
    int i;

    for(; m<end; m++) {
        for(i= 0; i<1; i++) { m->mu_le[i]= 0.0; m->last_hle[i]= 0.0; }
        for(i= 0; i<2; i++) { m->mu_li[i]= 1.0; m->last_hli[i]= 0.0; }
    }
    return 1;
}

static int init_multipliers_final(tOptSet *o) {
    multipliersFin_t *const m= &o->multipliers.f;
    int i;

    for(i= 0; i<2; i++) { m->mu_fe[i]= 0.0; m->last_hfe[i]= 0.0; }
    for(i= 0; i<1; i++) { m->mu_fi[i]= 1.0; m->last_hfi[i]= 0.0; }
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
    trajEl_t *t= o->nominal->t;
    multipliersEl_t *m= o->multipliers.t;
    const double w_pen= o->w_pen_l;
    double **const p= o->p;
    int stalled= 0, k;

    for(k= 0; k<o->n_hor; k++, t++, m++) {
        stalled|= violation_stalls(fabs(aux_hle_1), fabs(m->last_hle[0]), o);
        m->last_hle[0]= aux_hle_1;
        stalled|= violation_stalls(aux_hli_1, m->last_hli[0], o);
        m->last_hli[0]= aux_hli_1;
        stalled|= violation_stalls(aux_hli_2, m->last_hli[1], o);
        m->last_hli[1]= aux_hli_2;
        if(init) return 1;  /* solver entry: the violations of the first element are remembered, nothing else */
        m->mu_le[0]= aux_hle_1*w_pen + m->mu_le[0];
        if(isNANorINF(m->mu_le[0])) { PRNT("    @k %d: m->mu_le[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_le[0]); return 0; }
        if(aux_hli_1>=0) {
            m->mu_li[0]= m->mu_li[0]*(2.0*aux_hli_1*w_pen + 1.0);
            if(isNANorINF(m->mu_li[0])) { PRNT("    @k %d: m->mu_li[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_li[0]); return 0; }
        } else {
            m->mu_li[0]= m->mu_li[0]/((-aux_hli_1*w_pen + 1.0)*(-aux_hli_1*w_pen + 1.0));
            if(isNANorINF(m->mu_li[0])) { PRNT("    @k %d: m->mu_li[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_li[0]); return 0; }
        }
        if(aux_hli_2>=0) {
            m->mu_li[1]= m->mu_li[1]*(2.0*aux_hli_2*w_pen + 1.0);
            if(isNANorINF(m->mu_li[1])) { PRNT("    @k %d: m->mu_li[1] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_li[1]); return 0; }
        } else {
            m->mu_li[1]= m->mu_li[1]/((-aux_hli_2*w_pen + 1.0)*(-aux_hli_2*w_pen + 1.0));
            if(isNANorINF(m->mu_li[1])) { PRNT("    @k %d: m->mu_li[1] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_li[1]); return 0; }
        }
    }
    if(!init && stalled)
        o->w_pen_l= min(o->w_pen_max_l, o->w_pen_l*o->w_pen_fact1);
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
    stalled|= violation_stalls(fabs(aux_hfe_2), fabs(m->last_hfe[1]), o);
    m->last_hfe[1]= aux_hfe_2;
    stalled|= violation_stalls(aux_hfi_1, m->last_hfi[0], o);
    m->last_hfi[0]= aux_hfi_1;
    if(!init && stalled)
        o->w_pen_f= min(o->w_pen_max_f, o->w_pen_f*o->w_pen_fact1);
    if(init) return 1;
    m->mu_fe[0]= aux_hfe_1*w_pen + m->mu_fe[0];
    if(isNANorINF(m->mu_fe[0])) { PRNT("    @k %d: m->mu_fe[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_fe[0]); return 0; }
    m->mu_fe[1]= aux_hfe_2*w_pen + m->mu_fe[1];
    if(isNANorINF(m->mu_fe[1])) { PRNT("    @k %d: m->mu_fe[1] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_fe[1]); return 0; }
    if(aux_hfi_1>=0) {
        m->mu_fi[0]= m->mu_fi[0]*(2.0*aux_hfi_1*w_pen + 1.0);
        if(isNANorINF(m->mu_fi[0])) { PRNT("    @k %d: m->mu_fi[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_fi[0]); return 0; }
    } else {
        m->mu_fi[0]= m->mu_fi[0]/((-aux_hfi_1*w_pen + 1.0)*(-aux_hfi_1*w_pen + 1.0));
        if(isNANorINF(m->mu_fi[0])) { PRNT("    @k %d: m->mu_fi[0] in line %d is nan or inf: %g\n", k, __LINE__-1, m->mu_fi[0]); return 0; }
    }
    return 1;
}

/* iLQG.c:236,337: multipliers of the running constraints, then of the final ones */
int update_multipliers(tOptSet *o, int init) {
    return update_multipliers_running(o, init) && update_multipliers_final(o, init);
}

/* no outputs g are defined by this generator (iLQG_func.tem:511-521) */
int get_g_size() { return 0; }

int calcG(double g[], trajEl_t *t, int k, double **p) { return 1; }
