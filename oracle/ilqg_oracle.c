/* TEST INFRASTRUCTURE — CPU restatement of the reference's hot path.
 *
 * This file restates, in plain C and with the reference's operation order,
 * the algorithm of the reference files
 *     matMult.c, cholesky.c (plain part), boxQP.c, back_pass.c, line_search.c
 * (the outer loop iLQG.c is restated in ilqg_oracle_loop.c) behind the same C symbols, so that it can be (a) pinned bit-for-bit against
 * the reference's own sources compiled here (oracle/_ref, see
 * tests/test_oracle_vs_ref.py and the fixtures in tests/golden/), and
 * (b) shipped to the GPU box as the checker for the HIP kernels and as the CPU
 * baseline.  It is never linked into, imported by, or called from the product.
 *
 * Parity status: the dense helpers (matMult.c, cholesky.c) are PINNED — checked with exact
 * (bitwise) equality against those reference files compiled with no stand-in of any kind (oracle/_ref/libref_pure.so).
 * The solver proper (boxQP.c, back_pass.c, line_search.c, iLQG.c) equals the reference's own sources bit for bit on
 * every fixture and on fresh seeded inputs, both compiled with -ffp-contract=off — but that reference build needs a
 * stand-in mex.h and this repository's generated problem file (the reference commits none and Maxima is absent), so
 * by the rule that such a build pins nothing, PARITY OF THE SOLVER PROPER IS UNPINNED beyond the analytic known
 * answer of the Brachistochrone demo (tests/test_known_answer.py).
 *
 * All sums run in ascending index order, fp64, no fused multiply-add
 * (SURVEY.md Appendix D).  Matrices are column-major; symmetric matrices are
 * packed upper triangles (matMult.h).
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "mex.h"
#include "iLQG.h"
#include "back_pass.h"
#include "line_search.h"
#include "boxQP.h"
#include "cholesky.h"
#include "matMult.h"
#include "printMat.h"

/* packed index of symmetric element (i,j) in either order */
static inline int sy(int i, int j) { return (i > j) ? (i * (i + 1)) / 2 + j : (j * (j + 1)) / 2 + i; }
static inline int ut(int r, int c) { return (c * (c + 1)) / 2 + r; }

/* ========================================================================
 * small dense helpers                                   reference matMult.c
 * ======================================================================== */

/* base += b' a, accumulated in place with the row index ascending  (matMult.c:3-12) */
void addMulVec(double base[], const double a[], const double b[], const int n_r, const int n_c) {
    for(int c = 0; c < n_c; c++) {
        const double *col = b + c * n_r;
        for(int r = 0; r < n_r; r++) base[c] += a[r] * col[r];
    }
}

/* packed-upper base += a' b a.  First ba = b a (each entry summed from 0.0),
 * then per output entry one accumulator; an off-diagonal entry continues the
 * same accumulator with the transposed product and is halved (matMult.c:14-46) */
void addSquareTri(double base[], const double b[], const double a[], const int n_r, const int n_c,
                  double ba[]) {
    for(int r = 0; r < n_r; r++)
        for(int c = 0; c < n_c; c++) {
            double acc = 0.0;
            for(int s = 0; s < n_r; s++) acc += b[sy(r, s)] * a[s + c * n_r];
            ba[r + c * n_r] = acc;
        }
    for(int c = 0; c < n_c; c++)
        for(int r = 0; r <= c; r++) {
            double acc = 0.0;
            for(int s = 0; s < n_r; s++) acc += a[s + r * n_r] * ba[s + c * n_r];
            if(r != c) {
                for(int s = 0; s < n_r; s++) acc += a[s + c * n_r] * ba[s + r * n_r];
                acc *= 0.5;
            }
            base[ut(r, c)] += acc;
        }
}

/* full base (n_ca x n_cc) += a' b c with bc = b c formed first (matMult.c:48-72) */
void addMul2Tri(double base[], const double b[], const double a[], const int n_ra, const int n_ca,
                const double c[], const int n_rc, const int n_cc, double bc[]) {
    for(int r = 0; r < n_ra; r++)
        for(int q = 0; q < n_cc; q++) {
            double acc = 0.0;
            for(int s = 0; s < n_rc; s++) acc += b[sy(r, s)] * c[s + q * n_rc];
            bc[r + q * n_ra] = acc;
        }
    for(int r = 0; r < n_ca; r++)
        for(int q = 0; q < n_cc; q++) {
            double acc = 0.0;
            for(int s = 0; s < n_ra; s++) acc += a[s + r * n_ra] * bc[s + q * n_ra];
            base[r + q * n_ca] += acc;
        }
}

/* ========================================================================
 * Cholesky                                     reference cholesky.c:6-74
 * ======================================================================== */

/* A = U'U, packed upper.  Returns 0 as soon as a pivot is <= 0.  Off-diagonal
 * entries are (1/U_jj) * s: reciprocal, then multiply (cholesky.c:6-27) */
int cholesky_tri(const double *A, int n, double *U) {
    for(int i = 0; i < n; i++)
        for(int j = 0; j <= i; j++) {
            double dot = 0;
            for(int k = 0; k < j; k++) dot += U[ut(k, i)] * U[ut(k, j)];
            double s = A[ut(j, i)] - dot;
            if(i == j) {
                if(s <= 0.0) return 0;
                U[ut(j, i)] = sqrt(s);
            } else {
                U[ut(j, i)] = 1.0 / U[ut(j, j)] * s;
            }
        }
    return 1;
}

/* explicit packed inverse from the factor: for each unit vector e_l solve
 * U'y = e_l (forward, starting at l because of the leading zeros) and
 * U z = y (backward); z gives row l of the inverse (cholesky.c:51-74) */
void cholesky_tri_inv(const double *U, double *invA, const int n, double *x) {
    for(int l = 0; l < n; l++) {
        x[l] = 1.0;
        for(int k = l + 1; k < n; k++) x[k] = 0.0;
        for(int k = l; k < n; k++) {
            for(int i = l; i < k; i++) x[k] -= x[i] * U[ut(i, k)];
            x[k] /= U[ut(k, k)];
        }
        for(int k = n - 1; k >= l; k--) {
            for(int i = k + 1; i < n; i++) x[k] -= x[i] * U[ut(k, i)];
            x[k] /= U[ut(k, k)];
            invA[ut(l, k)] = x[k];
        }
    }
}

/* ========================================================================
 * box-constrained QP                           reference boxQP.c:39-238
 * ======================================================================== */

static double qp_value(const double *H, const double *g, const double *x, int n) {
    /* sum_i x_i (g_i + 1/2 sum_j H_ij x_j)   (boxQP.c:75-82, 210-217) */
    double v = 0.0;
    for(int i = 0; i < n; i++) {
        double hx = 0.0;
        for(int j = 0; j < n; j++) hx += H[sy(i, j)] * x[j];
        v += x[i] * (g[i] + 0.5 * hx);
    }
    return v;
}

/* optional instrumentation for tools/divergence_stats.py (not part of the algorithm): when a log is
 * attached, every call appends {outer iterations, factorisations, Armijo trials} */
int *ilqg_oracle_boxqp_log = NULL;
long ilqg_oracle_boxqp_log_cap = 0, ilqg_oracle_boxqp_log_n = 0;
static int boxqp_core(double *H, const double *g, const double *lower, const double *upper, double *x,
                      double *Hfree, double *U, double *grad, double *grad_clamped, double *search,
                      int *is_clamped, int *n_free_, double *invHfree, const int n, int *stat);

int boxQP(double *H, const double *g, const double *lower, const double *upper, double *x,
          double *Hfree, double *U, double *grad, double *grad_clamped, double *search,
          int *is_clamped, int *n_free_, double *invHfree, const int n) {
    int stat[3] = {0, 0, 0};
    const int rc = boxqp_core(H, g, lower, upper, x, Hfree, U, grad, grad_clamped, search, is_clamped, n_free_,
                              invHfree, n, stat);
    if(ilqg_oracle_boxqp_log && ilqg_oracle_boxqp_log_n < ilqg_oracle_boxqp_log_cap) {
        int *e = ilqg_oracle_boxqp_log + 3 * ilqg_oracle_boxqp_log_n++;
        e[0] = stat[0]; e[1] = stat[1]; e[2] = stat[2];
    }
    return rc;
}

static int boxqp_core(double *H, const double *g, const double *lower, const double *upper, double *x,
                      double *Hfree, double *U, double *grad, double *grad_clamped, double *search,
                      int *is_clamped, int *n_free_, double *invHfree, const int n, int *stat) {
    /* constants: boxQP.c:52-57 */
    const int max_iter = 100;
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    double value, oldvalue = 0.0;

    memset(Hfree, 0, sizeof(double) * (n * (n + 1)) / 2);

    /* project the warm start onto the box, upper bound first (boxQP.c:61-67) */
    for(int i = 0; i < n; i++) {
        if(x[i] > upper[i]) x[i] = upper[i];
        if(x[i] < lower[i]) x[i] = lower[i];
        is_clamped[i] = 0;
    }
    value = qp_value(H, g, x, n);

    for(int iter = 0; iter < max_iter; iter++) {
        if(iter > 0 && (oldvalue - value) < min_rel_improve * fabs(oldvalue)) return 4;
        oldvalue = value;
        stat[0]++;

        /* gradient and clamp flags (boxQP.c:91-117): a variable sitting on a
         * bound with the gradient pushing outwards is clamped */
        int all_clamped = 1, changed = 0, n_free = 0;
        double gnorm = 0.0;
        for(int i = 0; i < n; i++) {
            double hx = 0.0;
            for(int j = 0; j < n; j++) hx += H[sy(i, j)] * x[j];
            grad[i] = g[i] + hx;

            int was = is_clamped[i];
            if(x[i] <= lower[i] && grad[i] > 0)
                is_clamped[i] = 1;
            else if(x[i] >= upper[i] && grad[i] < 0)
                is_clamped[i] = 2;
            else {
                is_clamped[i] = 0;
                all_clamped = 0;
                gnorm += grad[i] * grad[i];
                n_free++;
            }
            if((!was) != (!is_clamped[i])) changed = 1;
        }
        n_free_[0] = n_free;
        if(all_clamped) return 6;

        /* factorise the free block when the free set changed (boxQP.c:129-146) */
        if(iter == 0 || changed) {
            int jf = 0;
            for(int j = 0; j < n; j++) {
                if(is_clamped[j]) continue;
                int jfi = 0;
                for(int i = 0; i <= j; i++) {
                    if(is_clamped[i]) continue;
                    Hfree[ut(jfi, jf)] = H[ut(i, j)];
                    jfi++;
                }
                jf++;
            }
            stat[1]++;
            if(!cholesky_tri(Hfree, n_free, U)) return -1;
            cholesky_tri_inv(U, invHfree, n_free, search);
        }

        if(gnorm < min_grad * min_grad) return 5;

        /* Newton direction on the free variables (boxQP.c:155-177):
         * grad_clamped = g + H (x .* clamped) restricted to free rows,
         * search(free) = -invHfree grad_clamped - x(free), search(clamped) = 0 */
        for(int i = 0, fi = 0; i < n; i++) {
            if(is_clamped[i]) continue;
            double hc = 0.0;
            for(int j = 0; j < n; j++)
                if(is_clamped[j]) hc += H[sy(i, j)] * x[j];
            grad_clamped[fi++] = g[i] + hc;
        }
        for(int i = 0, fi = 0; i < n; i++) {
            if(is_clamped[i]) {
                search[i] = 0.0;
                continue;
            }
            search[i] = -x[i];
            for(int fj = 0; fj < n_free; fj++) search[i] -= invHfree[sy(fi, fj)] * grad_clamped[fj];
            fi++;
        }

        double sdotg = 0.0;
        for(int i = 0; i < n; i++) sdotg += search[i] * grad[i];
        if(sdotg >= 0.0) {
            printTri(H, n, "H"); /* boxQP.c:194 prints unconditionally */
            return -2;
        }

        /* Armijo backtracking on the projected step; the candidate lives in
         * grad[] (boxQP.c:199-227) */
        double step = 1.0, vc;
        double *xc = grad;
        for(;;) {
            for(int i = 0; i < n; i++) {
                xc[i] = x[i] + step * search[i];
                if(xc[i] > upper[i]) xc[i] = upper[i];
                if(xc[i] < lower[i]) xc[i] = lower[i];
            }
            vc = qp_value(H, g, xc, n);
            stat[2]++;
            if(((vc - oldvalue) / (step * sdotg)) >= armijo) break;
            step = step * step_dec;
            if(step < min_step) return 2;
        }
        for(int i = 0; i < n; i++) x[i] = xc[i];
        value = vc;
    }
    return 1;
}

/* ========================================================================
 * backward Riccati sweep                       reference back_pass.c:38-257
 * ======================================================================== */
int back_pass(tOptSet *o) {
    const int N = o->n_hor;
    double Vx[N_X], Vxx[sizeofQxx];
    double Qx[N_X], Qu[N_U], Qxx[sizeofQxx], Qxu[sizeofQxu], Quu[sizeofQuu];
    double QuuF[sizeofQuu], Qxu_reg[sizeofQxu];
    double invH[sizeofQuu], chol[sizeofQuu], Hfree[sizeofQuu];
    double grad[N_U], grad_clamped[N_U], search[N_U];
    double scratch[N_X * N_X];
    int clamp[N_U], n_free;
    double gsum = 0.0;

    o->dV[0] = 0.0;
    o->dV[1] = 0.0;

    /* terminal condition: V = final cost (back_pass.c:66-67) */
    memcpy(Vx, o->nominal->f.cx, sizeof(Vx));
    memcpy(Vxx, o->nominal->f.cxx, sizeof(Vxx));

    for(int k = N - 1; k >= 0; k--) {
        trajEl_t *t = o->nominal->t + k;

        /* Q-function expansion, in the reference's order (back_pass.c:80-131) */
        memcpy(Qu, t->cu, sizeof(Qu));
        addMulVec(Qu, Vx, t->fu, N_X, N_U);
        memcpy(Qx, t->cx, sizeof(Qx));
        addMulVec(Qx, Vx, t->fx, N_X, N_X);

        memcpy(Qxu, t->cxu, sizeof(Qxu));
        addMul2Tri(Qxu, Vxx, t->fx, N_X, N_X, t->fu, N_X, N_U, scratch);
#if FULL_DDP
        for(int j = 0; j < N_X * N_U; j++) {
            double acc = 0.0;
            for(int i = 0; i < N_X; i++) acc += Vx[i] * t->fxu[j + i * sizeofQxu];
            Qxu[j] += acc;
        }
#endif
        memcpy(Quu, t->cuu, sizeof(Quu));
        addSquareTri(Quu, Vxx, t->fu, N_X, N_U, scratch);
#if FULL_DDP
        for(int j = 0; j < sizeofQuu; j++) {
            double acc = 0.0;
            for(int i = 0; i < N_X; i++) acc += Vx[i] * t->fuu[j + i * sizeofQuu];
            Quu[j] += acc;
        }
#endif
        memcpy(Qxx, t->cxx, sizeof(Qxx));
        addSquareTri(Qxx, Vxx, t->fx, N_X, N_X, scratch);
#if FULL_DDP
        for(int j = 0; j < sizeofQxx; j++) {
            double acc = 0.0;
            for(int i = 0; i < N_X; i++) acc += Vx[i] * t->fxx[j + i * sizeofQxx];
            Qxx[j] += acc;
        }
#endif

        /* regularisation (back_pass.c:134-159).  regType 2 reproduces the
         * reference's index expressions literally. */
        memcpy(QuuF, Quu, sizeof(QuuF));
        memcpy(Qxu_reg, Qxu, sizeof(Qxu_reg));
        if(o->regType == 2) {
            for(int j = 0; j < N_U; j++)
                for(int i = 0; i <= j; i++) {
                    double acc = 0.0;
                    for(int q = 0; q < N_U; q++) acc += t->fu[sy(q, i)] * t->fu[sy(q, j)];
                    QuuF[ut(i, j)] += acc * o->lambda;
                }
            for(int i = 0; i < N_X; i++)
                for(int j = 0; j < N_U; j++) {
                    double acc = 0.0;
                    for(int q = 0; q < N_X; q++) acc += t->fx[q + i * N_X] * t->fu[q + j * N_U];
                    Qxu_reg[i + j * N_X] += acc * o->lambda;
                }
        }
        if(o->regType == 1)
            for(int i = 0; i < N_U; i++) QuuF[ut(i, i)] += o->lambda;

        /* feed-forward term from the box QP, warm-started with the later
         * step's solution (back_pass.c:163-171) */
        if(k == N - 1)
            memset(t->l, 0, sizeof(t->l));
        else
            memcpy(t->l, (t + 1)->l, sizeof(t->l));
        if(boxQP(QuuF, Qu, t->lower, t->upper, t->l, Hfree, chol, grad, grad_clamped, search, clamp,
                 &n_free, invH, N_U) < 1)
            return 1;

        /* feedback gains (back_pass.c:175-201): free rows from the inverse of
         * the free block; clamped rows follow the active constraint's state
         * gradient */
        memset(t->L, 0, sizeof(t->L));
        for(int i = 0, fi = 0; i < N_U; i++) {
            if(clamp[i]) {
                const double *hx = (clamp[i] == 1) ? t->lower_hx : t->upper_hx;
                const double sg = (clamp[i] == 1) ? t->lower_sign[i] : t->upper_sign[i];
                for(int r = 0; r < N_X; r++) t->L[i + r * N_U] -= sg * hx[r + i * N_X];
                continue;
            }
            for(int j = 0, fj = 0; j < N_U; j++) {
                if(!clamp[j]) {
                    for(int r = 0; r < N_X; r++)
                        t->L[i + r * N_U] -= invH[sy(fi, fj)] * Qxu_reg[r + j * N_X];
                    fj++;
                } else {
                    double w = 0.0;
                    for(int q = 0, fq = 0; q < N_U; q++)
                        if(!clamp[q]) {
                            w -= invH[sy(fi, fq)] * QuuF[sy(q, j)];
                            fq++;
                        }
                    const double *hx = (clamp[j] == 1) ? t->lower_hx : t->upper_hx;
                    const double sg = (clamp[j] == 1) ? t->lower_sign[j] : t->upper_sign[j];
                    for(int r = 0; r < N_X; r++) t->L[i + r * N_U] -= w * (sg * hx[r + j * N_X]);
                }
            }
            fi++;
        }

        /* expected cost change (back_pass.c:205-214) */
        for(int i = 0; i < N_U; i++) o->dV[0] += Qu[i] * t->l[i];
        for(int i = 0; i < N_U; i++) {
            double acc = 0.0;
            for(int j = 0; j < N_U; j++) acc += t->l[j] * Quu[sy(j, i)];
            o->dV[1] += 0.5 * t->l[i] * acc;
        }

        /* value function, using the UNregularised Quu/Qxu (back_pass.c:219-241) */
        memcpy(Vx, Qx, sizeof(Vx));
        addMul2Tri(Vx, Quu, t->L, N_U, N_X, t->l, N_U, 1, scratch);
        for(int i = 0; i < N_X; i++)
            for(int j = 0; j < N_U; j++) Vx[i] += t->L[j + i * N_U] * Qu[j];
        for(int i = 0; i < N_X; i++)
            for(int j = 0; j < N_U; j++) Vx[i] += Qxu[i + j * N_X] * t->l[j];

        memcpy(Vxx, Qxx, sizeof(Vxx));
        addSquareTri(Vxx, Quu, t->L, N_U, N_X, scratch);
        for(int i = 0; i < N_X; i++)
            for(int j = 0; j < N_X; j++)
                for(int q = 0; q < N_U; q++) {
                    double term = t->L[q + i * N_U] * Qxu[j + q * N_X];
                    if(i == j) term *= 2.0;
                    Vxx[sy(i, j)] += term;
                }

        /* gradient norm contribution (back_pass.c:246-251) */
        double gmax = 0.0;
        for(int i = 0; i < N_U; i++) {
            double gi = fabs(t->l[i]) / (fabs(t->u[i]) + 1.0);
            if(gi > gmax) gmax = gi;
        }
        gsum += gmax;
    }

    /* N summands divided by N-1 (back_pass.c:254) */
    o->g_norm = gsum / ((double)(o->n_hor - 1));
    return 0;
}

/* ========================================================================
 * line search                                  reference line_search.c:33-78
 * ======================================================================== */
int line_search(tOptSet *o, int iter) {
    double expected = 0.0, z = 0.0, dcost = 0.0, cnew = 0.0;
    int i, ok = 0;

    /* first acceptable step wins (line_search.c:37-60) */
    for(i = 0; i < o->n_alpha; i++) {
        const double a = o->alpha[i];
        ok = forward_pass(o->candidates[0], o, a, &cnew, 0);
        if(!ok) continue;
        dcost = o->cost - cnew;
        expected = -a * (o->dV[0] + a * o->dV[1]);
        z = (expected > 0) ? dcost / expected : 0;
        if(z > o->zMin) break;
        ok = 0;
    }

    if(o->log_linesearch != NULL) o->log_linesearch[iter] = i + 1;
    if(o->log_z != NULL) o->log_z[iter] = z;
    if(o->log_cost != NULL) o->log_cost[iter] = cnew;
    o->new_cost = cnew;
    o->dcost = dcost;
    o->expected = expected;
    return ok;
}

/* ========================================================================
 * debug printers                                     reference printMat.c
 * ======================================================================== */
void printVec(const double *A, const int n, const char *nm) {
    PRNT("%s= [", nm);
    for(int i = 0; i < n; i++) PRNT(i ? ", %g" : "%g", A[i]);
    PRNT("]\n");
}

void printTri(const double *A, const int n, const char *nm) {
    PRNT("%s= [\n", nm);
    for(int r = 0; r < n; r++) {
        for(int c = 0; c < n; c++) PRNT(c ? ", %g" : "  %g", A[sy(r, c)]);
        PRNT("\n");
    }
    PRNT("]\n");
}

void printMat(const double *A, const int n, const int m, const char *nm) {
    PRNT("%s= [\n", nm);
    for(int r = 0; r < n; r++) {
        for(int c = 0; c < m; c++) PRNT(c ? ", %g" : "  %g", A[r + n * c]);
        PRNT("\n");
    }
    PRNT("]\n");
}

void ilqg_release(tOptSet *o) { (void)o; }
