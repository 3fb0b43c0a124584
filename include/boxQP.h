/* Projected-Newton box-constrained QP (replaces reference boxQP.h:4).
 *   min 1/2 x'Hx + g'x  s.t. lower <= x <= upper,  H packed upper triangle.
 * Return codes (reference boxQP.c:39-238): 6 all clamped, 5 free gradient
 * norm^2 < 1e-16, 4 relative improvement < 1e-8, 2 Armijo step < 1e-22,
 * 1 iteration limit (100), -1 Cholesky failed, -2 not a descent direction. */
#ifndef BOXQP_H
#define BOXQP_H
#ifdef __cplusplus
extern "C" {
#endif
int boxQP(double *H, const double *g, const double *lower, const double *upper, double *x,
          double *Hfree, double *L, double *grad, double *grad_clamped, double *search,
          int *is_clamped, int *n_free_, double *invHfree, const int n);
#ifdef __cplusplus
}
#endif
#endif
