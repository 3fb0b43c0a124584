/* Packed upper-triangular Cholesky A = U'U and explicit inverse
 * (replaces the plain part of reference cholesky.h:4-6; the experimental
 * MOD_CHOL entry points, cholesky.h:8-11, are out of scope). */
#ifndef CHOLESKY_H
#define CHOLESKY_H
#ifdef __cplusplus
extern "C" {
#endif
int cholesky_tri(const double *A, int n, double *L);
void cholesky_tri_inv(const double *L_, double *invA, const int n, double *x);
#ifdef __cplusplus
}
#endif
#endif
