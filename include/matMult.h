/* Index macros and small dense helpers shared by the solver and by generated
 * problem code (the generated iLQG_func.c includes "matMult.h" and uses
 * MAT_IDX, reference iLQG_func.tem:3,152).
 *
 * Replaces reference matMult.h:4-14.  Conventions (reference matMult.h:4-9):
 *   - dense matrices are column-major, element (r,c) of an nr-row matrix at r + c*nr
 *   - symmetric matrices are stored as the packed upper triangle, column by
 *     column: element (r,c), r<=c, at c*(c+1)/2 + r
 */
#ifndef MATMUL_H
#define MATMUL_H

#define MAT_IDX(r, c, nr) ((r) + (c) * (nr))
#define MAT_IDX3(r, c, b, nr, nc) ((r) + (c) * (nr) + (b) * (nr) * (nc))
#define MAT_IDX4(r, c, i3, i4, nr, nc, n3) \
    ((r) + (c) * (nr) + (i3) * (nr) * (nc) + (i4) * (nr) * (nc) * (n3))

#define UTRI_MAT_IDX(r, c) ((((c) * ((c) + 1)) / 2) + (r))
#define SYMTRI_MAT_IDX(r, c) (((r) > (c)) ? UTRI_MAT_IDX(c, r) : UTRI_MAT_IDX(r, c))

#ifdef __cplusplus
extern "C" {
#endif

/* base[c] += sum_r a[r] * b[r + c*n_r]                      (reference matMult.c:3)  */
void addMulVec(double base[], const double a[], const double b[], const int n_r, const int n_c);
/* packed-upper base += a' * b * a, b symmetric packed        (reference matMult.c:14) */
void addSquareTri(double base[], const double b[], const double a[], const int n_r, const int n_c,
                  double ba[]);
/* full base (n_ca x n_cc) += a' * b * c, b symmetric packed  (reference matMult.c:48) */
void addMul2Tri(double base[], const double b[], const double a[], const int n_ra, const int n_ca,
                const double c[], const int n_rc, const int n_cc, double bc[]);

#ifdef __cplusplus
}
#endif

#endif /* MATMUL_H */
