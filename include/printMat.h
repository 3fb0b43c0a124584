/* Debug printers through PRNT (replaces reference printMat.h). */
#ifndef PRINTMAT_H
#define PRINTMAT_H
#ifdef __cplusplus
extern "C" {
#endif
void printVec(const double *A, const int n, const char *nm);
void printTri(const double *A, const int n, const char *nm);
void printMat(const double *A, const int n, const int m, const char *nm);
#ifdef __cplusplus
}
#endif
#endif
