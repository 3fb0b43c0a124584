/* Backtracking line search over o->alpha (replaces reference line_search.h:6).
 * Returns 1 if a step was accepted (candidate left in o->candidates[0]). */
#ifndef LINE_SEARCH_H
#define LINE_SEARCH_H
#include "iLQG.h"
#ifdef __cplusplus
extern "C" {
#endif
int line_search(tOptSet *o, int iter);
#ifdef __cplusplus
}
#endif
#endif
