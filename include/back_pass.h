/* Backward Riccati sweep entry point (replaces reference back_pass.h:7).
 * Returns 0 = ok, 1 = box-QP failed at some step (caller raises lambda and
 * retries, reference iLQG.c:267-275). */
#ifndef BACK_PASS_H
#define BACK_PASS_H
#include "iLQG.h"
#ifdef __cplusplus
extern "C" {
#endif
int back_pass(tOptSet *o);
#ifdef __cplusplus
}
#endif
#endif
