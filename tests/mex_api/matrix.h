/* TEST INFRASTRUCTURE: MATLAB splits the mx* declarations off into matrix.h; tests/mex_api/mex.h declares both parts. */
#include "mex.h"
