/*
 * orpm_hess.c — CPU ORACLE Lagrangian Hessian (test infrastructure; PARITY UNPINNED, see orpm.h).
 * Placeholder until SURVEY §8 row f-1 (eval_h) is built: the exact-Hessian mode reports
 * zero entries.  Reference: Core/LpHessian.cpp:12-599,878-1018,1192-2161.
 */
#include <stdlib.h>
#include "orpm.h"

void* orpm_hess_create(orpm* o) { (void)o; return NULL; }
void orpm_hess_destroy(void* h) { (void)h; }
int orpm_hess_nnz(void* h) { (void)h; return 0; }
void orpm_hess_structure(orpm* o, int* iRow, int* jCol) { (void)o; (void)iRow; (void)jCol; }
void orpm_eval_h(orpm* o, const double* x, double obj_factor, const double* lambda, double* values) {
  (void)o; (void)x; (void)obj_factor; (void)lambda; (void)values;
}
