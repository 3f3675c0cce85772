// nk2d_precond.hip -- placeholder until the banded factorisation lands
#include "nk2d_common.h"
void nk2d_precond_free(nk2d_ctx* c) { (void)c; }
extern "C" int nk2d_precond_setup(nk2d_ctx* c) { return nk2d_fail(c, "nk2d_precond_setup: not built yet", -9); }
extern "C" int nk2d_precond_apply(nk2d_ctx* c, nk2d_vec v, nk2d_vec out) { (void)v; (void)out; return nk2d_fail(c, "nk2d_precond_apply: not built yet", -9); }
