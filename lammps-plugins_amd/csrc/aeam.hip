// aeam.hip -- placeholder until the AEAM kernels land (next milestone)
#include "mdp_common.h"
int mdp_aeam_prepare(mdp_ctx *c) { return mdp_fail(c, MDP_ENOTIMPL, "aeam: not built yet"); }
int mdp_aeam_run_density(mdp_ctx *c, int) { return mdp_fail(c, MDP_ENOTIMPL, "aeam: not built yet"); }
int mdp_aeam_run_force(mdp_ctx *c, int, int) { return mdp_fail(c, MDP_ENOTIMPL, "aeam: not built yet"); }
extern "C" {
int mdp_aeam_set_tables(mdp_ctx *c, const mdp_aeam_tables *) { return mdp_fail(c, MDP_ENOTIMPL, "aeam: not built yet"); }
int mdp_aeam_density_host(mdp_ctx *c, int, double *, double *, double *, double *) { return mdp_fail(c, MDP_ENOTIMPL, "aeam: not built yet"); }
int mdp_aeam_force_host(mdp_ctx *c, int, int, const double *, double *, double *, double *, double *) { return mdp_fail(c, MDP_ENOTIMPL, "aeam: not built yet"); }
}
