// md.hip -- resident-mode MD around the hot path (placeholder: lands in the next milestone)
#include "mdp_common.h"
#define NI(c) return mdp_fail(c, MDP_ENOTIMPL, "resident mode: not built yet")
int mdp_md_build_master_list(mdp_ctx *c) { NI(c); }
extern "C" {
int mdp_md_setup(mdp_ctx *c, const mdp_md_config *, const double *, const double *, const int *, const int *, const double *, const int *, const int *, const double *, const int *, const int *) { NI(c); }
int mdp_md_build_neighbors(mdp_ctx *c) { NI(c); }
int mdp_md_initial_integrate(mdp_ctx *c) { NI(c); }
int mdp_md_final_integrate(mdp_ctx *c) { NI(c); }
int mdp_md_compute(mdp_ctx *c, int, int) { NI(c); }
int mdp_md_pack_x(mdp_ctx *c, int, const int *, const double *, double *) { NI(c); }
int mdp_md_unpack_x(mdp_ctx *c, int, int, const double *) { NI(c); }
int mdp_md_pack_scalar(mdp_ctx *c, int, int, const int *, double *) { NI(c); }
int mdp_md_unpack_scalar(mdp_ctx *c, int, int, int, const double *) { NI(c); }
int mdp_md_pack_ghost_f(mdp_ctx *c, int, int, double *) { NI(c); }
int mdp_md_unpack_add_f(mdp_ctx *c, int, const int *, const double *) { NI(c); }
int mdp_md_fold_self_ghost_f(mdp_ctx *c) { NI(c); }
int mdp_md_aeam_density(mdp_ctx *c, int) { NI(c); }
int mdp_md_aeam_force(mdp_ctx *c, int, int) { NI(c); }
int mdp_md_thermo(mdp_ctx *c, double *) { NI(c); }
int mdp_md_download(mdp_ctx *c, double *, double *, double *, double *) { NI(c); }
int mdp_md_upload_x(mdp_ctx *c, const double *) { NI(c); }
void *mdp_md_ptr(mdp_ctx *, const char *) { return nullptr; }
int mdp_md_neighbor_stats(mdp_ctx *c, long long *) { NI(c); }
}
