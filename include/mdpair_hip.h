/*
 * mdpair_hip.h -- C-ABI of libmdpair_hip.so, the MI355X (gfx950) device layer behind the
 * LAMMPS pair styles `rebomos` and `aeam`.
 *
 * This is the "thin C-ABI layer" of the north star: plain pointers and sizes, no C++ or torch
 * types.  Host code (the PairREBOMoS / PairAEAM plugin adapters in lammps-plugins_amd/plugin/,
 * the mini-host, and the Python bench/test harness via ctypes) reaches the HIP kernels only
 * through these entry points.  Every entry point names the reference interface it replaces
 * (paths relative to the upstream repo lammps/lammps-plugins).
 *
 * Two ways to feed the hot path:
 *   host mode     -- what a LAMMPS `Pair::compute()` has in hand: host pointers to atom->x/type/tag
 *                    and the paged NeighList (SURVEY.md 8b "Data handed to compute").
 *   resident mode -- atoms, ghosts, neighbor lists, integrator and thermo all live on the GPU
 *                    (mdp_md_*); used by the mini-host and by bench.py.  Device buffers may be
 *                    owned by the caller (e.g. torch tensors) -- see mdp_md_ptr().
 *
 * All functions return 0 on success or a negative MDP_E* code; mdp_last_error() gives the text.
 * Nothing here falls back to the CPU: without a usable HIP device every call fails.
 */
#ifndef MDPAIR_HIP_H
#define MDPAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDP_ABI_VERSION 3

enum {
  MDP_OK = 0,
  MDP_EINVAL = -1,    /* bad argument / call order                                   */
  MDP_EHIP = -2,      /* HIP runtime error (text in mdp_last_error)                  */
  MDP_ENOMEM = -3,
  MDP_EOVERFLOW = -4, /* "Neighbor list overflow, boost neigh_modify one" (pair_rebomos.cpp:350) */
  MDP_ENOTIMPL = -5,  /* feature not available                                       */
  MDP_ESTATE = -6     /* potential / atoms / neighbors not set                       */
};

/* eflag / vflag bits follow LAMMPS ev_setup */
#define MDP_EFLAG_GLOBAL 1
#define MDP_EFLAG_ATOM 2
#define MDP_VFLAG_GLOBAL 1 /* explicit pair virial -> virial[6]; set by either VIRIAL_PAIR or VIRIAL_FDOTR */
#define MDP_VFLAG_ATOM 4

typedef struct mdp_ctx mdp_ctx;

/* ---- lifecycle ------------------------------------------------------------------------- */
int mdp_abi_version(void);
int mdp_device_count(void);
int mdp_create(mdp_ctx **ctx, int device);
int mdp_destroy(mdp_ctx *ctx);
const char *mdp_last_error(const mdp_ctx *ctx);
/* bytes of device memory THIS context holds (ctx = NULL: all contexts of the process): what Pair::memory_usage() should
 * add for the style's device-side lists and work arrays (the reference reports the lists it holds:
 * pair_rebomos.cpp:1113-1124) */
double mdp_device_bytes(const mdp_ctx *ctx);
/* host arrays are staged through pinned buffers of the context by default.  With MDP_HOST_REGISTER=1 large arrays
 * (atom->x) are page-locked in place instead; a host that opts in must call this before it frees or re-allocates
 * such an array (LAMMPS: when atom->nmax grows).  ptr = NULL releases every registration. */
int mdp_host_release(mdp_ctx *ctx, const void *ptr);
int mdp_set_stream(mdp_ctx *ctx, void *hip_stream); /* run everything on this hipStream_t (default: own stream) */
int mdp_sync(mdp_ctx *ctx);

/* ---- potentials --------------------------------------------------------------------------
 * REBO Mo-S: the 61 scalars of MoS.REBO.set5b after mixing, i.e. the protected members of
 * PairREBOMoS (USER-REBOMOS/pair_rebomos.h:54-60) as filled by read_file
 * (pair_rebomos.cpp:964-1066) and init_one (pair_rebomos.cpp:262-265).  Element 0 = Mo, 1 = S. */
typedef struct {
  double rcmin[2][2], rcmax[2][2], rcmaxsq[2][2];
  double Q[2][2], alpha[2][2], A[2][2], BIJc[2][2], Beta[2][2];
  double b[7][2], bg[7][2], a[4][2];
  double rcLJmin[2][2], rcLJmax[2][2], epsilon[2][2], sigma[2][2];
  double lj1[2][2], lj2[2][2], lj3[2][2], lj4[2][2];
} mdp_rebomos_params;

int mdp_rebomos_set_params(mdp_ctx *ctx, const mdp_rebomos_params *p);
/* replaces PairREBOMoS::read_file (pair_rebomos.cpp:857-1066): 61 scalars, one per non-comment line,
 * then the mixing rules and lj1..lj4.  err receives a message shaped like pair_rebomos.cpp:955-957. */
int mdp_rebomos_read_file(const char *path, mdp_rebomos_params *p, char *err, int errlen);
int mdp_rebomos_params_from_scalars(const double *v61, mdp_rebomos_params *p);

/* AEAM: the spline tables PairAEAM::array2spline builds (USER-AEAM/pair_aeam.cpp:876-942) and
 * the Setfl scalars (pair_aeam.h:66-76).  Tables are dense [table][row 0..nmax][7] doubles with
 * 1-based rows exactly as the reference stores them; type maps are (ntypes+1)^2 / (ntypes+1),
 * 1-based (pair_aeam.cpp:785-871). */
typedef struct {
  int ntypes, nelements, nnonangular;
  int nrhomax, nrmax, nfrho, nrhor, nz2r;
  const int *nrho;       /* [nelements]            */
  const double *drho;    /* [nelements]            */
  const int *nr;         /* [nelements*nelements]  */
  const double *dr;      /* [nelements*nelements]  */
  const double *cut;     /* [nelements*nelements]  */
  const int *type2frho;  /* [ntypes+1]             */
  const int *type2rhor;  /* [(ntypes+1)*(ntypes+1)]*/
  const int *type2z2r;   /* [(ntypes+1)*(ntypes+1)]*/
  const double *frho_spline; /* [nfrho][nrhomax+1][7] */
  const double *rhor_spline; /* [nrhor][nrmax+1][7]   */
  const double *z2r_spline;  /* [nz2r][nrmax+1][7]    */
} mdp_aeam_tables;

int mdp_aeam_set_tables(mdp_ctx *ctx, const mdp_aeam_tables *t);
/* replaces PairAEAM::read_file (pair_aeam.cpp:627-746), file2array (:752-872) and array2spline /
 * interpolate (:876-942).  map[1..ntypes] = element index of each atom type, -1 for NULL. */
typedef struct mdp_aeam_file mdp_aeam_file;
int mdp_aeam_file_read(const char *path, mdp_aeam_file **out, char *err, int errlen);
/* mass[mass_cap]: one per element (a file may define up to 64; call with mass = NULL to learn nelements first) */
int mdp_aeam_file_info(const mdp_aeam_file *f, int *nelements, int *nnonangular, int *nangular, double *mass,
                       int mass_cap, char *names, int nameslen);
int mdp_aeam_file_build(mdp_aeam_file *f, int ntypes, const int *map, mdp_aeam_tables *out); /* out points into f */
void mdp_aeam_file_free(mdp_aeam_file *f);

/* ---- host mode: the data a LAMMPS Pair::compute() holds ---------------------------------
 * replaces the reads of atom->x/type/tag/nlocal/nghost (pair_rebomos.cpp:288-290,370-374;
 * pair_aeam.cpp:141-145).  x is [nall][3] (LAMMPS double** is contiguous), type is 1-based,
 * map[1..ntypes] = element index (REBO-MoS: pair_rebomos.cpp:168-179; AEAM: identity-1). */
int mdp_set_atoms_host(mdp_ctx *ctx, int nlocal, int nghost, const double *x, const int *type, const int *tag,
                       int ntypes, const int *map);
int mdp_set_positions_host(mdp_ctx *ctx, const double *x); /* per step, same nlocal/nghost */
/* One periodic rank: every ghost is an image of an owned atom, and Comm::forward_comm gives it its owner's position
 * plus whole box vectors.  A host that hands over its box -- h = Domain::h = {xprd, yprd, zprd, yz, xz, xy}, before
 * mdp_set_atoms_host and again whenever the box changes (fix npt / deform: before each mdp_set_positions_host) -- lets
 * the library do that on the device: at mdp_set_atoms_host owner (by tag) and image counts of every ghost are derived
 * and checked against the uploaded positions (1e-8 A); if all ghosts pass, mdp_host_ghosts_derived() returns 1 and
 *   - mdp_set_positions_host reads x[0..nlocal) only (the images follow with the box of that step),
 *   - mdp_aeam_density_host accepts fp == NULL (and rho == NULL): fp stays on the device, images included,
 *   - mdp_aeam_force_host accepts fp_all == NULL, folds what the images collected onto their owners itself (the
 *     host's reverse_comm then carries zeros for this style) and adds into f[0..nlocal) / vatom[0..nlocal) only.
 * With a ghost of another rank's atom, without tags, or with MDP_HOST_GHOSTS=upload the answer is 0 and everything is
 * as before.  h == NULL withdraws the box. */
int mdp_set_box_host(mdp_ctx *ctx, const double *h /* [6] or NULL */);
int mdp_host_ghosts_derived(mdp_ctx *ctx);

/* replaces the reads of list->inum/gnum/ilist/numneigh/firstneigh (pair_rebomos.cpp:304-307,
 * pair_aeam.cpp:150-153).  Call when the host has rebuilt its list (neighbor->ago == 0).
 * firstneigh[i] points into LAMMPS' pages; entries are masked with NEIGHMASK here.
 * skin = neighbor->skin: the list was built at r <= cut+skin and stays valid until the next call;
 * the device repacks it (trimmed, coalesced) once per call. */
int mdp_set_neighbors_host(mdp_ctx *ctx, int inum, int gnum, const int *ilist, const int *numneigh,
                           int *const *firstneigh, double skin);
/* replaces the read of neighbor->skin alone.  The REBO-MoS device path derives its own trimmed lists
 * (REBO candidates, Lennard-Jones cluster pair lists) from the positions, as LAMMPS' own GPU/KOKKOS
 * packages do with `neigh yes`; the host's list is requested for API parity and for the ghost shell it
 * implies but its entries are not read, so a rebomos host only reports the skin (at every reneighboring,
 * after mdp_set_atoms_host).  neigh_modify exclusions are therefore not honoured. */
int mdp_set_skin(mdp_ctx *ctx, double skin);
/* guard that goes with mdp_set_skin: the reference iterates the host's entries (pair_rebomos.cpp:304-307, 328-330,
 * 490-495), so exclusions, special_bonds or skip lists change ITS result.  Compares the host's owned-row entry count
 * with the number of geometric pairs inside cutneigh (= sqrt(cutsq)+skin, the host's list cutoff) counted on the
 * device from the atoms last uploaded, and scans a sample of rows for special-bond bits; MDP_EINVAL + message on
 * any difference.  Call at every reneighboring after mdp_set_atoms_host.  MDP_SKIP_LIST_CHECK=1 disables it. */
int mdp_rebomos_check_host_list(mdp_ctx *ctx, int inum, const int *ilist, const int *numneigh, int *const *firstneigh,
                                double cutneigh);
/* rebomos, host mode: 1 = candidates (REBO_neigh, pair_rebomos.cpp:281-352) and Lennard-Jones rows (FLJ, :490-495)
 * are taken from the HOST's full + ghost neighbor list, handed over with mdp_set_neighbors_host (inum + gnum rows,
 * entries masked with NEIGHMASK as the reference does, :328-330) at every reneighboring -- subsets of what the host
 * listed, so `neigh_modify exclude` and special-bond settings act as in the reference instead of stopping the run
 * (mdp_rebomos_check_host_list is then not called).  The device keeps the host's atom order and the host's ghosts
 * (no mdp_set_box_host images, no fix nve/mdp) and pays the upload of the list: ~640 entries per atom at the
 * reference's 13.4 A list cutoff.  Call before mdp_set_atoms_host.  0 (default): lists from the positions. */
int mdp_rebomos_host_list(mdp_ctx *ctx, int on);
/* aeam, host mode: 1 = the style builds its lists on the device from the positions, exactly as rebomos does
 * (bins, tile lists, CSR rows of the angular centres) and the host only reports its skin (mdp_set_skin) at every
 * reneighboring; the device keeps its own Hilbert-sorted copy of the atoms and returns results in the host's order.
 * The host's list is then not read -- flattening its 86 entries per atom on a host thread and uploading them cost
 * more than ten steps per reneighboring at a million atoms -- so, as for rebomos, a list that is not the plain
 * geometric one must be refused: mdp_aeam_check_host_list (per-type-pair cutoffs cut[ti][tj] + skin,
 * pair_aeam.cpp:615-621).  Call before mdp_set_atoms_host.  0 (default) = stream the host's list
 * (mdp_set_neighbors_host), the reference's own data flow (pair_aeam.cpp:150-153). */
int mdp_aeam_device_lists(mdp_ctx *ctx, int on);
int mdp_aeam_check_host_list(mdp_ctx *ctx, int inum, const int *ilist, const int *numneigh, int *const *firstneigh,
                             double skin);
/* same as mdp_set_neighbors_host, from a CSR copy (tests / hosts that already hold a flat list): offset[nall+1] */
int mdp_set_neighbors_csr_host(mdp_ctx *ctx, int nall, const int *numneigh, const long long *offset,
                               const int *neigh, double skin);

/* replaces PairREBOMoS::compute (pair_rebomos.cpp:102-111): REBO_neigh + FREBO + FLJ + virial.
 * f[nlocal][3] and eatom[nlocal] are ACCUMULATED into (LAMMPS zeroes them); forces are complete
 * for owned atoms (owner-computes: nothing is written to ghosts, so the host's reverse_comm adds
 * zeros).  virial[6] is the explicit pair virial (xx,yy,zz,xy,xz,yz) of this rank's owned atoms,
 * to be used with no_virial_fdotr = 1.  vatom[nlocal][6] (vflag & 4) is the per-atom virial in the
 * reference's split (ev_tally halves, v_tally3 thirds, v_tally2 halves: pair_rebomos.cpp:444,554,707-725),
 * complete for owned atoms.  Any of eng/virial/eatom/vatom may be NULL. */
int mdp_rebomos_compute_host(mdp_ctx *ctx, int eflag, int vflag, double *f, double *eng_vdwl, double *virial,
                             double *eatom, double *vatom);

/* replaces PairAEAM::compute (pair_aeam.cpp:110-479) in two halves around the style's own
 * forward_comm (pair_aeam.cpp:307):
 *   density:  passes 1+2 (rho, F, F'); writes fp[0..nlocal) (host) = F'(rho) chain factor the
 *             neighbors need, adds the embedding energy to *eng_vdwl / eatom.
 *   force:    pass 3; fp_all[nall] must hold the owners' values on ghosts (after forward_comm).
 *             f[nall][3] is accumulated into: owned atoms fully, ghosts only with the angular
 *             three-body terms (folded by the host's reverse_comm).  vatom[nall][6] (vflag & 4, may be
 *             NULL) follows ev_tally / ev_tally3 (pair_aeam.cpp:393,472): halves / thirds, ghosts as for f.
 * With mdp_host_ghosts_derived() == 1 fp / rho / fp_all may be NULL (see mdp_set_box_host): no fp round trip. */
int mdp_aeam_density_host(mdp_ctx *ctx, int eflag, double *fp, double *rho, double *eng_vdwl, double *eatom);
int mdp_aeam_force_host(mdp_ctx *ctx, int eflag, int vflag, const double *fp_all, double *f, double *eng_vdwl,
                        double *virial, double *eatom, double *vatom);

/* ---- resident mode: device-side MD around the hot path -------------------------------------
 * (neighbor build, integrator, thermo.)  One sub-domain per context. */
typedef struct {
  int style;          /* 1 = rebomos, 2 = aeam                                         */
  int nlocal, nghost; /* owned atoms, ghost atoms (periodic self-images + remote halo) */
  int ntypes;
  double skin;
  double dt;
  double ftm2v, mvv2e;   /* unit constants (metal: SURVEY.md Appendix B)               */
  double bbox_lo[3], bbox_hi[3]; /* Cartesian bounds of owned+ghost atoms, for binning  */
  int nghost_self;    /* ghosts [0,nghost_self) are periodic self-images (refreshed on the device), the rest are
                         remote atoms filled by mdp_md_unpack_x; lists that reach none of the latter need not
                         wait for the halo (mdp_md_compute_begin / _end)                                   */
  int master_list;    /* 1 = also build the LAMMPS-style full list of every owned atom (rebomos: at 3*rcmax+skin,
                         log.rebomos-bulk.1:43; aeam: per-type-pair cutoffs + skin, 86 entries/atom) for its
                         statistics.  By default rebomos builds none (its kernels use the style's own lists) and
                         aeam with tile lists builds the rows of the angular centres only; a per-atom-virial step
                         (CSR kernels) switches the full aeam list on by itself.                              */
} mdp_md_config;

/* upload a sub-domain.  x/v/type/tag: owned atoms [nlocal]; ghosts: ghost_owner[g] = local index
 * of the owner (>=0: periodic self-image, refreshed on the device each step) or -1 (remote: filled
 * by mdp_md_unpack_ghosts), ghost_shift[g][3] Cartesian image shift, ghost_type/tag.
 * mass[1..ntypes].  map as in mdp_set_atoms_host. */
int mdp_md_setup(mdp_ctx *ctx, const mdp_md_config *cfg, const double *x, const double *v, const int *type,
                 const int *tag, const double *mass, const int *map, const int *ghost_owner,
                 const double *ghost_shift, const int *ghost_type, const int *ghost_tag);
/* binned full(+ghost) neighbor list on the device with the cutoffs the style's init_one() returns
 * (pair_rebomos.cpp:244-274, pair_aeam.cpp:615-621) + skin, then the style's repack. */
int mdp_md_build_neighbors(mdp_ctx *ctx);
int mdp_md_initial_integrate(mdp_ctx *ctx); /* fix nve: v += dt/2 f/m; x += dt v; refresh self-image ghosts */
int mdp_md_final_integrate(mdp_ctx *ctx);   /* v += dt/2 f/m */
/* final_integrate of the finished step and initial_integrate of the next in one pass over the atoms (same operations,
 * same order: bit-identical trajectory).  For a host that needs the full-step velocities of the finished step for
 * nothing (no thermo output, no dump at that step): it skips mdp_md_final_integrate there and opens the next step
 * with this call instead of mdp_md_initial_integrate. */
int mdp_md_final_initial_integrate(mdp_ctx *ctx);
/* the host's declaration that it skips mdp_md_final_integrate for the step just computed and will open the next step
 * with mdp_md_final_initial_integrate / mdp_md_integrate_check(with_final).  Until then the velocities on the device
 * are half-step velocities: mdp_md_thermo and a velocity download complete the kick themselves first (and the
 * with_final call that follows then applies only the initial half-kick), so nothing reads half-step values unnoticed. */
int mdp_md_defer_final(mdp_ctx *ctx);
int mdp_md_compute(mdp_ctx *ctx, int eflag, int vflag); /* force_clear + Pair::compute on the device */
/* the same in two halves for multi-GPU runs: _begin needs only owned atoms, self-image ghosts and LAST step's
 * remote ghosts (it runs while this step's halo exchange is in flight: the work whose lists reach no remote
 * ghost -- rebomos: the interior REBO centres; aeam: the density of the interior tiles); _end needs the unpacked
 * halo (aeam: _end is density + force + self-image fold for a host without exchanges of its own; a multi-GPU host
 * calls the four aeam phases below instead). */
int mdp_md_compute_begin(mdp_ctx *ctx, int eflag, int vflag);
int mdp_md_compute_end(mdp_ctx *ctx, int eflag, int vflag);
/* halo plumbing for multi-GPU: pack x (or the AEAM fp) of owned atoms sendlist[n] (+shift) into buf;
 * unpack recv buffers into ghosts [first, first+n).  buf/sendlist/shift are DEVICE pointers. */
int mdp_md_pack_x(mdp_ctx *ctx, int n, const int *d_sendlist, const double *d_shift, double *d_buf);
int mdp_md_unpack_x(mdp_ctx *ctx, int first_ghost, int n, const double *d_buf);
int mdp_md_pack_scalar(mdp_ctx *ctx, int which, int n, const int *d_sendlist, double *d_buf);
int mdp_md_unpack_scalar(mdp_ctx *ctx, int which, int first_ghost, int n, const double *d_buf);
int mdp_md_pack_ghost_f(mdp_ctx *ctx, int first_ghost, int n, double *d_buf);       /* reverse comm */
int mdp_md_unpack_add_f(mdp_ctx *ctx, int n, const int *d_sendlist, const double *d_buf);
int mdp_md_fold_self_ghost_f(mdp_ctx *ctx); /* reverse comm for periodic self-images */
/* AEAM only: compute in phases around the style's own exchanges -- forward_comm of fp (pair_aeam.cpp:307, 946-963)
 * and the host's reverse_comm of the forces that three-body terms put on ghosts (pair_aeam.cpp:462-470):
 *   mdp_md_compute_begin     [optional] density of the tiles that reach no remote ghost | position halo in flight
 *   mdp_md_aeam_density      the remaining density, angular centres, embedding; when _compute_begin did run, also the
 *                            three-body forces (they need the centre's own F' only), so that ghost forces are final
 *   mdp_md_aeam_force_begin  [optional] pair forces of the interior tiles           | fp and ghost forces in flight
 *   mdp_md_aeam_force        the remaining pair forces (three-body forces if not done yet); needs fp on the ghosts
 * Without the optional calls the two others do everything, as before.  eflag / vflag must agree between the phases
 * of one compute.  mdp_md_aeam_state: out[0] = phases done so far in the current compute (bit 1: interior density,
 * bit 2: three-body forces, bit 3: interior forces), [1] = tiles that reach no remote ghost, [2] = tiles,
 * [3] = 1 when an angular centre may have a remote ghost in its row (otherwise no ghost force is ever non-zero and
 * the reverse exchange can be skipped by all ranks alike). */
int mdp_md_aeam_density(mdp_ctx *ctx, int eflag);
int mdp_md_aeam_force_begin(mdp_ctx *ctx, int eflag, int vflag);
int mdp_md_aeam_force(mdp_ctx *ctx, int eflag, int vflag);
int mdp_md_aeam_state(mdp_ctx *ctx, int out[4]);
/* thermo: out[0]=KE(owned), [1]=PE (eng_vdwl of last compute), [2..7]=virial of last compute,
 * [8] = max squared displacement since the last neighbor build (neigh_modify check yes) */
int mdp_md_thermo(mdp_ctx *ctx, double out[9]);
int mdp_md_download(mdp_ctx *ctx, double *x, double *v, double *f, double *eatom); /* owned atoms; NULLs skipped */
int mdp_md_upload_x(mdp_ctx *ctx, const double *x);                             /* owned atoms [nlocal][3] */
int mdp_md_download_x_all(mdp_ctx *ctx, double *x_all); /* owned atoms then ghosts, [nlocal+nghost][3] (diagnostics) */
/* device pointers of resident arrays for zero-copy plumbing (name: "x","v","f","fp","eatom") */
void *mdp_md_ptr(mdp_ctx *ctx, const char *name);
/* statistics of the last neighbor build: out[0]=total master entries (owned; 0 if not built),
 * [1]=ghost-list entries, [2]=LJ row entries incl. padding (rebomos), [3]=REBO candidate entries, [4]=#centres,
 * [5]=#centres with at most three neighbours (one lane each), [6]=#centres in 12- and 16-lane groups, [7]=style-list
 * builds so far (rebomos) /
 * #angular atoms (aeam) */
int mdp_md_neighbor_stats(mdp_ctx *ctx, long long out[8]);
/* dynamic pruning of the tile rows in resident runs (between list builds the rows are re-filtered from the current
 * positions to the entries within window + buffer of a cluster atom; the reference walks its whole list every step,
 * pair_rebomos.cpp:490-521 -- same pairs evaluated, fewer entries tested):
 * out[0]=prunings so far, [1]=prunings that came late (an atom had moved more than half the buffer when the
 * deferred trigger was read; the analogue of LAMMPS' "dangerous builds"), [2]=1 if the kernels currently walk pruned
 * rows, [3]=buffer in units of 1e-6 Angstrom */
int mdp_md_prune_stats(mdp_ctx *ctx, long long out[4]);
/* out[0] = skin of the style's own lists in effect (rebomos: the inner skin, adaptive or MDP_INNER_SKIN; aeam: the
 * host's), [1] = cap on the inner skin, 0 = none (a candidate row outgrew the 64-bit active mask at that skin: the
 * lists were rebuilt with half of it), [2] = pruning buffer in effect (0: rows as built),
 * [3] = list builds that came late (an atom was beyond half the inner skin when the deferred trigger was read),
 * [4] = rebomos: centres that outgrew the lane-per-centre kernel (a fourth neighbour) in the last step whose count has
 * reached the host, [5] = 1 while such centres are collected on a device list and taken by the 8-lane-group kernel
 * (0: they go to the general kernel's list), [6..7] reserved */
int mdp_md_list_state(mdp_ctx *ctx, double out[8]);
/* shape of the rebomos style's own Lennard-Jones lists after the last build (host and resident mode):
 * out[0]=1 tile lists / 0 per-cluster lists (fallback), [1]=#tiles, [2]=union stride, [3]=largest union,
 * [4]=row entries incl. padding, [5]=#clusters, [6]=#tiles in the large-union launch classes,
 * [7]=style-list builds so far.  (No reference counterpart: the CPU style reads the host's list.) */
int mdp_rebomos_list_info(mdp_ctx *ctx, long long out[8]);
/* how the work of the last compute was spread over the kernel classes (diagnostics of a bench line; no reference
 * counterpart -- the CPU style has one loop, pair_rebomos.cpp:358-447, pair_aeam.cpp:337-475):
 * rebomos: out[0..19] = centres per launch class at the last style-list build, class = 2 * (lane-group index: 0 one lane
 * per centre, 1..4 groups of 8 / 12 / 16 / 32 lanes) + element, + 10 for centres that reach a remote ghost;
 * [20..23] = tiles per Lennard-Jones launch class (small unions, large unions, two unused); [24] = centres the general
 * kernel's list received in the last compute whose count has reached the host (lane-group overflow); [25..28] = centres
 * with a fourth neighbour per (interior / boundary, element) list of the lane-per-centre kernel; [29] = largest tile union
 * of the small launch class, [30] = largest union.
 * aeam: out[0] = angular centres, [1] = tiles, [2] = tiles that reach no remote ghost, [30] = largest union. */
int mdp_md_class_stats(mdp_ctx *ctx, long long out[32]);

/* ---- resident mode, domain decomposition on the device ----------------------------------------------------------
 * What the reference gets from the LAMMPS host at every reneighboring -- Domain::remap, Comm::exchange,
 * Comm::borders (the REQ_GHOST list of pair_rebomos.cpp:218 presupposes them; processor grid of
 * log.rebomos-bulk.4:22) -- done per rank on the GPU: one brick of the periodic (triclinic) box per context.
 * The library packs and unpacks DEVICE buffers; the caller moves the bytes between ranks (RCCL all-to-all) and sees
 * only per-rank counts.  Sequence at a reneighboring (every rank, same step):
 *     mdp_dd_migrate_begin  -> counts of atoms leaving for each rank        [exchange counts]
 *     mdp_dd_migrate_pack   -> 8 doubles per leaver, rank-major             [exchange records]
 *     mdp_dd_migrate_end    <- arrivals; owned atoms re-ordered along a Hilbert curve over the brick
 *     mdp_dd_borders_begin  -> counts of ghost entries for each rank        [exchange counts]
 *     mdp_dd_borders_pack   -> 6 doubles per entry, rank-major              [exchange records]
 *     mdp_dd_borders_end    <- remote ghosts; then mdp_md_build_neighbors
 * and per step mdp_dd_forward_pack / _unpack (3 doubles per send-list entry).  A one-rank run needs no transport:
 * mdp_dd_reneighbor does the whole sequence.  Dimensions are periodic (as in both bundled inputs) unless
 * mdp_dd_config.nonperiodic says otherwise. */
typedef struct {
  double boxlo[3];
  double h[6];        /* xprd, yprd, zprd, yz, xz, xy  (LAMMPS Domain::h) */
  int procgrid[3];    /* bricks per dimension; rank = (ix*py + iy)*pz + iz */
  int rank;
  double cutghost;    /* ghost-shell width: the host's list cutoff, pair cutoff + skin (log.rebomos-bulk.1:43) */
  int self_remote;    /* 0.  Testing aid: 1 = periodic self-images are treated as remote ghosts that travel through the
                         transport to the rank itself (exercises every exchange with a single rank / GPU) */
  int nonperiodic[3]; /* 0 0 0 (both bundled inputs: `boundary p p p`).  1 = this dimension is not periodic (LAMMPS
                         boundary f / s / m: a slab, a free surface, a cluster): no images across it, positions are
                         not wrapped, atoms beyond the box belong to the brick at that end */
} mdp_dd_config;

int mdp_dd_setup(mdp_ctx *ctx, const mdp_dd_config *cfg); /* after mdp_md_setup (owned atoms in any order, ghosts optional) */
int mdp_dd_reneighbor(mdp_ctx *ctx);                       /* one-rank runs: remap, order, self-image ghosts, lists */
int mdp_dd_migrate_begin(mdp_ctx *ctx, int *send_counts /* [nranks] */);
int mdp_dd_migrate_pack(mdp_ctx *ctx, double *d_buf);
int mdp_dd_migrate_end(mdp_ctx *ctx, int narrive, const double *d_buf);
int mdp_dd_borders_begin(mdp_ctx *ctx, int *send_counts /* [nranks] */);
int mdp_dd_borders_pack(mdp_ctx *ctx, double *d_buf);
int mdp_dd_borders_end(mdp_ctx *ctx, const int *recv_counts /* [nranks] */, const double *d_buf);
/* out[0]=nlocal [1]=periodic self-image ghosts [2]=send-list entries [3]=remote ghosts [4]=reneighborings so far
 * [5]=atoms that left at the last one [6]=nranks [7]=rank; per-rank counts of the per-step halo (may be NULL) */
int mdp_dd_info(mdp_ctx *ctx, long long out[8], int *send_counts, int *recv_counts);
int mdp_dd_forward_pack(mdp_ctx *ctx, double *d_buf);          /* x of the send list (+ image shift) */
int mdp_dd_forward_unpack(mdp_ctx *ctx, const double *d_buf);  /* -> remote ghosts */
int mdp_dd_forward_scalar_pack(mdp_ctx *ctx, double *d_buf);   /* AEAM fp (pair_aeam.cpp:307, 946-963) */
int mdp_dd_forward_scalar_unpack(mdp_ctx *ctx, const double *d_buf);
int mdp_dd_reverse_pack(mdp_ctx *ctx, double *d_buf);          /* forces on remote ghosts (AEAM angular terms) */
int mdp_dd_reverse_unpack(mdp_ctx *ctx, const double *d_buf);  /* += onto the send-list atoms */
/* ---- the same exchanges done by the library on RCCL (csrc/comm_rccl.hip): for C++ hosts ---------------------------
 * What LAMMPS' Comm brick does over MPI for the reference (exchange / borders / forward_comm / reverse_comm; "Comm"
 * is 5.67 % of log.rebomos-bulk.4:67) as grouped ncclSend/ncclRecv between the bricks' GPUs over xGMI.  RCCL is
 * bound at run time (dlopen of librccl.so.1); MDP_ENOTIMPL if it cannot be loaded.  Rank 0 creates the id, the host
 * distributes its 128 bytes (MPI_Bcast, a file, ...), every rank calls mdp_dd_comm_init after mdp_dd_setup.  All
 * calls below are collective over the ranks and run on the context's stream; the per-step position exchange uses
 * a stream of its own between _begin and _end, so that work launched in between (mdp_md_compute_begin) overlaps it. */
int mdp_dd_comm_unique_id(void *id128);
/* Which RCCL object is bound: the name goes to buf (may be NULL).  Returns 0 for a RCCL library, 1 when MDP_RCCL_LIBRARY
 * points at the repository's TEST DOUBLE (tests/native/fake_rccl.cpp: several ranks sharing one GPU, host-staged -- a
 * rehearsal of the exchange schedule whose timings mean nothing; every report has to say so), MDP_ENOTIMPL when nothing
 * loads.  MDP_RCCL_LIBRARY=<path> binds that one object instead of librccl.so.1; nothing else is tried then. */
int mdp_dd_comm_library(char *buf, int cap);
int mdp_dd_comm_init(mdp_ctx *ctx, const void *id128);
int mdp_dd_comm_destroy(mdp_ctx *ctx);
int mdp_dd_comm_reneighbor(mdp_ctx *ctx);       /* Comm::exchange + Comm::borders + Neighbor::build */
int mdp_dd_comm_forward_begin(mdp_ctx *ctx);    /* Comm::forward_comm: x of the send list -> remote ghosts */
int mdp_dd_comm_forward_end(mdp_ctx *ctx);
int mdp_dd_comm_forward_scalar(mdp_ctx *ctx);   /* AEAM fp (pair_aeam.cpp:307) */
int mdp_dd_comm_reverse(mdp_ctx *ctx);          /* Comm::reverse_comm of f (AEAM angular terms) */
/* the two aeam exchanges of a step in one group of sends and receives on the communication stream, between
 * mdp_md_aeam_density (after mdp_md_compute_begin: three-body forces done) and mdp_md_aeam_force; work launched in
 * between (mdp_md_aeam_force_begin) overlaps them.  with_reverse = 0 leaves the ghost forces out (all ranks alike:
 * no rank has an angular centre next to a remote ghost, mdp_md_aeam_state out[3] reduced over the ranks).  _begin
 * also folds the periodic self-images' forces (mdp_md_fold_self_ghost_f). */
int mdp_dd_comm_aeam_exchange_begin(mdp_ctx *ctx, int with_reverse);
int mdp_dd_comm_aeam_exchange_end(mdp_ctx *ctx);
int mdp_dd_comm_allreduce(mdp_ctx *ctx, double *vals, int n, int op /* 0 sum, 1 max */);
/* A whole step of a multi-GPU run in two calls (what the LAMMPS host does around Pair::compute in Verlet::run:
 * initial_integrate, neighbor->decide, exchange / borders or forward_comm, force, final_integrate;
 * log.rebomos-bulk.4:22,65-67 is such a run on four ranks).  The `neigh_modify every 1 check yes` decision is collective
 * without a collective of the host's: every rank's "an owned atom moved beyond the trigger" word of a step is gathered
 * behind that step's position exchange and read -- the same value on all ranks -- at the next step.
 *   _begin: with_final = complete the half-kick a deferred step left; force_rebuild: -1 decide from the gathered word,
 *           0 never, 1 reneighbor now; *reneighbored = 1 when the call reneighbored.  Integrates, reneighbors or starts
 *           the position exchange, launches what needs no remote ghost of this step.
 *   _end:   waits for the exchange, runs the rest of the compute (aeam: fp / ghost-force exchange behind the interior
 *           pair forces when possible, the same exchanges in the blocking order otherwise), final half-kick now
 *           (defer_final = 0) or fused into the next _begin (1).
 * mdp_dd_comm_step_info: out[0] = aeam steps on the phased order, [1] = 1 if ghost forces travel, [2] = 1 if the last
 * _begin reneighbored, [3] = reneighborings so far, [4] = checks that saw an owned atom beyond half the skin,
 * [5] = overlap policy, [6] = 1 if MDP_OVERLAP_POLICY fixed it, [7] = its mean device time per step in the trial (ns).
 * Overlap policy: how a step orders its compute against its exchanges -- 0 "split" (what needs no remote ghost of this
 * step is launched behind the start of the exchange), 1 "lead" (as 0, and the compute stream waits until the RCCL kernel
 * has started), 2 "blocking" (exchange, then the whole compute: rebomos with one launch per centre class over its
 * interior and boundary halves), 3 "first" (rebomos: only the first interior kernel behind the exchange), 4 "inline"
 * (as blocking, the position exchange queued on the context's own stream: no second stream, no events).  The sends and
 * receives are the same in all of them.  The first steps of a run are a trial (blocks of 4 steps per policy, two
 * rounds; -1 in out[5] while it runs): device time per step between two events, MAX over the ranks, cheapest policy
 * kept.  MDP_OVERLAP_POLICY = split | lead | blocking | first | inline skips the trial. */
int mdp_dd_comm_step_begin(mdp_ctx *ctx, int with_final, int force_rebuild, int eflag, int vflag, int *reneighbored);
int mdp_dd_comm_step_end(mdp_ctx *ctx, int eflag, int vflag, int defer_final);
int mdp_dd_comm_step_info(mdp_ctx *ctx, long long out[8]);

/* `neigh_modify every 1 delay 0 check yes` (sample.in:17-18, log.rebomos-bulk.1:46) without a host round trip per
 * step: *moved = outcome of the check launched by the PREVIOUS call (0 right after a reneighboring), then a check of
 * the current positions is launched.  Trigger: an owned atom moved more than skin/2 - 0.1 A since the last build
 * (the margin covers the step the answer is late by); *dangerous = an atom was beyond skin/2 itself. */
int mdp_md_moved_async(mdp_ctx *ctx, int *moved, int *dangerous);
/* mdp_md_initial_integrate (with_final != 0: mdp_md_final_initial_integrate) and mdp_md_moved_async in one call and one
 * pass over the atoms: same results, the check reads the new positions while they are in registers. */
int mdp_md_integrate_check(mdp_ctx *ctx, int with_final, int *moved, int *dangerous);
/* owned atoms' "tag" / "type" in device order (the device re-orders atoms at every reneighboring); "tile_nu"
 * (diagnostics): {members of the neighbour union, Mo / first-type members} of every 32-atom tile, 2 ints per tile */
int mdp_md_download_int(mdp_ctx *ctx, const char *name, int *out);

/* ---- host mode with the integrator on the device (the plugins' `fix nve/mdp`) -----------------------------------------
 * The reference's styles run under the host's fix nve, which reads atom->f and writes atom->x / atom->v on the host
 * every step (LAMMPS Verlet::run; the reference repository itself registers a fix style from a plugin,
 * USER-BFIELD/bfieldplugin.cpp:15-29, virtuals fix_bfield.h:33-38).  With these calls a host-mode context does the two
 * half-kicks and the drift on the device: between two reneighborings of the host nothing per atom crosses the link.
 * Needs the periodic images kept by the library (mdp_set_box_host, one periodic rank).  Sequence:
 *     mdp_hnve_setup (once: dt, force->ftm2v, atom->mass[1..ntypes])
 *     at every reneighboring: mdp_set_atoms_host (+ skin / list check as before), mdp_hnve_upload_v(atom->v)
 *     per step: mdp_hnve_initial -> [compute with f == NULL: forces stay on the device] -> mdp_hnve_final
 *     mdp_hnve_download(x, v, f) before anything on the host reads them (reneighboring, thermo / dump steps)
 * mdp_hnve_initial: *moved = an owned atom has moved half the host's skin minus a margin since the last
 * mdp_set_atoms_host, as seen by the PREVIOUS call (no wait for the kernel just queued) -- the host reneighbors at its
 * next step; *dangerous = one was beyond half the skin already.  rebomos: the style's own checks (device-built lists,
 * pruned rows) ride in the same integrate kernel and are read by the NEXT compute, as in resident runs, so no call of a
 * step waits for the stream.  mdp_*_compute_host / mdp_aeam_force_host accept
 * f == NULL while the integrator is on (per-atom tallies then need f).  mdp_hnve_off: back to plain host mode. */
int mdp_hnve_setup(mdp_ctx *ctx, double dt, double ftm2v, const double *mass_per_type /* [ntypes+1], 1-based */, int ntypes);
int mdp_hnve_off(mdp_ctx *ctx);
int mdp_hnve_upload_v(mdp_ctx *ctx, const double *v /* [nlocal][3], the host's atom order */);
int mdp_hnve_initial(mdp_ctx *ctx, int *moved, int *dangerous);
int mdp_hnve_final(mdp_ctx *ctx);
int mdp_hnve_download(mdp_ctx *ctx, double *x, double *v, double *f /* [nlocal][3] each, or NULL */);

/* per-phase device time of the last compute in ms (HIP events on the compute stream):
 * rebomos: [0]=REBO centre kernels of the lane-group classes, [1]=the general kernel (centres that outgrew their lane
 * group since the list build), [2]=row pruning (0 unless one was due), [3]=LJ+gather kernel;
 * aeam: [0]=density of the metal centres (tile kernel), [1]=density of the angular centres, [2]=embedding,
 * [3]=force tile kernel (incl. force_clear), [4]=angular three-body forces.
 * Enabled by mdp_set_timing(ctx,1). */
int mdp_set_timing(mdp_ctx *ctx, int on);
int mdp_get_timing(mdp_ctx *ctx, double ms[8]);

#ifdef __cplusplus
}
#endif
#endif /* MDPAIR_HIP_H */
