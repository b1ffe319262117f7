// mdp_common.h -- internal context and helpers of libmdpair_hip.so (gfx950 only)
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "mdpair_hip.h"

#define MDP_NEIGHMASK 0x1FFFFFFF
// global energy/virial accumulators: acc[0..15] final values (0 eng, 1..6 virial, 7 KE, 8 maxdisp2),
// followed by MDP_ACC_SLOTS partial slots of MDP_ACC_STRIDE doubles each
#define MDP_CLUSTER 2
#define MDP_ACC_SLOTS 512
#define MDP_ACC_STRIDE 16
// REBO centre classes: lane-group size (4, 8, 12, 16, 32) x element, x {interior, boundary}: classes 10..19 hold the
// centres whose candidate set reaches a REMOTE ghost (multi-GPU runs); the interior ones run while the halo is in flight
#define MDP_NOVF_LISTS 7 // {count, ids ...} lists in mdp_ctx::ovf, counts zeroed with the accumulators of every compute: [0] centres for the
                         // general kernel, [1..4] the lane-per-centre kernel's overflow per (part, element), [5] tiles with a pair on
                         // the cubic Lennard-Jones spline, [6] those of them whose pair queues overflowed (rows walked after all)
#define MDP_NCLASS 20
#define MDP_NCLASS_HALF 10

// bytes of device memory currently held by the library's buffers in this process (memory_usage(), pair_rebomos.cpp:1113-1124)
inline std::atomic<long long> &mdp_device_bytes_counter()
{
  static std::atomic<long long> bytes{0}; // several contexts (threads) allocate concurrently
  return bytes;
}

// device buffer that only grows
template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n, bool keep = false, hipStream_t s = nullptr)
  {
    if (n <= cap) return hipSuccess;
    size_t ncap = n + n / 8 + 64;
    T *q = nullptr;
    hipError_t e = hipMalloc((void **) &q, ncap * sizeof(T));
    if (e != hipSuccess) return e;
    if (keep && p && cap) {
      e = hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return e;
      e = hipStreamSynchronize(s);
      if (e != hipSuccess) return e;
    }
    if (p) (void) hipFree(p);
    mdp_device_bytes_counter() += (long long) ((ncap - cap) * sizeof(T));
    p = q;
    cap = ncap;
    return hipSuccess;
  }
  void release()
  {
    if (p) (void) hipFree(p);
    mdp_device_bytes_counter() -= (long long) (cap * sizeof(T));
    p = nullptr;
    cap = 0;
  }
  size_t bytes() const { return cap * sizeof(T); }
};

// REBO-MoS parameters as the kernels see them: pair tables flattened [ti*2+tj]
struct RebomosDev {
  double rcmin[4], rcmax[4], rcmaxsq[4], rcinv[4]; // rcinv = 1/(rcmax-rcmin)
  double Q[4], alpha[4], A[4], B[4], beta[4];
  double b[2][7], bg[2][7], a[2][4];
  // LJ: thresholds in rsq space chosen so that the branch taken is bit-identical to the
  // reference's comparisons on rij = sqrt(rsq) (pair_rebomos.cpp:518-532)
  double lj_rsq_lo[4];  // smallest rsq with sqrt(rsq) >= rcLJmin
  double lj_rsq_hi[4];  // largest  rsq with sqrt(rsq) <= rcLJmax
  double lj_rsq_sw[4];  // smallest rsq with sqrt(rsq) >= 0.95*sigma
  double lj1[4], lj2[4], lj3[4], lj4[4];
  double ljc2[4], ljc3[4], rcLJmin[4]; // cubic inner spline (pair_rebomos.cpp:533-543)
  double cand_cutsq[4];                // (rcmax+skin)^2 : REBO candidate list
  double ljlist_cutsq[4];              // (rcLJmax+skin)^2 : trimmed LJ list
};

// The style-level displacement checks of a resident run (has an atom moved half the inner skin since the style's lists
// were built / half the pruning buffer since the rows were pruned?) ride in kernels that touch the positions anyway:
// owned atoms in the integrate kernel, remote ghosts in the halo unpack; periodic self-images move with their owners.
// Flag words are pinned host memory, two sets used alternately (the words of step n are read during step n+1, after
// the integrate kernel of step n+1 was queued -- never in the step that writes them, which would be a host wait on a
// kernel just launched): [0] beyond the list trigger [1] beyond half the inner skin [2] beyond
// the pruning trigger [3] beyond half the buffer; [4..7] the same for remote ghosts.
// Reference positions of the displacement checks (at the last reneighboring, style-list build, row pruning): single
// precision -- the triggers have margins of 0.07-0.1 A, a float resolves 6e-5 A at |x| = 1000 A -- which takes 36 of
// the 216 bytes per atom out of the integrate kernel.
typedef float mdp_hold_t;
struct MdpStyleCheck {
  const mdp_hold_t *xa = nullptr, *xp = nullptr; // positions at the style-list build / at the last row pruning, [nall][3]
  double trig_a = 0, hard_a = 0, trig_p = 0, hard_p = 0; // squared distances
  int *flag = nullptr;
  // accumulators reset by the same kernel (mdp_acc_begin of the compute that follows): acc[0..nacc), flags, ovf[0]
  double *acc = nullptr;
  int nacc = 0;
  int *flags = nullptr, *ovf = nullptr;
  int ovf_stride = 0; // the MDP_NOVF_LISTS list counters (five overflow lists, two lists of cubic-spline tiles) sit at ovf[k * ovf_stride]
};
struct MdpStyleCheckMeta {
  bool has_style = false, has_prune = false;
  long long build_epoch = -1;
  int prune_epoch = -1;
};
// The deferred triggers are read one compute late, so they fire early by these margins (Angstrom; two steps of motion
// for the pruned rows: in a multi-GPU run the owned atoms are checked before the halo of the same step arrives)
constexpr double kStaleMargin = 0.1;
constexpr double kPruneMargin = 0.07;

// uniform Cartesian bin grid over the bounding box of owned+ghost atoms
struct MdpGrid {
  double lo[3], inv[3];
  int n[3];
  int range; // stencil half width in cells: range * cell width >= the cutoff the grid was made for
};

// The reference sizes everything from the potential file (pair_aeam.cpp:752-872; the bundled AlSi.aeam has two
// elements), and so does this library.  Up to MDP_AEAM_MAXT atom types the parameters of the type pairs also ride in
// the kernel arguments (1.8 KB of the 4 KB a launch may carry), where the tile kernels and the templated CSR kernels read
// them as scalars at fixed offsets; with more types only the block in device memory (g_*) exists and the generic CSR
// kernels (PairPar<0>) serve every step.
#define MDP_AEAM_MAXT 8
#define MDP_AEAM_MAXTYPES 64 // (a sanity bound on what a file may declare, not a table size)
struct AeamDev {
  int ntypes, nelements, nnonangular, nrhomax, nrmax;
  // per type pair [(ti-1)*ntypes + (tj-1)], filled when ntypes <= MDP_AEAM_MAXT
  double cut[MDP_AEAM_MAXT * MDP_AEAM_MAXT], rdr[MDP_AEAM_MAXT * MDP_AEAM_MAXT];
  int nr[MDP_AEAM_MAXT * MDP_AEAM_MAXT], t2rhor[MDP_AEAM_MAXT * MDP_AEAM_MAXT], t2z2r[MDP_AEAM_MAXT * MDP_AEAM_MAXT];
  double rdrho[MDP_AEAM_MAXT];
  int nrho[MDP_AEAM_MAXT], t2frho[MDP_AEAM_MAXT];
  // the same in device memory for any number of types: [ntypes*ntypes] / [ntypes]
  const double *g_cut, *g_rdr, *g_rdrho;
  const int *g_nr, *g_t2rhor, *g_t2z2r, *g_nrho, *g_t2frho;
  const double *frho, *rhor, *z2r; // device spline tables [table][row][7]
  const double4 *rhor_v4, *rhor_d4, *z2r_v4, *z2r_d4; // the same rows as aligned {c3..c6} / {c0..c2,0} records
  const double2 *rhor_ys, *z2r_ys; // [table][nrmax+1] (value, slope) = columns 6 and 5 of a row: the persistent tile kernels' tables
  const double2 *pair_d6; // [ntypes*ntypes][nrmax+1][3]: {rho' c0 c1 | rho' c2, phi' c0 | phi' c1 c2} of the pair type, 48 B per row
};

// 30-bit key of a cell (10 bits per axis) along a 3-D Hilbert curve (Skilling, "Programming the Hilbert curve":
// axes -> transposed index).  A Hilbert curve has no jumps: any run of consecutive keys is a compact blob, which
// is what bounds the neighbour unions of the tile lists.
__device__ __forceinline__ unsigned mdp_hilbert30(unsigned x0, unsigned x1, unsigned x2)
{
  constexpr int B = 10;
  unsigned X[3] = {x0, x1, x2};
  for (unsigned Q = 1u << (B - 1); Q > 1; Q >>= 1) {
    const unsigned P = Q - 1;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      if (X[d] & Q)
        X[0] ^= P;
      else {
        const unsigned t = (X[0] ^ X[d]) & P;
        X[0] ^= t;
        X[d] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned t = 0;
  for (unsigned Q = 1u << (B - 1); Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t;
  X[1] ^= t;
  X[2] ^= t;
  unsigned key = 0;
  for (int b = B - 1; b >= 0; b--)
#pragma unroll
    for (int d = 0; d < 3; d++) key = (key << 1) | ((X[d] >> b) & 1u);
  return key;
}

// ---- brick domain decomposition (domain.hip): geometry as the kernels see it
struct DdGeom {
  double lo[3];
  double h[6], hinv[6]; // LAMMPS Domain::h order: xprd, yprd, zprd, yz, xz, xy
  int g[3];             // bricks per dimension
  int me[3], rank, nranks;
  double cutl[3];       // ghost-shell width in lamda (fractional) units
  int ns[3];            // periodic image shifts tested per dimension: -ns .. ns
  int self_remote;      // testing: periodic self-images travel through the transport (to the rank itself)
  int nonper[3];        // 1: no periodic images / no wrap in this dimension (slab, free surface)
};

struct MdpDomain {
  bool on = false;
  DdGeom G;
  double cutghost = 0.0;
  std::vector<int> mig_send, bord_send, bord_recv; // per rank
  int mig_total = 0, nself = 0, nsend = 0, nrecv = 0, nent = 0;
  int nlocal_old = 0, nghost_old = 0;
  long long reneighbors = 0;
  DevBuf<int> dest, counters, idx_a, idx_b, ent_atom, ent_code, ent_cnt, ent_off, sendlist, type_tmp, tag_tmp;
  DevBuf<unsigned long long> key_a, key_b;
  DevBuf<double> sendshift, v_tmp;
  DevBuf<double4> xq_tmp;
  // deferred displacement trigger of the host-level skin (neigh_modify check yes)
  hipEvent_t ev_moved = nullptr, ev_moved_ref = nullptr; // (_ref: ev_moved, or the style-check event behind the same kernel)
  bool moved_pending = false;
  // RCCL transport inside the library (comm_rccl.hip)
  void *nccl_comm = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_arrived = nullptr, ev_lead = nullptr;
  DevBuf<double> sbuf, rbuf, abuf; // abuf: the all-reduce's own words
  bool fwd_pending = false;        // a position exchange is in flight between _forward_begin and _forward_end
  int aeam_pending = 0;            // the aeam fp (1) / fp + ghost force (2) exchange is in flight
  DevBuf<int> cnt_dev;
  // The collective `check yes` decision rides with the halo (mdp_dd_comm_step_begin / _end): the integrate kernel of a
  // step leaves "an owned atom of this rank has moved beyond the trigger" in a device word (two words alternate: the
  // kernel sets one and clears the other), an all-gather of that word follows the position exchange on the
  // communication stream, the halo unpack reduces the ranks' words into a pinned word, and every rank reads the SAME
  // answer at its next step -- no blocking thermo call, no host-side collective.
  DevBuf<double> flagbuf;          // [0..1] this rank's word of the even / odd steps, [2 .. 2 + nranks) the ranks' words as gathered
  int flag_par = 0;
  hipEvent_t ev_glob = nullptr, ev_glob_ref = nullptr; // (_ref: ev_glob, or the style-check event recorded behind the same kernel)
  bool glob_pending = false;
  bool step_mode = false;          // the host drives whole steps (mdp_dd_comm_step_begin / _end): the words are gathered every step
  bool fwd_gathered = false;       // the all-gather of the words was queued behind the position exchange in flight
  bool fresh_ghosts = false;       // the border exchange of a reneighboring carried the current positions
  bool ghost_forces = true;        // aeam: some rank has an angular centre next to a remote ghost (decided per reneighboring)
  long long dangerous = 0;         // step mode: checks that saw an owned atom beyond half the skin
  long long steps_phased = 0;      // aeam steps whose exchanges travelled behind the interior tiles
  // How a step orders its compute against its exchanges (mdp_dd_comm_step_begin / _end), chosen from measurements on
  // the machine the run is on (comm_rccl.hip, "overlap policy"):
  //   0 split     the work that needs no remote ghost of this step is launched behind the start of the exchange
  //   1 lead      as 0, and the compute stream waits until the RCCL kernel of the exchange has started (comm_lead)
  //   2 blocking  exchange first, then the whole compute in its one-GPU order
  //   3 first     rebomos only: of the interior work only the first kernel (lane-per-centre) runs behind the exchange
  //   4 inline    as blocking, with the position exchange queued on the context's own stream (no second stream, no events)
  int ov_policy = -1;              // the choice; -1 while the trial runs
  int ov_forced = -2;              // MDP_OVERLAP_POLICY: -2 not read yet, -1 auto, >= 0 fixed
  int ov_cur = 0;                  // policy of the step in flight
  bool lead = false;               // (policy 1 of the step in flight)
  long long ov_step = 0;           // steps the trial has seen
  double ov_sum[5] = {0, 0, 0, 0, 0}; // device time of the measured steps per policy, ms
  int ov_cnt[5] = {0, 0, 0, 0, 0};
  double ov_mean[5] = {0, 0, 0, 0, 0};
  bool inline_x = false, fwd_inline = false; // (policy 4 of the step in flight; how the exchange in flight was queued)
  hipEvent_t ov_ev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  int ov_slot_pol[4] = {-1, -1, -1, -1}; // policy of the step a slot's event pair brackets (-1: free)
  int ov_slot = -1;                // slot of the step in flight (-1: not measured)
};

constexpr int MDP_UP_RING = 2;      // host-mode upload: pinned staging chunks in flight
constexpr int MDP_DOWN_CHUNKS = 8;  // host-mode download: pieces of the force array
struct mdp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // ---- potentials
  bool have_rebomos = false, have_aeam = false;
  mdp_rebomos_params rebomos_host;
  RebomosDev rebomos;
  AeamDev aeam;
  DevBuf<double> aeam_frho, aeam_rhor, aeam_z2r;
  DevBuf<double4> aeam_rhor_v4, aeam_rhor_d4, aeam_z2r_v4, aeam_z2r_d4;
  DevBuf<double> aeam_pair_d8;
  DevBuf<double2> aeam_rhor_ys, aeam_z2r_ys;
  int lds_max = 0, num_cu = 0; // device limits (queried once)
  bool ptile_reported[3] = {false, false, false};
  DevBuf<int> aeam_maps;
  DevBuf<double> aeam_par_d; // cut | rdr | rdrho of the type pairs / types (AeamDev::g_*)
  DevBuf<int> aeam_par_i;    // nr | t2rhor | t2z2r | nrho | t2frho
  std::vector<double> aeam_hcut; // host copy of cut[ti*ntypes+tj] (list cutoffs, launch geometry)
  DevBuf<double> cut_tab;    // squared list cutoffs per type pair for more than MDP_AEAM_MAXT types (md.hip CutTables)

  // ---- atoms
  int nlocal = 0, nghost = 0, nall = 0, ntypes = 0;
  bool atoms_set = false;
  int map[16];
  DevBuf<double4> xq;   // x,y,z, w = element / type-1 as double
  DevBuf<double4> xq_cell; // the same in cell order (xq[cell_perm[p]]), as of the last mdp_bin_atoms
  DevBuf<double> xraw;  // staging for host x [nall][3]
  bool rebo_host_list = false;  // host mode, rebomos: candidates and LJ rows from the host's neighbor list (mdp_rebomos_host_list)
  bool host_sort = false;       // host mode, rebomos: device storage order = Hilbert order (host_perm: device -> host index)
  DevBuf<int> host_perm;
  DevBuf<double> host_stage;    // per-atom results back in host order before the download
  // host mode on one periodic rank: every ghost is an image of an owned atom.  Once the host has handed over its box
  // (mdp_set_box_host), owner (device order, in ghost_owner) and image counts (as doubles, in ghost_shift) are derived
  // from tags and positions at mdp_set_atoms_host, and the per-step upload carries the owned atoms only.
  bool host_box_set = false, host_ghosts_derived = false;
  bool host_check_armed = false; // rebomos host mode: the displacement check ran behind the last position upload
  double host_h[6] = {1, 1, 1, 0, 0, 0}; // xprd, yprd, zprd, yz, xz, xy (Domain::h)
  DevBuf<int> host_tagmap, host_inv, host_tag_dev; // tag -> owned host index; host -> device index; tags in device order
  DevBuf<double> host_img;                         // [nghost][3] image counts (ghost_shift: the Cartesian shift at the list build)
  std::vector<std::pair<const void *, size_t>> host_regs; // host arrays page-locked in place (large x arrays)
  char *h_up[MDP_UP_RING] = {};                // pinned upload staging (a ring of chunks)
  hipEvent_t ev_up[MDP_UP_RING] = {};
  hipEvent_t ev_down[MDP_DOWN_CHUNKS] = {};    // chunked force download (host mode)
  double *h_down = nullptr;     // pinned download buffer
  size_t h_down_cap = 0;
  DevBuf<int> tag, type;
  DevBuf<double> f;     // [nall][3]
  DevBuf<double> eatom; // [nall]
  DevBuf<double> vatom; // [nall][6] per-atom virial (allocated on first use)
  DevBuf<double> acc;   // [16] eng, virial[6], flags...
  DevBuf<int> flags;    // [4] overflow etc.
  double *h_pinned = nullptr; // pinned staging for small results (32 doubles)
  char *h_small = nullptr;    // pinned scratch of mdp_read_small / mdp_write_small (counts, totals, flag words)
  size_t h_small_cap = 0;

  // ---- master neighbor list (CSR over nall atoms)
  bool neigh_set = false;
  double skin = 0.0;
  DevBuf<long long> nb_off; // [nall+1]
  DevBuf<int> nb;           // [total]
  long long nb_total = 0, nb_owned_total = 0;
  std::vector<long long> h_off;
  std::vector<int> h_nb;

  // ---- REBO-MoS repacked structures
  bool rebo_packed = false;
  DevBuf<int> cand_cnt, cand_off; // [nall+1]
  DevBuf<int> cand;               // REBO candidates (r <= rcmax+skin)
  int cand_total = 0;
  // Lennard-Jones cluster pair list: MDP_CLUSTER consecutive owned atoms share one union list
  DevBuf<long long> lj_off;       // [nclus+1]
  DevBuf<int> lj_cnt;             // [nclus]
  DevBuf<int> lj;                 // union neighbours (r <= rcLJmax+skin of ANY atom of the cluster)
  long long lj_total = 0;
  int nclus = 0, cluster = MDP_CLUSTER;
  // halo overlap (multi-GPU): clusters whose lists reach no remote ghost come first in cl_order
  int remote_start = 1 << 30;
  bool centre_split = false;      // REBO centres split interior / boundary (default with remote ghosts)
  int centres_early = 0;          // rebomos: which interior centre kernels mdp_rebomos_run_begin launched (launch_centres `which`)
  int overlap_mode = 0;           // this step: 0 interior work in compute_begin, 2 nothing early (blocking order), 3 rebomos:
                                  // only the lane-per-centre kernel early (MdpDomain::ov_policy; set per step by the step calls)
  DevBuf<int> cl_flag, cl_pos, cl_order;
  DevBuf<int> lj_split;           // [nclus] number of Mo entries at the head of each row
  // Lennard-Jones tile lists (default): the MDP_TILE consecutive clusters one workgroup handles share
  // the UNION of their neighbours; its coordinates are staged in LDS once per step and the cluster rows
  // hold 16-bit indices into it (lj16), so every global gather is amortised over ~7 uses
  bool lj_tiled = false;
  int ntile = 0, tile_cap = 0, tile_maxu = 0, tile_rowmax = 0; // (rowmax: most row entries of one tile, generic builder)
  int tile_maxu_in = 0, tile_rowmax_in = 0; // the same among the tiles that reach no remote ghost (aeam, bricks)
  int lj_class_base[5] = {0, 0, 0, 0, 0}; // ranges of cl_order: small / large unions (the last two are empty)
  bool lj_ordered = false;        // cl_order in use (otherwise natural order, everything in class 0)
  int tile_small = 0;             // largest union of the "small" launch classes
  DevBuf<int> tu;                 // [ntile][tile_cap] union members (atom index), Mo first then S
  DevBuf<unsigned short> tmask;   // [ntile][tile_cap] bit g: cluster g of the tile lists the member
  // dynamic pruning of the tile rows (resident mode): between two list builds the rows are re-filtered, from the
  // current positions, to the entries within window + prune_buf of a cluster atom; the kernels walk the pruned rows
  // until an atom has moved prune_buf/2 since (second trigger of moved_kernel), then they are pruned again
  DevBuf<unsigned short> lj16_in; // pruned rows, at the offsets of the rows as built
  DevBuf<int> lj_len_in, lj_split_in; // their lengths and splits
  DevBuf<mdp_hold_t> xhold_prune; // [nall][3] positions at the last pruning
  bool prune_valid = false, prune_stale = false;
  double prune_buf = 0.3;
  int prune_epoch = 0, prune_copied_epoch = -1, prunes = 0, dangerous_prunes = 0;
  int computes_since_prune = 0;
  int tile_rows_cl = 2; // atoms per row of the current tile lists
  DevBuf<int> tile_nu;            // [ntile] members of each union
  DevBuf<int> tile_flag;          // [0] a union outgrew tile_cap   [1] largest union   [2] most row entries of a tile
  DevBuf<unsigned short> lj16;    // cluster rows, indices into the tile's union
  DevBuf<int> is_center;          // [nall]
  // fix nve on the device for a host-mode context (mdp_hnve_*, the plugins' `fix nve/mdp`)
  bool hn_on = false;
  double hn_dt = 0.0, hn_dtf = 0.0;
  double hn_mass[16] = {};
  DevBuf<double> hn_mass_dev;
  bool hn_v_current = false;      // c->v / c->rmass match the atoms of the last mdp_set_atoms_host
  int ovf_par = 0;                // which of the two sets of pinned overflow counts (h_pinned + 40) this compute uses
  bool hn_deferred_check = false; // host mode + device integrator, rebomos: the style checks ride in the integrate kernel, read a compute late
  int ang_list_n = -1;            // >= 0: ang_list holds exactly the owned angular centres of the current atoms (mdp_md_build_master_list)
  bool f_prezeroed = false;       // f[0 .. nall) was cleared by the integrate kernel / image refresh of this step (aeam)
  bool f_zero_remote_due = false; // ... except the remote ghosts' part, which this step's halo unpack clears (bricks, step mode)
  bool aeam_img_fp = false;       // the embedding kernel of this compute filled fp of the periodic self-images too
  DevBuf<int> lj_fix_stamp;       // [ntile] stamp of the compute that last listed the tile for rebo_lj_cubic_kernel
  int lj_stamp = 0;
  // hot systems: the flagged pairs of the Lennard-Jones tile kernel in per-group queues (rebo_lj_tile_kernel<.., QUEUE>)
  DevBuf<unsigned short> lj_cq;   // [ntile][16 groups][2 atoms][lj_cq_cap] items
  DevBuf<int> lj_cq_cnt;          // [ntile][16][2] entries (zero between computes)
  int lj_cq_cap = 0, lj_cq_tiles = 0;
  bool lj_queue_now = false;      // this compute's tile launches took the QUEUE variant
  DevBuf<double> lj_fixtab;       // [4][12] lo hi sw lj1 lj2 lj3 lj4 rcLJmin c2 c3 - - per pair type (cubic inner spline, rare path)
  DevBuf<int> cand_stage;         // [nall][64] candidate rows at a fixed stride, written by the counting sweep of a list build
  DevBuf<int> class_list;         // [MDP_NCLASS][nall]   class = 2 * (lane-group size index) + element
  DevBuf<int> class_count;        // [MDP_NCLASS]
  DevBuf<int> pk_cand;            // per class, per centre: its first UA*G candidates, contiguous in class order
  size_t pk_base[MDP_NCLASS] = {};
  DevBuf<int> class_merged;       // per class: interior centres then boundary centres, one list (launches over both halves)
  size_t merged_base[MDP_NCLASS_HALF] = {};
  int h_class_count[MDP_NCLASS] = {};
  DevBuf<unsigned long long> amask; // [nall] bit t: candidate t currently inside rcmax
  DevBuf<int> rev;                // [cand_total] absolute reverse slot (owned rows)
  DevBuf<int> rev16;              // [nlocal][16] the first 16 of them at a fixed stride
  DevBuf<int> ovf;                // 5 x [1+nall+1]: centres handed on this step -- to the general kernel / by the lane-per-centre kernel
  int ovf_stride = 0;
  int ovf3_hot[4] = {0, 0, 0, 0}; // computes left in list mode per (part, element) list of the lane-per-centre kernel
  DevBuf<mdp_hold_t> xhold_all;   // [nall][3] positions when the style lists were built
  double skin_inner = 0.0;        // the style lists' own skin (<= the host's)
  double skin_inner_auto = 1.0;   // adaptive default of it (grows when the displacement trigger fires too often)
  double skin_inner_cap = 1.0e9;  // a skin at which a candidate row outgrew the 64-bit active mask (dense systems)
  long long computes_since_build = 0;
  bool stale_rebuild = false;     // the pending rebuild was asked for by the displacement trigger
  long long style_builds = 0;     // number of style-list builds so far
  long long dangerous_builds = 0; // deferred check saw an atom beyond half the inner skin
  // fused style-level checks (MdpStyleCheck): set armed by the last integrate kernel, event of its last writer
  // The words of a set are READ one compute after they were written (mdp_sflag_collect takes the set of the step
  // before, whose event completed long ago): no host wait on the GPU in the step, on one GPU as on several.
  int sflag_set = 0;
  bool sflag_armed = false;
  bool sflag_committed[2] = {false, false}; // the set's last writer has been queued (event recorded), not yet collected
  MdpStyleCheck sflag_chk;
  MdpStyleCheckMeta sflag_meta[2];
  hipEvent_t ev_sflag[2] = {nullptr, nullptr};
  bool final_pending = false;      // the host deferred the final half-kick of the finished step (mdp_md_defer_final)
  bool final_deferred_seen = false; // the host uses mdp_md_defer_final at all (older hosts: with_final is authoritative)
  bool acc_prezeroed = false; // the integrate kernel reset the accumulators: the next mdp_acc_begin launches nothing
  bool check_now = false;         // positions were rewritten by the host (mdp_md_upload_x): check the lists before use
  DevBuf<double> fnbr;            // [cand_total][4] force on the slot's neighbour + its share of the pair energy
  DevBuf<double> fown;            // [nall][4] the centre's own share: -(sum of its slot forces), energy
  DevBuf<double> vslot;           // [cand_total][6] per-atom virial shares (allocated on first use)
  DevBuf<char> scan_tmp;

  // ---- AEAM work arrays
  DevBuf<double> rho, fp;         // [nall]
  DevBuf<int> ang_list;           // owned angular atoms
  DevBuf<int> ang_count;
  int h_ang_count = 0;
  int aeam_cl = 1;                // atoms per cluster of the AEAM tile lists
  bool aeam_tiled = false;        // resident mode: tile lists (tu / lj16 ...) serve the force-only steps
  bool aeam_device_lists = false; // host mode: lists built on the device from the positions (as rebomos), host list unused
  bool skin_set = false;
  bool csr_full = true;           // the CSR list holds rows for every owned atom (false: angular centres only)
  bool csr_want_full = false;     // a per-atom-virial step ran: keep building the full list
  int aeam_split = 0;             // tiles [0, aeam_split) reach no remote ghost (0: no remote ghosts / no tile lists)
  bool aeam_ang_remote = false;   // an angular centre may have a remote ghost in its row: ghost forces must travel
  int aeam_phase = 0;             // phases of the current compute already done (aeam.hip)
  int aeam_vflag = 0;             // vflag of the current compute (phase A hands it to the early three-body forces)

  // ---- binning (shared by the master-list builder and the cluster-list builder)
  MdpGrid grid;
  double bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0}; // of the atoms last uploaded (host mode)

  // ---- resident MD
  bool md = false;
  mdp_md_config cfg;
  DevBuf<double> v;          // [nlocal][3]
  DevBuf<mdp_hold_t> xhold;  // [nlocal][3] positions at last build
  DevBuf<double> rmass;      // [nlocal]
  DevBuf<int> ghost_owner;   // [nghost]
  DevBuf<double> ghost_shift;// [nghost][3]
  DevBuf<double> mass_type;  // [ntypes+1] per-type masses on the device (rmass of migrated atoms)
  double h_mass[16] = {};
  MdpDomain dd;
  // binning scratch
  DevBuf<int> cell_of, cell_perm, cell_start;
  DevBuf<unsigned> sort_keys_a, sort_keys_b;
  DevBuf<int> sort_vals_b;
  DevBuf<int> nb_cnt;
  double last_eng = 0.0, last_virial[6] = {0, 0, 0, 0, 0, 0};

  // ---- timing
  bool timing = false;
  hipEvent_t ev[8] = {};
  hipEvent_t ev_sb[8] = {}, ev_se[8] = {}; // spans
  unsigned span_mask = 0;                  // spans recorded since the last mark 0 / span reset
  bool timing_spans = false;
  bool ev_made = false;
  int ev_marks = 0;
  double t_ms[8] = {};
};

int mdp_fail(mdp_ctx *c, int code, const char *fmt, ...);

// The deferred displacement triggers are read one compute late, so they fire `margin` early.  The margins were
// measured for the reference inputs' 1 fs step (0.1 A covers two steps at 50 A/ps); they scale with the time step of a
// resident run (a 5 fs step moves atoms five times as far before the answer is read).
inline double mdp_margin_scale(const mdp_ctx *c) { return c->md && c->cfg.dt > 0.001 ? c->cfg.dt / 0.001 : 1.0; }

// Small device <-> host transfers (counts, totals, flag words) go through a pinned scratch buffer of the CONTEXT and
// are complete on return: no asynchronous copy ever targets a stack variable or a std::vector (pageable memory goes
// through the runtime's shared staging path, and a host that drives several contexts from several threads must not
// depend on how that path behaves under concurrency).
struct MdpRead {
  const void *d_src;
  size_t bytes;
  void *h_dst;
};
int mdp_read_small(mdp_ctx *c, const MdpRead *r, int n);                       // waits for the stream
int mdp_read_one(mdp_ctx *c, const void *d_src, size_t bytes, void *h_dst);    // waits for the stream
int mdp_write_small(mdp_ctx *c, void *d_dst, const void *h_src, size_t bytes); // waits for the stream

#define MDP_HIP(c, call)                                                                             \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return mdp_fail((c), MDP_EHIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

#define MDP_TRY(expr)                                                                                \
  do {                                                                                               \
    int rc_ = (expr);                                                                                \
    if (rc_ != MDP_OK) return rc_;                                                                   \
  } while (0)

// modules
int mdp_pack_xq(mdp_ctx *c, const double *d_x3, const int *d_type_or_null, int count = -1); // xraw/type -> xq (first `count` atoms; -1: all)
int mdp_rebomos_host_precheck(mdp_ctx *c);                 // host mode: displacement check behind the position upload
int mdp_host_ghost_scalar(mdp_ctx *c, double *d_a);        // host mode, derived images: a[image] = a[owner]
int mdp_host_ghost_fold(mdp_ctx *c, int w, double *d_a);   // a[owner] += a[image], a[image] = 0 (w doubles per atom)
int mdp_scan_exclusive_int(mdp_ctx *c, const int *d_in, int *d_out, int n);  // d_out[n] = total (n+1 entries)
int mdp_chunk_by_element(mdp_ctx *c, int n, int n_owned, const int *d_idx_in, int *d_idx_out, const double4 *d_xq,
                         const int *d_type, const int *d_map); // element-sorted runs of 32 owned atoms (stable)
int mdp_scan_exclusive_i64(mdp_ctx *c, const int *d_in, long long *d_out, int n);
int mdp_rebomos_repack(mdp_ctx *c);
int mdp_tile_prune(mdp_ctx *c, const double lim_rsq[4]);                 // (re-)prune the tile rows from the current positions
void mdp_prune_adapt(mdp_ctx *c, double buf_max, bool fired);
int mdp_prune_upkeep(mdp_ctx *c, const double cut[4], double skin, bool may_prune, bool *due); // trigger + pruning for a style without its own displacement check
int mdp_tile_lists_build(mdp_ctx *c, const double cutsq[4], int cl, bool *ok); // tile lists (cl atoms per cluster), two classes of atoms (type 0 | others); needs the bin grid
int mdp_rebomos_run(mdp_ctx *c, int eflag, int vflag, bool zero_f);
int mdp_rebomos_run_begin(mdp_ctx *c, int eflag, int vflag);
int mdp_rebomos_run_end(mdp_ctx *c, int eflag, int vflag);
int mdp_aeam_prepare(mdp_ctx *c);
int mdp_aeam_run_begin(mdp_ctx *c, int eflag, int vflag);       // interior density tiles (halo in flight)
int mdp_aeam_run_density(mdp_ctx *c, int eflag);
int mdp_aeam_run_force_begin(mdp_ctx *c, int eflag, int vflag); // interior force tiles (fp / ghost forces in flight)
int mdp_aeam_run_force(mdp_ctx *c, int eflag, int vflag);
int mdp_md_build_master_list(mdp_ctx *c);
int mdp_md_build_neighbors_impl(mdp_ctx *c);
void mdp_dd_release(mdp_ctx *c); // frees everything domain.hip / comm_rccl.hip hold
int mdp_bin_atoms(mdp_ctx *c, double cutoff, const double lo[3], const double hi[3]); // fills c->grid, cell_perm, cell_start
void mdp_time_mark(mdp_ctx *c, int k);
void mdp_span_begin(mdp_ctx *c, int k); // per-phase device time as independent (begin, end) event pairs: aeam
void mdp_span_end(mdp_ctx *c, int k);
int mdp_host_pinned_reserve(mdp_ctx *c, size_t ndoubles); // c->h_down: pinned download buffer (host mode)
int mdp_host_upload(mdp_ctx *c, void *d_dst, const void *h_src, size_t bytes); // pageable host array -> device, pipelined through pinned staging
int mdp_host_refresh_ghosts(mdp_ctx *c);                  // host mode, images kept by the library: owner + count * h of this step
int mdp_md_advance(mdp_ctx *c, bool with_final, int *flag, double trigsq, double hardsq); // integrate kernel (+ displacement check)
void mdp_host_add(double *dst, const double *src, size_t n); // dst += src, threaded for large arrays
int mdp_host_download_add(mdp_ctx *c, double *h_dst, double *h_stage, const double *d_src, size_t n); // chunked D2H + add
int mdp_to_host_order(mdp_ctx *c, int n, int w, const double *d_src, double *d_dst);   // per-atom arrays, device -> host order
int mdp_to_device_order(mdp_ctx *c, int n, int w, const double *d_src, double *d_dst); // host -> device order
int mdp_acc_begin(mdp_ctx *c, bool any); // zero acc (+ slots when any energy/virial is tallied)
void mdp_sflag_arm(mdp_ctx *c, MdpStyleCheck &sc);  // before the integrate kernel: which references, which flag set
int mdp_sflag_commit(mdp_ctx *c);                   // behind the last kernel that writes this step's flag words
void mdp_sflag_drop(mdp_ctx *c);                    // positions were rewritten outside the integrator: flag words in flight say nothing
int mdp_sflag_collect(mdp_ctx *c, bool *far, bool *toofar); // flags of the previous step (waits for their event): style part returned, pruning part applied
int mdp_acc_end(mdp_ctx *c, bool any);   // fold the slots into acc[0..6]
int mdp_flags_check(mdp_ctx *c, const int *hflags5); // overflow bits (last compute | sticky) -> MDP_EOVERFLOW
