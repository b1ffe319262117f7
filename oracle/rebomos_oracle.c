/*
 * oracle/rebomos_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, double precision, single thread) of the REBO Mo-S
 * pair-style hot path of lammps/lammps-plugins:
 *     PairREBOMoS::compute      USER-REBOMOS/pair_rebomos.cpp:102-111
 *     PairREBOMoS::REBO_neigh   USER-REBOMOS/pair_rebomos.cpp:281-352
 *     PairREBOMoS::FREBO        USER-REBOMOS/pair_rebomos.cpp:358-447
 *     PairREBOMoS::FLJ          USER-REBOMOS/pair_rebomos.cpp:453-558
 *     PairREBOMoS::bondorder    USER-REBOMOS/pair_rebomos.cpp:571-847
 *     gSpline / PijSpline / Sp  USER-REBOMOS/pair_rebomos.h:68-211
 *     read_file / init_one      USER-REBOMOS/pair_rebomos.cpp:857-1066, 244-274
 * plus the LAMMPS host bookkeeping the style calls (ev_tally, v_tally2,
 * v_tally3, virial_fdotr_compute; semantics as in SURVEY.md Appendix A).
 *
 * It keeps the reference's *scatter* formulation (tag-parity half selection,
 * forces written to owned AND ghost atoms, energy split half/half) so its raw
 * outputs are what LAMMPS would see after compute() and before reverse_comm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this file.  The product (libmdpair_hip.so) never links or loads it.
 *
 * PARITY PIN: the LAMMPS headers the reference needs are not in this image, so
 * the reference itself is unbuildable here.  This restatement
 * is pinned by the reference's only known-answer data,
 * USER-REBOMOS/log.rebomos-bulk.1:54-56 / log.rebomos-bulk.4:54-56 (thermo
 * rows at steps 0/10/20), reproduced by tests/test_oracle_rebomos.py.  Branches
 * that log never enters (switching-function interior, LJ cubic inner spline)
 * are pinned only by energy/force finite-difference consistency.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rebomos_oracle.h"

#define ORC_PI 3.14159265358979323846
#define ORC_2PI 6.28318530717958647692
#define TOL 1.0e-9 /* pair_rebomos.cpp:52 */

/* MathSpecial::powint: square-and-multiply */
static double powint(double x, int n)
{
  double yy, ww;
  int nn;
  if (x == 0.0) return 0.0;
  nn = (n > 0) ? n : -n;
  ww = x;
  for (yy = 1.0; nn != 0; nn >>= 1, ww *= ww)
    if (nn & 1) yy *= ww;
  return (n > 0) ? yy : 1.0 / yy;
}

/* ---- parameter file: one scalar per non-comment line, fixed order ---------
 * pair_rebomos.cpp:884-948 (order), :964-1066 (mixing), :262-265 (lj1..4)   */
int rebomos_oracle_read_params(const char *filename, rebomos_oracle_params *P)
{
  FILE *fp = fopen(filename, "r");
  char line[1024];
  double v[61];
  int n = 0;
  if (!fp) return -1;
  while (n < 61 && fgets(line, sizeof line, fp)) {
    char *s = line, *h;
    if ((h = strchr(s, '#'))) *h = 0; /* PotentialFileReader strips comments */
    while (*s == ' ' || *s == '\t') ++s;
    if (*s == 0 || *s == '\n' || *s == '\r') continue;
    v[n++] = strtod(s, NULL); /* first token of the line */
  }
  fclose(fp);
  if (n != 61) return -2;
  rebomos_oracle_params_from_scalars(v, P);
  return 0;
}

void rebomos_oracle_params_from_scalars(const double *v, rebomos_oracle_params *P)
{
  int k, a, b;
  memset(P, 0, sizeof *P);
#define SYM3(F, i)                                                                       \
  P->F[0][0] = v[i];                                                                     \
  P->F[0][1] = P->F[1][0] = v[i + 1];                                                    \
  P->F[1][1] = v[i + 2];
  SYM3(rcmin, 0)
  SYM3(rcmax, 3)
  SYM3(Q, 6)
  SYM3(alpha, 9)
  SYM3(A, 12)
  SYM3(BIJc, 15)
  SYM3(Beta, 18)
#undef SYM3
  for (a = 0; a < 2; a++)
    for (b = 0; b < 2; b++) P->rcmaxsq[a][b] = P->rcmax[a][b] * P->rcmax[a][b];
  for (k = 0; k < 7; k++) {
    P->b[k][0] = v[21 + k];  /* M_b0..6  */
    P->bg[k][0] = v[28 + k]; /* M_bg0..6 */
    P->b[k][1] = v[35 + k];  /* S_b0..6  */
    P->bg[k][1] = v[42 + k]; /* S_bg0..6 */
  }
  for (k = 0; k < 4; k++) {
    P->a[k][0] = v[49 + k];
    P->a[k][1] = v[53 + k];
  }
  {
    double eps_MM = v[57], eps_SS = v[58], sig_MM = v[59], sig_SS = v[60];
    P->sigma[0][0] = sig_MM;
    P->sigma[0][1] = P->sigma[1][0] = (sig_MM + sig_SS) / 2;
    P->sigma[1][1] = sig_SS;
    P->epsilon[0][0] = eps_MM;
    P->epsilon[0][1] = P->epsilon[1][0] = sqrt(eps_MM * eps_SS);
    P->epsilon[1][1] = eps_SS;
  }
  for (a = 0; a < 2; a++)
    for (b = 0; b < 2; b++) {
      P->rcLJmin[a][b] = P->rcmin[a][b];
      P->rcLJmax[a][b] = 2.5 * P->sigma[a][b];
      P->lj1[a][b] = 48.0 * P->epsilon[a][b] * powint(P->sigma[a][b], 12);
      P->lj2[a][b] = 24.0 * P->epsilon[a][b] * powint(P->sigma[a][b], 6);
      P->lj3[a][b] = 4.0 * P->epsilon[a][b] * powint(P->sigma[a][b], 12);
      P->lj4[a][b] = 4.0 * P->epsilon[a][b] * powint(P->sigma[a][b], 6);
    }
  P->cut3rebo = 3.0 * P->rcmax[0][0]; /* :257 */
}

/* ---- switching function, pair_rebomos.h:195-211 --------------------------- */
static double Sp(double X, double Xmin, double Xmax, double *dX)
{
  double t = (X - Xmin) / (Xmax - Xmin);
  if (t <= 0.0) {
    *dX = 0.0;
    return 1.0;
  }
  if (t >= 1.0) {
    *dX = 0.0;
    return 0.0;
  }
  *dX = (-0.5 * ORC_PI * sin(t * ORC_PI)) / (Xmax - Xmin);
  return 0.5 * (1.0 + cos(t * ORC_PI));
}

/* Horner evaluation of c[0]+c[1]x+..+c[6]x^6 and its derivative, in the
 * operation order of pair_rebomos.h:80-102 */
static double poly6(const double c[7][2], int t, double x, double *d)
{
  double g = c[6][t] * x, dg = 6.0 * c[6][t] * x;
  int k;
  for (k = 5; k >= 2; k--) {
    g += c[k][t];
    dg += (double) k * c[k][t];
    g *= x;
    dg *= x;
  }
  g += c[1][t];
  dg += c[1][t];
  g *= x;
  g += c[0][t];
  *d = dg;
  return g;
}

/* angular function G and dG/dcos, pair_rebomos.h:68-167 */
static double gspline(const rebomos_oracle_params *P, double c, int t, double *dgdc)
{
  if (c >= -1.0 && c < 0.5) { return poly6(P->b, t, c, dgdc); }
  if (c >= 0.5 && c <= 1.0) {
    double dgcos, dgamma;
    double gcos = poly6(P->b, t, c, &dgcos);
    double gamma = poly6(P->bg, t, c, &dgamma);
    double tmp = ORC_2PI * (c - 0.5);
    double psi = 0.5 * (1 - cos(tmp));
    double dpsi = ORC_PI * sin(tmp);
    *dgdc = dgcos + dpsi * (gamma - gcos) + psi * (dgamma - dgcos);
    return gcos + psi * (gamma - gcos);
  }
  *dgdc = 0.0;
  return 0.0;
}

/* coordination function P(N), pair_rebomos.h:173-179 */
static double pijspline(const rebomos_oracle_params *P, double NM, double NS, int t, double *dp)
{
  double N = NM + NS;
  *dp = -P->a[0][t] + P->a[1][t] * P->a[2][t] * exp(-P->a[2][t] * N);
  return -P->a[0][t] * (N - 1) - P->a[1][t] * exp(-P->a[2][t] * N) + P->a[3][t];
}

/* ---- working state --------------------------------------------------------- */
typedef struct {
  const rebomos_oracle_params *P;
  const double *x; /* [nall][3] */
  const int *elem; /* map[type]: 0 Mo, 1 S */
  const int *tag;
  int nlocal, nall;
  double *f;
  int eflag_global, eflag_atom, vflag_tally, vflag_atom;
  double eng, *eatom, vir_tally[6], *vatom;
  /* REBO neighbor list */
  int *rn_first, *rn_num, *rn;
  double *nM, *nS;
} work_t;

static void tally_v(work_t *W, const double v[6], const int *idx, int n, double share)
{
  int a, k;
  if (W->vflag_tally)
    for (k = 0; k < 6; k++) W->vir_tally[k] += v[k];
  if (W->vflag_atom)
    for (a = 0; a < n; a++)
      for (k = 0; k < 6; k++) W->vatom[6 * (size_t) idx[a] + k] += share * v[k];
}

/* Pair::ev_tally, newton_pair on */
static void ev_tally(work_t *W, int i, int j, double evdwl, double fpair, double dx, double dy, double dz)
{
  double v[6];
  int idx[2];
  if (W->eflag_global) W->eng += evdwl;
  if (W->eflag_atom) {
    W->eatom[i] += 0.5 * evdwl;
    W->eatom[j] += 0.5 * evdwl;
  }
  if (W->vflag_tally || W->vflag_atom) {
    v[0] = dx * dx * fpair;
    v[1] = dy * dy * fpair;
    v[2] = dz * dz * fpair;
    v[3] = dx * dy * fpair;
    v[4] = dx * dz * fpair;
    v[5] = dy * dz * fpair;
    idx[0] = i;
    idx[1] = j;
    tally_v(W, v, idx, 2, 0.5);
  }
}

/* Pair::v_tally2 */
static void v_tally2(work_t *W, int i, int j, double fpair, const double *d)
{
  double v[6];
  int idx[2];
  if (!(W->vflag_tally || W->vflag_atom)) return;
  v[0] = d[0] * d[0] * fpair;
  v[1] = d[1] * d[1] * fpair;
  v[2] = d[2] * d[2] * fpair;
  v[3] = d[0] * d[1] * fpair;
  v[4] = d[0] * d[2] * fpair;
  v[5] = d[1] * d[2] * fpair;
  idx[0] = i;
  idx[1] = j;
  tally_v(W, v, idx, 2, 0.5);
}

/* Pair::v_tally3(i,j,k,fi,fj,drik,drjk): v = drik (x) fi + drjk (x) fj */
static void v_tally3(work_t *W, int i, int j, int k, const double *fi, const double *fj, const double *drik,
                     const double *drjk)
{
  double v[6];
  int idx[3];
  if (!(W->vflag_tally || W->vflag_atom)) return;
  v[0] = drik[0] * fi[0] + drjk[0] * fj[0];
  v[1] = drik[1] * fi[1] + drjk[1] * fj[1];
  v[2] = drik[2] * fi[2] + drjk[2] * fj[2];
  v[3] = drik[0] * fi[1] + drjk[0] * fj[1];
  v[4] = drik[0] * fi[2] + drjk[0] * fj[2];
  v[5] = drik[1] * fi[2] + drjk[1] * fj[2];
  idx[0] = i;
  idx[1] = j;
  idx[2] = k;
  tally_v(W, v, idx, 3, 1.0 / 3.0);
}

/* tag-parity half selection, pair_rebomos.cpp:394-402 / 498-506.
 * returns 1 if the (i,j) visit is skipped */
static int parity_skip(const work_t *W, int i, int j)
{
  int itag = W->tag[i], jtag = W->tag[j];
  const double *xi = W->x + 3 * (size_t) i, *xj = W->x + 3 * (size_t) j;
  if (itag > jtag) {
    if ((itag + jtag) % 2 == 0) return 1;
  } else if (itag < jtag) {
    if ((itag + jtag) % 2 == 1) return 1;
  } else {
    if (xj[2] < xi[2]) return 1;
    if (xj[2] == xi[2] && xj[1] < xi[1]) return 1;
    if (xj[2] == xi[2] && xj[1] == xi[1] && xj[0] < xi[0]) return 1;
  }
  return 0;
}

/* ---- REBO_neigh, pair_rebomos.cpp:281-352 ---------------------------------- */
static void rebo_neigh(work_t *W, const int *numneigh, const long long *offset, const int *neigh)
{
  const rebomos_oracle_params *P = W->P;
  int i, jj, n, total = 0;
  for (i = 0; i < W->nall; i++) {
    const double *xi = W->x + 3 * (size_t) i;
    int it = W->elem[i];
    const int *jl = neigh + offset[i];
    double dS;
    n = 0;
    W->nM[i] = W->nS[i] = 0.0;
    W->rn_first[i] = total;
    for (jj = 0; jj < numneigh[i]; jj++) {
      int j = jl[jj] & 0x1FFFFFFF; /* NEIGHMASK */
      int jt = W->elem[j];
      double dx = xi[0] - W->x[3 * (size_t) j + 0];
      double dy = xi[1] - W->x[3 * (size_t) j + 1];
      double dz = xi[2] - W->x[3 * (size_t) j + 2];
      double rsq = dx * dx + dy * dy + dz * dz;
      if (rsq < P->rcmaxsq[it][jt]) {
        W->rn[total + n++] = j;
        if (jt == 0)
          W->nM[i] += Sp(sqrt(rsq), P->rcmin[it][jt], P->rcmax[it][jt], &dS);
        else
          W->nS[i] += Sp(sqrt(rsq), P->rcmin[it][jt], P->rcmax[it][jt], &dS);
      }
    }
    W->rn_num[i] = n;
    total += n;
  }
}

/* ---- bondorder, pair_rebomos.cpp:571-847 ------------------------------------
 * one side of the bond: centre c, partner o, neighbours k of c (k != o).
 * rco = x_c - x_o.  Returns p_co and scatters the many-body forces.
 * The i-side (pair_rebomos.cpp:606-725) and j-side (:731-843) of the reference
 * are the same mathematics with c<->o exchanged (cos_ijl = -rij.rjl/(|rij||rjl|)
 * is rji.rjl/...), so one routine serves both. */
static double bo_side(work_t *W, int c, int o, const double rco[3], double rmag, double VA, double dwco)
{
  const rebomos_oracle_params *P = W->P;
  const double *x = W->x;
  double *f = W->f;
  int ct = W->elem[c];
  const int *nb = W->rn + W->rn_first[c];
  int nn = W->rn_num[c], kk, d;
  double Etmp = 0.0, dp, PS, p, tmp, tmp2, dgdc, g;

  for (kk = 0; kk < nn; kk++) {
    int k = nb[kk];
    if (k != o) {
      int kt = W->elem[k];
      double rck[3], rckmag, wck, dS, cosv;
      for (d = 0; d < 3; d++) rck[d] = x[3 * (size_t) c + d] - x[3 * (size_t) k + d];
      rckmag = sqrt(rck[0] * rck[0] + rck[1] * rck[1] + rck[2] * rck[2]);
      wck = Sp(rckmag, P->rcmin[ct][kt], P->rcmax[ct][kt], &dS);
      cosv = (rco[0] * rck[0] + rco[1] * rck[1] + rco[2] * rck[2]) / (rmag * rckmag);
      if (cosv > 1.0) cosv = 1.0;
      if (cosv < -1.0) cosv = -1.0;
      g = gspline(P, cosv, ct, &dgdc);
      Etmp += wck * g;
    }
  }

  PS = pijspline(P, W->nM[c], W->nS[c], ct, &dp);
  p = 1.0 / sqrt(1.0 + Etmp + PS);
  tmp = -0.5 * p * p * p;

  for (kk = 0; kk < nn; kk++) {
    int k = nb[kk];
    if (k != o) {
      int kt = W->elem[k];
      double rck[3], rckmag, wck, dwck, cosv, rr = 0;
      double dcdc[3], dcdo[3], dcdk[3], fc[3], fo[3], fk[3];
      for (d = 0; d < 3; d++) rck[d] = x[3 * (size_t) c + d] - x[3 * (size_t) k + d];
      rckmag = sqrt(rck[0] * rck[0] + rck[1] * rck[1] + rck[2] * rck[2]);
      wck = Sp(rckmag, P->rcmin[ct][kt], P->rcmax[ct][kt], &dwck);
      cosv = (rco[0] * rck[0] + rco[1] * rck[1] + rco[2] * rck[2]) / (rmag * rckmag);
      if (cosv > 1.0) cosv = 1.0;
      if (cosv < -1.0) cosv = -1.0;
      rr = rmag * rckmag;
      for (d = 0; d < 3; d++) {
        /* d cos / d x_c, d x_k, d x_o  (pair_rebomos.cpp:648-665) */
        dcdc[d] = ((rco[d] + rck[d]) / rr) - (cosv * ((rco[d] / (rmag * rmag)) + (rck[d] / (rckmag * rckmag))));
        dcdk[d] = (-rco[d] / rr) + (cosv * (rck[d] / (rckmag * rckmag)));
        dcdo[d] = (-rck[d] / rr) + (cosv * (rco[d] / (rmag * rmag)));
      }
      g = gspline(P, cosv, ct, &dgdc);
      tmp2 = VA * 0.5 * (tmp * wck * dgdc);
      for (d = 0; d < 3; d++) {
        fo[d] = -tmp2 * dcdo[d];
        fc[d] = -tmp2 * dcdc[d];
        fk[d] = -tmp2 * dcdk[d];
      }
      /* d w_ck . g  (:683-689) */
      tmp2 = VA * 0.5 * (tmp * dwck * g) / rckmag;
      for (d = 0; d < 3; d++) {
        fc[d] -= tmp2 * rck[d];
        fk[d] += tmp2 * rck[d];
      }
      /* P'(N) d w_ck  (:693-699) */
      tmp2 = VA * 0.5 * (tmp * dp * dwck) / rckmag;
      for (d = 0; d < 3; d++) {
        fc[d] -= tmp2 * rck[d];
        fk[d] += tmp2 * rck[d];
      }
      for (d = 0; d < 3; d++) {
        f[3 * (size_t) c + d] += fc[d];
        f[3 * (size_t) o + d] += fo[d];
        f[3 * (size_t) k + d] += fk[d];
      }
      {
        double roc[3], rkc[3];
        for (d = 0; d < 3; d++) {
          roc[d] = -rco[d];
          rkc[d] = -rck[d];
        }
        /* v = (x_o-x_c) (x) f_o + (x_k-x_c) (x) f_k   (:707-711, :826-829) */
        v_tally3(W, c, o, k, fo, fk, roc, rkc);
      }
    }
  }

  /* P'(N) d w_co  (:716-725, :835-843) */
  tmp2 = -VA * 0.5 * (tmp * dp * dwco) / rmag;
  for (d = 0; d < 3; d++) {
    f[3 * (size_t) c + d] += rco[d] * tmp2;
    f[3 * (size_t) o + d] -= rco[d] * tmp2;
  }
  v_tally2(W, c, o, tmp2, rco);
  return p;
}

/* ---- FREBO, pair_rebomos.cpp:358-447 --------------------------------------- */
static void frebo(work_t *W)
{
  const rebomos_oracle_params *P = W->P;
  const double *x = W->x;
  double *f = W->f;
  int i, kk, d;
  for (i = 0; i < W->nlocal; i++) {
    int it = W->elem[i];
    const int *nb = W->rn + W->rn_first[i];
    for (kk = 0; kk < W->rn_num[i]; kk++) {
      int j = nb[kk], jt;
      double del[3], rsq, rij, wij, dwij, VR, pre, dVRdi, VA, dVA, bij, fpair, evdwl, pij, pji, rji[3];
      if (parity_skip(W, i, j)) continue;
      jt = W->elem[j];
      for (d = 0; d < 3; d++) del[d] = x[3 * (size_t) i + d] - x[3 * (size_t) j + d];
      rsq = del[0] * del[0] + del[1] * del[1] + del[2] * del[2];
      rij = sqrt(rsq);
      wij = Sp(rij, P->rcmin[it][jt], P->rcmax[it][jt], &dwij);
      if (wij <= TOL) continue;

      VR = wij * (1.0 + (P->Q[it][jt] / rij)) * P->A[it][jt] * exp(-P->alpha[it][jt] * rij);
      pre = wij * P->A[it][jt] * exp(-P->alpha[it][jt] * rij);
      dVRdi = pre * ((-P->alpha[it][jt]) - (P->Q[it][jt] / rsq) - (P->Q[it][jt] * P->alpha[it][jt] / rij));
      dVRdi += VR / wij * dwij;

      VA = -wij * P->BIJc[it][jt] * exp(-P->Beta[it][jt] * rij);
      dVA = -P->Beta[it][jt] * VA;
      dVA += VA / wij * dwij;

      pij = bo_side(W, i, j, del, rij, VA, dwij);
      for (d = 0; d < 3; d++) rji[d] = -del[d];
      pji = bo_side(W, j, i, rji, rij, VA, dwij);
      bij = 0.5 * (pij + pji);

      fpair = -(dVRdi + bij * dVA) / rij;
      for (d = 0; d < 3; d++) {
        f[3 * (size_t) i + d] += del[d] * fpair;
        f[3 * (size_t) j + d] -= del[d] * fpair;
      }
      evdwl = VR + bij * VA;
      ev_tally(W, i, j, evdwl, fpair, del[0], del[1], del[2]);
    }
  }
}

/* ---- FLJ, pair_rebomos.cpp:453-558 ----------------------------------------- */
static void flj(work_t *W, const int *numneigh, const long long *offset, const int *neigh)
{
  const rebomos_oracle_params *P = W->P;
  const double *x = W->x;
  double *f = W->f;
  int i, jj, d;
  for (i = 0; i < W->nlocal; i++) {
    int it = W->elem[i];
    const int *jl = neigh + offset[i];
    for (jj = 0; jj < numneigh[i]; jj++) {
      int j = jl[jj] & 0x1FFFFFFF, jt;
      double del[3], rsq, rij, VLJ = 0.0, dVLJ = 0.0, fpair;
      if (parity_skip(W, i, j)) continue;
      jt = W->elem[j];
      for (d = 0; d < 3; d++) del[d] = x[3 * (size_t) i + d] - x[3 * (size_t) j + d];
      rsq = del[0] * del[0] + del[1] * del[1] + del[2] * del[2];
      rij = sqrt(rsq);
      if (rij > P->rcLJmax[it][jt] || rij < P->rcLJmin[it][jt]) {
        VLJ = 0;
        dVLJ = 0;
      } else if (rij <= P->rcLJmax[it][jt] && rij >= 0.95 * P->sigma[it][jt]) {
        double r2inv = 1.0 / rsq, r6inv = r2inv * r2inv * r2inv;
        VLJ = r6inv * (P->lj3[it][jt] * r6inv - P->lj4[it][jt]);
        dVLJ = -r6inv * (P->lj1[it][jt] * r6inv - P->lj2[it][jt]) / rij;
      } else if (rij < 0.95 * P->sigma[it][jt] && rij >= P->rcLJmin[it][jt]) {
        double sg = P->sigma[it][jt], ep = P->epsilon[it][jt];
        double dr = 0.95 * sg - P->rcLJmin[it][jt];
        double r6 = powint((sg / (0.95 * sg)), 6);
        double vdw = 4 * ep * r6 * (r6 - 1.0);
        double dvdw = (-4 * ep / (0.95 * sg)) * r6 * (12.0 * r6 - 6.0);
        double c2 = ((3.0 / dr) * vdw - dvdw) / dr;
        double c3 = (vdw / (dr * dr) - c2) / dr;
        double drp = rij - P->rcLJmin[it][jt];
        VLJ = drp * drp * (drp * c3 + c2);
        dVLJ = drp * (3.0 * drp * c3 + 2.0 * c2);
      }
      fpair = -dVLJ / rij;
      for (d = 0; d < 3; d++) {
        f[3 * (size_t) i + d] += del[d] * fpair;
        f[3 * (size_t) j + d] -= del[d] * fpair;
      }
      ev_tally(W, i, j, VLJ, fpair, del[0], del[1], del[2]);
    }
  }
}

/* ---- compute, pair_rebomos.cpp:102-111 -------------------------------------- */
int rebomos_oracle_compute(const rebomos_oracle_params *P, int nlocal, int nghost, const double *x,
                           const int *elem, const int *tag, const int *numneigh, const long long *offset,
                           const int *neigh, int eflag, int vflag, double *f, double *eng_vdwl,
                           double *virial_fdotr, double *virial_tally, double *eatom, double *vatom,
                           double *nM_out, double *nS_out, int *rebo_numneigh_out, int phases)
{
  work_t W;
  int nall = nlocal + nghost, i, k;
  long long tot = 0;
  memset(&W, 0, sizeof W);
  W.P = P;
  W.x = x;
  W.elem = elem;
  W.tag = tag;
  W.nlocal = nlocal;
  W.nall = nall;
  W.f = f;
  W.eflag_global = eflag & 1;
  W.eflag_atom = (eflag & 2) && eatom;
  W.vflag_tally = (vflag & 1) && virial_tally;
  W.vflag_atom = (vflag & 4) && vatom;
  W.eatom = eatom;
  W.vatom = vatom;
  if (W.eflag_atom) memset(eatom, 0, sizeof(double) * nall);
  if (W.vflag_atom) memset(vatom, 0, sizeof(double) * 6 * nall);
  for (i = 0; i < nall; i++) tot += numneigh[i];
  W.rn_first = (int *) malloc(sizeof(int) * (nall + 1));
  W.rn_num = (int *) malloc(sizeof(int) * (nall + 1));
  W.rn = (int *) malloc(sizeof(int) * (size_t) (tot + 1));
  W.nM = (double *) malloc(sizeof(double) * (nall + 1));
  W.nS = (double *) malloc(sizeof(double) * (nall + 1));
  if (!W.rn_first || !W.rn_num || !W.rn || !W.nM || !W.nS) return -1;

  rebo_neigh(&W, numneigh, offset, neigh);
  if (phases & 1) frebo(&W);
  if (phases & 2) flj(&W, numneigh, offset, neigh);

  if (eng_vdwl) *eng_vdwl = W.eng;
  if (virial_tally)
    for (k = 0; k < 6; k++) virial_tally[k] = W.vir_tally[k];
  if (virial_fdotr) { /* Pair::virial_fdotr_compute over owned+ghost */
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (i = 0; i < nall; i++) {
      const double *xi = x + 3 * (size_t) i, *fi = f + 3 * (size_t) i;
      v[0] += xi[0] * fi[0];
      v[1] += xi[1] * fi[1];
      v[2] += xi[2] * fi[2];
      v[3] += xi[0] * fi[1];
      v[4] += xi[0] * fi[2];
      v[5] += xi[1] * fi[2];
    }
    for (k = 0; k < 6; k++) virial_fdotr[k] = v[k];
  }
  if (nM_out) memcpy(nM_out, W.nM, sizeof(double) * nall);
  if (nS_out) memcpy(nS_out, W.nS, sizeof(double) * nall);
  if (rebo_numneigh_out) memcpy(rebo_numneigh_out, W.rn_num, sizeof(int) * nall);
  free(W.rn_first);
  free(W.rn_num);
  free(W.rn);
  free(W.nM);
  free(W.nS);
  return 0;
}
