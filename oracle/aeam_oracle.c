/*
 * oracle/aeam_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, double precision, single thread) of the angular-EAM
 * pair-style hot path of lammps/lammps-plugins:
 *     PairAEAM::compute       USER-AEAM/pair_aeam.cpp:110-479
 *     read_file               USER-AEAM/pair_aeam.cpp:627-746
 *     file2array              USER-AEAM/pair_aeam.cpp:752-872
 *     array2spline/interpolate USER-AEAM/pair_aeam.cpp:876-942
 * plus the LAMMPS host bookkeeping it calls (ev_tally, ev_tally3,
 * virial_fdotr_compute; semantics as in SURVEY.md Appendix A).
 *
 * Scatter formulation exactly as the reference: every full-list (i,j) visit
 * writes f[i] and f[j] (ghosts included); outputs are what LAMMPS sees after
 * compute() and before reverse_comm.  The two per-pair comm calls
 * (pair_aeam.cpp:257,307) are inert with a full list and are
 * not modelled.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this file.  The product (libmdpair_hip.so) never links or loads it.
 *
 * PARITY UNPINNED: the reference ships no log, test or golden vector for AEAM
 * (USER-AEAM/sample.in has no log) and the reference itself cannot be built in
 * this image (LAMMPS headers absent).  This restatement is checked only by
 * self-consistency (forces = -dE/dx by central differences, sum f = 0, table
 * round trips), see tests/test_oracle_aeam.py and DESIGN.md.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aeam_oracle.h"

#define MAXLINE 1024

/* ---- interpolate, pair_aeam.cpp:915-942: 7-coefficient LAMMPS-EAM spline.
 * f[1..n] -> spline[(m)*7 + c], rows 1..n (row 0 unused) */
static void interpolate(int n, double delta, const double *f, double *spline)
{
  int m;
#define S(m, c) spline[(size_t) (m) *7 + (c)]
  for (m = 1; m <= n; m++) S(m, 6) = f[m];
  S(1, 5) = S(2, 6) - S(1, 6);
  S(2, 5) = 0.5 * (S(3, 6) - S(1, 6));
  S(n - 1, 5) = 0.5 * (S(n, 6) - S(n - 2, 6));
  S(n, 5) = S(n, 6) - S(n - 1, 6);
  for (m = 3; m <= n - 2; m++)
    S(m, 5) = ((S(m - 2, 6) - S(m + 2, 6)) + 8.0 * (S(m + 1, 6) - S(m - 1, 6))) / 12.0;
  for (m = 1; m <= n - 1; m++) {
    S(m, 4) = 3.0 * (S(m + 1, 6) - S(m, 6)) - 2.0 * S(m, 5) - S(m + 1, 5);
    S(m, 3) = S(m, 5) + S(m + 1, 5) - 2.0 * (S(m + 1, 6) - S(m, 6));
  }
  S(n, 4) = 0.0;
  S(n, 3) = 0.0;
  for (m = 1; m <= n; m++) {
    S(m, 2) = S(m, 5) / delta;
    S(m, 1) = 2.0 * S(m, 4) / delta;
    S(m, 0) = 3.0 * S(m, 3) / delta;
  }
#undef S
}

/* read n doubles across lines, skipping blank lines and '#' comments
 * (TextFileReader::next_dvector) */
static int next_dvector(FILE *fp, double *out, int n)
{
  char line[MAXLINE];
  int got = 0;
  while (got < n) {
    char *s, *h, *end;
    if (!fgets(line, MAXLINE, fp)) return -1;
    if ((h = strchr(line, '#'))) *h = 0;
    s = line;
    for (;;) {
      double v = strtod(s, &end);
      if (end == s) break;
      if (got < n) out[got++] = v;
      s = end;
    }
  }
  return 0;
}

void aeam_oracle_free(aeam_oracle_pot *T)
{
  if (!T) return;
  free(T->frho_spline);
  free(T->rhor_spline);
  free(T->z2r_spline);
  memset(T, 0, sizeof *T);
}

/* read_file + file2array + array2spline for ntypes == nelements with
 * type i <-> element i-1 (the only mapping coeff() accepts, pair_aeam.cpp:568-572) */
int aeam_oracle_read(const char *filename, aeam_oracle_pot *T)
{
  FILE *fp = fopen(filename, "r");
  char line[MAXLINE];
  int i, j, n, ne, nrmax = 0, nrhomax = 0;
  double *frho = NULL, *rhor = NULL, *z2r = NULL;
  memset(T, 0, sizeof *T);
  if (!fp) return -1;
  for (i = 0; i < 12; i++) /* nheader1 = 12, pair_aeam.cpp:645-648 */
    if (!fgets(line, MAXLINE, fp)) goto bad;
  {
    char *s = line, *end;
    T->nelements = (int) strtol(s, &end, 10);
    s = end;
    T->nnonangular = (int) strtol(s, &end, 10);
    s = end;
    T->nangular = (int) strtol(s, &end, 10);
    s = end;
    ne = T->nelements;
    if (ne < 1 || ne > AEAM_ORACLE_MAXEL) goto bad;
    for (i = 0; i < ne; i++) {
      if (sscanf(s, " %15s%n", T->elements[i], &n) != 1) goto bad;
      s += n;
    }
  }
  for (i = 0; i < ne; i++) {
    if (!fgets(line, MAXLINE, fp)) goto bad;
    if (sscanf(line, "%d %lf %lf", &T->nrho[i], &T->drho[i], &T->mass[i]) != 3) goto bad;
    if (T->nrho[i] > nrhomax) nrhomax = T->nrho[i];
  }
  for (i = 0; i < ne; i++)
    for (j = 0; j < ne; j++) {
      if (!fgets(line, MAXLINE, fp)) goto bad;
      if (sscanf(line, "%d %lf %lf", &T->nr[i][j], &T->dr[i][j], &T->cut[i][j]) != 3) goto bad;
      if (T->nr[i][j] > nrmax) nrmax = T->nr[i][j];
    }
  T->nrmax = nrmax;
  T->nrhomax = nrhomax;
  T->nfrho = ne + 1; /* + zero table for pair hybrid, :767 */
  T->nrhor = ne * ne;
  T->nz2r = ne * (ne + 1) / 2;

  frho = (double *) calloc((size_t) T->nfrho * (nrhomax + 1), sizeof(double));
  rhor = (double *) calloc((size_t) T->nrhor * (nrmax + 1), sizeof(double));
  z2r = (double *) calloc((size_t) T->nz2r * (nrmax + 1), sizeof(double));
  T->frho_spline = (double *) calloc((size_t) T->nfrho * (nrhomax + 1) * 7, sizeof(double));
  T->rhor_spline = (double *) calloc((size_t) T->nrhor * (nrmax + 1) * 7, sizeof(double));
  T->z2r_spline = (double *) calloc((size_t) T->nz2r * (nrmax + 1) * 7, sizeof(double));
  if (!frho || !rhor || !z2r || !T->frho_spline || !T->rhor_spline || !T->z2r_spline) goto bad;

  for (i = 0; i < ne; i++)
    if (next_dvector(fp, frho + (size_t) i * (nrhomax + 1) + 1, T->nrho[i])) goto bad;
  n = 0;
  for (i = 0; i < ne; i++)
    for (j = 0; j < ne; j++) {
      if (next_dvector(fp, rhor + (size_t) n * (nrmax + 1) + 1, T->nr[i][j])) goto bad;
      T->nrrho[n] = T->nr[i][j];
      T->drrho[n] = T->dr[i][j];
      n++;
    }
  n = 0;
  for (i = 0; i < ne; i++)
    for (j = 0; j <= i; j++) {
      if (next_dvector(fp, z2r + (size_t) n * (nrmax + 1) + 1, T->nr[i][j])) goto bad;
      T->nrz2r[n] = T->nr[i][j];
      T->drz2r[n] = T->dr[i][j];
      n++;
    }
  fclose(fp);
  fp = NULL;

  /* type maps, pair_aeam.cpp:785-871 (1-based types, ntypes == nelements) */
  n = 0;
  for (i = 1; i <= ne; i++) {
    T->type2frho[i] = i - 1;
    for (j = 1; j <= ne; j++) {
      int irow = i - 1, icol = j - 1, m, q = 0;
      T->type2rhor[i][j] = n++;
      if (irow < icol) {
        irow = j - 1;
        icol = i - 1;
      }
      for (m = 0; m < irow; m++) q += m + 1;
      T->type2z2r[i][j] = q + icol;
    }
  }

  /* array2spline, pair_aeam.cpp:889-910 */
  for (i = 0; i < T->nfrho; i++) {
    int nn = (i < T->nfrho - 1) ? T->nrho[i] : T->nrho[0];
    double dd = (i < T->nfrho - 1) ? T->drho[i] : T->drho[0];
    interpolate(nn, dd, frho + (size_t) i * (nrhomax + 1), T->frho_spline + (size_t) i * (nrhomax + 1) * 7);
  }
  for (i = 0; i < T->nrhor; i++)
    interpolate(T->nrrho[i], T->drrho[i], rhor + (size_t) i * (nrmax + 1),
                T->rhor_spline + (size_t) i * (nrmax + 1) * 7);
  for (i = 0; i < T->nz2r; i++)
    interpolate(T->nrz2r[i], T->drz2r[i], z2r + (size_t) i * (nrmax + 1),
                T->z2r_spline + (size_t) i * (nrmax + 1) * 7);
  free(frho);
  free(rhor);
  free(z2r);
  return 0;
bad:
  if (fp) fclose(fp);
  free(frho);
  free(rhor);
  free(z2r);
  aeam_oracle_free(T);
  return -2;
}

/* ---- compute, pair_aeam.cpp:110-479 ----------------------------------------- */
typedef struct {
  int eflag_global, eflag_atom, vflag_tally, vflag_atom;
  double eng, *eatom, vir[6], *vatom;
} tally_t;

static void tally_v(tally_t *W, const double v[6], const int *idx, int n, double share)
{
  int a, k;
  if (W->vflag_tally)
    for (k = 0; k < 6; k++) W->vir[k] += v[k];
  if (W->vflag_atom)
    for (a = 0; a < n; a++)
      for (k = 0; k < 6; k++) W->vatom[6 * (size_t) idx[a] + k] += share * v[k];
}

int aeam_oracle_compute(const aeam_oracle_pot *T, int nlocal, int nghost, const double *x, const int *type,
                        const int *numneigh, const long long *offset, const int *neigh, int eflag, int vflag,
                        double *f, double *eng_vdwl, double *virial_fdotr, double *virial_tally, double *eatom,
                        double *vatom, double *rho_out, double *fp_out)
{
  const double THIRD = 1.0 / 3.0, minrho = 0.0000000000001;
  const int nnon = T->nnonangular;
  const size_t rstride = (size_t) (T->nrmax + 1) * 7, fstride = (size_t) (T->nrhomax + 1) * 7;
  int nall = nlocal + nghost, i, jj, kk, d, k6;
  double *rho = (double *) calloc((size_t) nall + 1, sizeof(double));
  double *fp = (double *) calloc((size_t) nall + 1, sizeof(double));
  tally_t W;
  if (!rho || !fp) return -1;
  memset(&W, 0, sizeof W);
  W.eflag_global = eflag & 1;
  W.eflag_atom = (eflag & 2) && eatom;
  W.vflag_tally = (vflag & 1) && virial_tally;
  W.vflag_atom = (vflag & 4) && vatom;
  W.eatom = eatom;
  W.vatom = vatom;
  if (W.eflag_atom) memset(eatom, 0, sizeof(double) * nall);
  if (W.vflag_atom) memset(vatom, 0, sizeof(double) * 6 * nall);

  /* pass 1: density, :164-253 */
  for (i = 0; i < nlocal; i++) {
    int itype = type[i], jnum = numneigh[i];
    const int *jl = neigh + offset[i];
    const double *xi = x + 3 * (size_t) i;
    for (jj = 0; jj < jnum; jj++) {
      int j = jl[jj] & 0x1FFFFFFF, jtype = type[j], m1;
      double d1[3], rsq1, r1, CutDec, p1, fij;
      const double *c;
      for (d = 0; d < 3; d++) d1[d] = x[3 * (size_t) j + d] - xi[d];
      rsq1 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
      r1 = sqrt(rsq1);
      CutDec = (itype > nnon && jtype > nnon) ? 1.5 : 0;
      if (r1 > T->cut[itype - 1][jtype - 1] - CutDec) continue;
      p1 = r1 * (1 / T->dr[itype - 1][jtype - 1]) + 1.0;
      m1 = (int) p1;
      if (m1 > T->nr[itype - 1][jtype - 1] - 1) m1 = T->nr[itype - 1][jtype - 1] - 1;
      p1 -= m1;
      if (p1 > 1.0) p1 = 1.0;
      c = T->rhor_spline + T->type2rhor[itype][jtype] * rstride + (size_t) m1 * 7;
      fij = ((c[3] * p1 + c[4]) * p1 + c[5]) * p1 + c[6];
      if (itype <= nnon) {
        rho[i] += fij;
      } else {
        for (kk = jj + 1; kk < jnum; kk++) {
          int k = jl[kk] & 0x1FFFFFFF, ktype = type[k], m2;
          double d2[3], d3[3], rsq2, r2, rsq3, p2, fik, cs, delcs;
          for (d = 0; d < 3; d++) d2[d] = x[3 * (size_t) k + d] - xi[d];
          CutDec = (ktype > nnon) ? 1.5 : 0;
          rsq2 = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
          r2 = sqrt(rsq2);
          if (r2 > T->cut[itype - 1][ktype - 1] - CutDec) continue;
          for (d = 0; d < 3; d++) d3[d] = x[3 * (size_t) k + d] - x[3 * (size_t) j + d];
          rsq3 = d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2];
          p2 = r2 * (1 / T->dr[itype - 1][ktype - 1]) + 1.0;
          m2 = (int) p2;
          if (m2 > T->nr[itype - 1][ktype - 1] - 1) m2 = T->nr[itype - 1][ktype - 1] - 1;
          p2 -= m2;
          if (p2 > 1.0) p2 = 1.0;
          c = T->rhor_spline + T->type2rhor[itype][ktype] * rstride + (size_t) m2 * 7;
          fik = ((c[3] * p2 + c[4]) * p2 + c[5]) * p2 + c[6];
          cs = (rsq1 + rsq2 - rsq3) / (2 * r1 * r2);
          delcs = cs + THIRD;
          rho[i] += 2 * fij * fik * (delcs * delcs);
        }
      }
    }
  }

  /* pass 2: embedding, :264-303 */
  for (i = 0; i < nlocal; i++) {
    int itype = type[i], m;
    double ni = (itype <= nnon) ? 1 : 0.5;
    double p = pow(rho[i], ni) * (1 / T->drho[itype - 1]) + 1.0;
    const double *c;
    m = (int) p;
    if (m > T->nrho[itype - 1] - 1) m = T->nrho[itype - 1] - 1;
    if (m < 1) m = 1;
    p -= m;
    if (p > 1.0) p = 1.0;
    c = T->frho_spline + T->type2frho[itype] * fstride + (size_t) m * 7;
    fp[i] = (c[0] * p + c[1]) * p + c[2];
    if (eflag) {
      double poteam = ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
      if (W.eflag_global) W.eng += poteam;
      if (W.eflag_atom) eatom[i] += (itype <= nnon) ? poteam : THIRD * poteam;
    }
  }

  /* pass 3: forces, :309-476 */
  for (i = 0; i < nlocal; i++) {
    int itype = type[i], jnum = numneigh[i];
    const int *jl = neigh + offset[i];
    const double *xi = x + 3 * (size_t) i;
    double ni = (itype <= nnon) ? 1 : 0.5, deli = (itype <= nnon) ? 0 : 1, ci = (itype <= nnon) ? 0 : 2;
    double Fptmp = (rho[i] > minrho) ? ni * pow(rho[i], (ni - 1)) : 0;
    for (jj = 0; jj < jnum; jj++) {
      int j = jl[jj] & 0x1FFFFFFF, jtype = type[j], m1;
      double d1[3], rsq1, r1, p1, fij, dfij, phip, phi, recip, Feam, F2b, fpair;
      const double *c;
      for (d = 0; d < 3; d++) d1[d] = x[3 * (size_t) j + d] - xi[d];
      rsq1 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
      r1 = sqrt(rsq1);
      if (r1 > T->cut[itype - 1][jtype - 1]) continue;
      p1 = r1 * (1 / T->dr[itype - 1][jtype - 1]) + 1.0;
      m1 = (int) p1;
      if (m1 > T->nr[itype - 1][jtype - 1] - 1) m1 = T->nr[itype - 1][jtype - 1] - 1;
      p1 -= m1;
      if (p1 > 1.0) p1 = 1.0;
      c = T->rhor_spline + T->type2rhor[itype][jtype] * rstride + (size_t) m1 * 7;
      fij = ((c[3] * p1 + c[4]) * p1 + c[5]) * p1 + c[6];
      dfij = (c[0] * p1 + c[1]) * p1 + c[2];
      c = T->z2r_spline + T->type2z2r[itype][jtype] * rstride + (size_t) m1 * 7;
      phip = (c[0] * p1 + c[1]) * p1 + c[2];
      phi = ((c[3] * p1 + c[4]) * p1 + c[5]) * p1 + c[6];
      recip = 1 / r1;
      Feam = -(1 - deli) * Fptmp * fp[i] * (dfij * recip);
      F2b = -phip * recip;
      fpair = Feam + 0.5 * F2b;
      for (d = 0; d < 3; d++) {
        f[3 * (size_t) i + d] -= d1[d] * fpair;
        f[3 * (size_t) j + d] += d1[d] * fpair;
      }
      if (eflag) {
        if (W.eflag_global) W.eng += 0.5 * phi;
        if (W.eflag_atom) eatom[i] += 0.5 * phi;
      }
      if (W.vflag_tally || W.vflag_atom) { /* ev_tally(i,j,...,0,0,fpair,del) */
        double v[6];
        int idx[2];
        v[0] = d1[0] * d1[0] * fpair;
        v[1] = d1[1] * d1[1] * fpair;
        v[2] = d1[2] * d1[2] * fpair;
        v[3] = d1[0] * d1[1] * fpair;
        v[4] = d1[0] * d1[2] * fpair;
        v[5] = d1[1] * d1[2] * fpair;
        idx[0] = i;
        idx[1] = j;
        tally_v(&W, v, idx, 2, 0.5);
      }
      if (itype <= nnon) continue;
      for (kk = jj + 1; kk < jnum; kk++) {
        int k = jl[kk] & 0x1FFFFFFF, ktype = type[k], m2;
        double d2[3], d3[3], rsq2, r2, rsq3, r3, p2, fik, dfik, cs, dcosij, dcosik, dcosjk, delcs, ftet, delcs2;
        double DFij, DFik, DFjk, FFij, FFik, FFjk, fj[3], fk[3], CutDec;
        for (d = 0; d < 3; d++) d2[d] = x[3 * (size_t) k + d] - xi[d];
        CutDec = (ktype > nnon) ? 1.5 : 0;
        rsq2 = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
        r2 = sqrt(rsq2);
        if (r2 > T->cut[itype - 1][ktype - 1] - CutDec) continue;
        for (d = 0; d < 3; d++) d3[d] = x[3 * (size_t) k + d] - x[3 * (size_t) j + d];
        rsq3 = d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2];
        r3 = sqrt(rsq3);
        p2 = r2 * (1 / T->dr[itype - 1][ktype - 1]) + 1.0;
        m2 = (int) p2;
        if (m2 > T->nr[itype - 1][ktype - 1] - 1) m2 = T->nr[itype - 1][ktype - 1] - 1;
        p2 -= m2;
        if (p2 > 1.0) p2 = 1.0;
        c = T->rhor_spline + T->type2rhor[itype][ktype] * rstride + (size_t) m2 * 7;
        fik = ((c[3] * p2 + c[4]) * p2 + c[5]) * p2 + c[6];
        dfik = (c[0] * p2 + c[1]) * p2 + c[2];
        cs = (rsq1 + rsq2 - rsq3) / (2 * r1 * r2);
        dcosij = 1 / r2 - cs / r1;
        dcosik = 1 / r1 - cs / r2;
        dcosjk = -r3 / (r1 * r2);
        delcs = cs + THIRD;
        ftet = delcs * delcs;
        delcs2 = 2 * delcs;
        DFij = ci * (fik * dfij * ftet + fij * fik * delcs2 * dcosij);
        DFik = ci * (fij * dfik * ftet + fij * fik * delcs2 * dcosik);
        DFjk = ci * fij * fik * delcs2 * dcosjk;
        FFij = -Fptmp * fp[i] * DFij / r1;
        FFik = -Fptmp * fp[i] * DFik / r2;
        FFjk = -Fptmp * fp[i] * DFjk / r3;
        for (d = 0; d < 3; d++) {
          fj[d] = d1[d] * FFij - d3[d] * FFjk;
          fk[d] = d2[d] * FFik + d3[d] * FFjk;
          f[3 * (size_t) i + d] -= fj[d] + fk[d];
          f[3 * (size_t) j + d] += fj[d];
          f[3 * (size_t) k + d] += fk[d];
        }
        if (W.vflag_tally || W.vflag_atom) { /* ev_tally3(i,j,k,0,0,fj,fk,drji,drki) */
          double v[6];
          int idx[3];
          v[0] = d1[0] * fj[0] + d2[0] * fk[0];
          v[1] = d1[1] * fj[1] + d2[1] * fk[1];
          v[2] = d1[2] * fj[2] + d2[2] * fk[2];
          v[3] = d1[0] * fj[1] + d2[0] * fk[1];
          v[4] = d1[0] * fj[2] + d2[0] * fk[2];
          v[5] = d1[1] * fj[2] + d2[1] * fk[2];
          idx[0] = i;
          idx[1] = j;
          idx[2] = k;
          tally_v(&W, v, idx, 3, THIRD);
        }
      }
    }
  }

  if (eng_vdwl) *eng_vdwl = W.eng;
  if (virial_tally)
    for (k6 = 0; k6 < 6; k6++) virial_tally[k6] = W.vir[k6];
  if (virial_fdotr) {
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
    for (k6 = 0; k6 < 6; k6++) virial_fdotr[k6] = v[k6];
  }
  if (rho_out) memcpy(rho_out, rho, sizeof(double) * nall);
  if (fp_out) memcpy(fp_out, fp, sizeof(double) * nall);
  free(rho);
  free(fp);
  return 0;
}
