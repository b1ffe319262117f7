/* oracle/rebomos_oracle.h -- TEST INFRASTRUCTURE (see rebomos_oracle.c header). */
#ifndef REBOMOS_ORACLE_H
#define REBOMOS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* element index: 0 = Mo, 1 = S.  Layout mirrors pair_rebomos.h:54-60. */
typedef struct {
  double rcmin[2][2], rcmax[2][2], rcmaxsq[2][2];
  double Q[2][2], alpha[2][2], A[2][2], BIJc[2][2], Beta[2][2];
  double b[7][2], bg[7][2], a[4][2];
  double rcLJmin[2][2], rcLJmax[2][2], epsilon[2][2], sigma[2][2];
  double lj1[2][2], lj2[2][2], lj3[2][2], lj4[2][2];
  double cut3rebo;
} rebomos_oracle_params;

int rebomos_oracle_read_params(const char *filename, rebomos_oracle_params *P);
void rebomos_oracle_params_from_scalars(const double *v61, rebomos_oracle_params *P);

/* x[nall][3]; elem[nall] (0/1); tag[nall]; CSR neighbor list over all nall atoms
 * (owned: full list to the master cutoff; ghosts: list to rcmax+skin).
 * eflag: 1 global, 2 per-atom.  vflag: 1 explicit tally, 4 per-atom.
 * f[nall][3] is ACCUMULATED into (caller zeroes), incl. ghosts.
 * phases: bit0 FREBO, bit1 FLJ.  Any output pointer may be NULL. */
int rebomos_oracle_compute(const rebomos_oracle_params *P, int nlocal, int nghost, const double *x,
                           const int *elem, const int *tag, const int *numneigh, const long long *offset,
                           const int *neigh, int eflag, int vflag, double *f, double *eng_vdwl,
                           double *virial_fdotr, double *virial_tally, double *eatom, double *vatom,
                           double *nM_out, double *nS_out, int *rebo_numneigh_out, int phases);

#ifdef __cplusplus
}
#endif
#endif
