/* oracle/aeam_oracle.h -- TEST INFRASTRUCTURE (see aeam_oracle.c header). */
#ifndef AEAM_ORACLE_H
#define AEAM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define AEAM_ORACLE_MAXEL 16

/* mirrors PairAEAM::Setfl (pair_aeam.h:66-76) + the spline arrays (pair_aeam.h:55-62).
 * Spline tables are dense [table][row 0..nmax][7]; row 0 is unused (1-based rows). */
typedef struct {
  int nelements, nnonangular, nangular, nrhomax, nrmax;
  char elements[AEAM_ORACLE_MAXEL][16];
  double mass[AEAM_ORACLE_MAXEL], drho[AEAM_ORACLE_MAXEL];
  int nrho[AEAM_ORACLE_MAXEL];
  int nr[AEAM_ORACLE_MAXEL][AEAM_ORACLE_MAXEL];
  double dr[AEAM_ORACLE_MAXEL][AEAM_ORACLE_MAXEL], cut[AEAM_ORACLE_MAXEL][AEAM_ORACLE_MAXEL];
  int nfrho, nrhor, nz2r;
  int nrrho[AEAM_ORACLE_MAXEL * AEAM_ORACLE_MAXEL], nrz2r[AEAM_ORACLE_MAXEL * AEAM_ORACLE_MAXEL];
  double drrho[AEAM_ORACLE_MAXEL * AEAM_ORACLE_MAXEL], drz2r[AEAM_ORACLE_MAXEL * AEAM_ORACLE_MAXEL];
  int type2frho[AEAM_ORACLE_MAXEL + 1];
  int type2rhor[AEAM_ORACLE_MAXEL + 1][AEAM_ORACLE_MAXEL + 1];
  int type2z2r[AEAM_ORACLE_MAXEL + 1][AEAM_ORACLE_MAXEL + 1];
  double *frho_spline, *rhor_spline, *z2r_spline;
} aeam_oracle_pot;

int aeam_oracle_read(const char *filename, aeam_oracle_pot *T);
void aeam_oracle_free(aeam_oracle_pot *T);

/* x[nall][3]; type[nall] 1-based; CSR full neighbor list for the nlocal owned
 * atoms (numneigh/offset need nlocal entries).  eflag: 1 global, 2 per-atom.
 * vflag: 1 explicit tally, 4 per-atom.  f[nall][3] is ACCUMULATED into. */
int aeam_oracle_compute(const aeam_oracle_pot *T, int nlocal, int nghost, const double *x, const int *type,
                        const int *numneigh, const long long *offset, const int *neigh, int eflag, int vflag,
                        double *f, double *eng_vdwl, double *virial_fdotr, double *virial_tally, double *eatom,
                        double *vatom, double *rho_out, double *fp_out);

#ifdef __cplusplus
}
#endif
#endif
