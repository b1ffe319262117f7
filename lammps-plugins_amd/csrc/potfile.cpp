// potfile.cpp -- potential-file front ends (host C++, cold path), shared by the LAMMPS plugin
// adapters and the Python harness through the C-ABI.
//
//   mdp_rebomos_read_file  replaces PairREBOMoS::read_file (USER-REBOMOS/pair_rebomos.cpp:857-1066)
//                          + the lj1..lj4 prefactors of init_one (pair_rebomos.cpp:262-265)
//   mdp_aeam_file_*        replaces PairAEAM::read_file / file2array / array2spline / interpolate
//                          (USER-AEAM/pair_aeam.cpp:627-746, 752-872, 876-942)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mdpair_hip.h"

namespace {

void set_err(char *err, int errlen, const std::string &s)
{
  if (err && errlen > 0) {
    strncpy(err, s.c_str(), errlen - 1);
    err[errlen - 1] = 0;
  }
}

// MathSpecial::powint
double powint(double x, int n)
{
  double yy = 1.0, ww = x;
  if (x == 0.0) return 0.0;
  for (int nn = n > 0 ? n : -n; nn != 0; nn >>= 1, ww *= ww)
    if (nn & 1) yy *= ww;
  return n > 0 ? yy : 1.0 / yy;
}

// first whitespace-separated token of the next non-blank, non-comment line as a double
// (PotentialFileReader::next_double).  returns 0 ok, 1 eof, 2 not a number
int next_double(FILE *fp, double &out, std::string &bad)
{
  char line[1024];
  while (fgets(line, sizeof line, fp)) {
    if (char *h = strchr(line, '#')) *h = 0;
    char *s = line;
    while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n') ++s;
    if (!*s) continue;
    char *end = nullptr;
    out = strtod(s, &end);
    if (end == s || !(*end == 0 || *end == ' ' || *end == '\t' || *end == '\r' || *end == '\n')) {
      bad.assign(s, strcspn(s, " \t\r\n"));
      return 2;
    }
    return 0;
  }
  return 1;
}

} // namespace

struct mdp_aeam_file {
  int nelements = 0, nnonangular = 0, nangular = 0, nrhomax = 0, nrmax = 0;
  std::vector<std::string> elements;
  std::vector<double> mass, drho, dr, cut;
  std::vector<int> nrho, nr;
  std::vector<double> frho_raw, rhor_raw, z2r_raw; // [el][nrhomax+1], [el*el][nrmax+1], [tri][nrmax+1]
  // built tables
  int ntypes = 0, nfrho = 0, nrhor = 0, nz2r = 0;
  std::vector<int> type2frho, type2rhor, type2z2r;
  std::vector<double> frho_spline, rhor_spline, z2r_spline;
};

extern "C" {

int mdp_rebomos_params_from_scalars(const double *v, mdp_rebomos_params *p)
{
  if (!v || !p) return MDP_EINVAL;
  memset(p, 0, sizeof *p);
  auto sym3 = [&](double (*F)[2], int i) {
    F[0][0] = v[i];
    F[0][1] = F[1][0] = v[i + 1];
    F[1][1] = v[i + 2];
  };
  // file order, pair_rebomos.cpp:884-948
  sym3(p->rcmin, 0);
  sym3(p->rcmax, 3);
  sym3(p->Q, 6);
  sym3(p->alpha, 9);
  sym3(p->A, 12);
  sym3(p->BIJc, 15);
  sym3(p->Beta, 18);
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) p->rcmaxsq[a][b] = p->rcmax[a][b] * p->rcmax[a][b];
  for (int k = 0; k < 7; k++) {
    p->b[k][0] = v[21 + k];
    p->bg[k][0] = v[28 + k];
    p->b[k][1] = v[35 + k];
    p->bg[k][1] = v[42 + k];
  }
  for (int k = 0; k < 4; k++) {
    p->a[k][0] = v[49 + k];
    p->a[k][1] = v[53 + k];
  }
  const double eps_MM = v[57], eps_SS = v[58], sig_MM = v[59], sig_SS = v[60];
  // mixing, pair_rebomos.cpp:1048-1066
  p->sigma[0][0] = sig_MM;
  p->sigma[0][1] = p->sigma[1][0] = (sig_MM + sig_SS) / 2;
  p->sigma[1][1] = sig_SS;
  p->epsilon[0][0] = eps_MM;
  p->epsilon[0][1] = p->epsilon[1][0] = sqrt(eps_MM * eps_SS);
  p->epsilon[1][1] = eps_SS;
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) {
      p->rcLJmin[a][b] = p->rcmin[a][b];
      p->rcLJmax[a][b] = 2.5 * p->sigma[a][b];
      p->lj1[a][b] = 48.0 * p->epsilon[a][b] * powint(p->sigma[a][b], 12);
      p->lj2[a][b] = 24.0 * p->epsilon[a][b] * powint(p->sigma[a][b], 6);
      p->lj3[a][b] = 4.0 * p->epsilon[a][b] * powint(p->sigma[a][b], 12);
      p->lj4[a][b] = 4.0 * p->epsilon[a][b] * powint(p->sigma[a][b], 6);
    }
  return MDP_OK;
}

int mdp_rebomos_read_file(const char *path, mdp_rebomos_params *p, char *err, int errlen)
{
  if (!path || !p) return MDP_EINVAL;
  FILE *fp = fopen(path, "r");
  if (!fp) {
    set_err(err, errlen, std::string("cannot open rebomos potential file ") + path);
    return MDP_EINVAL;
  }
  double v[61];
  for (int n = 0; n < 61; n++) {
    std::string bad;
    const int rc = next_double(fp, v[n], bad);
    if (rc) {
      fclose(fp);
      // message shape of pair_rebomos.cpp:955-957
      set_err(err, errlen,
              std::string("reading rebomos potential file ") + path + "\nREASON: " +
                  (rc == 1 ? "unexpected end of file (61 parameters expected)" : "Not a valid floating-point number: '" + bad + "'") +
                  "\n");
      return MDP_EINVAL;
    }
  }
  fclose(fp);
  return mdp_rebomos_params_from_scalars(v, p);
}

// ---- AEAM ------------------------------------------------------------------------------------------
void mdp_aeam_file_free(mdp_aeam_file *f) { delete f; }

int mdp_aeam_file_read(const char *path, mdp_aeam_file **out, char *err, int errlen)
{
  if (!path || !out) return MDP_EINVAL;
  *out = nullptr;
  FILE *fp = fopen(path, "r");
  if (!fp) {
    set_err(err, errlen, std::string("Cannot open AEAM potential file ") + path); // pair_aeam.cpp:638
    return MDP_EINVAL;
  }
  mdp_aeam_file *F = new mdp_aeam_file();
  auto fail = [&](const std::string &why) {
    set_err(err, errlen, "AEAM potential file parser error: " + why); // pair_aeam.cpp:665,684,709
    fclose(fp);
    delete F;
    return MDP_EINVAL;
  };
  char line[1024];
  for (int i = 0; i < 12; i++) // nheader1 = 12, pair_aeam.cpp:645-648
    if (!fgets(line, sizeof line, fp)) return fail("file too short (header)");
  {
    char *s = line, *end;
    long v[3];
    for (int k = 0; k < 3; k++) {
      v[k] = strtol(s, &end, 10);
      if (end == s) return fail("Not a valid integer number in element-count line");
      s = end;
    }
    F->nelements = (int) v[0];
    F->nnonangular = (int) v[1];
    F->nangular = (int) v[2];
    if (F->nelements < 1 || F->nelements > 64) return fail("unsupported number of elements (1..64)");
    for (int i = 0; i < F->nelements; i++) {
      while (*s == ' ' || *s == '\t') ++s;
      const size_t len = strcspn(s, " \t\r\n");
      if (!len) return fail("Not enough tokens (element names)");
      F->elements.emplace_back(s, len);
      s += len;
    }
  }
  const int ne = F->nelements;
  F->mass.resize(ne);
  F->drho.resize(ne);
  F->nrho.resize(ne);
  F->nr.resize(ne * ne);
  F->dr.resize(ne * ne);
  F->cut.resize(ne * ne);
  for (int i = 0; i < ne; i++) {
    if (!fgets(line, sizeof line, fp) || sscanf(line, "%d %lf %lf", &F->nrho[i], &F->drho[i], &F->mass[i]) != 3)
      return fail("nrho drho mass line");
    if (F->nrho[i] < 5) return fail("nrho too small");
    if (F->nrho[i] > F->nrhomax) F->nrhomax = F->nrho[i];
  }
  for (int k = 0; k < ne * ne; k++) {
    if (!fgets(line, sizeof line, fp) || sscanf(line, "%d %lf %lf", &F->nr[k], &F->dr[k], &F->cut[k]) != 3)
      return fail("nr dr cut line");
    if (F->nr[k] < 5) return fail("nr too small");
    if (F->nr[k] > F->nrmax) F->nrmax = F->nr[k];
  }
  // TextFileReader::next_dvector: n doubles across lines, '#' comments and blank lines skipped
  auto next_dvector = [&](double *dst, int n) -> bool {
    int got = 0;
    while (got < n) {
      if (!fgets(line, sizeof line, fp)) return false;
      if (char *h = strchr(line, '#')) *h = 0;
      char *s = line, *end;
      for (;;) {
        const double v = strtod(s, &end);
        if (end == s) break;
        if (got < n) dst[got++] = v;
        s = end;
      }
    }
    return true;
  };
  const size_t fs = (size_t) F->nrhomax + 1, rs = (size_t) F->nrmax + 1;
  F->frho_raw.assign(fs * ne, 0.0);
  F->rhor_raw.assign(rs * ne * ne, 0.0);
  F->z2r_raw.assign(rs * (ne * (ne + 1) / 2), 0.0);
  for (int i = 0; i < ne; i++)
    if (!next_dvector(&F->frho_raw[fs * i + 1], F->nrho[i])) return fail("unexpected end of file in F(rho)");
  for (int k = 0; k < ne * ne; k++)
    if (!next_dvector(&F->rhor_raw[rs * k + 1], F->nr[k])) return fail("unexpected end of file in rho(r)");
  int n = 0;
  for (int i = 0; i < ne; i++)
    for (int j = 0; j <= i; j++, n++)
      if (!next_dvector(&F->z2r_raw[rs * n + 1], F->nr[i * ne + j])) return fail("unexpected end of file in phi(r)");
  fclose(fp);
  *out = F;
  return MDP_OK;
}

int mdp_aeam_file_info(const mdp_aeam_file *F, int *nelements, int *nnonangular, int *nangular, double *mass,
                       int mass_cap, char *names, int nameslen)
{
  if (!F) return MDP_EINVAL;
  if (nelements) *nelements = F->nelements;
  if (nnonangular) *nnonangular = F->nnonangular;
  if (nangular) *nangular = F->nangular;
  if (mass)
    for (int i = 0; i < F->nelements && i < mass_cap; i++) mass[i] = F->mass[i];
  if (names && nameslen > 0) {
    std::string s;
    for (int i = 0; i < F->nelements; i++) s += (i ? " " : "") + F->elements[i];
    strncpy(names, s.c_str(), nameslen - 1);
    names[nameslen - 1] = 0;
  }
  return MDP_OK;
}

// One tabulated function y[1..n] on a uniform grid of spacing h -> n spline rows of 7 doubles,
//   row[6] = value, row[5] = knot slope (per grid step), row[4], row[3] = quadratic / cubic Hermite terms of the
//   interval [m, m+1], row[2..0] = the derivative polynomial per unit of the abscissa.
// Same numbers, coefficient by coefficient, as PairAEAM::interpolate (pair_aeam.cpp:915-942) -- the operand order
// inside each expression is part of the parity contract -- produced here in one sweep over the knots.
static void spline_rows(int n, double h, const double *y, double *rows)
{
  // five-point slope inside, centred difference next to the ends, one-sided difference at the ends
  const auto knot_slope = [n, y](int m) -> double {
    if (m == 1) return y[2] - y[1];
    if (m == n) return y[n] - y[n - 1];
    if (m == 2 || m == n - 1) return 0.5 * (y[m + 1] - y[m - 1]);
    return ((y[m - 2] - y[m + 2]) + 8.0 * (y[m + 1] - y[m - 1])) / 12.0;
  };
  double s_here = knot_slope(1);
  for (int m = 1; m <= n; m++) {
    double *row = rows + (size_t) 7 * m;
    double quad = 0.0, cubic = 0.0, s_next = 0.0;
    if (m < n) { // the last row has no interval to its right
      s_next = knot_slope(m + 1);
      const double rise = y[m + 1] - y[m];
      quad = 3.0 * rise - 2.0 * s_here - s_next;
      cubic = s_here + s_next - 2.0 * rise;
    }
    row[6] = y[m];
    row[5] = s_here;
    row[4] = quad;
    row[3] = cubic;
    row[2] = s_here / h;
    row[1] = 2.0 * quad / h;
    row[0] = 3.0 * cubic / h;
    s_here = s_next;
  }
}

// file2array + array2spline for `ntypes` atom types, map[1..ntypes] = element index or -1 (NULL)
int mdp_aeam_file_build(mdp_aeam_file *F, int ntypes, const int *map, mdp_aeam_tables *out)
{
  if (!F || !map || !out || ntypes < 1 || ntypes > 64) return MDP_EINVAL;
  const int ne = F->nelements;
  const size_t fs = (size_t) F->nrhomax + 1, rs = (size_t) F->nrmax + 1;
  F->ntypes = ntypes;
  F->nfrho = ne + 1; // + zero table (pair hybrid), pair_aeam.cpp:767
  F->nrhor = ne * ne;
  F->nz2r = ne * (ne + 1) / 2;
  F->type2frho.assign(ntypes + 1, 0);
  F->type2rhor.assign((size_t) (ntypes + 1) * (ntypes + 1), 0);
  F->type2z2r.assign((size_t) (ntypes + 1) * (ntypes + 1), 0);
  const size_t ts = (size_t) ntypes + 1; // row stride of the 1-based type-pair maps
  for (int ti = 1; ti <= ntypes; ti++) {
    const int ei = map[ti];
    F->type2frho[ti] = ei >= 0 ? ei : F->nfrho - 1; // NULL types use the all-zero table (pair_aeam.cpp:785-790)
    for (int tj = 1; tj <= ntypes; tj++) {
      const int ej = map[tj];
      // rho(r) tables are numbered by TYPE pair with stride ntypes, exactly as the reference numbers them
      // (pair_aeam.cpp:816-821) -- not by element pair
      F->type2rhor[ti * ts + tj] = (ti - 1) * ntypes + (tj - 1);
      // phi(r) tables are stored for element pairs hi >= lo in packed lower-triangular order (:858-869);
      // a pair with a NULL partner points at table 0 and is never evaluated
      int tri = 0;
      if (ei >= 0 && ej >= 0) {
        const int hi = ei > ej ? ei : ej, lo = ei > ej ? ej : ei;
        tri = hi * (hi + 1) / 2 + lo;
      }
      F->type2z2r[ti * ts + tj] = tri;
    }
  }
  F->frho_spline.assign(fs * 7 * F->nfrho, 0.0);
  F->rhor_spline.assign(rs * 7 * F->nrhor, 0.0);
  F->z2r_spline.assign(rs * 7 * F->nz2r, 0.0);
  std::vector<double> zero(fs, 0.0);
  for (int i = 0; i < F->nfrho; i++) { // :889-898
    const bool last = i == F->nfrho - 1;
    spline_rows(last ? F->nrho[0] : F->nrho[i], last ? F->drho[0] : F->drho[i],
                last ? zero.data() : &F->frho_raw[fs * i], &F->frho_spline[fs * 7 * i]);
  }
  for (int k = 0; k < F->nrhor; k++) spline_rows(F->nr[k], F->dr[k], &F->rhor_raw[rs * k], &F->rhor_spline[rs * 7 * k]);
  int n = 0;
  for (int i = 0; i < ne; i++)
    for (int j = 0; j <= i; j++, n++)
      spline_rows(F->nr[i * ne + j], F->dr[i * ne + j], &F->z2r_raw[rs * n], &F->z2r_spline[rs * 7 * n]);
  out->ntypes = ntypes;
  out->nelements = ne;
  out->nnonangular = F->nnonangular;
  out->nrhomax = F->nrhomax;
  out->nrmax = F->nrmax;
  out->nfrho = F->nfrho;
  out->nrhor = F->nrhor;
  out->nz2r = F->nz2r;
  out->nrho = F->nrho.data();
  out->drho = F->drho.data();
  out->nr = F->nr.data();
  out->dr = F->dr.data();
  out->cut = F->cut.data();
  out->type2frho = F->type2frho.data();
  out->type2rhor = F->type2rhor.data();
  out->type2z2r = F->type2z2r.data();
  out->frho_spline = F->frho_spline.data();
  out->rhor_spline = F->rhor_spline.data();
  out->z2r_spline = F->z2r_spline.data();
  return MDP_OK;
}

} // extern "C"
