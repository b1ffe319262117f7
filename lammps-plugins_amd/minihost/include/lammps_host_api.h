// lammps_host_api.h -- the subset of the LAMMPS (2 Aug 2023) host API that the two pair-style
// plugins touch, as implemented by THIS repository's mini-host (minilmp).
//
// The plugin adapters in ../../plugin/ include LAMMPS' own header names ("pair.h", "atom.h", ...).
// Built with -DLAMMPS_SOURCE_DIR=... they see the real LAMMPS headers and the resulting .so loads
// into a real LAMMPS binary; built without, the forwarding headers next to this file resolve those
// names to the classes below and the .so loads into minilmp.  The two builds are NOT interchangeable
// (a plugin is coupled to its host's C++ class layout -- see INTEGRATION.md).
//
// This header is part of the product's own host; it is never used to compile upstream reference
// sources.
#ifndef MINILMP_LAMMPS_HOST_API_H
#define MINILMP_LAMMPS_HOST_API_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define FLERR __FILE__, __LINE__
#ifndef MIN
#define MIN(A, B) ((A) < (B) ? (A) : (B))
#endif
#ifndef MAX
#define MAX(A, B) ((A) > (B) ? (A) : (B))
#endif
#define NEIGHMASK 0x1FFFFFFF

typedef int MPI_Comm; // the communicator is a token: one process, `minilmp -np N` runs its ranks as threads
typedef int MPI_Datatype;
#define MPI_BYTE 1
// the one MPI call a style of these plugins makes (fix nve/mdp on several ranks hands RCCL's unique id from rank 0 to
// the others): defined by the host executable for its rank threads, resolved from it when the plugin is loaded
extern "C" int MPI_Bcast(void *buffer, int count, MPI_Datatype datatype, int root, MPI_Comm comm);

namespace LAMMPS_NS {

typedef int tagint;
typedef int64_t bigint;

class LAMMPS;
class Memory;
class Error;
class Atom;
class Comm;
class Domain;
class Force;
class Neighbor;
class NeighList;
class Pair;
class Update;
class Output;
class Fix;

namespace NeighConst {
  enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0, REQ_GHOST = 1 << 1 };
}

namespace utils {
  enum { NOCONVERT = 0, METAL2REAL = 1, REAL2METAL = 1 << 1 };
  enum { UNKNOWN = 0, ENERGY };
  int get_supported_conversions(const int property);
  std::string get_potential_file_path(const std::string &name);
}

class Error {
 public:
  // never return
  [[noreturn]] void all(const std::string &file, int line, const std::string &msg);
  [[noreturn]] void one(const std::string &file, int line, const std::string &msg);
  void warning(const std::string &file, int line, const std::string &msg);
};

class Memory {
 public:
  template <typename T> T **create(T **&array, int n1, int n2, const char *)
  {
    T *data = (T *) calloc((size_t) n1 * n2, sizeof(T));
    array = (T **) malloc(sizeof(T *) * n1);
    for (int i = 0; i < n1; i++) array[i] = data + (size_t) i * n2;
    return array;
  }
  template <typename T> T *create(T *&array, int n, const char *)
  {
    array = (T *) calloc((size_t) n, sizeof(T));
    return array;
  }
  template <typename T> void destroy(T **&array)
  {
    if (array) {
      free(array[0]);
      free(array);
    }
    array = nullptr;
  }
  template <typename T> void destroy(T *&array)
  {
    free(array);
    array = nullptr;
  }
};

class AtomVec { // Atom::avec: per-atom array storage of the atom style
 public:
  virtual ~AtomVec() = default;
  virtual void grow(int n) = 0; // room for n atoms (owned + ghost) in x, v, f, type, tag; contents kept (AtomVec::grow)
};

class Atom {
 public:
  AtomVec *avec = nullptr;
  double **x = nullptr, **f = nullptr, **v = nullptr;
  int *type = nullptr;
  tagint *tag = nullptr;
  double *mass = nullptr;
  int nlocal = 0, nghost = 0, nmax = 0, ntypes = 0, tag_enable = 1;
  bigint natoms = 0;
  void set_mass(const char *file, int line, int itype, double value);
};

class Domain { // the members of LAMMPS' Domain the adapters read (box of this step)
 public:
  int triclinic = 0;
  int xperiodic = 1, yperiodic = 1, zperiodic = 1;
  double xprd = 1.0, yprd = 1.0, zprd = 1.0, xy = 0.0, xz = 0.0, yz = 0.0;
  double boxlo[3] = {0, 0, 0}, boxhi[3] = {1, 1, 1};
  double h[6] = {1, 1, 1, 0, 0, 0}; // xprd, yprd, zprd, yz, xz, xy
  double sublo_lamda[3] = {0, 0, 0}, subhi_lamda[3] = {1, 1, 1}; // this rank's brick (`minilmp -np N`: one of N)
};

class Comm {
 public:
  int me = 0, nprocs = 1, nthreads = 1;
  int procgrid[3] = {1, 1, 1}, myloc[3] = {0, 0, 0};
  virtual ~Comm() = default;
  virtual void forward_comm(Pair *pair) = 0;
  virtual void reverse_comm(Pair *pair) = 0;
};

class Force {
 public:
  int newton_pair = 1;
  double ftm2v = 1.0 / 1.0364269e-4; // metal units
  double mvv2e = 1.0364269e-4;
  Pair *pair = nullptr;
};

class Update { // the members of LAMMPS' Update a fix style reads
 public:
  bigint ntimestep = 0, laststep = 0;
  double dt = 0.001;
};

class Output { // next step on which the host prints thermo / writes dumps (whichever comes first)
 public:
  bigint next = 0, next_thermo = 0;
};

class NeighList {
 public:
  int inum = 0, gnum = 0;
  int *ilist = nullptr, *numneigh = nullptr;
  int **firstneigh = nullptr;
};

class Neighbor {
 public:
  double skin = 2.0;
  int pgsize = 100000, oneatom = 2000;
  int ago = 0; // steps since the list was built (0: built this step)
  int every = 1, delay = 0, dist_check = 1; // neigh_modify every / delay / check
  int request_flags = 0;
  Pair *requestor = nullptr;
  void add_request(Pair *pair, int flags = 0)
  {
    requestor = pair;
    request_flags = flags;
  }
};

class LAMMPS {
 public:
  Memory *memory = nullptr;
  Error *error = nullptr;
  Atom *atom = nullptr;
  Comm *comm = nullptr;
  Domain *domain = nullptr;
  Force *force = nullptr;
  Neighbor *neighbor = nullptr;
  Update *update = nullptr;
  Output *output = nullptr;
  MPI_Comm world = 0;
};

class Pointers {
 public:
  explicit Pointers(LAMMPS *ptr) :
      lmp(ptr), memory(ptr->memory), error(ptr->error), atom(ptr->atom), comm(ptr->comm), domain(ptr->domain), force(ptr->force),
      neighbor(ptr->neighbor), update(ptr->update), output(ptr->output), world(ptr->world)
  {
  }
  virtual ~Pointers() = default;

 protected:
  LAMMPS *lmp;
  Memory *&memory;
  Error *&error;
  Atom *&atom;
  Comm *&comm;
  Domain *&domain;
  Force *&force;
  Neighbor *&neighbor;
  Update *&update;
  Output *&output;
  MPI_Comm &world;
};

class Pair : protected Pointers {
 public:
  enum { CENTROID_SAME = 0, CENTROID_AVAIL = 1, CENTROID_NOTAVAIL = 2 };
  double eng_vdwl = 0.0, eng_coul = 0.0;
  double virial[6] = {0, 0, 0, 0, 0, 0};
  double *eatom = nullptr, **vatom = nullptr;
  double cutforce = 0.0;
  double **cutsq = nullptr;
  int **setflag = nullptr;
  int comm_forward = 0, comm_reverse = 0;
  int single_enable = 1, restartinfo = 1, one_coeff = 0, manybody_flag = 0, ghostneigh = 0;
  int unit_convert_flag = 0, no_virial_fdotr = 0, centroidstressflag = CENTROID_SAME;
  double **cutghost = nullptr;
  NeighList *list = nullptr;
  int allocated = 0;

  explicit Pair(LAMMPS *lmp) : Pointers(lmp) {}
  ~Pair() override;

  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual void init_style() {}
  virtual double init_one(int, int) { return 0.0; }
  virtual int pack_forward_comm(int, int *, double *, int, int *) { return 0; }
  virtual void unpack_forward_comm(int, int, double *) {}
  virtual int pack_reverse_comm(int, int, double *) { return 0; }
  virtual void unpack_reverse_comm(int, int *, double *) {}
  virtual double memory_usage() { return 0.0; }
  virtual void *extract(const char *, int &) { return nullptr; }

  void init(); // host side: init_style + init_one for all i<=j, fills cutsq

 protected:
  int *map = nullptr;
  int evflag = 0, eflag_either = 0, eflag_global = 0, eflag_atom = 0;
  int vflag_either = 0, vflag_global = 0, vflag_atom = 0, vflag_fdotr = 0;
  int maxeatom = 0, maxvatom = 0;

  void ev_init(int eflag, int vflag)
  {
    if (eflag || vflag)
      ev_setup(eflag, vflag);
    else
      evflag = eflag_either = eflag_global = eflag_atom = vflag_either = vflag_global = vflag_atom = vflag_fdotr = 0;
  }
  void ev_setup(int eflag, int vflag);
  void virial_fdotr_compute();
};

// the part of LAMMPS' Fix a time-integration fix style touches (fix.h of the 2 Aug 2023 release; the reference
// repository's own plugin fix, USER-BFIELD/fix_bfield.h:33-38, overrides the same virtuals)
namespace FixConst {
  enum { INITIAL_INTEGRATE = 1 << 0, POST_INTEGRATE = 1 << 1, PRE_EXCHANGE = 1 << 2, PRE_NEIGHBOR = 1 << 3,
         POST_NEIGHBOR = 1 << 4, PRE_FORCE = 1 << 5, PRE_REVERSE = 1 << 6, POST_FORCE = 1 << 7, FINAL_INTEGRATE = 1 << 8,
         END_OF_STEP = 1 << 9 };
}

class Fix : protected Pointers {
 public:
  char *id = nullptr, *style = nullptr;
  int igroup = 0, groupbit = 1;
  int time_integrate = 0;
  int force_reneighbor = 0;     // 1: Neighbor::decide() reneighbors on the step next_reneighbor names
  bigint next_reneighbor = -1;

  Fix(LAMMPS *lmp, int narg, char **arg) : Pointers(lmp)
  {
    id = strdup(narg > 0 ? arg[0] : "");
    style = strdup(narg > 2 ? arg[2] : "");
  }
  ~Fix() override
  {
    free(id);
    free(style);
  }
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void setup(int) {}
  virtual void initial_integrate(int) {}
  virtual void final_integrate() {}
  virtual void post_run() {}    // Modify::post_run(): behind the last step of every run
  virtual void reset_dt() {}
};

} // namespace LAMMPS_NS

// lammpsplugin.h
extern "C" {
typedef void *(lammpsplugin_factory1)(void *);
typedef void *(lammpsplugin_factory2)(void *, int, char **);
typedef struct {
  const char *version;
  const char *style;
  const char *name;
  const char *info;
  const char *author;
  union {
    lammpsplugin_factory1 *v1;
    lammpsplugin_factory2 *v2;
  } creator;
  void *handle;
} lammpsplugin_t;
typedef void (*lammpsplugin_regfunc)(lammpsplugin_t *, void *);
typedef void (*lammpsplugin_initfunc)(void *, void *, void *);
void lammpsplugin_init(void *, void *, void *);
}


#define LAMMPS_VERSION "2 Aug 2023"

#endif
