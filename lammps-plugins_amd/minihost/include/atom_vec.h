// forwarding header: LAMMPS name -> this repository's mini-host API subset
#include "lammps_host_api.h"
