// forwards a LAMMPS header name to the mini-host API subset
#include "lammps_host_api.h"
