#!/bin/bash
# the mini-host's logs on several ranks (one GPU, ranks as threads; the bricks' RCCL through the test double)
cd $GRAFT_REPO_ROOT/lammps-plugins_amd; O=$GRAFT_REPO_ROOT/gpurun_out/r06_minilmp_ranks; mkdir -p $O
export MDP_FIX_STATS=1
./minilmp -np 4 -in examples/in.rebomos-bulk.mi355x > $O/log.rebomos-bulk.4.host_mode 2>&1
export MDP_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/native/libfake_rccl.so MDP_FAKE_RCCL_TIMEOUT_S=60
./minilmp -np 4 -in examples/in.rebomos-bulk.nve-mdp.mi355x > $O/log.rebomos-bulk.4.fix_nve_mdp_bricks 2>&1
./minilmp -np 8 -in examples/in.aeam-alsi.nve-mdp.mi355x > $O/log.aeam-alsi.8.fix_nve_mdp_bricks 2>&1
unset MDP_RCCL_LIBRARY
./minilmp -np 8 -in examples/in.aeam-alsi.mi355x > $O/log.aeam-alsi.8.host_mode 2>&1
./minilmp -in examples/in.aeam-alsi.mi355x > $O/log.aeam-alsi.1.host_mode 2>&1
grep -h -A6 "^ *Step" $O/log.aeam-alsi.* | head -40
