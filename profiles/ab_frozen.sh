#!/bin/bash
# profiles/aeam_frozen.py under several environments.  usage: profiles/ab_frozen.sh "VAR=a" "VAR=b VAR2=c" ...  ("-" = none)
set -u
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  ( [ "$v" != "-" ] && export $v; timeout -k 10 300 python3 profiles/aeam_frozen.py ${FROZEN_ARGS:-} 2>&1 | tail -1 )
done
