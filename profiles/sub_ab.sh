#!/bin/bash
# sub-domain step timeline (8 bricks, REBO-MoS) under several environments.  usage: profiles/sub_ab.sh "VAR=a" "VAR=b" ...
set -u
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$ROOT/gpurun_out/sub_ab; rm -rf $OUT; mkdir -p $OUT
  # (the setting is scoped to this one run -- a subshell -- so that variant B does not inherit variant A's variable;
  #  the program itself stays directly behind `--`)
  ( export $v
    rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/subdomain_step.py 24 30 8 > $OUT/subdomain.json 2> $OUT/subdomain.err ) || echo "trace failed"
  (cd $ROOT && python3 profiles/step_timeline.py $OUT 20 > $OUT/timeline.txt 2>&1)
  echo "== $v"; grep -E "start|^step|mean of" $OUT/timeline.txt
  rm -rf $OUT/trace
done
