for v in 0 1 2; do
  MDP_LJ_VARIANT=$v python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
