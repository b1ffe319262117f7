for v in 0 1 2 3 4 5; do
  MDP_LJ_VARIANT=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
for s in 1.0 0.6; do
  python bench.py --steps 300 --warmup 5 --temp 300 --inner-skin $s --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=300K inner $s', d['value'], d['ms_per_step'], d['config']['style_list_builds_in_timed_region_rank0'], d['config']['neighbor_rebuilds_in_timed_region'])"
done
