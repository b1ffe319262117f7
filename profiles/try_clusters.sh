for cl in 1 2 4; do
  MDP_CLUSTER=$cl python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cluster $cl', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
