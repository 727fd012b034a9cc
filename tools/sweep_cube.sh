#!/usr/bin/env bash
# tools/sweep_cube.sh -- light-cube resolution sweep (MIRT_CUBE_BINS) on the two soup workloads
for cb in 64 128 256; do
  MIRT_CUBE_BINS=$cb python bench.py --workload soup1m8k --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('1M cube $cb', d['ms_per_step'], d['value'], d['kernel_ms_rank0'], d['roofline']['tests_per_launch'])"
done
for cb in 64 128; do
  MIRT_CUBE_BINS=$cb python bench.py --workload soup100k --no-cpu-baseline 2>/dev/null > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('100k cube $cb', d['ms_per_step'], d['value'], d['kernel_ms_rank0'], d['roofline']['tests_per_launch'])"
done
