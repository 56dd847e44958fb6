#!/bin/bash
# KUKA 8192 / 65536 and LWA4D 16384: round-robin slicing + tail spreading (0), spreading only
# (GIK_DBG=1024), neither (512):
# kernel ms per step, 6 steps each.  Run on the GPU box: bash tools/gpu_tail.sh
for cfg in "--robot kuka --batch 8192" "--robot kuka --batch 65536" "--robot lwa4d --batch 16384" "--robot ur10 --batch 8192"; do
  for dbg in 0 1024 512; do
    GIK_DBG=$dbg python bench.py $cfg --steps 6 --warmup 2 --no-cpu-baseline --serving-streams 0 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'dbg=$dbg', round(d['value']), [round(x,1) for x in d['roofline']['kernel_ms_per_step']])"
  done
done
