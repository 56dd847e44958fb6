#!/bin/bash
# dev: one line for the table-scene bench -- solves/s, kernel ms, executed products, time per product and CU
timeout 300 python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --serving-streams 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['hv_products']['executed_per_gpu']; print(round(d['value'],1), 'solves/s, kernel', round(d['roofline']['kernel_ms'],1), 'ms, executed', e, ', ns per product per CU', round(d['roofline']['kernel_ms']*1e6*256/e,1))"
