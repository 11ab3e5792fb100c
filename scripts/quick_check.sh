#!/bin/bash
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "end_to_end or saturated or chain" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('chr20_21 ms_per_step %.3f kernels_only %.3f chain %.3f stage %.3f' % (d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['host']['stage_batch_ms_mean']))"
