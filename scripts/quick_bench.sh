#!/bin/bash
# short bench of the default workload + probes (gpurun -- bash scripts/quick_bench.sh)
mkdir -p gpurun_out/quick
python3 scripts/sweep_probe.py chr20_21 2>&1 | grep probe | tail -3
python bench.py --no-cpu-baseline --no-others > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/quick/bench.json'))
print("ms_per_step %.3f kernels_only %.3f" % (d["ms_per_step"], d["kernels_only_ms"]))
print("sweep ms %.4f frac %.3f | chain ms %.3f" % (d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["kernels"]["benefit_chain"]["avg_ms"]))
print("large stream %.3f gather %.3f" % (d["roofline_large"]["stream"]["frac"], d["roofline_large"]["gather"]["frac"]))
PY
