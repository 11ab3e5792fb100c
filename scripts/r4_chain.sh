#!/bin/bash
# chain tests + kernel trace of the chain kernels (gpurun -- bash scripts/r4_chain.sh)
mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "chain or saturat or end_to_end or random or incremental or golden" 2>&1 | tail -6 ) > gpurun_out/r4/chain_parity.log 2>&1
tail -6 gpurun_out/r4/chain_parity.log
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
python bench.py --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 > gpurun_out/r4/bench_chain.json 2> gpurun_out/r4/bench_chain.err || tail -5 gpurun_out/r4/bench_chain.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_chain.json'))
print("step %.3f lone %.3f resident %.3f | sweep %.4f | chain %.3f | form %s" % (d["ms_per_step"], d.get("lone_update_ms", 0), d["kernels_only_ms"], d["roofline"]["avg_launch_ms"], d["kernels"]["benefit_chain"]["avg_ms"], {k: v for k, v in d["benefit_chain_form"].items() if k != "note"}))
print(d.get("kernels_only_note"))
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_chain -o chain -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_chain/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("%-70s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
