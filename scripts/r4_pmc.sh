#!/bin/bash
# one PMC pass over the default bench, per-kernel averages of the given counters (gpurun -- bash scripts/r4_pmc.sh CTR1 CTR2 ...)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4 BOSSX_NO_OVERLAP=1
ARGS="--no-cpu-baseline --no-others --no-large --steps 10 --warmup 3"
python3 "$ROOT/bench.py" $ARGS --prepare-only
rm -rf /tmp/pmc_r4
timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_r4 -o r4 -- python3 "$ROOT/bench.py" $ARGS > /tmp/pmc_r4.log 2>&1 || tail -5 /tmp/pmc_r4.log
python3 - "$@" <<'PY'
import csv, glob, collections, sys
f = glob.glob('/tmp/pmc_r4/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-36:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    if any(x in k for x in ("chain", "sweep1")):
        print("%-38s" % k, "  ".join("%s=%.3g" % (c, acc[k][c] / max(cnt[k][c], 1)) for c in sys.argv[1:] if c in acc[k]), " (n=%d)" % max(cnt[k].values()))
PY
