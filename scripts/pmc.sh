#!/bin/bash
# Counter passes of the default bench for ONE kernel name pattern (each pass its own rocprofv3 run, --kernel-trace + --pmc only):
#   gpurun --timeout 1500 -- bash scripts/pmc.sh <kernel substring> "<CTR CTR ...>" ["<CTR ...>" ...]
# Environment: BENCH_ARGS (extra bench flags), BOSSX_LIB.  Prints per pass the average per launch of every counter for kernels whose
# name contains the substring.  `LIST=1` first dumps the available counter names to gpurun_out/pmc/counters.txt.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/pmc"; mkdir -p "$OUT"
K=$1; shift
cd /tmp && export TMPDIR=/tmp
export BOSSX_BATCH_CACHE=/tmp/bossx_batches BOSSX_NO_OVERLAP=1
ARGS="--no-cpu-baseline --no-large --no-others --no-entropy-off-run --no-cold --no-late --steps 8 --warmup 3 $BENCH_ARGS"
[ -n "$LIST" ] && rocprofv3 -L > "$OUT/counters.txt" 2>&1
python3 "$ROOT/bench.py" $ARGS --prepare-only > /dev/null 2>&1
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf "$OUT/p$i"
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i ($set) failed"; tail -3 "$OUT/p$i.log"; continue; }
  python3 - "$OUT/p$i" "$K" <<'PY'
import csv, glob, sys, os
d, k = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = {}
for r in csv.DictReader(open(f[0])):
    if k not in r["Kernel_Name"]: continue
    name = r["Kernel_Name"].split("(")[0][-40:]
    a = acc.setdefault((name, r["Counter_Name"]), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for (name, c), (t, n) in sorted(acc.items()):
    print("%-40s %-28s %16.1f  (%d launches)" % (name, c, t / n, n))
PY
done
