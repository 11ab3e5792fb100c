#!/bin/bash
# Kernel start / end timestamps of the LAST lone update of scripts/e2e_timeline.py (14 lone updates, PAF text -> masks), from a
# rocprofv3 kernel trace: what runs when, and the gaps between launches.   gpurun -- bash scripts/trace_timeline.sh [workload]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/trace_tl"
rm -rf "$OUT"; mkdir -p "$OUT"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
cd "$ROOT" && python3 bench.py --no-cpu-baseline --no-large --no-others --no-entropy-off-run --no-cold --no-late --prepare-only > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o tl -- python3 scripts/e2e_timeline.py ${1:-chr20_21} > "$OUT/run.log" 2>&1 )
tail -8 "$OUT/run.log"
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "cigar_walk_kernel<false>" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    busy += e - s
    print("%9.1f %9.1f %8.1f us  gap %6.1f  q%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, (s - (prev_end - t0)) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("bossx::", "")[:70]))
    prev_end = max(prev_end, int(r["End_Timestamp"]))
print("span %.1f us, kernels %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(rows)))
PY
