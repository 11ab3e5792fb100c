#!/bin/bash
# Kernel start/end timestamps of the last updates of the default bench (overlap picture).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/trace_tl"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o tl -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-large --no-others --no-entropy-off-run --steps 4 --warmup 2 > "$OUT/bench.log" 2>&1
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-30:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %8.1f us  q%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
PY
