#!/bin/bash
# which counters exist + one PMC pass over the GRCh38 single-GPU bench (address-translation counters)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/r3pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "utcl|tlb|translation|TCP_.*MISS|TCP_PENDING|TCP_TCC_READ_REQ_LATENCY|TCP_TA_TCP_STATE" | cut -c1-160 | sort -u | head -60 > "$OUT/avail.txt"
wc -l "$OUT/avail.txt"; head -40 "$OUT/avail.txt"
W=${1:-grch38}
ARGS="--workload $W --no-cpu-baseline --no-large --no-others --steps 4 --warmup 2"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_$W
python3 "$ROOT/bench.py" $ARGS --prepare-only
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$tag" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_$tag.log" 2>&1 || echo "pass $tag failed"
  f=$(find "$OUT/pmc_$tag" -name "*counter_collection.csv" | head -1)
  [ -f "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "site_sweep" in k or "expand" in k:
        print(k, {c: round(v / max(n[(k, c)], 1)) for c, v in d.items()}, "launches", max(n[(k, c)] for c in d))
PY
done
