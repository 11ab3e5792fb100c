"""One lone update out of a rocprofv3 kernel trace (scripts/regime_trace.sh leaves gpurun_out/regime_<d>/r_kernel_trace.csv): every launch
with start / end relative to the update's first launch, its stream, and the gap in front of the critical kernels.
   python3 scripts/update_timeline.py gpurun_out/regime_8/r_kernel_trace.csv [update index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# an update ends with strategy_mask_kernel
ends = [i for i, r in enumerate(rows) if "strategy_mask_kernel" in r["Kernel_Name"]]
hi = ends[-k]; lo = ends[-k - 1] + 1
sel = rows[lo:hi + 1]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = {}
print("update of %d launches, %.1f us from first start to last end" % (len(sel), (int(sel[-1]["End_Timestamp"]) - t0) / 1e3))
for r in sel:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("bossx::", "").split("(")[0][:44]
    print("%8.1f -> %8.1f  (%7.1f us)  q%-2s %s" % (s, e, e - s, r["Queue_Id"], name))
