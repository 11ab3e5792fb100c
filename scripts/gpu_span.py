"""GPU-side span of lone updates out of a rocprofv3 kernel trace: per update (ending with strategy_mask_kernel) the time from the first
upload / from build_groups_kernel to the end of the masks, and the mean duration of every kernel on the way — the low-noise figure for
changes to the device side of the front end (wall-clock per update moves by 0.1 ms from box to box and run to run).
   python3 scripts/gpu_span.py <kernel_trace.csv> [updates to skip at the start, default 6]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "strategy_mask_kernel" in r["Kernel_Name"]]
spans, spans_g, dur = [], [], collections.defaultdict(list)
for a, b in zip(ends[skip:-1], ends[skip + 1:]):
    sel = rows[a + 1:b + 1]
    if not any("build_groups_kernel" in r["Kernel_Name"] for r in sel):
        continue
    t_end = int(sel[-1]["End_Timestamp"])
    t_up = int(sel[0]["Start_Timestamp"])
    t_bg = min(int(r["Start_Timestamp"]) for r in sel if "build_groups_kernel" in r["Kernel_Name"])
    spans.append((t_end - t_up) / 1e3); spans_g.append((t_end - t_bg) / 1e3)
    per = collections.defaultdict(float)
    for r in sel:
        per[r["Kernel_Name"].replace("bossx::", "").split("(")[0][:40]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in per.items():
        dur[k].append(v)
import statistics as st
print("%d updates: first upload -> masks end  median %.1f us (min %.1f) | build_groups -> masks end  median %.1f us (min %.1f)" % (
    len(spans), st.median(spans), min(spans), st.median(spans_g), min(spans_g)))
for k, v in sorted(dur.items(), key=lambda kv: -st.median(kv[1])):
    if "upload" in k: continue
    print("   %-42s median %7.1f us per update" % (k, st.median(v)))
