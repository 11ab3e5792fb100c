#!/bin/bash
# ONE end-of-round collection (once per round): the GPU tier + smoke, the rocprofv3 summaries of three workloads (kernel
# trace + four PMC passes each, scripts/profile_gpu.sh), the bench lines of all four, the memory-pattern ceiling and the regime traces.
#   gpurun --timeout 2700 -- bash scripts/collect_round.sh [tier|profiles|bench|extras|all]
# Everything lands under gpurun_out/; scripts/collect_round.sh copy <rNN> then files the judged summaries under profiles/ (run locally).
what=${1:-all}
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
if [ "$what" = copy ]; then
  R=${2:?round tag, e.g. r05}
  for w in chr20_21 ecoli barcoded; do
    cp gpurun_out/prof_$w/${w}_summary.json profiles/${R}_${w}_rocprof_summary.json 2>/dev/null
    cp gpurun_out/prof_$w/${w}_kernel_stats.csv profiles/${R}_${w}_kernel_stats.csv 2>/dev/null
  done
  for f in default:chr20_21 ecoli:ecoli barcoded:barcoded grch38:grch38_1gpu; do cp gpurun_out/bench/${f%%:*}.json profiles/${R}_${f##*:}_bench.json 2>/dev/null; done
  cp gpurun_out/tier/gpu_tier_summary.txt profiles/${R}_gpu_tier.txt 2>/dev/null
  cp gpurun_out/extras/rw_pattern.txt profiles/${R}_rw_pattern.txt 2>/dev/null
  cp gpurun_out/extras/regimes.txt profiles/${R}_regimes.txt 2>/dev/null
  cp gpurun_out/extras/drift.txt profiles/${R}_drift.txt 2>/dev/null
  cp gpurun_out/grch38/trace/grch38_kernel_stats.csv profiles/${R}_grch38_1gpu_kernel_stats.csv 2>/dev/null
  ls -la profiles/${R}_*
  exit 0
fi
if [ "$what" = tier ] || [ "$what" = all ]; then
  mkdir -p gpurun_out/tier
  timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/tier/gpu_tier.log 2>&1
  ( grep -E "passed|failed|error" gpurun_out/tier/gpu_tier.log | tail -3; grep -E "^[0-9.]+s (call|setup)" gpurun_out/tier/gpu_tier.log | head -8
    timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee gpurun_out/tier/gpu_tier_summary.txt
fi
if [ "$what" = profiles ] || [ "$what" = all ]; then
  for w in chr20_21 ecoli barcoded; do timeout 900 bash scripts/profile_gpu.sh $w > gpurun_out/prof_$w.log 2>&1; tail -6 gpurun_out/prof_$w.log | cut -c1-260; done
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  mkdir -p gpurun_out/bench
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench/default.json 2> gpurun_out/bench/default.err     # the driver's command
  timeout 300 python bench.py --workload ecoli --no-others --no-large > gpurun_out/bench/ecoli.json 2> gpurun_out/bench/ecoli.err
  timeout 300 python bench.py --workload barcoded --no-others --no-large > gpurun_out/bench/barcoded.json 2> gpurun_out/bench/barcoded.err
  timeout 300 python bench.py --workload grch38 --steps 8 --warmup 3 > gpurun_out/bench/grch38.json 2> gpurun_out/bench/grch38.err
  for f in default ecoli barcoded grch38; do python3 -c "
import json,sys
d=json.load(open('gpurun_out/bench/$f.json'))
print('$f', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], 'kernels_only', d.get('kernels_only_ms'), 'roofline', d.get('roofline',{}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('ms_per_update'), 'equal', (d.get('cpu_baseline') or {}).get('masks_and_threshold_equal_to_gpu'), 'cold', (d.get('cold_update_ms') or {}).get('after_2s_idle_ms_median'), 'late', (d.get('late_regime') or {}).get('ms_per_step'))
"; done
fi
if [ "$what" = extras ] || [ "$what" = all ]; then
  mkdir -p gpurun_out/extras
  [ -x scripts/rw_pattern.bin ] && scripts/rw_pattern.bin > gpurun_out/extras/rw_pattern.txt 2>&1
  ( for d in 8 28; do bash scripts/regime_trace.sh $d; done; bash scripts/trace_timeline.sh | tail -28 ) > gpurun_out/extras/regimes.txt 2>&1
  bash scripts/grch38_trace.sh > gpurun_out/extras/grch38_trace.txt 2>&1
  ( python3 scripts/drift_steps.py chr20_21 60 2>&1 | tail -12; bash scripts/drift_trace.sh 2>&1 | tail -24 ) > gpurun_out/extras/drift.txt 2>&1      # lone updates as coverage accumulates
  tail -3 gpurun_out/extras/rw_pattern.txt
fi
