#!/bin/bash
# candidate rows left standing where their inputs did not change: parity, then on / off (gpurun -- bash scripts/r4_skip.sh)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
[ -z "$STATS" ] && unset BOSSX_SPEC_STATS   # (the counters cost an atomic per row: timings with STATS=1 are not timings)
if [ -n "$WITH_PARITY" ]; then
( timeout 1500 python -m pytest tests -m gpu -x -q -k "chain or saturat or movesum or random_scenarios or incremental or chr20_21_full or ecoli or lookahead" 2>&1 | tail -4 ) > gpurun_out/r4/parity_skip.log 2>&1
tail -4 gpurun_out/r4/parity_skip.log
fi
for off in "" 1; do
  export BOSSX_SPEC_NO_SKIP=$off; [ -z "$off" ] && unset BOSSX_SPEC_NO_SKIP
  for w in chr20_21 ecoli; do
  env ${STATS:+BOSSX_SPEC_STATS=1} timeout 600 python bench.py --workload $w --no-cpu-baseline --no-others --no-large --no-entropy-off-run --steps 20 --warmup 5 2> /tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skip %-3s %-8s lone %.3f resident %.3f chain %.3f' % ('off' if '$off' else 'on', '$w', d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms']), {a:c for a,c in d['benefit_chain_form'].items() if a not in ('note','per_loop')})"
  grep -o "chunk-parallel chain: [0-9]* (window, chunk) tables built + [0-9]* left standing" /tmp/err.txt | tail -1
  done
  env ${STATS:+BOSSX_SPEC_STATS=1} BOSSX_BATCH_CACHE=/tmp/bossx_batches_grch38 timeout 900 python bench.py --workload grch38 --steps 8 --warmup 3 2> /tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['grch38']; print('skip %-3s grch38 lone' % ('off' if '$off' else 'on'), d['ms_per_step'], 'chain', g['benefit_chain_ms_rank0'])"
  grep -o "chunk-parallel chain: [0-9]* (window, chunk) tables built + [0-9]* left standing" /tmp/err.txt | tail -1
done
