export BOSSX_BATCH_CACHE=/tmp/bc
for i in 1 2; do for w in chr20_21; do python bench.py --workload $w --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 2>/tmp/err_$w.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', \"step %.3f lone %.3f resident %.3f chain %.3f\"%(d[\"ms_per_step\"], d[\"lone_update_ms\"], d[\"kernels_only_ms\"], d[\"kernels\"][\"benefit_chain\"][\"avg_ms\"]), {k:v for k,v in d[\"benefit_chain_form\"].items() if k!=\"note\"}, 'resident chain', d['kernels_resident_loop']['benefit_chain']['avg_ms'])"; done; done
