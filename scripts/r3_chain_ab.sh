mkdir -p gpurun_out/r3exp
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "end_to_end or chain or random or awkward or schedule or saturated" 2>&1 | tail -4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_exp
for gc in 1 0 1 0; do
  BOSSX_CHAIN_GC=$gc timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 > gpurun_out/r3exp/bench_gc$gc.json 2>gpurun_out/r3exp/bench_gc$gc.err || tail -5 gpurun_out/r3exp/bench_gc$gc.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r3exp/bench_gc$gc.json')); print('gc=$gc step %.3f chain %.3f'%(d['ms_per_step'], d['kernels']['benefit_chain']['avg_ms']))"
done
PROBE_L=64444167 timeout 300 python3 scripts/probe_chain.py 2>&1 | tail -30
