#!/bin/bash
# barrier-free chain kernel: parity subset first, then timings against the barrier kernel
# (gpurun -- bash scripts/chain_flow_exp.sh)
mkdir -p gpurun_out/flow


timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --durations=5 > gpurun_out/flow/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/flow/pytest.log
tail -4 gpurun_out/flow/pytest.log
grep -q "failed" gpurun_out/flow/pytest.log && grep "FAILED" gpurun_out/flow/pytest.log
timeout 120 python3 scripts/probe_chain.py 2>&1 | tail -19
for wl in chr20_21 ecoli; do
for env in "X=1" "BOSSX_CHAIN_BARRIER=1" "BOSSX_OVERLAP=1" "BOSSX_NO_OVERLAP=1" "BOSSX_INCREMENTAL=1"; do
  echo "== $wl $env"
  env $env timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step %.3f kernels_only %.3f chain %.3f sweep %.3f ns/bin %.2f stage %.3f' % (d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['kernels']['site_sweep']['avg_ms'], d['chain_latency']['ns_per_bin_longest'], d['host']['stage_batch_ms_mean']))"
done
done

