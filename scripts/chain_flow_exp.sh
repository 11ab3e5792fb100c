#!/bin/bash
# barrier-free chain kernel: timings of alternative builds (boss-runs_amd/csrc/alt/libbossx_*.so), wave
# probe, then the parity file on the default build (gpurun -- bash scripts/chain_flow_exp.sh)
mkdir -p gpurun_out/flow
L=boss-runs_amd/csrc
cp $L/libbossx.so /tmp/libbossx_default.so
python3 bench.py --prepare-only --workload chr20_21 >/dev/null 2>&1
for lib in default $(ls $L/alt 2>/dev/null | sed 's/libbossx_//; s/\.so//') default; do
  if [ $lib = default ]; then cp /tmp/libbossx_default.so $L/libbossx.so; else cp $L/alt/libbossx_$lib.so $L/libbossx.so; fi
  out=$(timeout 600 python bench.py --workload chr20_21 --no-cpu-baseline --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step %.3f kernels_only %.3f chain %.3f ns/bin %.2f stage %.3f' % (d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['chain_latency']['ns_per_bin_longest'], d['host']['stage_batch_ms_mean']))")
  echo "== $lib: $out"
done
cp /tmp/libbossx_default.so $L/libbossx.so
timeout 120 python3 scripts/probe_chain.py 2>&1 | grep -E "wave (0|1|9|10|11|13|14):" | tail -7
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x > gpurun_out/flow/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/flow/pytest.log
