#!/bin/bash
# write-back experiments: benches of the variants, then the parity subset under the combined variant
bash scripts/r4_variants.sh wball entf both
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx_both.so timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "random_scenarios or incremental or counters_beyond or saturated or lookahead or ecoli" 2>&1 | tail -8 ) > gpurun_out/r4/parity_both.log 2>&1
tail -8 gpurun_out/r4/parity_both.log
