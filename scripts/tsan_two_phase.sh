#!/bin/bash
# ThreadSanitizer over the host half of staging (CPU only; sanitizers are not available on the GPU pool)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tsan
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from boss_runs_amd import synth
contigs = synth.make_reference([4_641_652], seed=1, names=["ecoli"])
b = synth.make_batch(contigs, 1500, seed=10)
open('gpurun_out/tsan/paf.txt', 'w').write(b["paf"])
with open('gpurun_out/tsan/reads.txt', 'w') as f:
    for n, s in b["seqs"].items():
        f.write("%s\t%d\n" % (n, len(s)))
PY
g++ -O1 -g -fsanitize=thread -pthread -std=c++17 -Iinclude -Iboss-runs_amd/csrc scripts/tsan_two_phase.cpp boss-runs_amd/csrc/paf_host.cpp -o gpurun_out/tsan/tsan_two_phase
for t in 2 8; do TSAN_OPTIONS="halt_on_error=0" gpurun_out/tsan/tsan_two_phase gpurun_out/tsan/paf.txt gpurun_out/tsan/reads.txt 4641652 $t 2>&1 | tail -15; done
