#!/bin/bash
# Host-only timing of the PAF/CIGAR front end on an E. coli 4000-read batch, 1..N threads.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pb
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from boss_runs_amd import synth
contigs = synth.make_reference([4_641_652], seed=1, names=["ecoli"])
b = synth.make_batch(contigs, 4000, seed=10)
open('gpurun_out/pb/paf.txt', 'w').write(b["paf"])
with open('gpurun_out/pb/reads.txt', 'w') as f:
    for n, s in b["seqs"].items():
        f.write("%s\t%d\n" % (n, len(s)))
PY
g++ -O2 -pthread -DBOSSX_PARSE_TIMING -std=c++17 -Iinclude -Iboss-runs_amd/csrc scripts/parse_bench.cpp boss-runs_amd/csrc/paf_host.cpp -o gpurun_out/pb/parse_bench
echo "nproc $(nproc)"
for t in 1 2 4 8 16; do echo "threads $t"; gpurun_out/pb/parse_bench gpurun_out/pb/paf.txt gpurun_out/pb/reads.txt 4641652 $t 2>&1 | grep -v "^rc" | tail -1; done
