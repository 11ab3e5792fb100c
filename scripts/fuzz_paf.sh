#!/bin/bash
# ASan + UBSan build of the native PAF / CIGAR front end and its mutation fuzzer (CPU only, no device):
#   scripts/fuzz_paf.sh [batches per process, default 125000] [processes, default 8]   (= 10^6 mutated batches)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/bossx_fuzz
mkdir -p "$OUT"
N=${1:-125000}
P=${2:-8}
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -pthread \
    -I"$ROOT/include" -I"$ROOT/boss-runs_amd/csrc" "$ROOT/scripts/fuzz/fuzz_paf.cpp" "$ROOT/boss-runs_amd/csrc/paf_host.cpp" -o "$OUT/fuzz_paf"
export ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 BOSSX_POOL_THREADS=3
pids=()
for i in $(seq 1 "$P"); do "$OUT/fuzz_paf" "$N" "$((1000 + i))" > "$OUT/out_$i.log" 2>&1 & pids+=($!); done
rc=0
for i in $(seq 1 "$P"); do wait "${pids[$((i - 1))]}" || rc=1; tail -3 "$OUT/out_$i.log"; done
exit $rc
