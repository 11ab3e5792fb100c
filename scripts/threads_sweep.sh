#!/bin/bash
make -C boss-runs_amd/csrc > /dev/null 2>&1
for t in 8 12 16 24 32 48; do echo "threads $t"; BOSSX_PARSE_THREADS=$t python3 scripts/e2e_timeline.py chr20_21 2>&1 | grep -E "^stage|^total"; done
