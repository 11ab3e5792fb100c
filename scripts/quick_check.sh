#!/bin/bash
BOSSX_STAGE_TIMING=1 timeout 300 python3 scripts/front_end_timing.py chr20_21 2>&1 | grep -E "stage_batch:|\[parse\]|pass1\] region|threads|process_batch|python:" | sed -n '18,60p'
