#!/bin/bash
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
python3 scripts/e2e_timeline.py 2>&1 | tail -9
BOSSX_STAGE_TIMING=1 python3 scripts/host_profile_e2e.py 2>&1 | grep -E "stage_batch|parse\]|pass1\]|ms per lone" | tail -14
python3 scripts/host_profile_e2e.py 2>&1 | tail -32
