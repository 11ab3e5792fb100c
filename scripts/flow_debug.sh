#!/bin/bash
# which stage slows the chain wave down? (timing only: results are wrong with BOSSX_FLOW_DEBUG bits 1/2 set)
for d in 0 3 64; do
  echo "== BOSSX_FLOW_DEBUG=$d"
  BOSSX_FLOW_DEBUG=$d timeout 120 python3 scripts/probe_chain.py 2>&1 | grep -E "wave 0:|wave 1:|wave 10:|wave 14:|benefit ms" | tail -7
done
