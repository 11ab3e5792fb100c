timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "chunk_parallel" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15
