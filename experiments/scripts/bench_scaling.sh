#!/bin/bash
# Frame-parallel throughput on 1, 2, 4, 8 GPUs of one node (BASELINE configs[4]; bench.py launches its own ranks).
set -e
for n in 1 2 4 8; do python bench.py --gpus $n --frames 1024 --cpu-frames 0 --profile-steps 0 --sustained-seconds 0 --skip-pcie; done
