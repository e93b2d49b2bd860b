#!/bin/bash
# tile sweep of selected shapes on the GPU box
python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -2
echo "== default"; python tools/bench_conv.py 2>&1 | grep "^\["
for t in 0 1 3 4 5 6 7; do
  echo "== CAPF_TILE=$t"
  for i in 1 2 3 5 8; do CAPF_TILE=$t python tools/bench_conv.py --only $i 2>&1 | grep "^\["; done
done
