#!/bin/bash
# tile sweep of selected conv shapes on the GPU box: tools/sweep_tiles.sh "<shape ids>" "<tile ids>"
shapes=${1:-"1 2 3 5 8"}
tiles=${2:-"0 1 2 3"}   # 0 128x64, 1 64x64, 2 128x128, 3 256x32
echo "== default"; for i in $shapes; do python tools/bench_conv.py --only $i 2>&1 | grep "^\["; done
for t in $tiles; do
  echo "== CAPF_TILE=$t"
  for i in $shapes; do CAPF_TILE=$t python tools/bench_conv.py --only $i 2>&1 | grep "^\["; done
done
