#!/bin/bash
# ablation of the fp32 conv kernel on selected shapes (GPU box): CAPF_ABLATE 0..6
for i in 1 2 5; do
  for ab in 0 1 3 4 5 6 2; do
    echo -n "abl=$ab "; CAPF_ABLATE=$ab python tools/bench_conv.py --only $i 2>&1 | grep "^\["
  done
done
