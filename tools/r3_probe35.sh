#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
for c in 0 4096 1024 0 4096; do
  CTTA_RING17=$c python3 $R/tools/prof_unet.py --batch 32 --guided 1 --iters 8 2>&1 | tail -1 | sed "s/^/ring17=$c /"
done
for c in 0 4096 1024; do
  CTTA_RING17=$c python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 8 2>&1 | tail -1 | sed "s/^/ring17=$c /"
done
