#!/bin/bash
# A/B of the fused sampler in ONE call on ONE box: the snapshot under _ab_old/ against the working tree, interleaved.
set -e
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  DRK_PKG_ROOT=_ab_old timeout -k 10 120 python -u tools/exp.py time 1024 128
  timeout -k 10 120 python -u tools/exp.py time 1024 128
done
