#!/bin/bash
# A/B of the working tree's library against an experiment build (drakes_amd/libexp_<name>.so), interleaved, one box.
set -e
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  timeout -k 10 120 python -u tools/exp.py time 1024 128
  DRK_LIB=$GRAFT_REPO_ROOT/drakes_amd/libexp_$1.so timeout -k 10 120 python -u tools/exp.py time 1024 128
done
