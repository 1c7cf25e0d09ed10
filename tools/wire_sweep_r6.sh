#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for nb in 16 32 48 64 96 128; do
  for g in default 24 32 48 64; do
    if [ "$g" = default ]; then python tools/wire_sweep.py 4 $nb registered; else
      if [ $g -le $nb ]; then BSW_REFBATCH_GROUP=$g python tools/wire_sweep.py 4 $nb registered; fi; fi
  done
done
