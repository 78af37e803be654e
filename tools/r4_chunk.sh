#!/bin/bash
cd $GRAFT_REPO_ROOT
for cs in 1024 512 256 128; do
  echo "== TEHMM_SPEC_CHUNK=$cs"
  export TEHMM_SPEC_CHUNK=$cs
  SINGLE=1 STAGES=viterbi,both timeout -k 10 200 python tools/stage_bench.py 10 2>/dev/null | cut -c1-420
done
for cs in 1024 512; do
  echo "== TEHMM_SPEC_CHUNK=$cs (100 Mb)"
  export TEHMM_SPEC_CHUNK=$cs
  STAGES=viterbi,both timeout -k 10 200 python tools/stage_bench.py 100 2>/dev/null | cut -c1-420
done
