#!/bin/bash
# arms alternated: product library (working tree) vs build/libchiron_beam_HEAD.so (the committed beam.hip)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "beam" 2>&1 | tail -2
for r in 1 2; do
  for arm in new old; do
    if [ $arm = old ]; then export CHIRON_AMD_LIB=$GRAFT_REPO_ROOT/build/libchiron_beam_HEAD.so; else unset CHIRON_AMD_LIB; fi
    echo "== $arm alone"; python tools/beam_peaked.py 1100 2>/dev/null | grep "beam 30\|beam 50"
    echo "== $arm mix"; BENCH_STEPS=150 python tools/beam_mix_peaked.py 2>/dev/null | tail -4
  done
done
