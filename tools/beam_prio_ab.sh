#!/bin/bash
# round-5 review item 4, second half: the decoder's wave priority.  Product (s_setprio 3 in the register beam kernels) against
# build/libchiron_beam_CHIRON_BEAM_PRIO_0.so (priority 0: the decoder yields to the other batches' network kernels), arms alternated:
# RNA configs[2] (beam 50, batch 400) and DNA beam 30 / greedy, three batches in flight (tools/bench_configs.py).
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for arm in prio3 prio0; do
    if [ $arm = prio0 ]; then export CHIRON_AMD_LIB=$GRAFT_REPO_ROOT/build/libchiron_beam_CHIRON_BEAM_PRIO_0.so; else unset CHIRON_AMD_LIB; fi
    BENCH_STEPS=300 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$arm', d['config'], d['ms_per_batch'], 'ctc_beam alone', d['kernels_ms'].get('ctc_beam'))"
  done
done
