#!/bin/bash
# same-box A/B of the zero-copy windows (signal_io.window_signal hands out overlapping views; CHIRON_WINDOW_COPY=1 = the copy of rounds 1..4)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for arm in view copy; do
  if [ $arm = copy ]; then export CHIRON_WINDOW_COPY=1; else unset CHIRON_WINDOW_COPY; fi
  echo "== $arm"
  python tools/e2e_bench.py 2048 0 - fp16 4096 fast5 2>&1 | tail -1 | cut -c1-150
  python tools/e2e_bench.py 2048 0 - fp16 4096 signal 2>&1 | tail -1 | cut -c1-150
  rm -rf /dev/shm/hc; mkdir -p /dev/shm/hc
  python tools/host_ceiling.py --ranks 1 --reads 2048 --inputs fast5 --workdir /dev/shm/hc 2>&1 | grep windows_per_s | cut -c1-120
done; done; rm -rf /dev/shm/hc
