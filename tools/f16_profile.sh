#!/bin/bash
# rocprofv3 kernel trace of the fp16 engine at BASELINE configs[4] (batch 4096): per-kernel averages, single stream
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/f16prof
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o f16 -- python tools/f16_probe.py "$@" > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/f16prof/**/*kernel_stats.csv"), recursive=True)
for row in list(csv.DictReader(open(f[0])))[:18]:
    print("%-88s calls %4s avg %9.1f us  %5s %%" % (row["Name"][:88], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
