#!/bin/bash
# usage: tools/ubench/pmc_probe.sh <outdir> <probe binaries...>  -- counter passes over the GEMM probe variants
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for bin in "$@"; do
  n=$(basename $bin)
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
     --output-format csv -d "$OUT" -o "${n}_a" -- $bin $n > "$OUT/$n.a.txt" 2> "$OUT/$n.a.err"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
     --output-format csv -d "$OUT" -o "${n}_b" -- $bin $n > "$OUT/$n.b.txt" 2> "$OUT/$n.b.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("==", f.split("/")[-1])
    for (k, c), v in sorted(acc.items()):
        # the probe launches K=256 (23 launches) then K=768 (23 launches) with the same kernel: split halves
        h = len(v) // 2
        print("  %-28s K256 %.4g   K768 %.4g" % (c, sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1)))
PY
