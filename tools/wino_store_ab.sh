#!/bin/bash
# Round-5 review, item 7: re-take the Winograd F(4,3) kernel's HBM counters once, with a store-order A/B.
#   product: per channel quad q, the four output rows' 32-byte pieces (a 128-byte line is completed by every fourth store instruction)
#   variant: build/libchiron_wino_CHIRON_WINO_ROWMAJOR_STORES_1.so -- row-major, the four pieces of one line in consecutive instructions
# Separate --pmc passes (FETCH_SIZE / WRITE_SIZE), kernel trace for the launch time; bench workload with one batch in flight.
#   usage (GPU box): tools/variants.sh --product wino CHIRON_WINO_ROWMAJOR_STORES 1   (build container)
#                    bash tools/wino_store_ab.sh gpurun_out/wino_ab
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="--steps 3 --rounds 1 --host-rounds 0 --warmup 1 --slots 1 --no-cpu-baseline --no-f16 --density-rounds 0"
for arm in product rowmajor product rowmajor; do
  if [ $arm = rowmajor ]; then export CHIRON_AMD_LIB=$GRAFT_REPO_ROOT/build/libchiron_wino_CHIRON_WINO_ROWMAJOR_STORES_1.so; else unset CHIRON_AMD_LIB; fi
  n=$(ls "$OUT" | grep -c "^$arm" || true)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o ${arm}_rd_$n -- python bench.py $ARGS > /dev/null 2> "$OUT/${arm}_rd_$n.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o ${arm}_wr_$n -- python bench.py $ARGS > /dev/null 2>> "$OUT/${arm}_rd_$n.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${arm}_t_$n -- python bench.py $ARGS > /dev/null 2>> "$OUT/${arm}_rd_$n.err"
done
unset CHIRON_AMD_LIB
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
def counter(path, name):
    tot, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if "wino_conv3_f4_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
            tot += float(row["Counter_Value"]); n += 1
    return tot / max(n, 1), n
for arm in ("product", "rowmajor"):
    for f in sorted(glob.glob(os.path.join(out, "**", arm + "_rd_*counter_collection.csv"), recursive=True)):
        v, n = counter(f, "FETCH_SIZE")
        print("%-9s FETCH_SIZE per launch %.1f KB-units (x 2 x 1024 B on gfx950 = %.1f MB), %d launches   %s" % (arm, v, v * 2 * 1024 / 1e6, n, os.path.basename(f)))
    for f in sorted(glob.glob(os.path.join(out, "**", arm + "_wr_*counter_collection.csv"), recursive=True)):
        v, n = counter(f, "WRITE_SIZE")
        print("%-9s WRITE_SIZE per launch %.1f KB-units (= %.1f MB), %d launches   %s" % (arm, v, v * 1024 / 1e6, n, os.path.basename(f)))
    for f in sorted(glob.glob(os.path.join(out, "**", arm + "_t_*kernel_stats.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if "wino_conv3_f4_kernel" in row["Name"]:
                print("%-9s launch time avg %.1f us over %s calls   %s" % (arm, float(row["AverageNs"]) / 1e3, row["Calls"], os.path.basename(f)))
PY
