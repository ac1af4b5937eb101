#!/bin/bash
# Same-box A/B of library builds on the headline bench (run on the GPU box): alternates the arms, REPS times each.
#   tools/ab_run.sh REPS name1=lib1.so name2=lib2.so ...     ("product" as a lib path = the tree's own library)
# Prints per arm the headline of every repetition and, from the single-slot profiling pass, the per-kernel averages.
REPS=$1; shift
OUT=${AB_OUT:-gpurun_out/ab}
mkdir -p "$OUT"
for r in $(seq 1 "$REPS"); do
  for arm in "$@"; do
    name=${arm%%=*}; lib=${arm#*=}
    if [ "$lib" = product ]; then unset CHIRON_AMD_LIB; else export CHIRON_AMD_LIB=$lib; fi
    CHIRON_ALLOW_TIMING_BUILD=1 python bench.py --steps 20 --rounds ${AB_ROUNDS:-15} --host-rounds 0 --no-f16 --no-cpu-baseline --density-rounds 0 > "$OUT/${name}_$r.json" 2> "$OUT/${name}_$r.err"
  done
done
unset CHIRON_AMD_LIB
python - "$OUT" "$REPS" "$@" <<'PY'
import json, sys
out, reps, arms = sys.argv[1], int(sys.argv[2]), [a.split("=")[0] for a in sys.argv[3:]]
for a in arms:
    vals, kern = [], {}
    for r in range(1, reps + 1):
        try:
            j = json.loads(open("%s/%s_%d.json" % (out, a, r)).read().strip().splitlines()[-1])
        except Exception as e:
            print(a, r, "failed", e); continue
        vals.append(j["value"])
        for k, v in j["extra"]["kernels"].items():
            kern.setdefault(k, []).append(v["avg_ms"])
    print("%-14s headline %s" % (a, " ".join("%.0f" % v for v in vals)))
    print("   " + "  ".join("%s %.3f" % (k, sum(v) / len(v)) for k, v in sorted(kern.items())))
PY
