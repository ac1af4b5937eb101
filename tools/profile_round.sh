#!/bin/bash
# Everything under profiles/rNN_* in one gpurun call (run from the repo root on the GPU box):
#   tools/profile_round.sh r02        ->  gpurun_out/r02/...   (copy what is to be judged into profiles/ afterwards)
# Order matters: the PMC record is written into profiles/ on the box BEFORE the bench runs, so that bench.py finds a record
# collected on this tree's kernel sources (roofline.traffic / roofline.pmc are withheld otherwise).
set -u
R=${1:-r02}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"

# 1. PMC passes (separate runs per counter group, --kernel-trace only) -> json + text summary
bash tools/pmc_pass.sh "$OUT/pmc" > "$OUT/pmc_pass.log" 2>&1
python tools/pmc_to_json.py "$OUT/pmc" "$OUT/${R}_pmc.json" > /dev/null 2> "$OUT/pmc_json.err"
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/${R}_pmc_summary.txt" 2>> "$OUT/pmc_json.err"
cp "$OUT/${R}_pmc.json" "profiles/${R}_pmc.json"

# 2. kernel trace of the bench command with one batch in flight (the per-kernel table) + its bench line
#    (--no-f16: the fp16 / split side runs of the default command would put their kernels into the same trace)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o slots1 -- \
  python bench.py --slots 1 --steps 5 --rounds 1 --host-rounds 0 --warmup 2 --no-f16 --density-rounds 0 > "$OUT/bench_slots1.json" 2> "$OUT/bench_slots1.err"
cp "$OUT"/prof/slots1_kernel_stats.csv "$OUT/${R}_kernel_stats_slots1.csv" 2>/dev/null || \
  find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/${R}_kernel_stats_slots1.csv" \;

# 2b. the same command with THREE batches in flight (the regime the headline is quoted in): kernel trace -> how many kernels are
#     resident over time, per-kernel durations in the mix (tools/overlap_trace.py)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof3" -o slots3 -- \
  python bench.py --slots 3 --steps 20 --rounds 2 --host-rounds 0 --warmup 3 --no-f16 --no-cpu-baseline --density-rounds 0 > "$OUT/bench_slots3_traced.json" 2> "$OUT/bench_slots3_traced.err"
python tools/overlap_trace.py "$(find "$OUT/prof3" -name '*kernel_trace.csv' | head -1)" 0.1 > "$OUT/${R}_overlap_slots3.txt" 2>> "$OUT/bench_slots3_traced.err"

# 3. the default bench command, three times
for i in 1 2 3; do python bench.py > "$OUT/bench_default_$i.json" 2> "$OUT/bench_default_$i.err"; done
cp "$OUT/bench_default_1.json" "$OUT/bench_default.json"

# 4. secondary configurations (configs[2], beam widths, configs[4] fp16, split dtype)
python tools/bench_configs.py > "$OUT/secondary.jsonl" 2> "$OUT/secondary.err"
python tools/bench_configs.py f16 >> "$OUT/secondary.jsonl" 2>> "$OUT/secondary.err"
python tools/bench_configs.py split >> "$OUT/secondary.jsonl" 2>> "$OUT/secondary.err"
python tools/bench_configs.py w2 >> "$OUT/secondary.jsonl" 2>> "$OUT/secondary.err"

# 5. end to end through the host pipeline (synthetic reads x 100k samples -> FASTQ tree): .signal input, fast5 input on the
#    direct path, fast5 through the reference's two passes; greedy and the default beam 30; behind the fp16 engine at batch 4096
python tools/e2e_bench.py 512 0 - fp32 1100 signal > "$OUT/e2e.txt" 2> "$OUT/e2e.err"
python tools/e2e_bench.py 512 0 - fp32 1100 fast5 >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"
python tools/e2e_bench.py 512 0 - fp32 1100 fast5-via-signal >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"
python tools/e2e_bench.py 512 30 - fp32 1100 signal >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"
python tools/e2e_bench.py 512 30 - fp32 1100 fast5 >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"
python tools/e2e_bench.py 2048 0 - fp16 4096 signal >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"
python tools/e2e_bench.py 2048 0 - fp16 4096 fast5 >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"
python tools/e2e_bench.py 2048 0 - fp16-w2 4096 fast5 >> "$OUT/e2e.txt" 2>> "$OUT/e2e.err"

# 5b. the beam-search kernel on its own: engine logits (flat posteriors) and trained-model-like peaked posteriors, kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/beam" -o beam -- python tools/beam_peaked.py 1100 > "$OUT/${R}_beam_kernel.txt" 2> "$OUT/beam.err"
python - "$OUT" >> "$OUT/${R}_beam_kernel.txt" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "beam/**/*kernel_stats.csv"), recursive=True)
for row in list(csv.DictReader(open(f[0])))[:8]:
    print("%-70s calls %4s avg %9.1f us  %5s %%" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY

echo "--- PMC of beam32x2_kernel (beam 30: two windows per wave) and beam64_kernel (beam 50), 8 launches each on peaked and flat posteriors; per-launch means" >> "$OUT/${R}_beam_kernel.txt"
bash tools/pmc_beam.sh "$OUT/pmc_beam" >> "$OUT/${R}_beam_kernel.txt" 2>> "$OUT/beam.err"

# 6. fp16 engine at configs[4], kernel trace
bash tools/f16_profile.sh > "$OUT/f16_kernels.txt" 2> "$OUT/f16_kernels.err"

# 7. dtype fp32-split (round 6): its kernel table with ONE batch in flight (the bench line's extra.f32_split_dtype.roofline uses HIP events;
#    this is the rocprofv3 view of the same launches) -- the split kernels are the MODE 2 instances of gemm_f32_dma_kernel and lstm32s_kernel
BENCH_SLOTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/split" -o split -- python tools/bench_configs.py split > "$OUT/split_bench.jsonl" 2> "$OUT/split.err"
find "$OUT/split" -name '*kernel_stats.csv' -exec cp {} "$OUT/${R}_split_kernel_stats_slots1.csv" \;

# 8. roctx ranges (CHIRON_ROCTX=1): the stages named by the engine itself in a marker trace next to the kernel trace
CHIRON_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d "$OUT/roctx" -o roctx -- \
  python bench.py --slots 1 --steps 3 --rounds 1 --host-rounds 0 --warmup 1 --no-f16 --density-rounds 0 --no-cpu-baseline > "$OUT/roctx_bench.json" 2> "$OUT/roctx.err"
find "$OUT/roctx" -name '*marker_api_stats.csv' -exec cp {} "$OUT/${R}_roctx_marker_stats.csv" \;

for i in 1 2 3; do python -c "
import json,sys
j=json.loads(open('$OUT/bench_default_$i.json').read().strip().splitlines()[-1]); r=j['roofline']
print('default run $i:', j['value'], j['ms_per_step'], r['kernel'], r['frac'], r['traffic'])"; done
cat "$OUT/e2e.txt"
