#!/bin/bash
# PMC passes for the bench workload (one batch in flight). Each counter group is its own rocprofv3 run
# (gpurun refuses --pmc combined with trace domains other than --kernel-trace).
# usage: tools/pmc_pass.sh <outdir> [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- \
    python bench.py --steps 3 --rounds 1 --host-rounds 0 --warmup 1 --slots 1 --no-cpu-baseline --no-f16 --density-rounds 0 > "$OUT/$name.bench.json" 2> "$OUT/$name.err"; }
run pmc_sq  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run pmc_rd  FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_wr  WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
ls "$OUT"
