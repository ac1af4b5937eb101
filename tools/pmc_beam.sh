#!/bin/bash
# PMC passes for the beam-search kernel on its own (tools/beam_peaked.py: peaked and flat posteriors through the decode-only
# entry).  Separate rocprofv3 runs per counter group, --kernel-trace only.   usage: tools/pmc_beam.sh <outdir>
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python tools/beam_peaked.py 1100 > "$OUT/$name.log" 2> "$OUT/$name.err"; }
run pmc_sq  SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run pmc_ins SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python tools/pmc_summary.py "$OUT" | awk '/^chiron::beam(64|32x2)_kernel/{p=1;print;next} /^[^ ]/{p=0} p{print}'
