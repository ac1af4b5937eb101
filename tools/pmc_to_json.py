#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (tools/pmc_pass.sh) into a small JSON that bench.py reads to fill
roofline.traffic: per kernel family, mean per-launch FETCH_SIZE / WRITE_SIZE (KB, as reported) and the
corrected HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE counts 128-byte requests
as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.
usage: pmc_to_json.py <pmc dir> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def family(name):
    n = name.split("(")[0]
    if "gemm_f32_dma_kernel" in n or "gemm_f32_kernel" in n:
        return n.replace("void ", "").replace("chiron::", "").strip()
    if "lstm_kernel<" in n or "lstm32w" in n or "conv1x1_f32_stream_kernel" in n:
        return n.replace("void ", "").replace("chiron::", "").strip()
    if "wino_conv3" in n:
        return n.replace("void ", "").replace("chiron::", "").strip()
    for key in ("pwl_conv_kernel", "lstm16_kernel", "lstm_kernel", "fc_kernel", "beam64_kernel", "greedy_kernel", "beam_kernel", "scan_kernel", "scatter_kernel"):
        if key in n:
            return key
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(src + "/*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            f = family(r["Kernel_Name"])
            if f:
                acc[f][r["Counter_Name"]].append(float(r["Counter_Value"]))
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    out = {"_kernel_sources_sha256_16": bench.kernel_sources_digest(),
           "_note": "mean per launch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction)",
           "_command": "tools/pmc_pass.sh: rocprofv3 --kernel-trace --pmc <group> -- python bench.py --steps 3 --rounds 1 --host-rounds 0 --warmup 1 --slots 1 --no-cpu-baseline --no-f16"}
    for f, c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        rec = {"launches_sampled": len(c.get("FETCH_SIZE", c.get("SQ_WAVE_CYCLES", [])))}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            rec.update({"FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
                        "hbm_bytes": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            rec["mfma_busy_frac_of_all_simds"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        # Issue slots on the CUs the kernel actually holds (a launch of 138 workgroups holds 138 of the 256 CUs): SQ_BUSY_CU_CYCLES is summed
        # over the CUs while they hold a wave of the kernel, four SIMDs each.  SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy.
        # SQ_ACTIVE_INST_VALU counts in units of four cycles the time a SIMD spends on VALU instructions -- the non-MFMA ones at their
        # full length (4 cycles; 16 for exp / rcp) and one issue cycle per MFMA.  In a saturated MFMA stream VALU work of a partner wave
        # adds to the stream's time (tools/ubench/mfma_valu_overlap.hip); in a kernel with idle gaps the two counters may overlap, so their
        # sum is an UPPER bound of the SIMD time that is spoken for (DESIGN 3.9 has the direct measurement for the recurrence).
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"] > 0:
            simd_cycles = 4.0 * m["SQ_BUSY_CU_CYCLES"]
            rec["mfma_busy_frac_of_busy_cus"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            if "SQ_ACTIVE_INST_VALU" in m:
                valu = max(0.0, 4.0 * m["SQ_ACTIVE_INST_VALU"] - m.get("SQ_INSTS_MFMA", 0.0))
                rec["valu_busy_frac_of_busy_cus"] = valu / simd_cycles
                rec["issue_busy_frac_of_busy_cus"] = rec["mfma_busy_frac_of_busy_cus"] + rec["valu_busy_frac_of_busy_cus"]
        if "TCC_HIT_sum" in m:
            rec["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_LDS_BANK_CONFLICT" in m:
            rec["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)
        out[f] = rec
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
