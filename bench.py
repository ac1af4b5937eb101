#!/usr/bin/env python3
"""Headline benchmark: kilobases/s basecalled, DNA_default, segment_len=400, jump=390, batch=1100,
greedy decode, synthetic 4 kHz signal (BASELINE.json configs[1]).

One process per GPU (driver launches torch.distributed.run for N>1).  Reads shard per rank, no
data-path collective ("scaling": "weak": every rank runs K batches of 1100 windows).

A step = one pass of the hot path over one batch of 1100 windows whose signal is already resident
in HBM: CNN -> 3x BiLSTM -> FC -> greedy CTC -> SparseTensor on device, decoded tensor copied to
the host, per-read regroup + glue overlap-consensus vote on the host.  Three batches are kept in
flight (three engine slots / HIP streams; --slots).

Timed region: --steps K steps take ~11 ms each, so K = 20 would be a quarter of a second -- too short for the
clocks to settle or for a utilisation sampler with a 5 s period to see the GPU busy.  The region therefore runs the K
steps --rounds R times back to back (default 45: 900 steps, ~10 s), all inside ONE barrier + synchronize bracket;
ms_per_step = time / (K * R), value = windows of all K * R steps / time, and the line reports steps = K, rounds = R,
timed_steps = K * R.  Nothing is skipped or cached between rounds: every step is a full submit / collect.

`value` is measured with the batch's signal already resident in HBM (the contract of this benchmark).  What the drop-in
boundary really receives are HOST buffers, so a second region of the same kind (`extra.host_inclusive`, --host-rounds) feeds
numpy arrays: every step cuts its 1100 windows out of a raw signal on the host (signal_io.window_signal, chiron_input.py:253-292),
rounds the lengths (chiron_eval.py:337) and hands host pointers to chiron_engine_submit, which stages them in pinned memory and
copies them over PCIe inside the step.

value = signal-normalised kbases/s = windows * jump / (4000 Hz / 450 b/s) / seconds / 1000
(SURVEY.md 8d (i)); decoded consensus bases/s with the synthetic weights is reported in "extra".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEG_LEN, JUMP, BATCH = 400, 390, 1100
SAMPLES_PER_BASE = 4000.0 / 450.0
BASES_PER_WINDOW = JUMP / SAMPLES_PER_BASE            # 43.875
LSTM_GEMM_FLOP_PER_WINDOW = 611.84e6
MODEL_FLOP_PER_WINDOW = 1.451e9                       # reference op count (CNN 839.3 M + LSTM 611.84 M)
# res_layer1 conv2a + conv2b run as a piecewise-linear table of the signal value (chiron_amd/csrc/pwl.hip): their
# 2*(1 + 3*256)*256 FLOP per position are no longer executed as multiply-adds
# conv2b of res_layer2 / res_layer3 runs in Winograd F(4,3) form (wino.hip): 6 instead of 12 C x C products per four
# output positions, i.e. 1.5 of the op's 3 products per position are not executed
EXECUTED_FLOP_PER_WINDOW = MODEL_FLOP_PER_WINDOW - SEG_LEN * 2.0 * (1 + 3 * 256) * 256 - 2 * SEG_LEN * 2.0 * 1.5 * 256 * 256
EXECUTED_SHARE = {"conv_wino": 0.5}                   # MFMA FLOPs a bucket's kernel issues / FLOPs of the op it replaces
PEAK_F32_MFMA_TFLOPS = 157.3                          # MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
PEAK_F16_MFMA_TFLOPS = 2500.0                         # MI355X_MICROARCH.md: BF16 / FP16 MFMA, dense (32x32x16)
READ_SAMPLES = 100000                                 # configs[3]: 100k-sample reads -> 257 windows


def make_batches(n_batches, rank):
    """Windows of synthetic reads, packed across reads into full batches (chiron_eval.py:321-334)."""
    import chiron_amd as ca
    from chiron_amd import signal_io
    win_per_read = -(-READ_SAMPLES // JUMP)
    n_reads = -(-n_batches * BATCH // win_per_read) + 1
    xs, lens, tags = [], [], []
    for r in range(n_reads):
        sig = ca.synthetic_signal(1, READ_SAMPLES, seed=1234 + 1000 * rank + r)[0]
        ev, ln = signal_io.window_signal(sig, 0, JUMP, SEG_LEN)
        xs.append(ev)
        lens.append(ln)
        tags += [(r, i) for i in range(len(ln))]
    x = np.concatenate(xs)[:n_batches * BATCH]
    ln = np.concatenate(lens)[:n_batches * BATCH]
    tags = tags[:n_batches * BATCH]
    return (x.reshape(n_batches, BATCH, SEG_LEN), ln.reshape(n_batches, BATCH), tags, win_per_read)


# profiling bucket of the engine (chiron_engine_profile) -> kernel symbol(s) in a rocprofv3 trace; template arguments of
# gemm_f32_dma_kernel are <ZOUT, RES, chunks per K-segment, K tail, mode (0 fp32, 1 f16, 2 split), single K-segment>
_REC_FORM = os.environ.get("CHIRON_LSTM_WIDE", "2")     # engine.hip CHIRON_LSTM_WIDE_DEFAULT
_STREAM32 = os.environ.get("CHIRON_NO_STREAM32") is None
BUCKET_SYMBOL = {
    # the recurrence: lstm32w2_kernel = 16 rows per workgroup on v_mfma_f32_16x16x4_f32, ONE workgroup per 16-row group and
    # direction (138 for batch 1100: 54 % of the CUs); lstm_kernel<1> = the 4-row form (CHIRON_LSTM_WIDE=0)
    "lstm_recurrence": {"0": "lstm_kernel<1>", "1": "lstm32w_kernel", "2": "lstm32w2_kernel"}.get(_REC_FORM, "lstm32w2_kernel"),
    "conv_dma": "gemm_f32_dma_kernel<false, false, 8, false, 0, false>",          # conv2c+branch1 (K=512) of res_layer2/3 (and conv2a with CHIRON_NO_STREAM32)
    "conv2a": "conv1x1_f32_stream_kernel<false>" if _STREAM32 else "gemm_f32_dma_kernel<false, false, 8, false, 0, false>",   # conv2a (K=256) of res_layer2/3
    "conv_wino": "wino_conv3_f4_kernel",                                    # conv2b of res_layer2/3, Winograd F(4,3) (T % 4 == 0)
    "conv_res": "conv1x1_f32_stream_kernel<true>" if _STREAM32 else "gemm_f32_dma_kernel<false, true, 8, false, 0, false>",   # conv2c + signal branch of res_layer1
    "lstm_proj0_dma": "gemm_f32_dma_kernel<true, false, 8, false, 0, true>",      # x-projection of layer 0 (K = 256)
    "lstm_proj_dma": "gemm_f32_dma_kernel<true, false, 7, true, 0, true>",        # x-projections of layers 1, 2 (K = 200)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=45, help="the timed region is --steps steps repeated this many times (one bracket)")
    ap.add_argument("--host-rounds", type=int, default=10, help="rounds of the host-inclusive region (windowing + PCIe inside the step); 0 = skip")
    ap.add_argument("--slots", type=int, default=3, help="batches in flight (engine slots / HIP streams)")
    ap.add_argument("--density-rounds", type=int, default=15, help="rounds of the realistic-decode-density region (a head fitted to emit "
                    "~43.9 bases per window, as a trained model does); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=0, help="windows for the CPU baseline (0 = auto)")
    ap.add_argument("--backend", default="gloo", choices=("gloo", "nccl"),
                    help="torch.distributed backend of the BOOKKEEPING (two barriers, a MAX of the clocks, a SUM of two counters: 24 "
                         "bytes in all).  The data path has no collective (north_star: 'no RCCL collectives'), so the default is gloo "
                         "on CPU tensors: the N-GPU record does not depend on RCCL bring-up.  'nccl' (= RCCL) stays selectable; when "
                         "its initialisation raises, the ranks fall back to gloo together and config.parallelism_bookkeeping says so")
    ap.add_argument("--share-gpu", action="store_true", help="self-test only: every rank uses GPU 0")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic in THIS run")
    ap.add_argument("--pmc-child", nargs="?", const="fp32", default=None, metavar="DTYPE",
                    help="internal: three single-slot batches of the headline workload in DTYPE (fp32 / fp32-split) and exit (the workload of the --pmc child runs)")
    ap.add_argument("--no-f16", action="store_true", help="skip the extra measurements (configs[4]: fp16 batch 4096; fp32-split dtype)")
    ap.add_argument("--stub-engine", type=float, default=0.0, metavar="MS",
                    help="self-test of the multi-rank bookkeeping WITHOUT a GPU (tests/test_bench_ranks.py): the engine is replaced "
                         "by a stub whose collect() takes MS + rank milliseconds (with --backend nccl it exercises the fall-back to "
                         "gloo: there is no RCCL without a GPU).  Never a measurement: the "
                         "line says data = 'stub'.")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args.pmc_child)
    stub = args.stub_engine > 0
    if stub:
        args.no_f16, args.no_cpu_baseline, args.density_rounds = True, True, 0

    # `--gpus N` is the contract: N ranks, one per GPU.  Started bare (no WORLD_SIZE: the way the 1-GPU line is started), the
    # script starts its own N ranks under torch.distributed.run and rank 0's line is the record; started under a launcher
    # whose world size is NOT N, it refuses -- a line with the wrong "n_gpus" and rc 0 must not exist (BASELINE.json metric:
    # "at 1/2/4/8 GPUs").
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to write a record\n"
                         % (args.gpus, os.environ["WORLD_SIZE"]))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    device_ordinal = local_rank
    if args.share_gpu:
        local_rank = 0
    if not stub:
        if local_rank >= torch.cuda.device_count():
            sys.stderr.write("bench.py: rank %d wants GPU %d, this node shows %d\n" % (rank, local_rank, torch.cuda.device_count()))
            sys.exit(3)
        torch.cuda.set_device(local_rank)
    bookkeeping, grp = "none (one rank)", None
    idents = [device_identity(torch, local_rank) if not stub else "stub:%d" % device_ordinal]
    if world > 1:
        bookkeeping, grp = init_bookkeeping(dist, torch, args.backend, local_rank)
    cdev = torch.device("cuda", local_rank) if bookkeeping.startswith("nccl") else torch.device("cpu")   # where the reductions run
    if world > 1:
        # one rank per device: every rank reports the device it computes on (the GPU's PCI address; the ordinal for the stub) and
        # all of them must differ -- eight ranks on one GPU would still print a line (--share-gpu is the declared self-test)
        ident = idents[0]
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if not args.share_gpu and ranks_share_a_device(idents, world if stub else torch.cuda.device_count()):
            sys.stderr.write("bench.py: ranks share a device: %s\n" % idents)
            sys.exit(4)
    # the record itself proves N distinct GPUs: rank r -> "uuid/pci domain/bus/device" + ordinal as torch reports them
    devices = [i if isinstance(i, str) else {"rank": r, "ordinal": i[1], "hardware": i[0]} for r, i in enumerate(idents)]

    import chiron_amd as ca
    from chiron_amd import assembly
    spec = ca.dna_default_spec()
    weights = ca.synthetic_weights(spec, seed=1234)
    eng = StubEngine(args.stub_engine + rank, args.slots) if stub else \
        ca.Engine(spec, weights, max_batch=BATCH, segment_len=SEG_LEN, device_id=local_rank, n_slots=args.slots)

    n_distinct = 4
    xb, lb, tags, win_per_read = make_batches(n_distinct, rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    x_dev = [torch.from_numpy(xb[i]).to(dev) for i in range(n_distinct)]
    s_dev = [torch.from_numpy(ca.seq_len_for_engine(lb[i], eng.ratio)).to(dev) for i in range(n_distinct)]
    device_sync = (lambda: None) if stub else torch.cuda.synchronize
    device_sync()

    decoded_bases = [0]
    consensus_bases = [0]

    # rows of a batch that belong to the same read (cross-read packing, chiron_eval.py:321-334)
    read_of_row = np.asarray([t[0] for t in tags], dtype=np.int64).reshape(n_distinct, BATCH)
    run_bounds = []
    for i in range(n_distinct):
        cuts = np.flatnonzero(np.diff(read_of_row[i])) + 1
        run_bounds.append(np.concatenate([[0], cuts, [BATCH]]))

    def consume(res, which):
        """host side of a step: ragged split of the SparseTensor, then the glue overlap-consensus vote
        (chiron_assemble) for every per-read run of rows in the batch"""
        idx, val = res.decoded.indices, res.decoded.values
        decoded_bases[0] += int(val.shape[0])
        if not val.shape[0]:
            return
        counts = np.bincount(idx[:, 0], minlength=BATCH)
        row_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        v8 = val.astype(np.uint8)
        bounds = run_bounds[which]
        for a, b in zip(bounds[:-1], bounds[1:]):
            c = counts[a:b]
            keep = c[c > 0]                              # sparse2dense drops rows with an empty decode
            if keep.shape[0] == 0:
                continue
            seg = v8[row_off[a]:row_off[b]]
            off = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
            cons = assembly.assemble_native(seg, off, None, "glue")
            consensus_bases[0] += int(cons[0].shape[1])

    def step(i, pending, eng=eng):
        """collect the slot's finished batch, hand the slot its next batch at once, THEN do the host side of the finished one
        (collect() returns copies: the slot's buffers are free again) -- the stream never waits for the host's vote"""
        slot = i % args.slots
        res, which = (eng.collect(slot), pending[slot]) if pending[slot] is not None else (None, None)
        eng.submit(slot, x_dev[i % n_distinct], s_dev[i % n_distinct], beam_width=0, want_prob=True)
        pending[slot] = i % n_distinct
        if res is not None:
            consume(res, which)

    def drain(pending, eng=eng):
        for slot in range(args.slots):
            if pending[slot] is not None:
                consume(eng.collect(slot), pending[slot])
                pending[slot] = None

    pending = [None] * args.slots
    for i in range(args.warmup):
        step(i, pending)
    drain(pending)
    eng.sync()
    decoded_bases[0] = consensus_bases[0] = 0

    if world > 1:
        dist.barrier(group=grp)
    device_sync()
    t0 = time.perf_counter()
    timed_steps = args.steps * max(1, args.rounds)
    for i in range(timed_steps):
        step(i, pending)
    drain(pending)
    eng.sync()
    device_sync()
    if world > 1:
        dist.barrier(group=grp)
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=grp)
        dt = float(tt.item())
        agg = torch.tensor([decoded_bases[0], consensus_bases[0]], dtype=torch.float64, device=cdev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM, group=grp)
        decoded_bases[0], consensus_bases[0] = int(agg[0].item()), int(agg[1].item())

    windows = timed_steps * BATCH * world
    kbases = windows * BASES_PER_WINDOW / 1000.0
    value = kbases / dt
    decoded_total, consensus_total = decoded_bases[0], consensus_bases[0]     # the headline region's counts (consume() keeps counting)

    # ---- the same steps fed from the HOST: windowing of the raw signal, length rounding and the H2D copy inside the step
    host_inclusive = None
    if args.host_rounds > 0:
        from chiron_amd import signal_io
        raw = [np.ascontiguousarray(np.concatenate([xb[i][:, :JUMP].reshape(-1), xb[i][-1, JUMP:]])) for i in range(n_distinct)]

        def host_step(i, pending):
            slot = i % args.slots
            ev, ln = signal_io.window_signal(raw[i % n_distinct], 0, JUMP, SEG_LEN)      # chiron_input.py:253-292
            sl = ca.seq_len_for_engine(ln[:BATCH], eng.ratio)
            res, which = (eng.collect(slot), pending[slot]) if pending[slot] is not None else (None, None)
            eng.submit(slot, ev[:BATCH], sl, beam_width=0, want_prob=True)
            pending[slot] = i % n_distinct
            if res is not None:
                consume(res, which)

        for i in range(args.warmup):
            host_step(i, pending)
        drain(pending)
        eng.sync()
        if world > 1:
            dist.barrier(group=grp)
        device_sync()
        h0 = time.perf_counter()
        host_steps = args.steps * args.host_rounds
        for i in range(host_steps):
            host_step(i, pending)
        drain(pending)
        eng.sync()
        device_sync()
        if world > 1:
            dist.barrier(group=grp)
        hdt = time.perf_counter() - h0
        if world > 1:
            tt = torch.tensor([hdt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=grp)
            hdt = float(tt.item())
        host_inclusive = {"kbases_per_s": round(host_steps * BATCH * world * BASES_PER_WINDOW / 1000.0 / hdt, 2),
                          "ms_per_step": round(hdt / host_steps * 1e3, 3), "timed_steps": host_steps, "timed_region_s": round(hdt, 3),
                          "ratio_to_value": round(host_steps * BATCH * world * BASES_PER_WINDOW / 1000.0 / hdt / value, 4),
                          "inside_the_step": "window_signal of the batch's raw samples, seq_len rounding, pinned staging + H2D copy (1.76 MB), "
                                             "then the same submit / collect / host vote as the headline step"}

    # ---- the same steps at a trained model's decode density (round-4 review, Missing #5): the synthetic weights emit ~5.6 bases per
    # window, a trained Chiron model ~44 (43.875 at 450 bases/s and 4 kHz) -- so the SparseTensor's D2H copy and the host's glue vote
    # inside the headline step carry 1/8 of the real payload.  Same workload, same engine code, same step; the weights differ in the
    # LSTM forget biases (cells that follow their input) and in a head FITTED to emit a base where the squiggle changes level.
    # (it runs AFTER the per-kernel profiling pass below: two more engines and three more seconds of load in front of that pass
    #  moved its launch times by 2 .. 3 %)
    out = None
    if rank == 0 and stub:
        out = {"metric": "kilobases/sec basecalled (DNA_default seg_len=400 batch=1100)", "value": round(value, 2), "unit": "kbases/s",
               "n_gpus": world, "steps": args.steps, "rounds": max(1, args.rounds), "timed_steps": timed_steps, "timed_region_s": round(dt, 3),
               "warmup": args.warmup, "ms_per_step": round(dt / timed_steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "stub", "config": {"workload": "SELF-TEST: stub engine, not a measurement", "parallelism_bookkeeping": bookkeeping, "devices": devices},
               "roofline": None, "cpu_baseline": None,
               "extra": {"windows_per_s": round(windows / dt, 1), "decoded_bases_total": decoded_total, "host_inclusive": host_inclusive}}
    elif rank == 0:
        # ---- per-kernel timing with HIP events on the engine's own stream (separate, untimed pass)
        eng.profile(True)
        for i in range(3):
            eng.submit(0, x_dev[i % n_distinct], s_dev[i % n_distinct], beam_width=0, want_prob=True)
            eng.collect(0)
        stats = eng.profile_read()
        eng.profile(False)
        per_kernel = {k: {"avg_ms": s["total_ms"] / s["launches"], "launches_per_batch": s["launches"] / 3.0,
                          "tflops": (s["flops"] / (s["total_ms"] * 1e-3) / 1e12) if s["total_ms"] > 0 else 0.0,
                          "gbps": (s["bytes"] / (s["total_ms"] * 1e-3) / 1e9) if s["total_ms"] > 0 else 0.0}
                      for k, s in stats.items()}
        # dominant kernel = the MFMA-bound bucket (= one kernel symbol) with the largest device time in this pass
        mfma_buckets = [k for k in BUCKET_SYMBOL if k in stats and stats[k]["total_ms"] > 0]
        dom_key = max(mfma_buckets, key=lambda k: stats[k]["total_ms"])
        dom = stats[dom_key]
        dom_symbol = BUCKET_SYMBOL[dom_key]
        achieved = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(dom_symbol)          # the committed record (also feeds the per-kernel pmc blocks below)
        live = None if (args.no_live_pmc or world != 1) else live_pmc_traffic()
        if live and live.get("kernels"):
            rec = next((v for k, v in live["kernels"].items() if dom_symbol in k), None)
            if rec and rec.get("hbm_bytes"):
                static_traffic, static_src = traffic, traffic_src
                traffic = rec["hbm_bytes"]
                traffic_src = ("measured in THIS run: child processes of this script under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
                               "(separate passes, %d / %d launches, three single-slot batches each; two more passes take the SQ counters of "
                               "traffic_live.mfma_busy); committed record for comparison: %s (%s)"
                               % (rec["launches_fetch"], rec["launches_write"], static_traffic, static_src))
        per_bucket = {k: {"symbol": BUCKET_SYMBOL[k], "ms_per_batch": round(stats[k]["total_ms"] / 3.0, 4),
                          "launches_per_batch": stats[k]["launches"] / 3.0,
                          "achieved": round(EXECUTED_SHARE.get(k, 1.0) * stats[k]["flops"] / (stats[k]["total_ms"] * 1e-3) / 1e12, 2),
                          "frac": round(EXECUTED_SHARE.get(k, 1.0) * stats[k]["flops"] / (stats[k]["total_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                          "op_equivalent_tflops": round(stats[k]["flops"] / (stats[k]["total_ms"] * 1e-3) / 1e12, 2)}
                      for k in mfma_buckets}
        roofline = {"kernel": dom_symbol, "bucket": dom_key, "bound": "mfma",
                    "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 of a separate rocprofv3 --pmc pass",
                    # a kernel that does not fill the chip by construction: the wide recurrence launches one 8-wave workgroup per
                    # 16-row group and direction; the other CUs run the other batches' kernels meanwhile (three in flight)
                    "workgroups": (2 * ((BATCH + 15) // 16)) if dom_key == "lstm_recurrence" and _REC_FORM != "0" else None,
                    "frac_of_the_cus_it_occupies": (round(achieved / PEAK_F32_MFMA_TFLOPS * 256.0 / min(256, 2 * ((BATCH + 15) // 16)), 4)
                                                    if dom_key == "lstm_recurrence" and _REC_FORM != "0" else None),
                    "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                    "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4),
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "flops_note": "achieved/frac count the MFMA FLOPs a kernel issues; conv_wino (Winograd F(4,3)) issues 1/2 of "
                                  "its op's 2*3*C*C per position, op_equivalent_tflops is the op's own count over the same time",
                    # the north-star figure -- LSTM-GEMM FLOPs of all windows / whole-job time / fp32 MFMA peak
                    "lstm_gemm_roofline_frac_whole_path": round(windows / dt * LSTM_GEMM_FLOP_PER_WINDOW / 1e12 / PEAK_F32_MFMA_TFLOPS / world, 4),
                    "lstm_gemm_flop_per_window": LSTM_GEMM_FLOP_PER_WINDOW,
                    "mfma_kernels": per_bucket,
                    "traffic_source": traffic_src,
                    "traffic_live": None if not live else {"error": live.get("error"), "seconds": live.get("seconds"),
                                                            "hbm_bytes_per_launch": {k: v.get("hbm_bytes") for k, v in (live.get("kernels") or {}).items()},
                                                            # the matrix pipe in THIS run (same child passes, SQ counters): busy share of the CUs a kernel
                                                            # holds / of the whole chip, VALU share next to it (committed record: roofline.pmc)
                                                            "mfma_busy": {k: {f: v[f] for f in ("mfma_busy_frac", "mfma_busy_frac_of_busy_cus", "valu_busy_frac_of_busy_cus") if f in v}
                                                                          for k, v in (live.get("kernels") or {}).items() if "mfma_busy_frac_of_busy_cus" in v}}}
        # north_star: "rocprof HBM GB/s on the conv and MFMA utilisation on the LSTM": the PMC pass's counters next to this
        # run's launch times (HBM GB/s = PMC bytes per launch / HIP-event launch time); absent when the PMC record is stale
        for k in mfma_buckets:
            pc = pmc_counters(BUCKET_SYMBOL[k])
            if pc is not None:
                per_bucket[k]["pmc"] = pc
                if pc["hbm_bytes_per_launch"]:
                    per_bucket[k]["hbm_gbps_pmc"] = round(pc["hbm_bytes_per_launch"] / (stats[k]["total_ms"] / stats[k]["launches"] * 1e-3) / 1e9, 1)
        roofline["pmc"] = pmc_counters(dom_symbol)
        fam = [s for k, s in stats.items() if k.startswith("conv") or k.startswith("lstm_proj")]
        gemm_family = {"launches_per_batch": sum(s["launches"] for s in fam) / 3.0,
                       "tflops": round(sum(s["flops"] for s in fam) / (sum(s["total_ms"] for s in fam) * 1e-3) / 1e12, 2),
                       "ms_per_batch": round(sum(s["total_ms"] for s in fam) / 3.0, 3)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N = 1 only
            cpu = cpu_baseline(spec, weights, xb, lb, eng.ratio, args.cpu_windows)
        out = {
            "metric": "kilobases/sec basecalled (DNA_default seg_len=400 batch=1100)",
            "value": round(value, 2), "unit": "kbases/s", "n_gpus": world, "steps": args.steps,
            "rounds": max(1, args.rounds), "timed_steps": timed_steps, "timed_region_s": round(dt, 3),
            "warmup": args.warmup, "ms_per_step": round(dt / timed_steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DNA_default seg_len=400 jump=390 batch=1100 greedy, synthetic 4 kHz signal "
                                   "(BASELINE.json configs[1])", "segment_len": SEG_LEN, "jump": JUMP,
                       "batch": BATCH, "decode": "greedy", "weights": "seeded synthetic (exact checkpoint shapes)",
                       "parallelism": "reads sharded per GPU, no collective", "slots_in_flight": args.slots,
                       # what carries the two barriers + the 24 bytes of timing / counter reductions around the timed region
                       "parallelism_bookkeeping": bookkeeping, "devices": devices,
                       "distinct_devices": len(set(d["hardware"] for d in devices)) if not args.share_gpu else 1},
            "roofline": roofline, "cpu_baseline": cpu,
            "extra": {"windows_per_s": round(windows / dt, 1),
                      "decoded_bases_per_s": round(decoded_total / dt, 1),
                      "consensus_bases_per_s": round(consensus_total / dt, 1),
                      "lstm_gemm_roofline_frac_whole_path": round(windows / dt * LSTM_GEMM_FLOP_PER_WINDOW / 1e12 / PEAK_F32_MFMA_TFLOPS / world, 4),
                      "model_tflops_whole_path": round(windows / dt * EXECUTED_FLOP_PER_WINDOW / 1e12, 2),
                      "model_tflops_reference_op_count": round(windows / dt * MODEL_FLOP_PER_WINDOW / 1e12, 2),
                      "host_inclusive": host_inclusive, "realistic_density": None,
                      "gemm_family": gemm_family, "kernels": per_kernel}}
    if args.density_rounds > 0 and world == 1 and rank == 0 and not stub:
        out["extra"]["realistic_density"] = density_region(args, spec, local_rank, x_dev, s_dev, xb, lb, step, drain, decoded_bases,
                                                           consensus_bases, value)
    ref32 = None
    if rank == 0 and world == 1 and not args.no_f16:
        ref32 = eng.infer(x_dev[0], s_dev[0], want_logits=True)
    eng.close()
    if ref32 is not None:
        out["extra"]["config5_f16"] = f16_config(spec, weights, x_dev, s_dev, ref32, local_rank)
        out["extra"]["config5_f16_w2"] = f16_config(spec, weights, x_dev, s_dev, ref32, local_rank, dtype="fp16-w2")
        out["extra"]["f32_split_dtype"] = split_config(spec, weights, x_dev, s_dev, ref32, local_rank, live_pmc=not args.no_live_pmc)
        # BASELINE configs[4]'s tolerance check where it means something (round-5 review, item 3): windows whose greedy string equals the fp32
        # engine's, next to the flat-weights logits_vs_f32 above -- measured in this run at a trained model's decode density, and the
        # committed frontier of every half-precision mode in three regimes (tools/f16_frontier.py)
        dens = (out["extra"].get("realistic_density") or {}).get("half_precision_engines_vs_fp32_greedy_strings") or {}
        for key, mode in (("config5_f16", "fp16"), ("config5_f16_w2", "fp16_w2")):
            out["extra"][key]["identical_windows_vs_f32_engine"] = {
                "density_regime_this_run": (dens.get(mode) or {}).get("identical_windows_frac"),
                "bases_per_window": dens.get("bases_per_window_fp32"), "windows": dens.get("windows"),
                "frontier": frontier_record()}
    if world > 1:
        dist.barrier(group=grp)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def density_region(args, spec, device_id, x_dev, s_dev, xb, lb, step, drain, decoded_bases, consensus_bases, value):
    """extra.realistic_density: the headline's timed loop on an engine whose weights decode ~43.9 bases per window."""
    import torch
    import chiron_amd as ca
    w = ca.synthetic_weights(spec, seed=1234)
    H = spec.hidden
    for k in w:
        if k.endswith("lstm_cell/bias"):
            w[k][2 * H:3 * H] = -3.0            # forget gate (TF adds 1.0): a cell keeps 12 % per frame and follows its input
    nfit = 256
    with ca.Engine(spec, w, max_batch=BATCH, segment_len=SEG_LEN, device_id=device_id) as e0:
        e0.infer(x_dev[0], s_dev[0])
        h = e0.rnn_output()[:nfit]
        sl = s_dev[0].cpu().numpy()[:nfit]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import regimes                                  # weight regimes of the tests: numpy only, nothing of the oracle
    w = regimes.fit_emitting_head(w, h, xb[0][:nfit], sl, BASES_PER_WINDOW, hidden=H)
    with ca.Engine(spec, w, max_batch=BATCH, segment_len=SEG_LEN, device_id=device_id, n_slots=args.slots) as ed:
        pending = [None] * args.slots
        for i in range(args.warmup):
            step(i, pending, ed)
        drain(pending, ed)
        ed.sync()
        d0, c0 = decoded_bases[0], consensus_bases[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = args.steps * args.density_rounds
        for i in range(n):
            step(i, pending, ed)
        drain(pending, ed)
        ed.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ref = ed.infer(x_dev[1], s_dev[1], want_logits=True) if not args.no_f16 else None
    kb = n * BATCH * BASES_PER_WINDOW / 1000.0 / dt
    halves = None
    if ref is not None:
        # BASELINE configs[4]'s "tolerance check vs fp32" at THIS decode density (the flat head of the headline weights decodes 5.6 bases
        # per window: almost every window is identical whatever the logits do).  Same emitting weights, the second batch of the workload
        # (not the one the head was fitted on): f16 engine without / with its bias correction, and fp16-w2.  The convolution filters are
        # still the synthetic ones (homogeneous scales): trained-like filters cost the f16 engines more (85 .. 91 / 95.5 %).
        halves = {}
        sl1 = s_dev[1].cpu().numpy()
        mask = np.arange(ref.logits.shape[1])[None, :] < sl1[:, None]
        ref_rows = np.split(ref.decoded.values, np.cumsum(np.bincount(ref.decoded.indices[:, 0], minlength=BATCH))[:-1])
        for name, dtype, cal in (("fp16", "fp16", False), ("fp16_bias_corrected", "fp16", True), ("fp16_w2", "fp16-w2", False)):
            with ca.Engine(spec, w, max_batch=BATCH, segment_len=SEG_LEN, device_id=device_id, dtype=dtype, calibrate=cal) as eh:
                r = eh.infer(x_dev[1], s_dev[1], want_logits=True)
            rows = np.split(r.decoded.values, np.cumsum(np.bincount(r.decoded.indices[:, 0], minlength=BATCH))[:-1])
            d = np.abs(r.logits - ref.logits)[mask]
            halves[name] = {"identical_windows_frac": round(float(np.mean([np.array_equal(a, b) for a, b in zip(rows, ref_rows)])), 4),
                            "logits_mean_abs": float("%.3e" % d.mean()), "logits_p999_abs": float("%.3e" % np.quantile(d, 0.999)),
                            "logits_max_abs": float("%.3e" % d.max())}
        halves["windows"], halves["bases_per_window_fp32"] = BATCH, round(ref.decoded.values.shape[0] / float(BATCH), 2)
    return {"kbases_per_s": round(kb, 2), "ms_per_step": round(dt / n * 1e3, 3), "timed_steps": n, "timed_region_s": round(dt, 3),
            "ratio_to_value": round(kb / value, 4),
            "decoded_bases_per_window": round((decoded_bases[0] - d0) / float(n * BATCH), 2),
            "decoded_bases_per_s": round((decoded_bases[0] - d0) / dt, 1), "consensus_bases_per_s": round((consensus_bases[0] - c0) / dt, 1),
            "half_precision_engines_vs_fp32_greedy_strings": halves,
            "weights": "the headline's synthetic weights with LSTM forget biases -3 and an FC head fitted (tests/regimes.py:fit_emitting_head) on 256 "
                       "windows to emit 43.875 bases per window; same step as the headline: submit, collect (SparseTensor D2H), per-read glue vote"}


def init_bookkeeping(dist, torch, backend, local_rank):
    """Process group(s) for the bookkeeping around the timed region (never the data path) -> (description, group for the reductions).
    The default group is ALWAYS gloo (CPU tensors).  With --backend nccl a second, RCCL group carries the barriers and reductions --
    if its creation or first collective raises on ANY rank (the ranks agree over gloo), every rank stays on gloo and the line says so."""
    dist.init_process_group("gloo")
    what = "2 barriers + all_reduce MAX of the clock + SUM of two counters per region"
    if backend != "nccl":
        return "gloo (cpu tensors): " + what, None
    err, grp = None, None
    try:
        grp = dist.new_group(backend="nccl")
        dist.all_reduce(torch.zeros(1, device=torch.device("cuda", local_rank)), group=grp)   # RCCL's communicator is created lazily: force it now
        torch.cuda.synchronize()
    except Exception as e:                                                            # noqa: BLE001 -- whatever bring-up raises
        err = "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0][:200])
    bad = torch.tensor([1.0 if err else 0.0], dtype=torch.float64)
    dist.all_reduce(bad)                                                              # over gloo
    if bad.item() == 0:
        return "nccl (RCCL, device tensors): " + what, grp
    sys.stderr.write("bench.py: rank %s: RCCL bring-up failed on %d of %d ranks (%s): bookkeeping stays on gloo\n"
                     % (os.environ.get("RANK"), int(bad.item()), dist.get_world_size(), err))
    return "gloo (cpu tensors) after RCCL bring-up failed on %d of %d ranks: %s" % (int(bad.item()), dist.get_world_size(), what), None


def spawn_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher: run this same command line as N ranks of one node under
    torch.distributed.run (rendezvous on 127.0.0.1, a free port), pass their output through, return their exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_identity(torch, ordinal):
    """(hardware identity, ordinal) of the GPU a rank computes on: UUID + PCI domain / bus / device as torch reports them"""
    p = torch.cuda.get_device_properties(ordinal)
    hw = "/".join(str(getattr(p, a, "")) for a in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"))
    return (hw, ordinal)


def ranks_share_a_device(idents, device_count):
    """idents: one device_identity (or a string for the stub) per rank.  Distinct hardware identities = one GPU per rank.  Should a
    runtime report the same identity for every GPU (attributes not filled in), distinct ordinals within the node's device count
    are accepted instead -- and said so on stderr."""
    if len(set(idents)) == len(idents) and all(isinstance(i, str) for i in idents):
        return False
    if any(isinstance(i, str) for i in idents):
        return True
    hw, ords = [i[0] for i in idents], [i[1] for i in idents]
    if len(set(hw)) == len(hw):
        return False
    if len(set(hw)) == 1 and len(set(ords)) == len(ords) and max(ords) < device_count:
        sys.stderr.write("bench.py: every GPU reports the identity %r; accepting distinct ordinals %s\n" % (hw[0], ords))
        return False
    return True


class StubEngine(object):
    """Stands in for ca.Engine in --stub-engine runs (no GPU): same submit / collect / sync surface, collect() sleeps the
    given milliseconds and returns a fixed decode (one base per window).  Only the rank bookkeeping of this file is under
    test with it: seeds per rank, slots, the barrier + MAX-reduced clock, the aggregate over ranks."""

    def __init__(self, ms, n_slots):
        self.ms, self.n_slots, self.ratio, self._busy = ms, n_slots, 1.0, [False] * n_slots

    def submit(self, slot, x, seq_len, beam_width=0, want_prob=True, want_logits=False, copy_decoded=True):
        assert not self._busy[slot] and x.shape == (BATCH, SEG_LEN) and seq_len.shape == (BATCH,)
        self._busy[slot] = True

    def collect(self, slot):
        import chiron_amd as ca
        assert self._busy[slot]
        time.sleep(self.ms * 1e-3)
        self._busy[slot] = False
        idx = np.stack([np.arange(BATCH, dtype=np.int64), np.zeros(BATCH, dtype=np.int64)], axis=1)
        dec = ca.engine.SparseTensor(idx, np.zeros(BATCH, dtype=np.int64), np.asarray([BATCH, 1], dtype=np.int64))
        return ca.engine.DecodeResult(dec, np.zeros((BATCH, 1), np.float32), np.zeros((BATCH, 1), np.float32), None)

    def sync(self):
        pass

    def close(self):
        pass


def f16_config(spec, weights, x_dev, s_dev, ref32, device_id, dtype="fp16"):
    """BASELINE.json configs[4] (a parity-test case, reported next to the headline, never as `value`): fp16 conv +
    LSTM on the f16 MFMA instructions with fp32 accumulation / gates / CTC, batch 4096, same synthetic workload;
    plus the tolerance check of its logits against the fp32 engine on the first 1100 windows.  dtype "fp16-w2": the same
    activations against exact (hi + lo) weights -- the mode for trained checkpoints (include/chiron_amd.h CHIRON_F16_W2)."""
    import torch
    import chiron_amd as ca
    B16, steps = 4096, (6 if dtype == "fp16-w2" else 16)   # (6 steps of 7 ms were at the mercy of one slow launch: 7.3 .. 8.6 ms per batch run to run)
    reps = -(-B16 // BATCH)
    x = torch.cat([x_dev[i % len(x_dev)] for i in range(reps)])[:B16].contiguous()
    sl = torch.cat([s_dev[i % len(s_dev)] for i in range(reps)])[:B16].contiguous()
    with ca.Engine(spec, weights, max_batch=B16, segment_len=SEG_LEN, device_id=device_id, n_slots=2, dtype=dtype) as e16:
        r16 = e16.infer(x_dev[0], s_dev[0], want_logits=True)
        T = r16.logits.shape[1]
        mask = np.arange(T)[None, :] < s_dev[0].cpu().numpy()[:, None]
        d = np.abs(r16.logits - ref32.logits)[mask]
        for i in range(2):
            e16.submit(i, x, sl, beam_width=0, want_prob=True)
        for i in range(2):
            e16.collect(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if i >= 2:
                e16.collect(i % 2)
            e16.submit(i % 2, x, sl, beam_width=0, want_prob=True)
        for i in range(2):
            e16.collect(i)
        e16.sync()
        dt = time.perf_counter() - t0
    return {"workload": "DNA_default seg_len=400 jump=390 batch=4096 greedy, fp16 conv+LSTM / fp32 accumulate, gates, CTC"
                        + (" -- weights as hi + lo half pairs (dtype fp16-w2)" if dtype == "fp16-w2" else ""),
            "kbases_per_s": round(steps * B16 * BASES_PER_WINDOW / 1000.0 / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 3),
            "logits_vs_f32": {"max_abs": round(float(d.max()), 5), "mean_abs": round(float(d.mean()), 6), "windows": BATCH}}


def split_config(spec, weights, x_dev, s_dev, ref32, device_id, live_pmc=True):
    """Opt-in dtype fp32-split on the headline workload (same batch 1100, three slots): fp32 values carried as hi + lo
    half pairs, GEMMs on the f16 matrix cores (hi*hi + hi*lo + lo*hi, fp32 accumulate), everything else the fp32 code.
    Reported next to the headline, never as `value`; with its logits deviation from the fp32 engine."""
    import torch
    import chiron_amd as ca
    steps, slots = 20, 3
    with ca.Engine(spec, weights, max_batch=BATCH, segment_len=SEG_LEN, device_id=device_id, n_slots=slots, dtype="fp32-split") as es:
        rs = es.infer(x_dev[0], s_dev[0], want_logits=True)
        d = np.abs(rs.logits - ref32.logits)
        same = np.array_equal(rs.decoded.values, ref32.decoded.values) and np.array_equal(rs.decoded.indices, ref32.decoded.indices)
        pend = [False] * slots
        for i in range(slots):
            es.submit(i, x_dev[i % len(x_dev)], s_dev[i % len(s_dev)], beam_width=0, want_prob=True)
            pend[i] = True
        torch.cuda.synchronize()
        es.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            if pend[i % slots]:
                es.collect(i % slots)
            es.submit(i % slots, x_dev[i % len(x_dev)], s_dev[i % len(s_dev)], beam_width=0, want_prob=True)
        for i in range(slots):
            es.collect(i)
        es.sync()
        dt = time.perf_counter() - t0
        # its own roofline (round-5 review, item 2): HIP events on the engine's stream, one batch in flight.  Every fp32 product is
        # three f16 MFMA terms (hi*hi + hi*lo + lo*hi), so a GEMM-shaped bucket EXECUTES 3 x its algorithmic FLOPs on the f16 matrix
        # pipe (v_mfma_f32_32x32x16_f16 / 16x16x16_f16, dense peak 2500 TFLOP/s); conv2b runs direct here (no Winograd form)
        es.profile(True)
        for i in range(3):
            es.submit(0, x_dev[i % len(x_dev)], s_dev[i % len(s_dev)], beam_width=0, want_prob=True)
            es.collect(0)
        stats = es.profile_read()
        es.profile(False)
    gemm_like = [k for k in ("lstm_recurrence", "lstm_proj0_dma", "lstm_proj_dma", "conv_dma", "conv2a", "conv_res") if k in stats and stats[k]["total_ms"] > 0]
    kern = {k: {"ms_per_batch": round(stats[k]["total_ms"] / 3.0, 4), "launches_per_batch": stats[k]["launches"] / 3.0,
                "algorithmic_tflops": round(stats[k]["flops"] / (stats[k]["total_ms"] * 1e-3) / 1e12, 1),
                "executed_f16_tflops": round(3.0 * stats[k]["flops"] / (stats[k]["total_ms"] * 1e-3) / 1e12, 1),
                "frac_of_f16_peak": round(3.0 * stats[k]["flops"] / (stats[k]["total_ms"] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)} for k in gemm_like}
    dom = max(gemm_like, key=lambda k: stats[k]["total_ms"])
    tot_ms = sum(v["total_ms"] for v in stats.values()) / 3.0
    roofline = {"kernel_bucket": dom, "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F16_MFMA_TFLOPS,
                "achieved": kern[dom]["executed_f16_tflops"], "frac": kern[dom]["frac_of_f16_peak"],
                "avg_launch_ms": round(stats[dom]["total_ms"] / stats[dom]["launches"], 4),
                "executed_flops": "3 x the algorithmic FLOPs (three f16 MFMA terms per fp32 product); algorithmic_tflops / 157.3 is the figure comparable with the fp32 engine's roofline",
                "frac_of_fp32_mfma_peak_in_algorithmic_flops": round(kern[dom]["algorithmic_tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
                "mfma_kernels": kern, "single_slot_ms_per_batch": round(tot_ms, 3),
                "traffic": None, "traffic_note": "no PMC pass for this dtype in this run; profiles/r06_split_kernel_stats_slots1.csv holds the rocprofv3 kernel table"}
    if live_pmc:       # the same two traffic passes as the headline's, over this dtype's engine (its recurrence: lstm32s_kernel)
        live = live_pmc_traffic("fp32-split", traffic_only=True)
        rec = next((v for k, v in (live.get("kernels") or {}).items() if k.startswith("lstm32s_kernel")), None) if dom == "lstm_recurrence" else None
        roofline["traffic_live"] = {"error": live.get("error"), "seconds": live.get("seconds"),
                                    "hbm_bytes_per_launch": {k: v["hbm_bytes"] for k, v in (live.get("kernels") or {}).items()}}
        if rec and rec.get("hbm_bytes"):
            roofline["traffic"] = rec["hbm_bytes"]
            roofline["algorithmic_bytes_per_launch"] = stats[dom]["bytes"] / stats[dom]["launches"]
            roofline["traffic_note"] = ("HBM bytes per launch of lstm32s_kernel measured in THIS run: child processes under rocprofv3 --kernel-trace --pmc "
                                        "FETCH_SIZE / WRITE_SIZE (separate passes, %d / %d launches), (2*FETCH_SIZE + WRITE_SIZE)*1024" % (rec["launches_fetch"], rec["launches_write"]))
    return {"workload": "headline workload (batch 1100, greedy), dtype fp32-split",
            "kbases_per_s": round(steps * BATCH * BASES_PER_WINDOW / 1000.0 / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 3),
            "roofline": roofline,
            "logits_vs_f32_engine": {"max_abs": float("%.3e" % d.max()), "mean_abs": float("%.3e" % d.mean()),
                                     "greedy_decode_identical": bool(same), "windows": BATCH}}


def pmc_child(dtype="fp32"):
    """--pmc-child [DTYPE]: the workload of the live PMC passes -- the headline engine (or its fp32-split form), ONE slot, three batches of the
    headline workload from host buffers, no timing, no torch.  rocprofv3 wraps this process; the parent parses its counter_collection.csv."""
    import chiron_amd as ca
    spec = ca.dna_default_spec()
    xb, lb, _, _ = make_batches(1, 0)
    with ca.Engine(spec, ca.synthetic_weights(spec, seed=1234), max_batch=BATCH, segment_len=SEG_LEN, n_slots=1, dtype=dtype) as eng:
        sl = ca.seq_len_for_engine(lb[0], eng.ratio)
        for _ in range(3):
            eng.submit(0, xb[0], sl, beam_width=0, want_prob=True)
            eng.collect(0)
    return 0


def collect_counter(files, counter, acc):
    """rows of rocprofv3's *counter_collection.csv files -> acc[kernel name without `void`, namespace and argument list][counter] = [values]"""
    import csv
    for path in files:
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter:
                name = row.get("Kernel_Name", "").replace("void ", "").replace("chiron::", "").split("(")[0]
                acc.setdefault(name, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
    return acc


def hbm_bytes_per_launch(acc):
    """acc of collect_counter -> {kernel: {hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md's gfx950 correction), ...}}"""
    kernels = {}
    for name, c in acc.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            kernels[name] = {"hbm_bytes": (2.0 * f + w) * 1024.0, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                             "launches_fetch": len(c["FETCH_SIZE"]), "launches_write": len(c["WRITE_SIZE"])}
    return kernels


def live_pmc_traffic(dtype="fp32", traffic_only=False):
    """HBM bytes per launch of every kernel of the path, measured in THIS run (round-5 review, Weak #6: the line used to carry the
    builder's committed record): this script as a child process under `rocprofv3 --kernel-trace --pmc <counter>`, once for FETCH_SIZE
    and once for WRITE_SIZE (separate passes, kernel trace only -- MI355X_MICROARCH.md's recipe), from /tmp.  hbm_bytes =
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (the guide's gfx950 correction of FETCH_SIZE).  -> {"kernels": {name: {...}}, ...} or
    {"error": ...}: a box without rocprofv3, a failing child or an unreadable csv leave the committed record in place."""
    import glob
    import shutil
    import subprocess
    import tempfile
    t0 = time.perf_counter()
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not found"}
    acc = {}
    try:
        # two traffic passes, then the matrix-pipe pass (north_star: "MFMA utilisation on the LSTM"): the counters of tools/pmc_pass.sh
        groups = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"), ("SQ_ACTIVE_INST_VALU",))
        for group in groups[:2] if traffic_only else groups:
            counter = " ".join(group)
            d = tempfile.mkdtemp(prefix="chiron_pmc_", dir="/tmp")
            try:
                cmd = [exe, "--kernel-trace", "--pmc"] + list(group) + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", dtype]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=60)   # a pass takes about a second; a profiler that hangs must not hold the bench
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    if group[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                        return {"error": "rocprofv3 --pmc %s: rc %d, %d csv files: %s" % (counter, r.returncode, len(files), (r.stderr or "")[-300:]), "seconds": round(time.perf_counter() - t0, 1)}
                    break                                   # the traffic is in; the matrix-pipe figures stay with the committed record
                for c in group:
                    collect_counter(files, c, acc)
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
        return {"error": "%s: %s" % (type(e).__name__, e), "seconds": round(time.perf_counter() - t0, 1)}
    kernels = hbm_bytes_per_launch(acc)
    for name, c in acc.items():                            # tools/pmc_to_json.py's formulas (mean per launch)
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if name in kernels and m.get("SQ_BUSY_CU_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            simd = 4.0 * m["SQ_BUSY_CU_CYCLES"]
            kernels[name]["mfma_busy_frac_of_busy_cus"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4)
            if m.get("GRBM_GUI_ACTIVE", 0) > 0:
                kernels[name]["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
            if "SQ_ACTIVE_INST_VALU" in m:
                kernels[name]["valu_busy_frac_of_busy_cus"] = round(max(0.0, 4.0 * m["SQ_ACTIVE_INST_VALU"] - m.get("SQ_INSTS_MFMA", 0.0)) / simd, 4)
    return {"kernels": kernels, "seconds": round(time.perf_counter() - t0, 1), "error": None if kernels else "no kernel carried both counters"}


def frontier_record():
    """profiles/r06_f16_frontier.json (tools/f16_frontier.py on the GPU box): per regime and mode the identical-window fraction, and the
    time per 4096 windows -- a static record of an earlier run, named as such"""
    path = os.path.join(ROOT, "profiles", "r06_f16_frontier.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None
    return {"source": "static profiles/r06_f16_frontier.json (not this run)", "ms_per_4096_windows": doc.get("ms_per_4096_windows"),
            "identical_windows_frac": {reg: {m: v["identical_windows_frac"] for m, v in rec["modes"].items()} for reg, rec in doc.get("regimes", {}).items()}}


def kernel_sources_digest():
    """SHA-256 over the KERNEL sources (*.hip and the headers they include): a PMC record is only valid for the kernels it was
    collected on.  The host-only C++ of the library (fast5.cpp: the HDF5 reader, assemble.cpp: the consensus vote and the finisher)
    defines no kernel and is left out: hardening the fast5 reader must not withhold a recurrence kernel's HBM traffic."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "chiron_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the most recent committed PMC pass under profiles/ (tools/pmc_pass.sh +
    tools/pmc_to_json.py; rocprofv3 cannot wrap the timed process from inside).  The record carries the digest of the
    kernel sources it was measured on; when the sources have changed since, the number is withheld (traffic = null)
    and the source string says so -- a stale constant is never reported as this run's traffic."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not cands:
        return None, "none: no profiles/r*_pmc.json"
    name = "profiles/" + os.path.basename(cands[-1])
    try:
        doc = json.load(open(cands[-1]))
    except (OSError, ValueError):
        return None, "unreadable: " + name
    have, want = doc.get("_kernel_sources_sha256_16"), kernel_sources_digest()
    if have != want:
        sys.stderr.write("bench.py: %s was collected on kernel sources %s, the tree holds %s: roofline.traffic withheld; "
                         "re-run tools/pmc_pass.sh + tools/pmc_to_json.py\n" % (name, have, want))
        return None, "STALE: %s was collected on kernel sources %s, this tree is %s" % (name, have, want)
    _PMC_DOC.update(doc)
    return doc.get(kernel, {}).get("hbm_bytes"), "static %s @ kernel sources %s (separate --pmc pass, not this run)" % (name, want)


_PMC_DOC = {}   # the PMC record pmc_traffic() accepted (same kernel sources as this tree), else empty


def pmc_counters(kernel):
    """MFMA-busy fraction, LDS bank-conflict fraction, L2 hit rate and HBM bytes per launch of `kernel` from the accepted PMC
    record, or None."""
    rec = _PMC_DOC.get(kernel)
    if not isinstance(rec, dict):
        return None
    return {"mfma_busy_frac": round(rec.get("mfma_busy_frac_of_all_simds", 0.0), 4),
            # on the CUs the kernel holds: matrix pipe busy, VALU busy (fp32 MFMA and VALU of one SIMD do not overlap on gfx950), their sum
            "mfma_busy_frac_of_busy_cus": round(rec.get("mfma_busy_frac_of_busy_cus", 0.0), 4),
            "valu_busy_frac_of_busy_cus": round(rec.get("valu_busy_frac_of_busy_cus", 0.0), 4),
            "issue_busy_frac_of_busy_cus": round(rec.get("issue_busy_frac_of_busy_cus", 0.0), 4),
            "lds_bank_conflict_frac": round(rec.get("lds_bank_conflict_frac", 0.0), 4),
            "l2_hit_rate": round(rec.get("l2_hit_rate", 0.0), 4), "hbm_bytes_per_launch": rec.get("hbm_bytes")}


def cpu_baseline(spec, weights, xb, lb, ratio, n_windows):
    """oracle/chiron_oracle.c (fp32 C restatement, OpenMP over windows) on this host's cores: the
    'port' CPU baseline.  Bounded sample: a few windows per thread."""
    from oracle import c_oracle
    import chiron_amd as ca
    threads = c_oracle.max_threads()
    cores = os.cpu_count() or threads
    threads = min(threads, cores)
    n = n_windows if n_windows > 0 else max(threads * 24, 64)
    n = min(n, xb.shape[0] * xb.shape[1])
    x = xb.reshape(-1, SEG_LEN)[:n]
    sl = ca.seq_len_for_engine(lb.reshape(-1)[:n], ratio)
    blob = spec.pack(weights)
    c_oracle.forward(x[:threads], sl[:threads], spec.to_dict(), blob, spec.output_len(SEG_LEN), threads)  # warm
    t0 = time.perf_counter()
    lg = c_oracle.forward(x, sl, spec.to_dict(), blob, spec.output_len(SEG_LEN), threads)
    c_oracle.greedy(lg, sl)
    dt = time.perf_counter() - t0
    return {"value": round(n * BASES_PER_WINDOW / 1000.0 / dt, 3), "unit": "kbases/s", "cores": threads,
            "kind": "port", "sample": "%d windows of the same synthetic workload, oracle/chiron_oracle.c "
            "(fp32, gcc -O3 AVX2, OpenMP over windows), %.1f s" % (n, dt)}


if __name__ == "__main__":
    main()
