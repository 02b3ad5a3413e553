#!/usr/bin/env python3
"""bench.py — frames/sec of the HDN homography hot path on MI355X + roofline of the correlation kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path's HIP kernels over one batch of 64 synthetic 127/255 template/search
pairs per GPU (BASELINE.json configs[1]; inputs of SURVEY.md §8d cfg 2, resident in HBM before the timed
region starts):
    1 x xcorr_depthwise           [64,256,61,61] (x) [64,256,31,31]     the north-star correlation kernel
    6 x xcorr_depthwise           [64,256,29,29] (x) [64,256,5,5]       MultiBAN: 3 levels x {cls,loc}  (one launch)
    6 x xcorr_depthwise_circular  [64,256,13,13] (x) [64,256,13,13]     MultiCircBAN                    (one launch)
    PreShareFeature(template,search) -> fused DLT+warp (offsets ~ N(0, 8^2) px) -> PreShareFeature(warped) -> 2 scores
    N > 1: one RCCL all-gather of the [64,8] corner offsets per rank (the path's only exchange step), issued through the
           C ABI (hdn_allgather_offsets on a communicator bootstrapped over torch.distributed's nccl group)
Pairs are independent, so ranks hold disjoint batches (weak scaling) and `value` = N*64*K / max-over-ranks time.
Before the W warm-up steps the GPU is kept busy for --prewarm-ms (150 ms, untimed, rank-local): from idle an MI355X needs
~40 ms of sustained load to reach steady clocks, and the first ~80 steps would otherwise be timed on the ramp.

The JSON line also carries
    "roofline"      for the dominant kernel (the 31x31 (x) 61x61 correlation), measured with events on the launch stream around every
                    launch of the TIMED region (mean / min / median); with a --head-stream other than the default, in
                    --roofline-steps separate after-north steps behind it
    "cpu_baseline"  the CPU oracle (PyTorch-CPU restatement of the reference path) timed on this box's host cores:
                    all 64 pairs, warm-up 3, 5 passes, at 1 thread pinned to one core (the reference's own setting,
                    tools/test.py:51) and at all cores; headline = the better of the two, best and median of both listed
    "full_head"     BASELINE configs[2] per GPU, timed after the headline region: the whole HomoModelBuilder head incl. the
                    ResNet-34 trunk on the same 64 pairs (the trunk is >90 % of it), with its own CPU figure
    "sequence"      BASELINE configs[3] (N = 1): the end-to-end tracker loop over a 501-frame synthetic 1280x720 sequence with a
                    production-shaped model — ms per frame, frames/s, per-component table, share owned by the hand-written kernels
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20260928
PAIRS = 64  # per GPU
C = 256
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_VALU_PEAK_TFLOPS = 157.3

# algorithmic (compulsory) bytes and flops per pair, fp32 — SURVEY.md §8d / BASELINE.md §4
NORTH_BYTES_PER_PAIR = 4 * C * (61 * 61 + 31 * 31 + 31 * 31)      # 5,778,432
NORTH_FLOPS_PER_PAIR = 2 * C * 31 * 31 * 31 * 31                  # 472,842,752
PROD_BYTES_PER_PAIR = 4 * C * (29 * 29 + 5 * 5 + 25 * 25)         # 1,526,784
CIRC_BYTES_PER_PAIR = 4 * C * 3 * 13 * 13                         # 519,168
SF_BYTES_PER_IMG = 2 * 4 * 127 * 127                              # 129,032
WARP_BYTES_PER_PAIR = 2 * 4 * 127 * 127 + 64 + 36                 # 129,132


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="before the W warm-up steps, keep the GPU busy with untimed steps for this long: an idle MI355X "
                         "needs ~40 ms of sustained load to reach its steady clocks (tools/experiments/exp_warmup.py: 0.48 -> 0.38 ms/step)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle timing (profiling runs)")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the per-kernel breakdown after the timed region")
    ap.add_argument("--no-full-head", action="store_true", help="skip the full_head block (configs[2] per-GPU workload)")
    ap.add_argument("--full-head-steps", type=int, default=30)
    ap.add_argument("--no-sequence", action="store_true", help="skip the `sequence` block (BASELINE configs[3]: the end-to-end tracker loop "
                                                               "with a production-shaped model, N = 1 only)")
    ap.add_argument("--sequence-frames", type=int, default=501, help="frames of the synthetic 1280x720 sequence (POT: 501, hdn/core/config.py:285)")
    ap.add_argument("--multi-sequence", type=str, default="16", help="comma-separated n: the `sequence.lockstep` rows — n sequences advancing in lock step on this GPU "
                    "(hdn_amd.batched_tracker); '' skips them.  Default 16 only (MIOpen's find at every new batch size costs ~30 s); the 1 / 4 / 16 / 32 table: "
                    "profiles/round6_multi_sequence.txt")
    ap.add_argument("--collective", choices=["c_abi", "torch", "oneshot"], default="c_abi",
                    help="N > 1: hdn_allgather_offsets of the C ABI on RCCL (default), torch.distributed.all_gather_into_tensor, or the "
                         "direct-write hdn_gather_offsets_oneshot (hipIpc windows; validated on one device only)")
    ap.add_argument("--only-north", action="store_true", help="step = the north-star correlation only (profiling aid)")
    ap.add_argument("--head-stream", choices=["inline", "after-north", "parallel"], default="after-north",
                    help="where the homography head runs in the TIMED region: in line with the correlations; on its own stream beside "
                         "the 13x13 and 5x5 launches only (default: the fastest schedule since the streaming hints, and the 31x31 "
                         "launch has the chip to itself, so the roofline brackets ARE the timed region's); or on its own stream from "
                         "the start of the step (then the roofline block times the 31x31 kernel in --roofline-steps separate "
                         "after-north steps)")
    ap.add_argument("--roofline-steps", type=int, default=30, help="untimed steps after the timed region in which the 31x31 launch is bracketed")
    ap.add_argument("--north", choices=["fft", "direct"], default=None,
                    help="kernel for the 31x31 (x) 61x61 correlation (default: the library's default, fft = the column-first FFT kernel); A/B runs")
    ap.add_argument("--config", type=int, choices=[2, 5], default=2,
                    help="2 = BASELINE configs[1] (default, the headline); 5 = configs[4]: 303-px search window (6x xcorr 5x5 (x) 35x35 "
                         "-> 31x31) + the 2-iteration refinement loop, 64 pairs per GPU (256 over 4 GPUs); kernels only")
    ap.add_argument("--workload", choices=["kernels", "full"], default="kernels",
                    help="kernels = BASELINE configs[1] (default); full = configs[2]: the whole HomoModelBuilder head "
                         "incl. the PyTorch-ROCm ResNet-34 trunk on 64 pairs per GPU, then the offsets all-gather")
    return ap.parse_args()


def make_inputs(dev, rank):
    """Synthetic inputs of SURVEY §8d cfg 2; generated on the CPU with a fixed seed (per rank) then copied."""
    g = torch.Generator().manual_seed(SEED + rank)
    relu_n = lambda *s: torch.randn(*s, generator=g).clamp_min_(0)
    d = {}
    d["north_x"] = relu_n(PAIRS, C, 61, 61).to(dev)
    d["north_k"] = relu_n(PAIRS, C, 31, 31).to(dev)
    d["prod_x"] = [relu_n(PAIRS, C, 29, 29).to(dev) for _ in range(6)]
    d["prod_k"] = [relu_n(PAIRS, C, 5, 5).to(dev) for _ in range(6)]
    d["circ_x"] = [relu_n(PAIRS, C, 13, 13).to(dev) for _ in range(6)]
    d["circ_k"] = [relu_n(PAIRS, C, 13, 13).to(dev) for _ in range(6)]
    d["imgs"] = torch.randn(PAIRS, 2, 127, 127, generator=g).to(dev)  # template, search (normalised gray)
    d["h4p"] = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(PAIRS, 1).to(dev)
    d["off"] = (8.0 * torch.randn(PAIRS, 8, generator=g)).to(dev)
    return d


def build_full_head(dev, imgs, h4p):
    """The BASELINE configs[2] network exactly as the `full_head` block and --workload full time it: seeded HomoModelBuilder,
    MIOpen find mode, BN-folded NHWC trunk with the fused first stage and fused block epilogues.  -> (net on dev, data, CPU
    state_dict of the un-folded net).  tests/test_gpu_parity.py::test_benchmarked_full_head_configuration_parity holds THIS
    configuration to the CPU oracle."""
    import hdn_amd

    torch.manual_seed(SEED + 7)
    net = hdn_amd.HomoModelBuilder().eval()
    net.fc.weight.data.mul_(0.01)
    cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    torch.backends.cudnn.benchmark = True  # MIOpen find mode for the trunk's fixed shapes (searched during warm-up)
    net = net.to(dev).optimize_for_inference(channels_last=True)
    n = imgs.shape[0]
    data = {"org_imgs": imgs, "input_tensors": imgs, "h4p": h4p,
            "patch_indices": torch.arange(127 * 127, dtype=torch.float32, device=dev).repeat(n, 1)}
    return net, data, cpu_sd


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # test hook: HDN_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and uses gloo, so the N>1 code path can be exercised on a
    # 1-GPU box (RCCL refuses two ranks on one device); never set by the driver
    one_device = os.environ.get("HDN_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import hdn_amd
    from hdn_amd import dist as hdist
    comm = None
    if world > 1 and not one_device and args.collective == "c_abi":
        comm = hdist.RcclComm.from_process_group(dev)
    elif world > 1 and args.collective == "oneshot":     # (works between processes sharing a device too)
        comm = hdist.OneShotGather.from_process_group(PAIRS, dev)
    from hdn_amd import homography as G
    from hdn_amd import share_feature as SF
    from hdn_amd import xcorr as X

    if args.north:
        X.north_variant(args.north)  # process-wide
    d = make_inputs(dev, rank)
    torch.manual_seed(SEED)
    sf = hdn_amd.PreShareFeature().eval()
    for m in sf.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.8, 1.2)
    sf_cpu_sd = {"ShareFeature." + k: v.clone() for k, v in sf.ShareFeature.state_dict().items()}
    sf = sf.to(dev)
    folded = sf.folded(dev)
    imgs2 = d["imgs"].reshape(PAIRS * 2, 1, 127, 127)
    tmpl = d["imgs"][:, :1].contiguous()

    if args.config == 5:
        return run_config5(args, dev, rank, world, dist, hdist, comm, sf, folded)

    north_ev = []
    full_net = full_data = None
    # Brackets of the 31x31 launch.  Default: the launch itself carries a start / stop hipEvent (hdn_xcorr_north_launch_events -> hipExtLaunchKernelGGL):
    # the dispatch's own timestamps, on the stream the kernel runs on, no marker packets around it.  HDN_BENCH_BRACKETS=record: a torch.cuda.Event pair
    # recorded on the stream before / after the call (rounds 1-5; each record is a marker packet: 7.5 us per step, profiles/round6_experiments.txt section 3).
    bracket_mode = os.environ.get("HDN_BENCH_BRACKETS", "launch")
    if (args.north or os.environ.get("HDN_NORTH", "fft")) != "fft":
        bracket_mode = "record"      # (the hook is the FFT kernel's launch site's)
    hip_rt = ctypes.CDLL("libamdhip64.so") if bracket_mode == "launch" else None

    class LaunchEvent:
        """hipEvent_t with torch.cuda.Event's elapsed_time()."""
        def __init__(self):
            self.h = ctypes.c_void_p()
            if hip_rt.hipEventCreate(ctypes.byref(self.h)) != 0:
                raise RuntimeError("hipEventCreate failed")

        def elapsed_time(self, other):
            ms = ctypes.c_float()
            rc = hip_rt.hipEventElapsedTime(ctypes.byref(ms), self.h, other.h)
            if rc != 0:
                raise RuntimeError(f"hipEventElapsedTime -> {rc}")
            return ms.value

    from hdn_amd import _lib as hlib
    fork_on_event = os.environ.get("HDN_BENCH_FORK", "event") == "event"
    join_every_step = os.environ.get("HDN_BENCH_JOIN", "step") == "step"      # A/B switch: "fence" = the head stream is only joined by the region's synchronize

    def build_full():
        return build_full_head(dev, d["imgs"], d["h4p"])

    from hdn_amd.homo_model import homo_stages
    full_cpu_sd = None
    if args.workload == "full":
        full_net, full_data, full_cpu_sd = build_full()

    def step_full(record, collective):
        st = homo_stages(full_net, full_data)
        if world > 1 and collective:
            hdist.all_gather_offsets(st["x"], PAIRS * world, comm=comm)

    # HDN_BENCH_HEAD_PRIORITY (A/B switch, tools/experiments/ab_priority.sh): queue priority of the head stream relative to the
    # correlation stream (torch: lower number = higher priority)
    _prio = os.environ.get("HDN_BENCH_HEAD_PRIORITY")
    head_stream = torch.cuda.Stream(device=dev) if _prio is None else torch.cuda.Stream(device=dev, priority=int(_prio))

    def step(record, collective=True, mode=None, sink=None):
        if args.workload == "full":
            return step_full(record, collective)
        main = torch.cuda.current_stream()

        def head():
            feats = SF.share_feature(imgs2, folded).reshape(PAIRS, 2, 127, 127)
            Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
            pf = SF.share_feature(warped, folded)
            G.l1_score2(feats[0, 1], pf[0, 0], feats[0, 0], 1.0 / (127 * 127))

        def fork_head(after=None):
            # `after`: the 31x31 launch's own stop event (launch-carried brackets): the head stream waits for THAT, and the main stream gets no marker
            # packet between the 31x31 launch and the 13x13 one (HDN_BENCH_FORK=stream: the ordinary wait_stream; A/B in profiles/round6_experiments.txt section 4)
            if after is not None and fork_on_event:
                if hip_rt.hipStreamWaitEvent(ctypes.c_void_p(head_stream.cuda_stream), after.h, 0) != 0:
                    raise RuntimeError("hipStreamWaitEvent failed")
            else:
                head_stream.wait_stream(main)
            with torch.cuda.stream(head_stream):
                head()
                if world > 1 and collective:  # the head's results are gathered while the correlation stream is still busy
                    hdist.all_gather_offsets(d["off"], PAIRS * world, comm=comm)

        # The step's two branches are independent, as in the tracker (similarity-branch correlations | homography head).
        # Rounds 2-3 (tools/experiments/exp_streams2.py): one stream 0.324 ms; head beside the 13x13 and 5x5 launches 0.317; head from
        # the start of the step 0.301.  Since the correlation kernels use nontemporal loads / stores the order is the other way
        # round (tools/experiments/ab_head_stream.sh, three alternations on one box): after-north 0.2778-0.2783 ms, parallel
        # 0.2830-0.2852 ms - the issue-bound 31x31 kernel is best left alone, the head hides behind the bandwidth-bound launches.
        mode = "inline" if args.only_north else (mode or args.head_stream)
        if mode == "parallel":
            fork_head()
        stop_only = record == "stop"          # (A/B: HDN_BENCH_BRACKET_EVERY) the launch carries only the stop event the head stream forks on
        if record and bracket_mode == "launch":
            e0, e1 = (None if stop_only else LaunchEvent()), LaunchEvent()
            hlib.load().hdn_xcorr_north_launch_events(e0.h if e0 is not None else None, e1.h)
        elif record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        X.xcorr_depthwise(d["north_x"], d["north_k"])
        if record:
            if bracket_mode != "launch":
                e1.record()
            if not stop_only:
                (north_ev if sink is None else sink).append((e0, e1))
        if args.only_north:
            return
        if mode == "after-north":
            fork_head(e1 if (record and bracket_mode == "launch") else None)
        # 13x13 before the write-heavy 5x5 (x) 29x29 launch: 2 % faster than the other way round (tools/experiments/exp_step_order.py)
        X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
        if mode == "inline":
            head()
        X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
        if mode != "inline" and join_every_step:
            main.wait_stream(head_stream)
        elif world > 1 and collective:
            hdist.all_gather_offsets(d["off"], PAIRS * world, comm=comm)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:  # clock / power ramp from idle, untimed
        for _ in range(5):
            step(False, collective=False)  # wall-clock bounded, so rank-local: no collective in here
        torch.cuda.synchronize()
    for _ in range(args.warmup + (5 if args.workload == "full" else 0)):
        step(False)
    fence()
    no_brackets = os.environ.get("HDN_BENCH_NO_BRACKETS") == "1"      # A/B switch (profiles/round6_experiments.txt section 3): what the event pairs cost the step
    t0 = time.perf_counter()
    every = max(1, int(os.environ.get("HDN_BENCH_BRACKET_EVERY", "1")))     # A/B switch: start + stop events on every N-th timed step only
    for i_ in range(args.steps):
        step(False if no_brackets else (True if (i_ % every == 0 or bracket_mode != "launch") else "stop"))
    fence()
    elapsed = time.perf_counter() - t0
    if no_brackets and args.workload != "full":
        for _ in range(max(1, args.roofline_steps)):
            step(True, collective=False)
        torch.cuda.synchronize()
    if hasattr(comm, "check_status"):
        comm.check_status()       # one-shot gather: a peer that did not deliver within 2 s leaves NaN rows and a sticky status: fail loudly
    oneshot_used = type(comm).__name__ == "OneShotGather"   # (from_process_group hands back an RcclComm when a pair of ranks lacks peer access)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if args.workload == "full":
        coll = None
        if world > 1:
            x_local = homo_stages(full_net, full_data)["x"]
            coll = collective_block(x_local, dev, rank, world, dist, hdist, comm)
            if rank == 0:      # SURVEY 8d cfg 3: parity against the CPU on a 16-pair sample of this rank's shard
                coll["parity_16"] = parity_16(x_local, d, full_cpu_sd)
        if rank == 0:
            print(json.dumps({
                **({"collective": coll} if coll is not None else {}),
                "metric": "frames/sec on 127/255 template/search pairs", "value": PAIRS * world * args.steps / elapsed,
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BASELINE configs[2] per GPU: full HomoModelBuilder head (PreShareFeature x2 -> PyTorch-ROCm "
                                       "ResNet-34 trunk -> fused DLT+warp -> PreShareFeature) on 64 pairs, offsets all-gathered",
                           "pairs_per_gpu": PAIRS}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # Roofline of the 31x31 (x) 61x61 kernel: HIP events on its launch stream around the launch.  Under the default schedule
    # (after-north: the head stream is forked behind that launch, which therefore has the chip to itself) the brackets of the
    # TIMED region are the measurement.  Under another schedule the timed region's brackets are reported beside a short separate
    # loop of after-north steps.
    timed_is_clean = args.only_north or args.head_stream == "after-north"
    solo_ev = []
    if timed_is_clean:
        solo_ev = list(north_ev)
    else:
        for _ in range(3):
            step(False, collective=False, mode="after-north")
        for _ in range(max(1, args.roofline_steps)):
            step(True, collective=False, mode="after-north", sink=solo_ev)
    torch.cuda.synchronize()
    solo = np.array([a.elapsed_time(b) for a, b in solo_ev])
    in_region = np.array([a.elapsed_time(b) for a, b in north_ev])
    north_ms = float(solo.mean())
    north_gbps = NORTH_BYTES_PER_PAIR * PAIRS / (north_ms * 1e-3) / 1e9
    north_tflops = NORTH_FLOPS_PER_PAIR * PAIRS / (north_ms * 1e-3) / 1e12

    # HBM traffic of the dominant kernel: measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs,
    # gfx950 x2 correction on the 16 B/lane stream) and committed under profiles/; bench.py cannot collect PMCs itself.
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    north_variant = X.last_variant()
    north_kernel = {"north_fftc_61x61_31x31": "xcorr_north_fft4_kernel", "north_61x61_31x31": "xcorr_north_kernel"}[north_variant]
    traffic, traffic_note, traffic_fresh = None, None, None
    for rnd in ("round6", "round5", "round4", "round3", "round2", "round1"):  # the newest committed PMC measurement of this kernel
        path = os.path.join(ROOT, "profiles", rnd + "_pmc_hbm_traffic.json")
        if not os.path.exists(path) or traffic is not None:
            continue
        with open(path) as f:
            for name, rec in json.load(f).items():
                if north_kernel in name and "traffic_calibrated_bytes" in rec and traffic is None:
                    traffic = rec["traffic_calibrated_bytes"]
                    traffic_note = rec.get("how", "profiles/%s_pmc_hbm_traffic.txt" % rnd)
                    # the PMC passes are tied to the kernel source they measured (tools/pmc_traffic.py records its SHA-256)
                    want = rec.get("kernel_source_sha256")
                    if want:
                        import hashlib
                        src = os.path.join(ROOT, "hdn_amd", "csrc", rec.get("kernel_source", "xcorr_fft.hip"))
                        have = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
                        traffic_fresh = have == want
                        traffic_note += "; measured on %s sha256 %s" % (rec.get("kernel_source", "xcorr_fft.hip"), want[:16])
                        if not traffic_fresh and rank == 0:
                            print("bench.py: WARNING: hdn_amd/csrc/%s changed since %s measured `roofline.traffic` (re-run tools/profile_round.sh + "
                                  "tools/pmc_traffic.py)" % (rec.get("kernel_source", "xcorr_fft.hip"), os.path.basename(path)), file=sys.stderr)
    if traffic is None:
        # a renamed / replaced kernel must not silently carry `traffic: null` (or a stale figure): re-run tools/profile_round.sh
        raise SystemExit("bench.py: no committed PMC traffic measurement under profiles/*_pmc_hbm_traffic.json names the kernel the "
                         "roofline block times (" + north_kernel + "); run tools/profile_round.sh and commit its output")

    result = {
        "metric": "frames/sec on 127/255 template/search pairs",
        "value": PAIRS * world * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_ms": args.prewarm_ms,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,  # the reference publishes no number for this metric
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("north-star correlation only" if args.only_north else
                         "BASELINE configs[1]: batch=64 synthetic 127/255 pairs per GPU, 256-ch features; HIP kernels only: "
                         "1x xcorr 31x31(x)61x61 + 6x xcorr 5x5(x)29x29 + 6x circular xcorr 13x13(x)13x13 + "
                         "3x PreShareFeature 127x127 + fused DLT/warp + the 2 L1 scores (one launch)" + ((" + direct-write all-gather of [64,8] offsets (hdn_gather_offsets_oneshot)" if oneshot_used else " + RCCL all-gather of [64,8] offsets") if world > 1 else "")),
            "pairs_per_gpu": PAIRS,
            "channels": C,
            "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective except the offsets all-gather",
            "streams": {"inline": "one stream",
                        "after-north": "two streams: the homography head (PreShareFeature, DLT+warp, scores) runs beside the 13x13 and "
                                       "5x5 correlation launches, the 31x31 launch alone (kernel durations of the two streams overlap: "
                                       "their sum exceeds the step)",
                        "parallel": "two streams: the homography head (PreShareFeature, DLT+warp, scores) runs beside all three correlation "
                                    "launches from the start of the step (kernel durations of the two streams overlap: their sum exceeds the step)"}[
                            "inline" if args.only_north else args.head_stream],
        },
        "roofline": {
            "kernel": "hdn::%s (hdn_xcorr_depthwise_f32, 31x31 (x) 61x61, variant %s)" % (north_kernel, north_variant),
            "bound": "hbm",
            "achieved": north_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": north_gbps / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_note,
            "traffic_source_matches_kernel_source": traffic_fresh,   # None: the committed PMC file predates the source hash (round <= 4)
            "algorithmic_bytes_per_launch": NORTH_BYTES_PER_PAIR * PAIRS,
            "avg_launch_ms": north_ms,
            "min_launch_ms": float(solo.min()),
            "median_launch_ms": float(np.median(solo)),
            "bracket": ("hipEvent pair carried by the launch itself (hipExtLaunchKernelGGL through hdn_xcorr_north_launch_events): the dispatch's start / end "
                        "timestamps on its own stream" if bracket_mode == "launch" else "torch.cuda.Event pair recorded on the launch stream before / after the call"),
            "timing": ("in-step, %d brackets: HIP events on the launch stream around every 31x31 launch of the timed region (after-north "
                       "schedule: the head stream is forked behind that launch, it has the chip to itself); achieved / frac use the mean"
                       % len(solo)) if timed_is_clean else
                      ("in-step, %d brackets: HIP events on the launch stream around the 31x31 launch in %d separate steps run after the "
                       "timed region with the head stream forked AFTER that launch (it has the chip to itself); achieved / frac use the mean"
                       % (len(solo), len(solo))),
            "timed_region_launch_ms": {"schedule": "inline" if args.only_north else args.head_stream, "mean": float(in_region.mean()),
                                       "min": float(in_region.min()), "median": float(np.median(in_region)),
                                       "note": "the brackets inside the timed region (identical to the above under the default schedule; under "
                                               "the parallel schedule the kernel shares the chip with the homography head there)"},
        },
    }
    # The in-step brackets see the kernel at the step's duty cycle (~1/3 of the time; the chip cools between launches).  Ten launches
    # back to back inside one hipGraph show what it sustains (round 4: 110 us = 0.42 against 92 us = 0.50 in the step): both are reported.
    sustained_ms = graph_timed(lambda: X.xcorr_depthwise(d["north_x"], d["north_k"]))
    result["roofline"].update({
        "sustained_launch_ms": sustained_ms,
        "sustained_frac": NORTH_BYTES_PER_PAIR * PAIRS / (sustained_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "sustained_note": "ten launches back to back in one hipGraph (100 % duty cycle: the power-limited clock of a dense packed-FMA stream); "
                          "`frac` above is the same kernel at the step's duty cycle"})
    # the practical ceiling beside the nominal peak: a 16-byte-per-lane nontemporal copy of the launch's own byte count
    result["roofline"].update(measured_copy(dev))
    if north_variant.startswith("north_fft"):
        result["roofline"].update(north_issue_fractions(north_ms, sustained_ms, lambda: X.xcorr_depthwise(d["north_x"], d["north_k"]), dev, rank))
        result["roofline"]["note"] = (
            "64x64 fp32 FFT per pair of planes in registers + LDS (~1,380 packed VALU ops per plane instead of the direct "
            "sum's 7,688); one wave per SIMD (32 KB of LDS per wave), issue-bound: DESIGN.md section 4/6")
    else:
        result["roofline"].update({
            "note": ("direct fp32 sum at 81.8 FLOP/B is fp32-FMA-bound (ridge 19.7 FLOP/B), and the packed-FMA pipe is "
                     "saturated at the clock the power budget allows (profiles/round1_pmc_sq_north.txt): see valu_*"),
            "valu_achieved_tflops": north_tflops, "valu_peak_tflops": FP32_VALU_PEAK_TFLOPS,
            "valu_frac": north_tflops / FP32_VALU_PEAK_TFLOPS})

    if world > 1:
        result["collective"] = collective_block(d["off"], dev, rank, world, dist, hdist, comm)
    if rank == 0 and not args.no_breakdown and not args.only_north:
        result["kernels"] = breakdown(d, imgs2, tmpl, folded, X, SF, G)
    if not args.no_full_head and not args.only_north:
        # BASELINE configs[2] per GPU, outside the headline region: every rank runs it (same max-over-ranks rule)
        full_net, full_data, full_cpu_sd = build_full()
        for _ in range(8):  # MIOpen find + clocks
            homo_stages(full_net, full_data)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.full_head_steps):
            st = homo_stages(full_net, full_data)
            if world > 1:
                hdist.all_gather_offsets(st["x"], PAIRS * world, comm=comm)
        fence()
        el = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        result["full_head"] = {
            "workload": "BASELINE configs[2] per GPU: full HomoModelBuilder head (PreShareFeature x2 -> PyTorch-ROCm ResNet-34 "
                        "trunk, BN-folded NHWC under MIOpen find mode, first stage and block epilogues as HIP kernels -> fused DLT+warp -> PreShareFeature) on 64 pairs"
                        + (", offsets all-gathered" if world > 1 else ""),
            "value": PAIRS * world * args.full_head_steps / el, "unit": "frames/s", "steps": args.full_head_steps,
            "ms_per_step": el / args.full_head_steps * 1e3, "n_gpus": world,
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["full_head"]["cpu_baseline"] = cpu_full_head(d, full_cpu_sd)
    if rank == 0 and world == 1 and not args.no_sequence and not args.only_north:
        result["sequence"] = sequence_block(args.sequence_frames, dev, args.multi_sequence)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(d, sf_cpu_sd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def collective_block(x_local, dev, rank, world, dist, hdist, comm):
    """What a driver needs to audit an N > 1 line (round-4 verdict): which exchange ran, how many ranks the communicator ITSELF reports,
    whether every rank ended up with the same gathered [N * 64, 8] array, and whether this rank's rows arrived where they belong.
    One extra exchange after the timed region; every rank calls this."""
    kind = type(comm).__name__ if comm is not None else "torch.distributed." + dist.get_backend()
    if comm is not None and hasattr(comm, "comm_count"):
        comm_ranks = comm.comm_count()                     # ncclCommCount through the C ABI (hdn_rccl_comm_count)
    elif comm is not None:
        comm_ranks = int(comm.world)                       # the one-shot gather's window count
    else:
        comm_ranks = dist.get_world_size()
    x_local = x_local.detach().to(torch.float32).contiguous()
    got = hdist.all_gather_offsets(x_local, x_local.shape[0] * world, comm=comm)
    torch.cuda.synchronize()
    rows = x_local.shape[0]
    own_ok = bool(torch.equal(got[rank * rows:(rank + 1) * rows], x_local))
    stats = torch.tensor([got.double().sum().item(), got.double().abs().sum().item(), float(own_ok), float(comm_ranks)], dtype=torch.float64)
    red_dev = dev if dist.get_backend() == "nccl" else "cpu"
    lo, hi = stats.clone().to(red_dev), stats.clone().to(red_dev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    lo, hi = lo.cpu(), hi.cpu()
    return {"kind": {"RcclComm": "RCCL ncclAllGather through the C ABI (hdn_allgather_offsets)",
                     "OneShotGather": "direct-write one-shot gather (hdn_gather_offsets_oneshot)"}.get(kind, kind),
            "comm_ranks": comm_ranks, "comm_ranks_equal_across_ranks": bool(lo[3] == hi[3]), "world_size": world,
            "gathered_shape": list(got.shape),
            "checksum": float(stats[0]), "checksum_equal_across_ranks": bool(lo[0] == hi[0] and lo[1] == hi[1]),
            "own_rows_in_place_on_every_rank": bool(lo[2] == 1.0),
            "distinct_rows_per_rank": bool(rows < 2 or world < 2 or not torch.equal(got[:rows], got[rows:2 * rows]))}


def parity_16(x_gpu, d, net_sd):
    """max |x - CPU oracle| over the first 16 pairs of this rank's shard (full workload): the oracle's track_proj around a PyTorch-CPU
    ResNet-34 with the same weights (the north star's bound on the corner offsets is 1e-4)."""
    import hdn_amd
    from oracle import hdn_oracle as O

    n = 16
    net = hdn_amd.HomoModelBuilder().eval()
    net.load_state_dict(net_sd)
    sf_sd = {k: v for k, v in net.ShareFeature.state_dict().items()}
    imgs = d["imgs"][:n].detach().cpu()
    data = {"org_imgs": imgs, "input_tensors": imgs, "h4p": d["h4p"][:n].detach().cpu()}
    with torch.no_grad():
        _, _, _, aux = O.track_proj(data, sf_sd, lambda f: net.fc(net.avgpool(net.backbone(f)).flatten(1)))
    err = float((x_gpu[:n].detach().cpu() - aux["x"]).abs().max())
    return {"pairs": n, "max_abs_err_px": err, "bound_px": 1e-4, "ok": bool(err <= 1e-4)}


def run_config5(args, dev, rank, world, dist, hdist, comm, sf, folded):
    """BASELINE configs[4], the HIP kernels only: per GPU 64 pairs (= 256 over 4 GPUs), INSTANCE_SIZE 303 => the six
    correlations are 5x5 (x) 35x35 -> 31x31 (hdn_tracker_proj_e2e.py:24-25), and the refinement loop runs twice
    (hdn_tracker_proj_e2e.py:242-250 with trip count 2): PreShareFeature(search) -> fused DLT+warp -> PreShareFeature(warped)
    -> 2 scores -> refine warp (restated cv2.warpPerspective, BORDER_REPLICATE).  Offsets are synthetic (N(0, 8^2) px)."""
    from hdn_amd import homography as G
    from hdn_amd import refine as R
    from hdn_amd import share_feature as SF
    from hdn_amd import xcorr as X
    g = torch.Generator().manual_seed(SEED + 50 + rank)
    relu_n = lambda *s_: torch.randn(*s_, generator=g).clamp_min_(0)
    xs = [relu_n(PAIRS, C, 35, 35).to(dev) for _ in range(6)]
    ks = [relu_n(PAIRS, C, 5, 5).to(dev) for _ in range(6)]
    imgs = torch.randn(PAIRS, 2, 127, 127, generator=g).to(dev)
    tmpl, srch = imgs[:, :1].contiguous(), imgs[:, 1:].contiguous()
    h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(PAIRS, 1).to(dev)
    offs = [(8.0 * torch.randn(PAIRS, 8, generator=g)).to(dev) for _ in range(2)]
    p1 = SF.share_feature(tmpl, folded)
    Hc = torch.eye(3, dtype=torch.float64, device=dev).repeat(PAIRS, 1, 1).contiguous()
    ev = []

    def step(record):
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        X.xcorr_depthwise_multi(xs, ks)
        if record:
            e1.record()
            ev.append((e0, e1))
        cur = srch
        for it in range(2):
            p2 = SF.share_feature(cur, folded)
            Hm, warped = G.dlt_warp(h4p, offs[it], tmpl)
            pf = SF.share_feature(warped, folded)
            G.l1_score2(p2[0, 0], pf[0, 0], p1[0, 0], 1.0 / (127 * 127))
            cur = R.refine_warp(Hm, cur, Hc)
        if world > 1:
            hdist.all_gather_offsets(offs[1], PAIRS * world, comm=comm)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(5):
            step(False) if world == 1 else X.xcorr_depthwise_multi(xs, ks)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    nbytes = 6 * 4 * C * (35 * 35 + 25 + 31 * 31) * PAIRS
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec on 127/255 template/search pairs", "value": PAIRS * world * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] per GPU (64 pairs = 256 over 4 GPUs), HIP kernels only: 6x xcorr 5x5(x)35x35 "
                                   "(303-px search window) + 2 refinement iterations x (PreShareFeature, fused DLT/warp, "
                                   "PreShareFeature, 2 scores, restated cv2.warpPerspective)", "pairs_per_gpu": PAIRS, "channels": C},
            "roofline": {"kernel": "hdn::xcorr_cfg5_kernel x6 in one launch (" + X.last_variant() + ")", "bound": "hbm",
                         "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms}}), flush=True)


def sequence_block(n_frames, dev, multi=""):
    """BASELINE configs[3]: the end-to-end per-frame loop — hdn_amd.tracker.DeviceTrackerHomo(model), the object
    install(tracker=True) registers, one hipGraph per frame, the host reading the 4 corners every frame — over a synthetic
    1280x720 sequence of `n_frames` frames with a PRODUCTION-SHAPED model (tests/production_standin.py: ResNet-50 / stride-8
    dilated backbone twice per frame on PyTorch-ROCm / MIOpen, 1x1 necks, 256-channel heads, ResNet-34 homography estimator;
    seeded weights, fp32).  Sequences are independent (H_total recurrence): replicas only, N = 1.  Parity of the same loop against
    the CPU restatement: tests/test_gpu_tracker.py and `tests/tools/sequence_bench.py --production-shape --parity 61`
    (profiles/round4_sequence.txt); nothing under oracle/ is touched here."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import sequence_bench as SB
    t0 = time.perf_counter()
    res = SB.run_production(n_frames=n_frames, n_parity=0, components=True, dev=dev, quiet=True)
    res["wall_s_incl_generation_and_miopen_find"] = time.perf_counter() - t0
    res["metric"], res["unit"], res["value"] = "end-to-end frames/sec, tracker.track() loop, 1280x720 frames, 127/255 crops", "frames/s", res["fps"]
    ns = tuple(int(x) for x in multi.split(",") if x)
    if ns:
        # the reference's only inference-time parallelism: several videos at once (tools/test.py:91-103, hand-split ranges, one process each).  Here n
        # sequences advance one frame per step on ONE GPU: one pinned [n,720,1280,3] upload, one hipGraph replay, one host read of [n, 9] per step.
        t0 = time.perf_counter()
        m = SB.run_multi(ns, n_steps=40, dev=dev, quiet=True)
        for r in m["rows"]:
            r["speedup_vs_single_sequence_loop"] = r["frames_per_s"] / res["fps"]
        res["lockstep"] = {"tracker": m["tracker"], "rows": m["rows"], "wall_s": time.perf_counter() - t0,
                           "bound": "the frame is ~110 GFLOP of fp32 convolutions in the PyTorch-ROCm backbone (89 at 255 px + 21 at 127 px), which the library runs at "
                                    "78-82 % of the 157 TFLOP/s fp32 matrix peak from n = 16 on (profiles/round6_multi_sequence.txt): at most ~4.0 x the n = 1 loop at 100 %"}
    return res


def graph_timed(fn, inner=10, iters=10):
    """ms per launch of `fn`, `inner` launches captured as one hipGraph and replayed back to back between two events: no host side of a
    call (output allocations, pointer tables of the multi-problem launches) and no per-replay fixed cost in the figure."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [fn() for _ in range(inner)]     # `inner` launches per graph: a replay's fixed ~10 us is not in the figure
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    del keep
    return e0.elapsed_time(e1) / (iters * inner)


def measured_copy(dev, mbytes=370):
    """`roofline.measured_copy_GBps`: hdn_ubench_copy_f32 (csrc/ubench.hip) over ~the 31x31 launch's byte count (read + write = 2 x 185 MB),
    ten launches per hipGraph, HIP events around the replays.  The 8 TB/s `peak` stays the denominator of `frac`; this is the figure a
    pure copy reaches on THIS box in THIS run."""
    from hdn_amd import _lib
    n = (mbytes * 1000 * 1000 // 2 // 4) // 4 * 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    lib = _lib.load()

    def launch():
        _lib.check(lib.hdn_ubench_copy_f32(_lib.ptr(src), _lib.ptr(dst), n, _lib.stream_ptr(dev)), "ubench copy")
    ms = graph_timed(launch)
    assert torch.equal(src, dst)
    return {"measured_copy_GBps": 2 * 4 * n / (ms * 1e-3) / 1e9, "measured_copy_bytes": 2 * 4 * n, "measured_copy_ms": ms,
            "measured_copy_note": "hdn_ubench_copy_f32: 16 B per lane, nontemporal loads and stores, 8,192 workgroups; read + write bytes / time, "
                                  "ten launches back to back in one hipGraph (csrc/ubench.hip)"}


def _hwmon_of(dev):
    """The hwmon freq1_input (shader clock) of THIS device: the drm card whose PCI address is the device's.  A host with several GPUs exposes
    all of them under /sys whichever one the process was given; reading card0 then samples an idle neighbour (95 MHz on a round-6 box)."""
    import glob
    cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    if not cands:
        return None
    try:
        p = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x." % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        for c in cands:
            if os.path.basename(os.path.realpath(c.split("/hwmon/")[0])).startswith(want):
                return c
    except Exception:
        pass
    return cands[0] if len(cands) == 1 else None


def _sclk_sampler(dev=None):
    """(start, stop) around a region: shader clock in MHz from hwmon every 10 ms -> list (empty when the box does not expose it)."""
    import threading
    hw = _hwmon_of(dev if dev is not None else torch.cuda.current_device())
    samples, halt = [], threading.Event()

    def run():
        while not halt.is_set():
            try:
                samples.append(float(open(hw).read()) / 1e6)
            except Exception:
                return
            time.sleep(0.01)
    th = threading.Thread(target=run, daemon=True)

    def start():
        if hw:
            th.start()

    def stop():
        halt.set()
        if hw:
            th.join(timeout=1.0)
        return samples
    return start, stop


def north_issue_fractions(in_step_ms, sustained_ms, launch, dev, rank):
    """SURVEY.md section 8d "report both fractions" for the FFT form of the 31x31 (x) 61x61 kernel, which is neither HBM- nor FMA-bound but
    ISSUE-bound: one wave per SIMD, one issue slot per 4 clocks.
        issue_frac = 4 clk x instructions issued per pair / clocks a pair takes          (1.0 = the lone wave never waits)
        valu_frac  = 4 clk x packed fp32 VALU instructions per pair / clocks a pair takes (the share of issue slots that is arithmetic)
    clocks per pair = launch time x shader clock / pairs per SIMD (64 x 256 planes / 2 per pair / 1,024 SIMDs = 8).  Instruction counts:
    static, from the shipped code object (tools/north_instr_count.py -> profiles/round6_north_instr.json, tied to the kernel source by
    SHA-256).  Shader clock: hwmon samples (10 ms) during 0.3 s of back-to-back launches; the in-step launch runs at a higher clock than
    that (the chip boosts between launches), so the in-step fractions computed with this clock are UPPER bounds; the sustained_ pair
    (sustained time, sustained clock) is the matched one."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "round6_north_instr.json")
    if not os.path.exists(path):
        return {"issue_frac": None, "issue_note": "profiles/round6_north_instr.json missing (tools/north_instr_count.py)"}
    rec = json.load(open(path))
    src = os.path.join(ROOT, "hdn_amd", "csrc", rec["kernel_source"])
    fresh = hashlib.sha256(open(src, "rb").read()).hexdigest() == rec["kernel_source_sha256"] if os.path.exists(src) else None
    if fresh is False and rank == 0:
        print("bench.py: WARNING: hdn_amd/csrc/%s changed since tools/north_instr_count.py counted its instructions" % rec["kernel_source"], file=sys.stderr)
    # shader clock under the kernel, sustained
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        keep = [launch() for _ in range(10)]
    start, stop = _sclk_sampler(dev)
    g.replay()
    torch.cuda.synchronize()
    start()
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
    mhz = stop()
    del keep
    # a reading outside what this chip can run at under load (its range is 0.5 ... 2.4 GHz) is another device's or a stale one: not used
    if len(mhz) >= 5 and 1000.0 <= float(np.median(mhz)) <= 2600.0:
        clock_ghz, clock_src = float(np.median(mhz)) / 1e3, "hwmon freq1_input of this device, median of %d samples over 0.3 s of back-to-back launches" % len(mhz)
    else:
        clock_ghz = 2.05
        clock_src = ("hwmon not usable on this box (%s); 2.05 GHz = s_memtime / s_memrealtime per worker of the instrumented build, "
                     "profiles/round3_north_experiments.txt" % ("median %.0f MHz" % float(np.median(mhz)) if mhz else "no freq1_input for this device"))
    pairs_per_simd = PAIRS * C / 2 / 1024.0
    out = {"instructions_per_pair": rec["issued_per_pair"], "valu_packed_per_pair": rec["valu_packed_per_pair"],
           "instruction_count_source": "profiles/round6_north_instr.json (static, hot loop of the shipped code object)",
           "instruction_count_matches_kernel_source": fresh, "pairs_per_simd": pairs_per_simd,
           "shader_clock_GHz": clock_ghz, "shader_clock_source": clock_src}
    for tag, ms in (("", in_step_ms), ("sustained_", sustained_ms)):
        clocks = ms * 1e-3 * clock_ghz * 1e9 / pairs_per_simd
        out[tag + "clocks_per_pair"] = clocks
        out[tag + "issue_frac"] = 4.0 * rec["issued_per_pair"] / clocks
        out[tag + "valu_frac"] = 4.0 * rec["valu_packed_per_pair"] / clocks
    tflops = rec["flops_per_pair"] * PAIRS * C / 2 / (in_step_ms * 1e-3) / 1e12
    out.update({"valu_achieved_tflops": tflops, "valu_peak_tflops": FP32_VALU_PEAK_TFLOPS, "valu_flops_frac": tflops / FP32_VALU_PEAK_TFLOPS,
                "issue_note": "issue_frac / valu_frac: share of a lone wave's issue slots (one per 4 clocks) that hold an instruction / a packed fp32 "
                              "math instruction; the in-step pair uses the SUSTAINED clock sample (the step's clock is higher and not observable at "
                              "microsecond resolution), so the in-step fractions are upper bounds and the sustained_ ones are the matched pair"})
    return out


def breakdown(d, imgs2, tmpl, folded, X, SF, G, iters=10):
    """Per-kernel timing outside the timed region (informational; algorithmic GB/s per kernel): ten launches captured as one
    hipGraph and replayed back to back between two events, so the host side of a call (output allocations, pointer tables of
    the multi-problem launches) is not in the figure."""
    def timed(fn, inner=10):
        return graph_timed(fn, inner, iters)

    warped = G.dlt_warp(d["h4p"], d["off"], tmpl)[1]
    rows = {
        "xcorr_31x31_61x61": (lambda: X.xcorr_depthwise(d["north_x"], d["north_k"]), NORTH_BYTES_PER_PAIR * PAIRS),
        "xcorr_5x5_29x29_x6": (lambda: X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"]), 6 * PROD_BYTES_PER_PAIR * PAIRS),
        "xcorr_circ_13x13_x6": (lambda: X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True), 6 * CIRC_BYTES_PER_PAIR * PAIRS),
        "share_feature_x128": (lambda: SF.share_feature(imgs2, folded), 2 * PAIRS * SF_BYTES_PER_IMG),
        "share_feature_x64": (lambda: SF.share_feature(warped, folded), PAIRS * SF_BYTES_PER_IMG),
        "dlt_warp_x64": (lambda: G.dlt_warp(d["h4p"], d["off"], tmpl), PAIRS * WARP_BYTES_PER_PAIR),
    }
    out = {}
    for name, (fn, nbytes) in rows.items():
        ms = timed(fn)
        out[name] = {"ms": ms, "algorithmic_GBps": nbytes / (ms * 1e-3) / 1e9}
    prod_ms = sum(v["ms"] for k, v in out.items() if k != "xcorr_31x31_61x61")
    out["production_kernels_frames_per_s"] = PAIRS / (prod_ms * 1e-3)
    return out


def _host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def _timed_cpu(one_pass, units, warm=3, reps=5):
    """warm-up 3, 5 timed passes, time.perf_counter; measured twice:
    (a) torch.set_num_threads(1) with the process pinned to ONE core — the reference's own setting (tools/test.py:51);
    (b) all cores (capped at 64 threads), affinity restored.
    Returns a dict: best and median units/s of both, the thread count of (b), the relative spread of the 1-thread repeats."""
    have_aff = hasattr(os, "sched_getaffinity")
    cpus = sorted(os.sched_getaffinity(0)) if have_aff else list(range(os.cpu_count() or 1))

    def run(threads):
        torch.set_num_threads(threads)
        for _ in range(warm):
            one_pass()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            one_pass()
            ts.append(time.perf_counter() - t0)
        return ts

    if have_aff:
        os.sched_setaffinity(0, {cpus[len(cpus) // 2]})  # one core, away from core 0's interrupt load
    try:
        t1 = run(1)
    finally:
        if have_aff:
            os.sched_setaffinity(0, set(cpus))
    all_threads = min(len(cpus), 64)
    tall = run(all_threads) if all_threads > 1 else t1
    torch.set_num_threads(all_threads)
    return {"best_1": units / min(t1), "median_1": units / float(np.median(t1)), "best_all": units / min(tall),
            "median_all": units / float(np.median(tall)), "threads_all": all_threads, "spread_1": (max(t1) - min(t1)) / min(t1)}


def _cpu_record(t, sample):
    """Headline = the better of the two configurations (best of 5); both, with their medians, beside it."""
    one_wins = t["best_1"] >= t["best_all"]
    return {"value": t["best_1"] if one_wins else t["best_all"], "unit": "frames/s", "cores": 1 if one_wins else t["threads_all"],
            "kind": "port", "sample": sample,
            "frames_per_s_1_thread": {"best_of_5": t["best_1"], "median_of_5": t["median_1"], "repeat_spread": t["spread_1"]},
            "frames_per_s_all_cores": {"best_of_5": t["best_all"], "median_of_5": t["median_all"], "threads": t["threads_all"]},
            "host_cpu": _host_cpu_model(), "host_logical_cpus": os.cpu_count() or 1}


def cpu_baseline(d, sf_sd):
    """The CPU oracle on the SAME workload as the GPU step: all 64 pairs, every kernel of the step."""
    from oracle import hdn_oracle as O

    n = PAIRS
    c = lambda t: t[:n].detach().cpu()
    nx, nk = c(d["north_x"]), c(d["north_k"])
    px, pk = [c(t) for t in d["prod_x"]], [c(t) for t in d["prod_k"]]
    cx, ck = [c(t) for t in d["circ_x"]], [c(t) for t in d["circ_k"]]
    imgs, h4p, off = c(d["imgs"]), c(d["h4p"]), c(d["off"])

    def one_pass():
        with torch.no_grad():
            O.xcorr_depthwise(nx, nk)
            for a, b in zip(px, pk):
                O.xcorr_depthwise(a, b)
            for a, b in zip(cx, ck):
                O.xcorr_depthwise_circular(a, b)
            p1 = O.share_feature(imgs[:, :1], sf_sd)
            p2 = O.share_feature(imgs[:, 1:], sf_sd)
            _, w = O.dlt_warp(h4p, off, imgs[:, :1])
            pf = O.share_feature(w, sf_sd)
            (p2 - pf).abs()[0][0].sum() / (127 * 127)
            (p2 - p1).abs()[0][0].sum() / (127 * 127)

    return _cpu_record(_timed_cpu(one_pass, n),
                       f"all {n} pairs of the step, every kernel of the step; warm-up 3, 5 timed passes; headline = the better of "
                       "1 thread pinned to one core (the reference's torch.set_num_threads(1), tools/test.py:51) and all cores; "
                       "oracle/hdn_oracle.py, PyTorch-CPU fp32")


def cpu_full_head(d, net_sd):
    """configs[2] on the host: the oracle's track_proj stages around a PyTorch-CPU ResNet-34 with the same weights,
    on 16 of the 64 pairs (the trunk is ~36 ms per pair on one core)."""
    import hdn_amd
    from oracle import hdn_oracle as O

    n = 16
    net = hdn_amd.HomoModelBuilder().eval()
    net.load_state_dict(net_sd)
    sf_sd = {k: v for k, v in net.ShareFeature.state_dict().items()}
    imgs = d["imgs"][:n].detach().cpu()
    data = {"org_imgs": imgs, "input_tensors": imgs, "h4p": d["h4p"][:n].detach().cpu()}
    regress = lambda f: net.fc(net.avgpool(net.backbone(f)).flatten(1))

    def one_pass():
        with torch.no_grad():
            O.track_proj(data, sf_sd, regress)

    return _cpu_record(_timed_cpu(one_pass, n, warm=1, reps=3),
                       f"{n} of the 64 pairs, whole head incl. the ResNet-34 trunk on PyTorch-CPU; warm-up 1, 3 timed passes; "
                       "headline = the better of 1 pinned thread and all cores")


if __name__ == "__main__":
    main()
