#!/usr/bin/env python3
"""BASELINE configs[3] on the synthetic sequence of tools/synth_sequence.py (no POT data, no cv2 here): a 1280x720
planar-target sequence streamed through the device-resident tracker loop (hdn_amd.tracker: upload the uint8 frame once,
full-frame warp, crops, networks, decodes, track_proj, 3x3 bookkeeping, ONE host read of the 4 corners per frame).

Three model choices for the similarity branch:
    (default)            none: the similarity estimate is the identity (the homography half of a frame alone)
    --similarity         tests/standin_model.py: a 16-channel, one-convolution-per-level toy (everything of a frame EXCEPT the networks)
    --production-shape   tests/production_standin.py: ResNet-50 / stride 8 / dilated backbone, 1x1 necks, 256-channel heads — the
                         shapes of the shipped configuration with seeded weights, driven through DeviceTrackerHomo(model), the
                         object install(tracker=True) registers.  This is the honest end-to-end frame: two backbone passes
                         (PyTorch-ROCm / MIOpen, as north_star assigns them) + everything hand-written around them.

Reported: wall time per frame with the host reading the corners EVERY frame (what tools/test.py's loop sees, :115-174), eager
and as one hipGraph per frame; with --production-shape also a per-component table (every stage of the frame body captured as
its own hipGraph and replayed) and the share of the frame the hand-written kernels own; corner error (success_4pts_error) of
the device loop against the CPU restatement of the same loop on the first --parity frames, free-running and with the
recurrence state (H_total) re-synchronised to the CPU loop's before every frame.

    python tests/tools/sequence_bench.py --production-shape [--frames 501] [--parity 61] [--nchw]
"""
import argparse, copy, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import hdn_amd
from hdn_amd import _lib, frame as FR
from hdn_amd.tracker import DeviceTrackerHomo, HomoTracker
from synth_sequence import LONG_WALK, make_sequence, success_4pts_error


def seeded_net(fc_bias_scale=1.0):
    torch.manual_seed(1)
    net = hdn_amd.HomoModelBuilder().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.8, 1.2)
    net.fc.weight.data.mul_(0.01)
    net.fc.bias.data.mul_(fc_bias_scale)
    return net


# --------------------------------------------------------------------------------------------------- timing helpers
def graph_ms(fn, reps=20, warm=3, inner=10):
    """fn() captured `inner` times back to back in ONE hipGraph (static inputs), replayed `reps` times: mean ms per fn().  (A
    replay has a fixed cost of ~10 us on this stack whatever the graph holds: one fn() per graph would put it into every row.)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [fn() for _ in range(inner)]
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    del keep
    return a.elapsed_time(b) / (reps * inner)


def stream_loop(new_tracker, frames, sync, graph=False, warm=6):
    """The tracker over the whole sequence: ms per frame (wall), host syncs per frame, p99 per-frame latency."""
    t = new_tracker(graph)
    for i in range(1, warm):
        t.track_new(i, frames[i])                    # warm-up (MIOpen find, clocks, graph capture)
    t.H_total.copy_(torch.eye(3, dtype=torch.float64, device=t.dev)); torch.cuda.synchronize(); s0 = t.host_syncs
    t0 = time.perf_counter(); per = []
    n = len(frames)
    for i in range(1, n):
        a = time.perf_counter(); t.track_new(i, frames[i], sync=sync); per.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n - 1) * 1e3, (t.host_syncs - s0) / (n - 1), float(np.percentile(per, 99) * 1e3), t


# --------------------------------------------------------------------------------------------------- production-shaped model
def build_production_model(frames, init, dev, nchw=False, instance_size=255, crops="product"):
    """tests/production_standin.ProductionStandIn on `dev`, BatchNorm statistics calibrated on crops of the first frames.
    crops="product": the calibration crops are cut by hdn_amd.frame on the device (bench.py: no oracle outside its cpu_baseline
    leg); "oracle": by oracle/frame_oracle.py (bit-identical crops; the CPU twin is then built from the same weights)."""
    import production_standin as PS
    net = seeded_net(0.1)
    twin = PS.ProductionStandIn(net, instance_size=instance_size)
    if crops == "oracle":
        c255, c127 = PS.calibration_crops(frames, init)
    else:
        poly = init["poly"]
        s_z = float(np.floor(np.sqrt((poly[2] + 0.5 * (poly[2] + poly[3])) * (poly[3] + 0.5 * (poly[2] + poly[3])))))
        c255, c127 = [], []
        for f in frames[:3]:
            fr, avg = FR.upload(f), np.mean(f, axis=(0, 1))
            c255.append(FR.get_subwindow(fr, poly[:2], 255, float(np.floor(2 * s_z)), avg)[0].cpu())
            c127.append(FR.get_subwindow(fr, poly[:2], 127, s_z, avg)[0].cpu())
        c255, c127 = torch.stack(c255), torch.stack(c127)
    twin.calibrate(c255, c127)
    cpu_twin_src = copy.deepcopy(twin) if crops == "oracle" else None
    twin = twin.to(dev).eval()
    twin.hm_net.optimize_for_inference(channels_last=True)
    if not nchw:
        twin.backbone.to(memory_format=torch.channels_last)
        twin.neck.to(memory_format=torch.channels_last); twin.neck_lp.to(memory_format=torch.channels_last)
    return twin, cpu_twin_src


def component_table(trk, frame_u8, reps=20):
    """Every stage of DeviceTrackerHomo's frame body as its own hipGraph on static inputs (the outputs of the stages before it,
    computed once; 10 executions per graph, see graph_ms), replayed `reps` times: (name, ms, owner) rows.  owner: 'hip' = hand-written kernels of this repo only,
    'rocm' = PyTorch-ROCm / MIOpen / hipBLASLt only, 'mixed' = a packed head whose convolutions / matrix products fell back to the
    libraries (since round 4 the 256-channel heads are all HIP: conv_search, correlations and tail are three launches; the correlation
    launch is timed again on its own in the row below each head)."""
    from hdn_amd.refine import homo_refine
    from hdn_amd.xcorr import xcorr_depthwise_multi
    sim, model, c, dev = trk.similarity, trk.model, trk.cfg, trk.dev
    lib, st = _lib.load(), (lambda: _lib.stream_ptr(dev))
    dec = sim.decoder(dev)
    static = torch.empty_like(FR.upload(frame_u8)); static.copy_(FR.upload(frame_u8))
    rows = []

    def add(name, fn, owner):
        with torch.no_grad():
            rows.append((name, graph_ms(fn, reps), owner))

    # host -> device copy of the frame and device -> host read of the result (eager, synchronised: PCIe, not kernels)
    host = torch.from_numpy(np.ascontiguousarray(frame_u8))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        static.copy_(host, non_blocking=True); torch.cuda.synchronize()
    rows.append(("frame upload (2.76 MB pageable, synchronised)", (time.perf_counter() - t0) / 50 * 1e3, "pcie"))

    def prepare_and_warp():
        _lib.check(lib.hdn_track_prepare_f64(_lib.ptr(trk.H_total), _lib.ptr(trk._Ht), _lib.ptr(trk._Hinv), 1, st()), "prepare")
        return FR.warp_perspective(static, trk._Hinv.view(-1))
    add("singular check + inverse, stabilising full-frame warp", prepare_and_warp, "hip")
    stab = prepare_and_warp()
    add("search crop 255 px (crop + pad + resize)", lambda: FR.get_subwindow(stab, None, c.instance_size, None, None, params=sim._params0), "hip")
    x_crop = FR.get_subwindow(stab, None, c.instance_size, None, None, params=sim._params0)
    add("backbone (ResNet-50, 255 px) + necks", lambda: model.neck(model.feature_extractor(x_crop)), "rocm")
    with torch.no_grad():
        xf = model.neck(model.feature_extractor(x_crop))
        out = model.track_new(x_crop)
    def head_owner(head):   # all-HIP when the packed head runs its convolutions and its tail on the HIP kernels (hdn_head_conv3x3_f32, hdn_head_tail_f32)
        pk = getattr(head, "_hdn_packed_head", None)
        return "hip" if pk is not None and pk.wsp is not None and pk.w1p is not None else "mixed"
    add("MultiBAN head, packed (2 x 3 levels)", lambda: model.head(model.zf, xf), head_owner(model.head))
    s6 = [torch.randn(1, 256, 29, 29, device=dev).relu_() for _ in range(6)]
    k6 = [torch.randn(1, 256, 5, 5, device=dev).relu_() for _ in range(6)]
    o6 = [torch.empty(1, 256, 25, 25, device=dev) for _ in range(6)]
    add("  of which: the 6 correlations 5x5 (x) 29x29 (one launch)", lambda: xcorr_depthwise_multi(s6, k6, circular=False, outs=o6), "hip")
    add("translation decode (softmax, window, argmax, gate)", lambda: dec.translation(out["cls"], out["loc_c"], sim.seq, sim.state), "hip")
    row = sim.state.view(-1)
    add("moved crop 255 px", lambda: FR.get_subwindow(stab, None, c.instance_size, None, None, params=row[8:14]), "hip")
    x_moved = FR.get_subwindow(stab, None, c.instance_size, None, None, params=row[8:14])
    polar = torch.zeros((1, 2), dtype=torch.float32, device=dev)
    add("log-polar sampler 255 -> 127 px", lambda: model.logpolar_instance(x_moved, polar, [0, 0]), "hip")
    x_lp, _ = model.logpolar_instance(x_moved, polar, [0, 0])
    add("backbone (ResNet-50, 127 px) + necks", lambda: model.neck_lp(model.feature_extractor(x_lp)), "rocm")
    with torch.no_grad():
        xf_lp = model.neck_lp(model.feature_extractor(x_lp))
        out_lp = model.track_new_lp(x_moved, [0, 0])
    add("MultiCircBAN head, packed", lambda: model.head_lp(model.zf_lp, xf_lp), head_owner(model.head_lp))
    s6 = [torch.randn(1, 256, 13, 13, device=dev).relu_() for _ in range(6)]
    k6 = [torch.randn(1, 256, 13, 13, device=dev).relu_() for _ in range(6)]
    o6 = [torch.empty(1, 256, 13, 13, device=dev) for _ in range(6)]
    add("  of which: the 6 circular correlations 13x13 (one launch)", lambda: xcorr_depthwise_multi(s6, k6, circular=True, outs=o6), "hip")
    add("log-polar decode, H_sim, crop / rotation records", lambda: dec.logpolar(out_lp["cls_lp"], out_lp["loc_lp"], sim.seq, sim.state), "hip")
    from hdn_amd.similarity import state_fields
    f = state_fields(row)
    add("rotate back (bicubic full-frame warp)", lambda: FR.warp_affine_cubic(stab, f["rot_matrix"]), "hip")
    rot = FR.warp_affine_cubic(stab, f["rot_matrix"])
    add("homography crop 127 px (crop + resize + normalise)", lambda: FR.get_search_info(rot, None, None, None, model_sz=c.exemplar_size, params=f["params_homo"]), "hip")
    search = FR.get_search_info(rot, None, None, None, model_sz=c.exemplar_size, params=f["params_homo"])
    add("homography estimator (PreShareFeature x3, ResNet-34 trunk, DLT, warp, scores, refinement warp)",
        lambda: homo_refine(trk.net, trk.init_homo_tmp, search, iterations=trk.iterations), "hip")
    with torch.no_grad():
        H_comp, score, _ = homo_refine(trk.net, trk.init_homo_tmp, search, iterations=trk.iterations)
    sc = score.detach().reshape(-1).to(torch.float32).contiguous()
    Hout = torch.empty_like(trk.H_total)

    def accumulate():
        _lib.check(lib.hdn_track_accumulate_f64(_lib.ptr(trk._Ht), _lib.ptr(sim.state), _lib.ptr(H_comp), _lib.ptr(sc), _lib.ptr(trk._consts),
                                                _lib.ptr(trk.init_points), trk.init_points.shape[1], _lib.ptr(Hout), _lib.ptr(trk._out), 1, st()), "acc")
    add("un-scale / un-shift, gate, accumulate, project the corners", accumulate, "hip")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        trk._out.cpu()
    rows.append(("host read of 4 corners + score (synchronising)", (time.perf_counter() - t0) / 200 * 1e3, "pcie"))
    return rows


def parity(frames, init, model_cpu_src, dev, model, n, iterations=1):
    """Device loop vs the CPU restatement on frames 1..n-1: free-running, and with H_total re-synchronised before every frame."""
    import production_standin as PS
    from oracle.tracker_oracle import HomoTrackerOracle, SimilarityOracle
    cpu = PS.ProductionStandInCPU(model_cpu_src)
    net_cpu = copy.deepcopy(model_cpu_src.hm_net).cpu().eval()
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    ref = HomoTrackerOracle(sd, lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1)), iterations, similarity=SimilarityOracle(cpu))
    free, forced = DeviceTrackerHomo(model, graph=True, iterations=iterations), DeviceTrackerHomo(model, graph=True, iterations=iterations)
    for t in (ref, free, forced):
        t.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    e_free, e_forced = [], []
    for i in range(1, n):
        forced.H_total.copy_(torch.from_numpy(np.asarray(ref.H_total, np.float64)).to(dev))
        b = ref.track_new(i, frames[i])
        e_free.append(success_4pts_error(free.track_new(i, frames[i])["points"], b["points"]))
        e_forced.append(success_4pts_error(forced.track_new(i, frames[i])["points"], b["points"]))
    q = lambda e: {"first": e[0], "median": float(np.median(e)), "max": max(e), "frames": len(e)}
    return {"free_running": q(e_free), "h_total_resynchronised_each_frame": q(e_forced)}


def kernel_profile(trk, frames, n=20):
    """torch.profiler over `n` frames of the hipGraph loop: device time per kernel name, split into the hand-written hdn::
    kernels of this repo and everything else (MIOpen / CK / hipBLASLt / ATen), per frame."""
    from torch.profiler import ProfilerActivity, profile
    for i in range(1, 4):
        trk.track_new(i, frames[i])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(4, 4 + n):
            trk.track_new(i, frames[i])
        torch.cuda.synchronize()
    agg = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            t = e.device_time if hasattr(e, "device_time") else e.cuda_time
            a = agg.setdefault(e.name, [0.0, 0])
            a[0] += t; a[1] += 1
    own = sum(t for k, (t, c) in agg.items() if "hdn::" in k)
    cp = sum(t for k, (t, c) in agg.items() if "memcpy" in k.lower() or "copyBuffer" in k)
    tot = sum(t for t, c in agg.values())
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]
    return {"frames": n, "device_ms_per_frame": tot / n / 1e3, "hdn_kernels_ms_per_frame": own / n / 1e3, "copies_ms_per_frame": cp / n / 1e3,
            "library_kernels_ms_per_frame": (tot - own - cp) / n / 1e3, "hdn_kernel_launches_per_frame": sum(c for k, (t, c) in agg.items() if "hdn::" in k) / n,
            "launches_per_frame": sum(c for t, c in agg.values()) / n,
            "top": [{"kernel": k[:120], "us_per_frame": round(t / n, 2), "calls_per_frame": round(c / n, 2)} for k, (t, c) in top]}


def run_production(n_frames=501, n_parity=0, nchw=False, components=True, dev=None, quiet=False, kprofile=False):
    dev = dev or torch.device("cuda:0")
    log = (lambda *a: None) if quiet else (lambda *a: print(*a, file=sys.stderr, flush=True))
    t0 = time.perf_counter()
    frames, corners, init = make_sequence(n_frames=n_frames, frame_hw=(720, 1280), target_wh=(300, 200), **LONG_WALK)
    log(f"[sequence] {n_frames} frames generated in {time.perf_counter() - t0:.1f} s")
    torch.backends.cudnn.benchmark = os.environ.get("HDN_SEQ_FIND", "1") != "0"     # MIOpen find mode (A/B switch: HDN_SEQ_FIND=0)
    t0 = time.perf_counter()
    model, cpu_src = build_production_model(frames, init, dev, nchw=nchw, crops="oracle" if n_parity else "product")

    def new_tracker(graph=False):
        t = DeviceTrackerHomo(model, graph=graph)
        t.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
        return t
    ms_graph, syncs, p99_graph, trk = stream_loop(new_tracker, frames, True, graph=True)
    log(f"[sequence] model built, graph loop over {n_frames} frames: {ms_graph:.3f} ms per frame ({time.perf_counter() - t0:.1f} s incl. MIOpen find)")
    ms_eager, _, p99_eager, _ = stream_loop(new_tracker, frames[:min(n_frames, 121)], True, graph=False)
    res = {"sequence": f"{n_frames} frames 1280x720, 300x200 target, seed 20260928, LONG_WALK (tools/synth_sequence.py)",
           "model": "tests/production_standin.py: ResNet-50 stride-8 dilated backbone x2 per frame (255 px, 127 px) + 1x1 necks + 256-channel MultiBAN / "
                    "MultiCircBAN + ResNet-34 homography estimator; seeded weights; fp32; " + ("NCHW" if nchw else "channels-last") + ", MIOpen find mode",
           "tracker": "hdn_amd.tracker.DeviceTrackerHomo(model) — the object install(tracker=True) registers; one hipGraph per frame, host reads 4 corners + score every frame",
           "frames": n_frames, "ms_per_frame": ms_graph, "fps": 1e3 / ms_graph, "p99_ms": p99_graph, "host_syncs_per_frame": syncs,
           "ms_per_frame_eager": ms_eager, "p99_ms_eager": p99_eager, "reference_host_syncs_per_frame": 6}
    if components:
        rows = component_table(trk, frames[1])
        own = {"hip": 0.0, "rocm": 0.0, "mixed": 0.0, "pcie": 0.0}
        corr = 0.0
        for name, ms, owner in rows:
            if name.startswith("  of which"):
                corr += ms
            else:
                own[owner] += ms
        total = sum(own.values())
        res["components"] = [{"stage": n, "ms": round(ms, 4), "owner": o} for n, ms, o in rows]
        res["component_sum_ms"] = total
        res["share"] = {"hand_written_hip_stages": own["hip"] / total, "pytorch_rocm_backbone_and_necks": own["rocm"] / total,
                        "packed_heads_mixed": own["mixed"] / total, "of_which_hip_correlations": corr / total, "pcie_copies": own["pcie"] / total}
    if kprofile:
        res["kernel_profile"] = kernel_profile(trk, frames)
    if n_parity:
        res["corner_error_device_vs_cpu_loop_px"] = parity(frames, init, cpu_src, dev, model, n_parity)
    return res


# --------------------------------------------------------------------------------------------------- n sequences in lock step
def init_from_corners(c0, target_wh):
    """The init record tools/synth_sequence.make_sequence builds for frame 0, for ANY frame of the sequence (its ground-truth corners)."""
    tw, th = target_wh
    c0 = np.asarray(c0, np.float32)
    return {"bbox": [float(c0[:, 0].min()), float(c0[:, 1].min()), float(tw), float(th)],
            "poly": [float(c0[:, 0].mean()), float(c0[:, 1].mean()), float(tw), float(th), 0.0],
            "gt_points": c0.reshape(-1).tolist(), "first_point": c0[0].tolist()}


def run_multi(ns=(1, 4, 16, 32), n_steps=60, dev=None, quiet=False, offset=3, distinct_steps=8, kprofile_n=None, simi=False):
    """hdn_amd.batched_tracker.BatchedDeviceTracker(model, n) over n sequences in lock step on ONE GPU, production-shaped model.
    Sequence b = the synthetic 1280x720 sequence started `offset * b` frames later (its own first frame, template, corners, H_total).
    A step's n frames arrive as ONE pinned uint8 [n,720,1280,3] buffer (what a decoder / loader thread hands over); the timed loop
    does, per step: one host->device copy, one hipGraph replay, one host read of [n, 9].  `distinct_steps` pinned step buffers are
    cycled (the seeded stand-in does not track, so which frame follows which does not matter for the timing).
    simi=True: hdn_amd.simi_tracker.BatchedSimiTracker (the similarity-only tracker with its per-step template refresh) the same way.
    -> rows {n, ms_per_step, frames_per_s, ms_per_step_frames_resident, ...}."""
    from hdn_amd.batched_tracker import BatchedDeviceTracker
    from hdn_amd.simi_tracker import BatchedSimiTracker
    from hdn_amd import backbone as BB
    dev = dev or torch.device("cuda:0")
    log = (lambda *a: None) if quiet else (lambda *a: print(*a, file=sys.stderr, flush=True))
    nmax = max(ns)
    target = (300, 200)
    frames, corners, init = make_sequence(n_frames=distinct_steps + 1 + (nmax - 1) * offset, frame_hw=(720, 1280), target_wh=target, **LONG_WALK)
    torch.backends.cudnn.benchmark = os.environ.get("HDN_SEQ_FIND", "1") != "0"
    model, _ = build_production_model(frames, init, dev)
    rows = []
    for n in ns:
        t0 = time.perf_counter()
        offs = [b * offset for b in range(n)]
        inits = [init_from_corners(corners[o], target) for o in offs]
        if simi:
            BB.optimize_similarity_model(model)
            trk = BatchedSimiTracker(model, n, graph=True)
            trk.init([frames[o] for o in offs], [i["bbox"] for i in inits], [i["poly"] for i in inits], [np.array([i["first_point"]]) for i in inits])
        else:
            trk = BatchedDeviceTracker(model, n)
            trk.init([frames[o] for o in offs], [i["bbox"] for i in inits], [i["poly"] for i in inits], [i["gt_points"] for i in inits])
        steps = [torch.from_numpy(np.stack([frames[o + 1 + k] for o in offs])).pin_memory() for k in range(distinct_steps)]
        for k in range(6):                                  # MIOpen find at this batch size, graph capture, clocks
            trk.track_new(k, steps[k % distinct_steps])
        torch.cuda.synchronize(); s0 = trk.host_syncs
        per = []
        t1 = time.perf_counter()
        for k in range(n_steps):
            a = time.perf_counter(); res = trk.track_new(k, steps[k % distinct_steps]); per.append(time.perf_counter() - a)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / n_steps * 1e3
        syncs_per_step = (trk.host_syncs - s0) / n_steps
        assert len(res) == n and all(np.isfinite(r["polygon" if simi else "points"]).all() for r in res)
        # the same with the frames already on the device (no PCIe in the step): what the kernels + networks alone sustain
        dsteps = [s.to(dev) for s in steps[:4]]
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for k in range(n_steps):
            trk.track_new(k, dsteps[k % len(dsteps)])
        torch.cuda.synchronize()
        ms_res = (time.perf_counter() - t2) / n_steps * 1e3
        row = {"n": n, "ms_per_step": ms, "frames_per_s": n * 1e3 / ms, "ms_per_frame": ms / n, "p99_ms_per_step": float(np.percentile(per, 99) * 1e3),
               "host_syncs_per_step": syncs_per_step, "graph": trk._graph is not None,
               "ms_per_step_frames_resident": ms_res, "frames_per_s_frames_resident": n * 1e3 / ms_res,
               "upload_MB_per_step": steps[0].numel() / 1e6}
        if kprofile_n is not None and n == kprofile_n:
            row["kernel_profile"] = kernel_profile(trk, [None] + [dsteps[k % len(dsteps)] for k in range(30)], n=20)
        rows.append(row)
        log(f"[multi] n={n:3d}: {ms:8.3f} ms per step = {n * 1e3 / ms:8.1f} frames/s  (frames resident: {ms_res:8.3f} ms = {n * 1e3 / ms_res:8.1f} frames/s)  "
            f"graph={row['graph']}  [{time.perf_counter() - t0:.1f} s]")
        del trk, steps, dsteps
        torch.cuda.empty_cache()
    base = next((r for r in rows if r["n"] == 1), None)
    if base is not None:
        for r in rows:
            r["speedup_vs_n1"] = r["frames_per_s"] / base["frames_per_s"]
    return {"tracker": ("hdn_amd.simi_tracker.BatchedSimiTracker(model, n): n independent sequences of the similarity-only tracker (template refreshed every "
                        "step) advance one frame per step; one pinned [n,720,1280,3] upload, one hipGraph replay, one host read of [n, 20] per step") if simi else
                       "hdn_amd.batched_tracker.BatchedDeviceTracker(model, n): n independent sequences advance one frame per step; one pinned "
                       "[n,720,1280,3] upload, one hipGraph replay, one host read of [n, 9] per step",
            "model": "tests/production_standin.py (ResNet-50 stride-8 dilated backbone x2, 256-channel heads, ResNet-34 estimator; seeded; fp32; channels-last; MIOpen find)",
            "steps_timed": n_steps, "rows": rows}



def run_simi(n_frames=201, dev=None):
    """hdn_amd.simi_tracker.DeviceTrackerSimi(model) — what install(tracker=True) registers under TRACKS['hdnTracker'] — over the synthetic 1280x720
    sequence with the production-shaped model; the host reads the result every frame."""
    from hdn_amd.simi_tracker import DeviceTrackerSimi
    dev = dev or torch.device("cuda:0")
    frames, corners, init = make_sequence(n_frames=n_frames, frame_hw=(720, 1280), target_wh=(300, 200), **LONG_WALK)
    model, _ = build_production_model(frames, init, dev)
    out = {"tracker": "hdn_amd.simi_tracker.DeviceTrackerSimi (TRACKS['hdnTracker']): search crop, backbone, head, decode, moved crop, backbone, log-polar head, decode, "
                      "recurrences + polygon, template refresh (rotate the first frame, crop, the two template passes as one batch of 2 unless HDN_SIMI_BATCH_TEMPLATE=0); one host read of 20 doubles per frame",
           "frames": n_frames}
    for graph in (True, False):
        t = DeviceTrackerSimi(model, graph=graph)
        t.init(frames[0], init["bbox"], init["poly"], np.array([init["first_point"]]))
        for i in range(1, 8):
            t.track_new(i, frames[i])
        torch.cuda.synchronize(); s0 = t.host_syncs
        n = n_frames if graph else min(n_frames, 81)
        t0 = time.perf_counter()
        for i in range(8, n):
            r = t.track_new(i, frames[i])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (n - 8) * 1e3
        assert np.isfinite(r["polygon"]).all()
        out["ms_per_frame" + ("" if graph else "_eager")] = ms
        if graph:
            out["fps"], out["host_syncs_per_frame"], out["graph"] = 1e3 / ms, (t.host_syncs - s0) / (n - 8), t._graph is not None
    return out


def format_table(res):
    lines = [f"{res['frames']} frames, {res['ms_per_frame']:.3f} ms per frame as one hipGraph ({res['fps']:.0f} frames/s), {res['ms_per_frame_eager']:.3f} ms eager",
             f"{'stage':<100s} {'ms':>8s}  owner"]
    for r in res.get("components", []):
        lines.append(f"{r['stage']:<100s} {r['ms']:8.4f}  {r['owner']}")
    if "share" in res:
        lines.append(f"{'sum of the stage graphs':<100s} {res['component_sum_ms']:8.4f}")
        lines.append("share: " + ", ".join(f"{k} {v * 100:.1f} %" for k, v in res["share"].items()))
    if "kernel_profile" in res:
        k = res["kernel_profile"]
        lines.append(f"torch.profiler over {k['frames']} graph frames: device time {k['device_ms_per_frame']:.3f} ms per frame in {k['launches_per_frame']:.0f} launches = "
                     f"hdn:: kernels {k['hdn_kernels_ms_per_frame']:.3f} ms ({k['hdn_kernel_launches_per_frame']:.0f} launches) + library kernels "
                     f"{k['library_kernels_ms_per_frame']:.3f} ms + copies {k['copies_ms_per_frame']:.3f} ms")
        for r in k["top"]:
            lines.append(f"  {r['kernel']:<120s} {r['us_per_frame']:9.2f} us  x{r['calls_per_frame']}")
    if "corner_error_device_vs_cpu_loop_px" in res:
        lines.append("corner error vs the CPU loop (px): " + json.dumps(res["corner_error_device_vs_cpu_loop_px"]))
    return "\n".join(lines)


# --------------------------------------------------------------------------------------------------- the earlier modes
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=None); ap.add_argument("--parity", type=int, default=None)
    ap.add_argument("--iterations", type=int, default=1); ap.add_argument("--similarity", action="store_true")
    ap.add_argument("--production-shape", action="store_true"); ap.add_argument("--nchw", action="store_true")
    ap.add_argument("--no-components", action="store_true", help="skip the per-stage table (profiling runs)")
    ap.add_argument("--kernel-profile", action="store_true", help="torch.profiler over 20 graph frames: device time per kernel, hdn:: vs library")
    ap.add_argument("--simi-tracker", action="store_true", help="TRACKS['hdnTracker'] (hdn_amd.simi_tracker.DeviceTrackerSimi) around the production-shaped model: "
                    "ms per frame as one hipGraph and eagerly (4 backbone passes per frame: two searches + the template refresh's two)")
    ap.add_argument("--multi", type=str, default=None, help="comma-separated n: BatchedDeviceTracker over n sequences in lock step (production-shaped model)")
    ap.add_argument("--multi-profile", type=int, default=None, help="with --multi: torch.profiler kernel table at this n")
    args = ap.parse_args()
    if args.simi_tracker and not args.multi:
        res = run_simi(args.frames or 201)
        print(json.dumps(res))
        return
    if args.multi:
        res = run_multi(tuple(int(x) for x in args.multi.split(",")), n_steps=args.frames or 60, kprofile_n=args.multi_profile, simi=args.simi_tracker)
        for r in res["rows"]:
            k = r.get("kernel_profile")
            if k:
                print(f"n={r['n']}: device time {k['device_ms_per_frame']:.3f} ms per step in {k['launches_per_frame']:.0f} launches = hdn:: {k['hdn_kernels_ms_per_frame']:.3f} ms "
                      f"({k['hdn_kernel_launches_per_frame']:.0f}) + library {k['library_kernels_ms_per_frame']:.3f} ms + copies {k['copies_ms_per_frame']:.3f} ms", file=sys.stderr)
                for t in k["top"]:
                    print(f"  {t['kernel']:<120s} {t['us_per_frame']:9.2f} us  x{t['calls_per_frame']}", file=sys.stderr)
        print(json.dumps(res))
        return
    if args.production_shape:
        res = run_production(args.frames or 501, 61 if args.parity is None else args.parity, nchw=args.nchw, components=not args.no_components, kprofile=args.kernel_profile)
        print(format_table(res), file=sys.stderr)
        print(json.dumps(res))
        return
    args.frames, args.parity = args.frames or 200, 12 if args.parity is None else args.parity
    from hdn_amd.graph import GraphedTrackProj
    from oracle.tracker_oracle import HomoTrackerOracle, SimilarityOracle
    dev = torch.device("cuda:0")
    frames, corners, init = make_sequence(n_frames=args.frames, frame_hw=(720, 1280), target_wh=(300, 200))
    net = seeded_net(); net_cpu = copy.deepcopy(net)
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    sim_cpu = twin = None
    if args.similarity:
        import standin_model as SM
        from hdn_amd.similarity import DeviceSimilarity
        net.fc.bias.data.mul_(0.1); net_cpu = copy.deepcopy(net)
        twin = SM.StandInSiamese(net, loc_scale_lp=0.01).eval()
        sim_cpu = SimilarityOracle(SM.StandInSiameseCPU(twin))
        twin = twin.to(dev)
    netd = net.to(dev)
    torch.backends.cudnn.benchmark = True
    netd.optimize_for_inference(channels_last=True)

    def new_tracker(graph=False):
        t = HomoTracker(netd, iterations=args.iterations, graph=graph, similarity=DeviceSimilarity(twin) if twin is not None else None)
        t.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
        return t

    # parity of the whole per-frame chain, device vs CPU restatement
    ref = HomoTrackerOracle(sd, lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1)), args.iterations, similarity=sim_cpu)
    ref.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    trk = new_tracker()
    errs = [success_4pts_error(trk.track_new(t, frames[t])["points"], ref.track_new(t, frames[t])["points"])
            for t in range(1, min(args.parity, args.frames))]

    ms_sync, syncs, p99, _ = stream_loop(new_tracker, frames, True)
    ms_async, _, _, _ = stream_loop(new_tracker, frames, False)
    ms_graph, _, p99_graph, _ = stream_loop(new_tracker, frames, True, graph=True)

    # the B=1 head alone (track_proj), eager vs hipGraph, synchronised EVERY frame as the tracker's score read does
    t = new_tracker()
    tmpl = t.init_homo_tmp
    searches = [FR.get_search_info(FR.upload(frames[i]), t.init_pos, t.init_s_z_sm, t.channel_average) for i in range(1, min(60, args.frames))]
    h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32, device=dev)
    pidx = torch.arange(127 * 127, dtype=torch.float32, device=dev).unsqueeze(0)
    dd = [{"org_imgs": torch.cat((tmpl, s), 1), "input_tensors": torch.cat((tmpl, s), 1), "h4p": h4p, "patch_indices": pidx} for s in searches]

    def head(fn):
        for fd in dd[:5]: float(fn(fd)[1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for fd in dd: float(fn(fd)[1])                                # per-frame sync: latency, not pipelined throughput
        return (time.perf_counter() - t0) / len(dd) * 1e3
    eager = head(lambda fd: netd.track_proj(fd, None))
    gr = GraphedTrackProj(netd, dd[0], template_constant=True)
    graphed = head(gr)
    print(json.dumps({
        "sequence": f"{args.frames} frames 1280x720, 300x200 target, seed 20260928 (tools/synth_sequence.py)",
        "refinement_iterations": args.iterations, "similarity_branch": "stand-in heads (tests/standin_model.py), device decode" if args.similarity else "identity",
        "ms_per_frame_loop_host_reads_corners_each_frame": ms_sync, "p99_ms": p99, "fps": 1e3 / ms_sync,
        "host_syncs_per_frame": syncs, "reference_host_syncs_per_frame": 6,
        "ms_per_frame_loop_no_host_read": ms_async,
        "ms_per_frame_loop_one_hipgraph_per_frame": ms_graph, "p99_ms_hipgraph": p99_graph, "fps_hipgraph": 1e3 / ms_graph,
        "corner_error_device_vs_cpu_loop_px": {"first": errs[0], "max": max(errs), "frames": len(errs)},
        "head_only_ms_per_frame_eager_synced": eager, "head_only_ms_per_frame_hipgraph_synced": graphed}))


if __name__ == "__main__":
    main()
