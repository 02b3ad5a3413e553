#!/usr/bin/env python3
"""BASELINE configs[3] on the synthetic sequence of tools/synth_sequence.py (no POT data, no cv2 here): a 1280x720
planar-target sequence streamed through the device-resident tracker loop (hdn_amd.tracker.HomoTracker: upload the uint8
frame once, full-frame warp, crop + normalise, track_proj, 3x3 bookkeeping, ONE host read of the 4 corners per frame).

Reported: wall time per frame with the host reading the corners EVERY frame (what tools/test.py's loop sees, :115-174),
host syncs per frame, the same loop without the per-frame read (pipelined), corner error (success_4pts_error) of the device
loop against the CPU restatement of the same loop on the first --parity frames, and — for the B=1 head alone — eager vs
hipGraph replay, each synchronised per frame.
--similarity runs the loop WITH the similarity branch (crop, heads, device decode, moved crop, log-polar heads, decode, H_sim,
rotate-back) around tests/standin_model.py's seeded stand-in for the reference's ModelBuilder: the ResNet-50 backbone of the
deployment is PyTorch-ROCm's and is not shipped here, so these times cover everything of a frame EXCEPT two backbone passes.
    python tests/tools/sequence_bench.py [--frames 200] [--parity 12] [--similarity]
"""
import argparse, copy, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hdn_amd
from hdn_amd.graph import GraphedTrackProj
from hdn_amd.tracker import HomoTracker
from oracle.tracker_oracle import HomoTrackerOracle, SimilarityOracle
from synth_sequence import make_sequence, success_4pts_error


def seeded_net():
    torch.manual_seed(1)
    net = hdn_amd.HomoModelBuilder().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.8, 1.2)
    net.fc.weight.data.mul_(0.01)
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200); ap.add_argument("--parity", type=int, default=12)
    ap.add_argument("--iterations", type=int, default=1); ap.add_argument("--similarity", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    frames, corners, init = make_sequence(n_frames=args.frames, frame_hw=(720, 1280), target_wh=(300, 200))
    net = seeded_net(); net_cpu = copy.deepcopy(net)
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    sim_cpu = twin = None
    if args.similarity:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
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

    def stream(sync, graph=False):
        t = new_tracker(graph)
        for i in range(1, 6): t.track_new(i, frames[i])            # warm-up (MIOpen find, clocks, graph capture)
        t.H_total.copy_(torch.eye(3, dtype=torch.float64, device=dev)); torch.cuda.synchronize(); s0 = t.host_syncs
        t0 = time.perf_counter(); per = []
        for i in range(1, args.frames):
            a = time.perf_counter(); t.track_new(i, frames[i], sync=sync); per.append(time.perf_counter() - a)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (args.frames - 1) * 1e3, (t.host_syncs - s0) / (args.frames - 1), np.percentile(per, 99) * 1e3

    ms_sync, syncs, p99 = stream(True)
    ms_async, _, _ = stream(False)
    ms_graph, _, p99_graph = stream(True, graph=True)

    # the B=1 head alone (track_proj), eager vs hipGraph, synchronised EVERY frame as the tracker's score read does
    t = new_tracker()
    tmpl = t.init_homo_tmp
    from hdn_amd import frame as FR
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
