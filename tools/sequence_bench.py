#!/usr/bin/env python3
"""Synthetic sequence stream through the homography head at B=1 (a reduced BASELINE config 4: no POT data, no cv2 here).

A band-limited random texture is the template; frame t's search crop is the template warped by a smooth random-walk
homography (our own warp kernel) plus noise.  Each frame runs track_proj on the GPU (eager and hipGraph replay) and on
the CPU oracle with the same seeded weights; reported: corner-offset difference GPU vs CPU (success_4pts_error of the
offset vectors), and per-frame latency.
    python tools/sequence_bench.py [--frames 100]
"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hdn_amd
from hdn_amd import homography as G
from hdn_amd.graph import GraphedTrackProj
from oracle import hdn_oracle as O


def texture(g, n=127):
    f = torch.fft.rfft2(torch.randn(n, n, generator=g))
    ky, kx = torch.meshgrid(torch.fft.fftfreq(n), torch.fft.rfftfreq(n), indexing="ij")
    return torch.fft.irfft2(f * torch.exp(-((kx ** 2 + ky ** 2) / (2 * 0.06 ** 2))), s=(n, n))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=100); args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(20260928)
    torch.manual_seed(1)
    net = hdn_amd.HomoModelBuilder().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.8, 1.2)
    net.fc.weight.data.mul_(0.01)
    sd = {k: v.clone() for k, v in net.ShareFeature.state_dict().items()}
    cpu_regress = lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1))
    import copy
    net_cpu = copy.deepcopy(net)
    netd = net.to(dev)
    tmpl = texture(g); tmpl = (tmpl - tmpl.mean()) / tmpl.std()
    h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32)
    pidx = torch.arange(127 * 127, dtype=torch.float32).unsqueeze(0)
    off = torch.zeros(1, 8)
    frames = []
    for t in range(args.frames):
        off = 0.9 * off + 1.5 * torch.randn(1, 8, generator=g)           # smooth random walk of the 4 corners
        _, search = G.dlt_warp(h4p.to(dev), off.to(dev), tmpl.reshape(1, 1, 127, 127).to(dev))
        search = search.cpu() + 0.02 * torch.randn(1, 1, 127, 127, generator=g)
        pair = torch.cat([tmpl.reshape(1, 1, 127, 127), search], dim=1)
        frames.append({"org_imgs": pair, "input_tensors": pair.clone(), "h4p": h4p, "patch_indices": pidx})
    dd = [{k: v.to(dev) for k, v in f.items()} for f in frames]
    from hdn_amd.homo_model import homo_stages
    # parity per frame
    errs = []
    with torch.no_grad():
        for f, fd in zip(frames[:min(20, args.frames)], dd):
            x_gpu = homo_stages(netd, fd)["x"].cpu().numpy()
            _, _, _, aux = O.track_proj(f, sd, cpu_regress)
            errs.append(float(O.corner_error(x_gpu, aux["x"].numpy())[0]))
    def run(fn):
        for fd in dd[:5]: fn(fd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for fd in dd: r = fn(fd)
        float(r[1]); torch.cuda.synchronize()   # the tracker reads the score on the host every frame
        return (time.perf_counter() - t0) / len(dd) * 1e3
    eager = run(lambda fd: netd.track_proj(fd, None))
    netd.optimize_for_inference()
    eager_folded = run(lambda fd: netd.track_proj(fd, None))
    gr = GraphedTrackProj(netd, dd[0], template_constant=True)
    H0, s0, _ = netd.track_proj(dd[3], None); Hg, sg, _ = gr(dd[3])
    graph_ok = float((H0 - Hg).abs().max()) < 1e-5
    graphed = run(gr)
    print(json.dumps({"frames": args.frames, "corner_error_gpu_vs_cpu_max": max(errs), "ms_per_frame_eager": eager,
                      "ms_per_frame_eager_folded_trunk": eager_folded, "ms_per_frame_hipgraph": graphed,
                      "graph_matches_eager": graph_ok}))


if __name__ == "__main__":
    main()
