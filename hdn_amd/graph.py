"""hipGraph capture of the per-frame homography head (B small, launch-latency bound).

At B=1 the head is ~10 short HIP kernels plus the PyTorch-ROCm trunk (~150 MIOpen launches): eager execution is bound
by host launch latency, not by the kernels.  `GraphedTrackProj` captures one `track_proj` call into a hipGraph (through
torch.cuda.CUDAGraph: the C-ABI kernels launch on the capturing stream, outputs come from the graph's private pool)
and replays it per frame on static input buffers.
"""
from __future__ import annotations

import torch

from .homo_model import homo_stages, track_proj


class GraphedTrackProj:
    """Capture `track_proj(net, data)` for a fixed batch size / image size; call it with new data every frame.

    `net` must be in eval mode on its device.  Pass `template_constant=True` when `input_tensors[:, :1]` (the template)
    does not change between frames: ShareFeature(template) is then computed once at capture time and only
    the search channel is re-extracted per frame.
    """

    def __init__(self, net, example: dict, template_constant: bool = False, warmup: int = 3):
        if net.training:
            raise ValueError("capture the head in eval mode")
        self.net = net
        self.static = {k: example[k].detach().clone() for k in ("org_imgs", "input_tensors", "h4p", "patch_indices")}
        self.template_constant = bool(template_constant)
        self._patch_1 = None
        if self.template_constant:
            self._patch_1 = homo_stages(net, self.static)["patch_1"].contiguous().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                track_proj(net, self.static, None, self._patch_1)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = track_proj(net, self.static, None, self._patch_1)

    def __call__(self, data: dict):
        for k, buf in self.static.items():
            buf.copy_(data[k], non_blocking=True)
        self.graph.replay()
        return self.out  # (H_mat [B,3,3], score, score_simi): static tensors, overwritten by the next call
