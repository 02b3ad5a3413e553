"""N independent sequences advancing in lock step on ONE GPU: the per-frame tracker loop of hdn_amd.tracker at batch N.

    BatchedHomoTracker(hm_net, n, ...)       N x HomoTracker        <- hdnTrackerHomo, hdn/tracker/hdn_tracker_proj_e2e.py:60-285
    BatchedDeviceTracker(model, n)           N x DeviceTrackerHomo  (the reference's ModelBuilder interface; BN-folded backbone, MIOpen find mode)

Why.  The reference's only inference-time parallelism is "several videos at once": tools/test.py pins one GPU (:49) and its authors
split the video list by hand across processes (:91-103).  One sequence is a B = 1 latency problem — 2.9 ms per frame on an MI355X of
which 2.1 ms are a ResNet-50 at batch 1 using < 5 % of the chip (profiles/round5_bench_line.json `sequence`) — and the H_total
recurrence forbids batching ALONG a sequence.  ACROSS sequences nothing is shared, so N of them run as one batch: every kernel of the
frame body already takes a batch dimension (the similarity decode, hdn_track_prepare / accumulate_f64, the correlation heads, the
homography estimator), the three frame kernels got one (hdn_*_batch_*, blockIdx.y = sequence, per-sequence parameter records read
straight out of the [N, 48] similarity state), and the two L1 scores — which the reference takes from sample 0 only (`[0][0]`,
model_builder_e2e_unconstrained_v2.py:213-216) — are computed per sequence (hdn_l1_score2_batch_f32) so that each has its own gate.
One frame of all N sequences = ONE upload of [N,H,W,3], ONE hipGraph replay, ONE host read of [N, 2 * points + 1].

Parity.  Every kernel computes sequence b of a batch exactly as it computes it alone (same code, blockIdx.y / the batch index only
selects the data); what may differ from N separate B = 1 runs is the rounding of the PyTorch-ROCm convolutions (MIOpen picks other
algorithms at another batch size) and of the B = 1-only packed head / chained trunk forms.  tests/test_gpu_tracker.py holds every
sequence of a batch to its own B = 1 run (<= 1e-4 px on the corners with the stand-in networks; bit-exact frame kernels) and to the CPU
restatement of the loop.

Frames of all sequences of a step must have one size (they share the [N,H,W,3] buffer); sequences of different lengths: keep feeding
the last frame of a finished one (its results are simply not read).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import frame as FR
from .refine import homo_refine
from .similarity import DeviceSimilarity, TrackerConfig
from .tracker import TRACK_CONST_DOUBLES


class BatchedHomoTracker:
    def __init__(self, hm_net, n: int, iterations: int = 1, similarity=None, score_gate: float = 2.5, graph: bool = False, cfg: TrackerConfig = None):
        """hm_net: hdn_amd.HomoModelBuilder (or the reference's, after install()) in eval mode on the GPU; n: sequences per step.
        similarity: None (identity) or a DeviceSimilarity (its model then holds n templates).  graph: one hipGraph per step."""
        if n < 1:
            raise ValueError("n must be >= 1")
        self.net, self.n = hm_net, int(n)
        self.cfg = cfg or (similarity.cfg if similarity is not None and hasattr(similarity, "cfg") else TrackerConfig())
        self.use_graph, self._graph = bool(graph), None
        self.iterations, self.similarity, self.score_gate = int(iterations), similarity, float(score_gate)
        self.host_syncs = 0
        self._staging = self._copy_done = None

    # ------------------------------------------------------------------------------------------------ frames: one upload per step
    def _upload(self, imgs, into=None):
        """n frames -> uint8 device tensor [n,H,W,3].  A list of numpy frames goes through ONE pinned staging buffer and one
        asynchronous copy; a stacked uint8 tensor (pageable, pinned or already on the device) is copied / used as it is."""
        n = self.n
        if isinstance(imgs, torch.Tensor):
            t = imgs
            if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[0] != n:
                raise TypeError(f"expected a uint8 [{n},H,W,C] tensor of frames, got {t.dtype} {tuple(t.shape)}")
        else:
            if len(imgs) != n:
                raise ValueError(f"this tracker advances {n} sequences per step, got {len(imgs)} frames")
            a0 = np.asarray(imgs[0])
            if a0.dtype != np.uint8 or a0.ndim != 3:
                raise TypeError(f"expected uint8 [H,W,C] frames, got {a0.dtype} {a0.shape}")
            shape = (n,) + a0.shape
            if self._staging is None or tuple(self._staging.shape) != shape:
                self._staging = torch.empty(shape, dtype=torch.uint8).pin_memory()
                self._copy_done = None
            if self._copy_done is not None:
                self._copy_done.synchronize()          # the previous step's copy has left the staging buffer
            host = self._staging.numpy()
            for b, im in enumerate(imgs):
                im = np.asarray(im)
                if im.shape != a0.shape or im.dtype != np.uint8:
                    raise ValueError(f"all frames of a step must be uint8 {a0.shape} (they share one buffer); frame {b} is {im.dtype} {im.shape}")
                host[b] = im
            t = self._staging
        if not torch.cuda.is_available():
            raise _lib.HdnHipError("hdn_amd runs on the GPU only; there is no CPU fallback")
        if into is not None:
            if tuple(t.shape) != tuple(into.shape):
                raise ValueError(f"graph mode was captured for uint8 frames of shape {tuple(into.shape)}, got {tuple(t.shape)}")
            into.copy_(t, non_blocking=True)
            dst = into
        else:
            dst = t.contiguous() if t.is_cuda else t.contiguous().to(self.dev, non_blocking=True)
        if t is self._staging:
            self._copy_done = torch.cuda.Event()
            self._copy_done.record()
        return dst

    # -------------------------------------------------------------------------------------------------- init
    def init(self, imgs, bboxes, polys, gt_points, first_points=None):
        """Per sequence what hdnTrackerHomo.init takes (hdn_tracker_proj_e2e.py:60): imgs n x BGR uint8 [H,W,3]; bboxes n x (x, y, w, h);
        polys n x (cx, cy, w, h, theta); gt_points n x the initial corners (the same number of points for every sequence)."""
        c, n = self.cfg, self.n
        if not (len(bboxes) == len(polys) == len(gt_points) == n):
            raise ValueError(f"init takes {n} bboxes / polys / gt_points")
        self.dev = next(self.net.parameters()).device
        polys = np.asarray([np.asarray(p, np.float64).reshape(-1)[:4] for p in polys], np.float64)
        self.init_pos = polys[:, 0:2].copy()
        self.size = polys[:, 2:4].copy()
        ctx = c.context_amount * self.size.sum(axis=1)
        self.init_s_z = np.floor(np.sqrt((self.size[:, 0] + ctx) * (self.size[:, 1] + ctx)))
        self.init_s_z_sm = np.floor(np.sqrt(self.size[:, 0] * self.size[:, 1]))
        frames = self._upload(imgs)
        # np.mean(img, axis=(0, 1)) of every first frame: one reduction on the device, read once
        self.channel_average = frames.to(torch.float64).mean(dim=(1, 2)).cpu().numpy()
        self.host_syncs += 1
        _, H, W, _ = frames.shape
        self.z_crop_points_sm = [FR.crop_points(self.init_pos[b], self.init_s_z_sm[b], H, W) for b in range(n)]
        # get_template_info(get_subwindow_for_homo(...)[:, 0:3]): the normalised gray templates, constant for the sequences
        self._const_params = torch.from_numpy(np.concatenate([self.init_pos, self.init_s_z_sm[:, None], self.channel_average], axis=1)).to(self.dev)
        self.init_homo_tmp = FR.get_search_info(frames, None, None, None, model_sz=c.exemplar_size, params=self._const_params)
        with torch.no_grad():
            self.init_patch_1 = self.net.ShareFeature(self.init_homo_tmp).detach()
        if self.similarity is not None:
            self.similarity.init(frames, self.init_pos, self.init_s_z, self.init_s_z_sm, self.channel_average)
        pts = [np.asarray(g, np.float64).reshape(-1, 2) for g in gt_points]
        if any(p.shape != pts[0].shape for p in pts):
            raise ValueError("every sequence must have the same number of initial points")
        self.n_points = pts[0].shape[0]
        self.init_points = torch.from_numpy(np.stack(pts)).to(self.dev).contiguous()
        self.H_total = torch.eye(3, dtype=torch.float64, device=self.dev).repeat(n, 1, 1).contiguous()
        self._Ht, self._Hinv = (torch.empty((n, 9), dtype=torch.float64, device=self.dev) for _ in range(2))
        self._out = torch.empty((n, 2 * self.n_points + 1), dtype=torch.float32, device=self.dev)
        self._graph = None
        E = c.exemplar_size
        consts = np.zeros((n, TRACK_CONST_DOUBLES), np.float64)
        for b, zp in enumerate(self.z_crop_points_sm):       # (:251-258, as HomoTracker.init builds them: float32 matrices, float32 inverses)
            S = np.diag([E / (zp[2] - zp[0] + 1), E / (zp[3] - zp[1] + 1), 1.0]).astype(np.float32)
            Sh = np.array([[1, 0, -zp[0]], [0, 1, -zp[1]], [0, 0, 1]], np.float32)
            consts[b, 0:9], consts[b, 9:18] = np.linalg.inv(S).astype(np.float64).reshape(-1), S.astype(np.float64).reshape(-1)
            consts[b, 18:27], consts[b, 27:36] = np.linalg.inv(Sh).astype(np.float64).reshape(-1), Sh.astype(np.float64).reshape(-1)
            consts[b, 36] = self.score_gate
        self._consts = torch.from_numpy(consts).to(self.dev)

    # -------------------------------------------------------------------------------------------------- one step = one frame of every sequence
    def _body(self, frames):
        """HomoTracker._body at batch n: no host reads, no data-dependent allocations -> capturable as one hipGraph.
        -> (out float32 [n, 2 * points + 1] = corners (x, y) + best_score per sequence, homo_score [n])."""
        lib, st, n = _lib.load(), _lib.stream_ptr(self.dev), self.n
        with _lib.device_guard(self.dev):
            _lib.check(lib.hdn_track_prepare_f64(_lib.ptr(self.H_total), _lib.ptr(self._Ht), _lib.ptr(self._Hinv), n, st), "track prepare")
        frames = FR.warp_perspective(frames, self._Hinv)                            # :150-155, every sequence by its own inv(H_total)
        sim_state = None
        if self.similarity is not None:
            sim = self.similarity(frames)                                           # :157-214
            sim_state = self.similarity.state
            params = sim["params_homo"]
            rot_img = FR.warp_affine_cubic(frames, sim["rot_matrix"])               # :223
        else:
            params, rot_img = self._const_params, frames
        search = FR.get_search_info(rot_img, None, None, None, model_sz=self.cfg.exemplar_size, params=params)     # :224-239
        H_comp, homo_score, _ = homo_refine(self.net, self.init_homo_tmp, search, iterations=self.iterations, patch_1=self.init_patch_1,
                                            per_sample=True)                        # :242-250
        score = homo_score.detach().reshape(-1).to(torch.float32).contiguous()
        with _lib.device_guard(self.dev):                                           # :251-272
            _lib.check(lib.hdn_track_accumulate_f64(_lib.ptr(self._Ht), _lib.ptr(sim_state) if sim_state is not None else None, _lib.ptr(H_comp),
                                                    _lib.ptr(score), _lib.ptr(self._consts), _lib.ptr(self.init_points), self.n_points,
                                                    _lib.ptr(self.H_total), _lib.ptr(self._out), n, st), "track accumulate")
        return self._out, score

    def _capture(self, frame_shape):
        self._static_frames = torch.empty(frame_shape, dtype=torch.uint8, device=self.dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        H0 = self.H_total.clone()
        try:
            with torch.cuda.stream(side):
                for _ in range(3):      # warm-up on the side stream (MIOpen find at this batch size, lazy initialisations)
                    self._body(self._static_frames)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._g_out, self._g_score = self._body(self._static_frames)
            self._graph = graph
        finally:
            torch.cuda.current_stream().wait_stream(side)
            self.H_total.copy_(H0)

    def track_new(self, fr_idx, imgs, sync: bool = True):
        """One frame of every sequence.  -> list of n result dictionaries with hdnTrackerHomo.track_new's keys (sync=True; one host read
        for all of them), or device views {'points' [n, P, 2], 'best_score' [n]} (sync=False)."""
        n, P = self.n, self.n_points
        if self.use_graph:
            if self._graph is None:
                shape = tuple(imgs.shape) if isinstance(imgs, torch.Tensor) else (n,) + tuple(np.asarray(imgs[0]).shape)
                try:
                    self._capture(shape)
                except RuntimeError as e:
                    import warnings
                    warnings.warn(f"hdn_amd: the batched per-frame body could not be captured as a hipGraph ({type(e).__name__}: {e}); running it eagerly")
                    self.use_graph, self._graph = False, None
                    return self.track_new(fr_idx, imgs, sync=sync)
            self._upload(imgs, into=self._static_frames)
            self._graph.replay()
            out, score = self._g_out, self._g_score
            if not sync:
                out, score = out.clone(), score.clone()
        else:
            out, score = self._body(self._upload(imgs))
            out = out.clone()
        self.last_points, self.last_score = out[:, :2 * P].view(n, P, 2), score
        if not sync:
            return {"points": self.last_points, "polygon": self.last_points, "best_score": out[:, 2 * P]}
        host = out.cpu().numpy()
        self.host_syncs += 1
        res = []
        for b in range(n):
            pn, best = host[b, :2 * P].reshape(P, 2), host[b, 2 * P]
            mx, mn = pn.max(0), pn.min(0)
            bbox = [mn[0], mn[1], mx[0] - mn[0], mx[1] - mn[1]]
            res.append({"bbox_aligned": bbox, "best_score": best, "polygon": pn, "points": pn, "bbox": bbox})
        return res

    def track(self, imgs):
        return self.track_new(None, imgs)


class BatchedDeviceTracker(BatchedHomoTracker):
    """n x DeviceTrackerHomo in lock step: `model` is the reference's ModelBuilder (hm_net = the homography estimator; template /
    track_new / track_new_lp = the similarity branch, called at batch n — the reference's own forward code is batch-general, its heads
    correlate sample b with template b).  Backbone / necks BatchNorm-folded, MIOpen find mode around this tracker's calls, one hipGraph
    per step (graph=False or HDN_TRACKER_GRAPH=0: eager), exactly as DeviceTrackerHomo does for one sequence."""

    def __init__(self, model, n: int, graph: bool = None, iterations: int = 1, cfg: TrackerConfig = None, fold_backbone: bool = None):
        if cfg is None:
            cfg = TrackerConfig()
            try:
                from hdn.core.config import cfg as ref_cfg
                cfg = TrackerConfig.from_reference(ref_cfg)
            except ImportError:
                pass
        if graph is None:
            graph = os.environ.get("HDN_TRACKER_GRAPH", "1") not in ("", "0")
        model.eval()
        self.model = model
        self.miopen_find = os.environ.get("HDN_MIOPEN_FIND", "1") not in ("", "0") and next(model.parameters()).is_cuda
        from . import backbone as BB
        self.folded = BB.optimize_similarity_model(model) if (BB.enabled() if fold_backbone is None else fold_backbone) else []
        super().__init__(model.hm_net, n, iterations=iterations, similarity=DeviceSimilarity(model, cfg), graph=graph, cfg=cfg)

    def _find_mode(self):
        import contextlib
        if not self.miopen_find:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def only_benchmark():        # (only this flag: torch.backends.cudnn.flags() would reset the others to its defaults)
            before = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            try:
                yield
            finally:
                torch.backends.cudnn.benchmark = before
        return only_benchmark()

    def init(self, imgs, bboxes, polys, gt_points, first_points=None):
        with self._find_mode():
            return super().init(imgs, bboxes, polys, gt_points, first_points)

    def track_new(self, fr_idx, imgs, sync: bool = True):
        with self._find_mode():
            return super().track_new(fr_idx, imgs, sync=sync)
