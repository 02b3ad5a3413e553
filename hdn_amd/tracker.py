"""The per-frame tracker loop, device-resident (BASELINE config 4).

    HomoTracker.init(img, bbox, poly, gt_points, first_point)        <- hdnTrackerHomo.init,      hdn/tracker/hdn_tracker_proj_e2e.py:60-120
    HomoTracker.track_new(fr_idx, img, gt_box, gt_poly, gt_points)   <- hdnTrackerHomo.track_new, hdn_tracker_proj_e2e.py:141-285
    DeviceTrackerHomo(model)                                         <- the class hdn/tracker/tracker_builder.py:12-19 registers as
                                                                        TRACKS['hdnTrackerHomoProje2e'] (hdn_amd.install.install(tracker=True))

Same call signatures and the same result dictionary ('bbox_aligned', 'best_score', 'polygon', 'points', 'bbox') as the
reference's tracker.  Per frame: undo the accumulated motion (full-frame warp by inv(H_total), :150-155); the similarity
estimate on the stabilised frame (:157-214: search crop, ModelBuilder.track_new, decode, moved crop, track_new_lp, decode, H_sim —
hdn_amd.similarity); rotate back and cut the 127-px homography crop (:223-239); the refinement loop around
ModelBuilder.track_proj (:242-250); un-scale / un-shift the residual (:251-258), gate it (`homo_score > 2.5`, :261-264),
accumulate H_total and project the initial corners (:266-272).

The networks of the similarity branch (ResNet-50 backbone, necks, the 3x3 / 1x1 convolutions of the heads) are PyTorch-ROCm's, as
north_star assigns them; they are reached through `similarity` (hdn_amd.similarity.DeviceSimilarity around an object with the
reference's ModelBuilder interface).  Without one the similarity estimate is the identity.

What the reference does on the host per frame — cv2 warps of the full frame, numpy crops, three crop uploads, numpy decodes of the
head maps and six .cpu().numpy() syncs — is here ONE upload of the uint8 frame, kernels (hdn_amd.frame, .similarity, .refine) and
two one-lane kernels for the 3x3 float64 bookkeeping (hdn_track_prepare_f64 / hdn_track_accumulate_f64); the only host read is the 4 projected corners + best_score the caller asks for
(`sync=True`).  Nothing in the body depends on host values, so it replays as one hipGraph (`graph=True`) with or without the
similarity branch.  Sequences are independent: N GPUs run N sequences (replicas only, no collective).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import frame as FR
from .refine import homo_refine
from .similarity import DeviceSimilarity, TrackerConfig

TRACK_CONST_DOUBLES = 40     # HDN_TRACK_CONST_DOUBLES


class HomoTracker:
    def __init__(self, hm_net, iterations: int = 1, similarity=None, score_gate: float = 2.5, graph: bool = False, cfg: TrackerConfig = None):
        """hm_net: hdn_amd.HomoModelBuilder (or the reference's, after install()) in eval mode on the GPU.
        iterations: trip count of the refinement loop (1 in the shipped tracker, :242; 2 in BASELINE config 5).
        similarity: None (identity) or a DeviceSimilarity: `.init(frame, init_pos, init_s_z, init_s_z_sm, avg)` once per
        sequence, `(stabilised_frame) -> views into its device state record` per frame (no host values).
        graph: replay the whole per-frame body as ONE hipGraph (several hundred launches at B = 1 are launch-latency bound)."""
        self.net = hm_net
        self.cfg = cfg or (similarity.cfg if similarity is not None and hasattr(similarity, "cfg") else TrackerConfig())
        self.use_graph = bool(graph)
        self._graph = None
        self.iterations = int(iterations)
        self.similarity = similarity
        self.score_gate = float(score_gate)
        self.host_syncs = 0  # device->host reads this object has caused (the reference: six per frame)

    # -------------------------------------------------------------------------------------------------- init
    def init(self, img, bbox, poly, gt_points, first_point=None):
        """img: BGR uint8 [H,W,3]; bbox (x, y, w, h); poly (cx, cy, w, h, theta); gt_points: the 4 corners."""
        c = self.cfg
        self.dev = next(self.net.parameters()).device
        self.init_pos = np.array([poly[0], poly[1]], np.float64)
        self.center_pos = self.init_pos.copy()
        self.size = np.array([poly[2], poly[3]], np.float64)
        w_z = self.size[0] + c.context_amount * np.sum(self.size)
        h_z = self.size[1] + c.context_amount * np.sum(self.size)
        self.init_s_z = float(np.floor(np.sqrt(w_z * h_z)))
        self.init_s_z_sm = float(np.floor(np.sqrt(self.size[0] * self.size[1])))
        frame = FR.upload(img)
        # np.mean(img, axis=(0, 1)) of the first frame: one reduction on the device, read once per sequence
        self.channel_average = frame.to(torch.float64).mean(dim=(0, 1)).cpu().numpy()
        self.host_syncs += 1
        H, W, _ = frame.shape
        self.z_crop_points_sm = FR.crop_points(self.center_pos, self.init_s_z_sm, H, W)
        # get_template_info(get_subwindow_for_homo(...)[:, 0:3]) : the normalised gray template, constant for the sequence
        self.init_homo_tmp = FR.get_search_info(frame, self.center_pos, self.init_s_z_sm, self.channel_average, model_sz=c.exemplar_size)
        with torch.no_grad():   # ShareFeature(template): constant for the sequence, so not part of the per-frame work
            self.init_patch_1 = self.net.ShareFeature(self.init_homo_tmp).detach()
        if self.similarity is not None:
            self.similarity.init(frame, self.init_pos, self.init_s_z, self.init_s_z_sm, self.channel_average)   # model.template(z_crop), :99-107
        self.init_points = torch.tensor(np.asarray(gt_points, np.float64).reshape(1, -1, 2), dtype=torch.float64, device=self.dev).contiguous()
        self.H_total = torch.eye(3, dtype=torch.float64, device=self.dev)
        self._Ht, self._Hinv = (torch.empty((3, 3), dtype=torch.float64, device=self.dev) for _ in range(2))
        self._out = torch.empty((1, 2 * self.init_points.shape[1] + 1), dtype=torch.float32, device=self.dev)
        self._const_params = FR._dev_f64([self.init_pos[0], self.init_pos[1], self.init_s_z_sm] + [float(a) for a in self.channel_average], self.dev)
        self._graph = None
        # un-scale / un-shift of the residual (:251-258): the four matrices are constants of the sequence (float32 as the reference
        # builds them, their inverses float32 by numpy's dtype rule), handed to hdn_track_accumulate_f64 with the gate
        cw = self.z_crop_points_sm[2] - self.z_crop_points_sm[0] + 1
        ch = self.z_crop_points_sm[3] - self.z_crop_points_sm[1] + 1
        E = c.exemplar_size
        S = np.diag([E / cw, E / ch, 1.0]).astype(np.float32)
        Sh = np.array([[1, 0, -self.z_crop_points_sm[0]], [0, 1, -self.z_crop_points_sm[1]], [0, 0, 1]], np.float32)
        consts = np.zeros(TRACK_CONST_DOUBLES, np.float64)
        consts[0:9], consts[9:18] = np.linalg.inv(S).astype(np.float64).reshape(-1), S.astype(np.float64).reshape(-1)
        consts[18:27], consts[27:36] = np.linalg.inv(Sh).astype(np.float64).reshape(-1), Sh.astype(np.float64).reshape(-1)
        consts[36] = self.score_gate
        self._consts = torch.from_numpy(consts).to(self.dev)

    # -------------------------------------------------------------------------------------------------- one frame
    def _body(self, frame):
        """Everything of a frame that runs on the device, from the uploaded frame to the 4 projected corners; no host
        reads, no allocations that depend on data: capturable in a hipGraph when the frame is a static buffer.
        -> (H_total' [3,3] float64 (a buffer of this object), out float32 [9] = 4 corners (x, y) + best_score, homo_score)."""
        lib, st = _lib.load(), _lib.stream_ptr(self.dev)
        # :150-155  undo the accumulated motion (a singular H_total is reset to the identity, as the reference does).
        # hdn_frame_warp_perspective_u8 inverts its matrix itself (cv2 semantics), so it is handed inv(H_total).
        with _lib.device_guard(self.dev):
            _lib.check(lib.hdn_track_prepare_f64(_lib.ptr(self.H_total), _lib.ptr(self._Ht), _lib.ptr(self._Hinv), 1, st), "track prepare")
        frame = FR.warp_perspective(frame, self._Hinv.view(-1))
        sim_state = None
        if self.similarity is not None:
            # :157-214 on the STABILISED frame; :223 rotate back by -rot_delta about the new centre (bicubic; rot 0 = identity)
            sim = self.similarity(frame)
            sim_state, params = self.similarity.state, sim["params_homo"]
            rot_img = FR.warp_affine_cubic(frame, sim["rot_matrix"])
        else:
            params, rot_img = self._const_params, frame
        # :224-239  cut the homography crop, normalise
        search = FR.get_search_info(rot_img, None, None, None, model_sz=self.cfg.exemplar_size, params=params)
        # :242-250  refinement loop around track_proj
        H_comp, homo_score, _ = homo_refine(self.net, self.init_homo_tmp, search, iterations=self.iterations, patch_1=self.init_patch_1)
        # :251-272  un-scale, un-shift, gate, accumulate, project the initial corners
        score = homo_score.detach().reshape(-1).to(torch.float32).contiguous()
        with _lib.device_guard(self.dev):
            _lib.check(lib.hdn_track_accumulate_f64(_lib.ptr(self._Ht), _lib.ptr(sim_state) if sim_state is not None else None, _lib.ptr(H_comp),
                                                    _lib.ptr(score), _lib.ptr(self._consts), _lib.ptr(self.init_points),
                                                    self.init_points.shape[1], _lib.ptr(self.H_total), _lib.ptr(self._out), 1, st),
                       "track accumulate")     # (in place: the kernel reads Ht, the copy hdn_track_prepare_f64 made)
        return self.H_total, self._out.view(-1), homo_score

    def _capture(self, frame_shape):
        """hipGraph of the whole per-frame body.  The frame lands in a static device buffer; H_total is carried in a static
        tensor updated by the graph itself; the similarity state record is a static tensor of the DeviceSimilarity."""
        self._static_frame = torch.empty(frame_shape, dtype=torch.uint8, device=self.dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        H0 = self.H_total.clone()
        try:
            with torch.cuda.stream(side):
                for _ in range(3):  # warm-up on the side stream (MIOpen find, lazy initialisations); H_total is put back below
                    self._body(self._static_frame)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                H, out, score = self._body(self._static_frame)      # (H_total is updated in place: the recurrence lives inside the graph)
                self._g_out, self._g_score = out, score
            self._graph = graph
        finally:
            torch.cuda.current_stream().wait_stream(side)
            self.H_total.copy_(H0)

    def track_new(self, fr_idx, img, gt_box=None, gt_poly=None, gt_points=None, sync: bool = True):
        if self.use_graph:
            t = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
            if self._graph is None:
                try:
                    self._capture(tuple(t.shape))
                except RuntimeError as e:
                    # stream capture reports what it cannot hold (e.g. a model whose forward makes host round trips) as a
                    # RuntimeError; the same kernels are then launched one by one.  Anything else is a real error and propagates,
                    # as does whatever the eager retry raises.
                    import warnings
                    warnings.warn(f"hdn_amd: the per-frame body could not be captured as a hipGraph ({type(e).__name__}: {e}); running it eagerly")
                    self.use_graph, self._graph = False, None
                    return self.track_new(fr_idx, img, gt_box, gt_poly, gt_points, sync=sync)
            if tuple(t.shape) != tuple(self._static_frame.shape) or t.dtype != torch.uint8:
                raise ValueError(f"graph mode was captured for uint8 frames of shape {tuple(self._static_frame.shape)}, got {t.dtype} {tuple(t.shape)}")
            self._static_frame.copy_(t, non_blocking=True)
            self._graph.replay()
            out, homo_score = self._g_out, self._g_score
            if not sync:  # the graph's static outputs are overwritten by the next replay: hand out copies
                out, homo_score = out.clone(), homo_score.clone()
        else:
            H, out, homo_score = self._body(FR.upload(img))
            out = out.clone()
        n = self.init_points.shape[1]          # any number of initial points (cv2.perspectiveTransform takes any; POT gives 4)
        pts = out[:2 * n].view(n, 2)
        self.last_points, self.last_score = pts, homo_score
        if not sync:
            return {"points": pts, "polygon": pts, "best_score": out[2 * n]}
        host = out.cpu().numpy()
        self.host_syncs += 1
        pn, best_score = host[:2 * n].reshape(n, 2), host[2 * n]
        mx, mn = pn.max(0), pn.min(0)
        bbox = [mn[0], mn[1], mx[0] - mn[0], mx[1] - mn[1]]
        return {"bbox_aligned": bbox, "best_score": best_score, "polygon": pn, "points": pn, "bbox": bbox}


    def track(self, img):
        """BaseTracker.track (hdn/tracker/base_tracker.py:28-37, abstract there): the next frame, nothing else known."""
        return self.track_new(None, img)

    def similarity_state(self) -> dict:
        """The last frame's similarity record read back to the host (one extra device->host read, counted): centre, deltas,
        scale / rotation increments, scores.  The reference keeps `center_pos`, `rot`, `scale` as host attributes updated every
        frame (hdn_tracker_proj_e2e.py:185,215-216); nothing outside the tracker reads them and here they live on the device
        only, so they are produced on demand instead of costing every frame a synchronisation."""
        if self.similarity is None or self.similarity.state is None:
            return {}
        from .similarity import state_fields
        row = self.similarity.state.view(-1).cpu()
        self.host_syncs += 1
        f = {k: v.numpy().copy() for k, v in state_fields(row).items()}
        self.center_pos = f["center"].astype(np.float64).copy()
        return f


class DeviceTrackerHomo(HomoTracker):
    """Drop-in for hdnTrackerHomo (hdn_tracker_proj_e2e.py:22-285) behind build_tracker(model)
    (hdn/tracker/tracker_builder.py:18-19): same constructor argument, same init / track_new signatures and result keys.
    `model` is the reference's ModelBuilder (hm_net = the homography estimator, template / track_new / track_new_lp = the
    similarity branch).  Each frame is replayed as ONE hipGraph, captured at a sequence's first frame (2.9 against 5.3 ms per frame with the
    production-shaped model: at B = 1 the loop is launch-bound; a body that cannot be captured falls back to eager launches with a warning;
    graph=False or HDN_TRACKER_GRAPH=0: eager).  The model's backbone and necks are switched to their
    BatchNorm-folded, epilogue-fused form (hdn_amd.backbone; fold_backbone=False or HDN_FOLD_BACKBONE=0: left as they are)."""

    def __init__(self, model, graph: bool = None, iterations: int = 1, cfg: TrackerConfig = None, fold_backbone: bool = None):
        if cfg is None:
            cfg = TrackerConfig()
            try:
                from hdn.core.config import cfg as ref_cfg     # the reference's node, after tools/test.py merged the YAML
                cfg = TrackerConfig.from_reference(ref_cfg)    # (incl. cfg.BAN.KWARGS.cls_out_channels: 2 = softmax, 1 = sigmoid decode)
            except ImportError:
                pass
        if graph is None:
            graph = os.environ.get("HDN_TRACKER_GRAPH", "1") not in ("", "0")
        model.eval()
        self.model = model
        # The backbone's convolutions are PyTorch-ROCm's, and their shapes are fixed for a whole sequence: let MIOpen search for its kernels once
        # (find mode; the search runs in the first frames / the capture warm-up).  Measured with the production-shaped model: 2.9 against 4.5 ms
        # per frame (profiles/round4_experiments.txt item 11).  The reference's inference scripts leave torch's default (off); HDN_MIOPEN_FIND=0 does too.
        # The flag is process-global in torch, so it is raised only AROUND this tracker's own calls (init / track_new, incl. the graph
        # capture) and put back afterwards: other models in the process keep the setting they had (round-4 ADVICE).
        self.miopen_find = os.environ.get("HDN_MIOPEN_FIND", "1") not in ("", "0") and next(model.parameters()).is_cuda
        # backbone + necks stay PyTorch-ROCm's convolutions; their BatchNorm / ReLU / add launches (a third of the B = 1 frame) are folded away
        from . import backbone as BB
        self.folded = BB.optimize_similarity_model(model) if (BB.enabled() if fold_backbone is None else fold_backbone) else []
        super().__init__(model.hm_net, iterations=iterations, similarity=DeviceSimilarity(model, cfg), graph=graph, cfg=cfg)
        if (self.folded or self.miopen_find) and not DeviceTrackerHomo._announced:
            import sys
            DeviceTrackerHomo._announced = True      # once per process
            print("hdn_amd: DeviceTrackerHomo" +
                  (" switched %s of the model it was given to their BatchNorm-folded form (parameters / state_dict unchanged; "
                   "hdn_amd.backbone.restore_similarity_model(model) or HDN_FOLD_BACKBONE=0 undoes it)" % " / ".join(self.folded) if self.folded else "") +
                  (" and" if self.folded and self.miopen_find else "") +
                  (" runs its networks under MIOpen find mode (torch.backends.cudnn.benchmark, raised around this tracker's calls only; "
                   "HDN_MIOPEN_FIND=0: torch's setting as it is)" if self.miopen_find else ""), file=sys.stderr)

    _announced = False

    def _find_mode(self):
        import contextlib
        if not self.miopen_find:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def only_benchmark():
            # torch.backends.cudnn.flags() sets EVERY flag (the ones not named fall to its defaults: deterministic=False,
            # allow_tf32=True), which would override a user's settings for the backbone's convolutions and bake them into the
            # captured graph (round-5 ADVICE).  Only `benchmark` is touched here.
            before = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            try:
                yield
            finally:
                torch.backends.cudnn.benchmark = before

        return only_benchmark()

    def init(self, img, bbox, poly, gt_points, first_point=None):
        with self._find_mode():
            return super().init(img, bbox, poly, gt_points, first_point)

    def track_new(self, fr_idx, img, gt_box=None, gt_poly=None, gt_points=None, sync: bool = True):
        with self._find_mode():
            return super().track_new(fr_idx, img, gt_box, gt_poly, gt_points, sync=sync)
