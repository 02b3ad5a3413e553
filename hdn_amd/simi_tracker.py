"""TRACKS['hdnTracker'] on the device: the similarity-only tracker with a per-frame template refresh.

    SimiTracker.init(img, bbox, poly, first_point)          <- hdnTracker.init,            hdn/tracker/hdn_tracker.py:109-154
    SimiTracker.track_new(fr_idx, img, gt_box, gt_poly)     <- hdnTracker.track_new,       hdn_tracker.py:174-301
    (inside track_new)                                       <- hdnTracker.update_template, hdn_tracker.py:156-162
    DeviceTrackerSimi(model)                                 <- the class hdn/tracker/tracker_builder.py:13 registers as TRACKS['hdnTracker']
                                                               (cfg.TRACK.TYPE's default, hdn/core/config.py:517; experiments/siamban_r50_l234_pot)

What the reference's loop does per frame: a search crop about the CURRENT centre with a side that follows the current size
(:176-192), ModelBuilder.track_new, numpy decode, a second crop about the moved centre, track_new_lp, numpy decode, the size / rotation /
scale recurrences and the result polygon in numpy (:213-281) — then update_template: the FIRST frame rotated by the accumulated
rotation (cv2.warpAffine, bicubic, full frame), cropped (+ cv2.logPolar channels) and pushed through ModelBuilder.template again
(two more backbone passes).  Four .cpu().numpy() reads, three crop uploads and a full-frame host warp per frame.

Here: one upload of the uint8 frame, the crops / the rotation of the resident first frame as kernels (hdn_amd.frame), the two decodes
(hdn_similarity_*_f32, given THIS frame's centre / s_z / s_x record), ONE one-lane kernel for everything between the second decode and
the next frame's first crop (hdn_simi_track_update_f64: recurrences, clamps, polygon, update_template's matrix, next s_z / s_x), and one
host read of 20 doubles — or none (`sync=False`).  The networks are the model's own (PyTorch-ROCm backbone / necks, the packed heads
around the HIP correlations).  Template features live in static buffers that the refresh overwrites in place, so a frame replays as one
hipGraph (`graph=True`) and the heads' template cache sees the change through the tensors' version counters.

Signatures: the reference's launchers call `init(img, bbox, poly, gt_points, first_point)` / `track_new(idx, img, bbox, poly, gt_points)`
(tools/test.py:130,153) — one argument more than hdnTracker's own methods take; both spellings are accepted here.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import frame as FR
from .similarity import SEQ_DOUBLES, SimilarityDecoder, TrackerConfig

TRACK_DOUBLES, OUT_DOUBLES = 48, 20     # HDN_SIMI_TRACK_DOUBLES / HDN_SIMI_OUT_DOUBLES


def _center2poly(c):
    """cetner2poly, hdn/utils/bbox.py:40-56."""
    x, y, w, h = c[0], c[1], c[2], c[3]
    return np.array([x - w * 0.5, y - h * 0.5, x + w * 0.5, y - h * 0.5, x + w * 0.5, y + h * 0.5, x - w * 0.5, y + h * 0.5])


def _rot_matrix(cx, cy, rot):
    """getRotMatrix, hdn/utils/bbox.py:58-75."""
    cc, ss = np.cos(rot), np.sin(rot)
    return np.array([[cc, -ss, cx - cx * cc + cy * ss], [ss, cc, cy - cy * cc - cx * ss], [0, 0, 1]])


def _transform_poly(polygon, m):
    """transformPoly, hdn/utils/bbox.py:77-90."""
    polygon = polygon.reshape(-1, 2)
    out = np.ones([polygon.shape[0], 3])
    out[:, 0:2] = polygon
    return (out @ m.transpose(1, 0))[:, 0:2]


def sequence_records(poly, first_point, channel_average, cfg: TrackerConfig):
    """The host arithmetic of hdnTracker.init for ONE sequence (hdn_tracker.py:117-136) -> (track record [48], seq record [8], poly_shift_l, init_s_z)."""
    poly = [float(p) for p in np.asarray(poly, np.float64).reshape(-1)]
    theta = poly[4] if len(poly) > 4 else 0.0
    first_point = np.asarray(first_point, np.float64).reshape(-1)[:2]
    size = np.array([poly[2], poly[3]], np.float64)
    polygon = _transform_poly(_center2poly(poly[:4]), _rot_matrix(poly[0], poly[1], theta))           # :120-125
    fir = (polygon - first_point) ** 2
    shift = int(np.argmin(fir[:, 0] + fir[:, 1]))
    w_z = size[0] + cfg.context_amount * np.sum(size)                                                  # :132-136
    h_z = size[1] + cfg.context_amount * np.sum(size)
    s_z = float(np.floor(np.sqrt(w_z * h_z)))
    s_x = float(np.floor(s_z * float(np.round(cfg.instance_size / cfg.exemplar_size))))
    avg = [float(a) for a in np.asarray(channel_average).reshape(-1)]
    tr = np.zeros(TRACK_DOUBLES, np.float64)
    tr[0:2], tr[2:4], tr[4], tr[6], tr[8] = poly[0:2], size, theta, 1.0, 1.0
    tr[14:16], tr[16], tr[17:19], tr[19], tr[20:23] = size, s_z, poly[0:2], shift, avg
    tr[24:30] = [poly[0], poly[1], s_x] + avg
    tr[32:38] = [1, 0, 0, 0, 1, 0]
    tr[40:46] = [poly[0], poly[1], s_z] + avg
    return tr, np.array([poly[0], poly[1], s_z, s_x, 0.0] + avg, np.float64), shift, s_z


class SimiTracker:
    def __init__(self, model, cfg: TrackerConfig = None, graph: bool = False, scale_score_thresh: float = 0.5, batch_template: bool = None):
        """model: the reference's ModelBuilder interface (template / track_new / track_new_lp) in eval mode on the GPU.
        scale_score_thresh: cfg.TRACK.SCALE_SCORE_THRESH (0.5 in hdn/core/config.py:527 and every shipped YAML).
        batch_template: refresh the template with one backbone pass over (crop, log-polar crop) instead of template()'s two (checked against
        template() at init; default: HDN_SIMI_BATCH_TEMPLATE, on)."""
        self.model = model
        self.batch_template = (os.environ.get("HDN_SIMI_BATCH_TEMPLATE", "1") not in ("", "0")) if batch_template is None else bool(batch_template)
        self._batched_ok = False
        self.cfg = cfg or TrackerConfig()
        self.use_graph, self._graph = bool(graph), None
        self.scale_score_thresh = float(scale_score_thresh)
        self.host_syncs = 0
        self._zf_static = None
        self.n = 1          # sequences advanced per call (BatchedSimiTracker: n > 1)

    # -------------------------------------------------------------------------------------------------- template (init + refresh)
    def _template_features(self, z_crop):
        """(zf, zf_lp) of ModelBuilder.template(z_crop) (model_builder_e2e_unconstrained_v2.py:87-96).  With `batch_template` the crop and its
        log-polar image go through the backbone as ONE batch of 2n instead of two passes of n (the 127-px passes are launch-bound; the necks keep
        their own halves) — used only after _check_batched_template found it equal to the model's own template() on this sequence's first crop."""
        m = self.model
        with torch.no_grad():
            if not self._batched_ok:
                m.template(z_crop)
                return m.zf, m.zf_lp
            n = z_crop.shape[0]
            f = m.feature_extractor(torch.cat((z_crop[:, 0:3], z_crop[:, 3:6]), 0))
            levels = isinstance(f, (list, tuple))
            a, b = ([t[:n] for t in f], [t[n:] for t in f]) if levels else (f[:n], f[n:])
            if hasattr(m, "neck"):                                 # cfg.ADJUST.ADJUST (the attribute exists exactly then, :47-51)
                a, b = m.neck(a), m.neck_lp(b)
        return a, b

    def _check_batched_template(self, z_crop, rtol: float = 1e-4) -> bool:
        """Is the one-pass form this model's template()?  Compared on the sequence's first crop (max difference against the largest magnitude of the
        level: a different convolution algorithm at 2n is rounding, a model whose template() does something else is not); on any doubt the
        model's own method stays."""
        m = self.model
        if not (self.batch_template and hasattr(m, "feature_extractor") and hasattr(m, "neck") == hasattr(m, "neck_lp")):
            return False
        flat = lambda f: list(f) if isinstance(f, (list, tuple)) else [f]
        self._batched_ok = False
        own = [t.detach().clone() for f in self._template_features(z_crop) for t in flat(f)]
        try:
            self._batched_ok = True
            got = [t for f in self._template_features(z_crop) for t in flat(f)]
        except Exception:
            return False
        finally:
            self._batched_ok = False
        if len(got) != len(own) or any(g.shape != o.shape for g, o in zip(got, own)):
            return False
        return all(float((g - o).abs().max()) <= rtol * max(float(o.abs().max()), 1e-30) for g, o in zip(got, own))

    def _template(self, z_crop, first: bool = False):
        """ModelBuilder.template(z_crop) with the resulting features COPIED into buffers that stay where they are: the per-frame refresh
        is then an in-place update (capturable; the heads' template cache keys on the tensors' version counters)."""
        m = self.model
        if first:
            self._batched_ok = False
            self._batched_ok = self._check_batched_template(z_crop)       # (two host reads per level, once per sequence)
        new = list(self._template_features(z_crop))
        if self._zf_static is None:
            self._zf_static = [[t.detach().clone() for t in f] if isinstance(f, (list, tuple)) else f.detach().clone() for f in new]
        else:
            for dst, src in zip(self._zf_static, new):
                if isinstance(dst, list):
                    for d, s in zip(dst, src):
                        d.copy_(s)
                else:
                    dst.copy_(src)
        m.zf, m.zf_lp = self._zf_static

    # -------------------------------------------------------------------------------------------------- init
    def init(self, img, bbox, poly, *rest):
        """img: BGR uint8 [H,W,3]; bbox (x, y, w, h); poly (cx, cy, w, h, theta); then first_point — hdnTracker.init's own signature — or
        gt_points, first_point (what tools/test.py passes); first_point: (x, y) of the first ground-truth corner."""
        if not rest:
            raise TypeError("init(img, bbox, poly, first_point) or init(img, bbox, poly, gt_points, first_point)")
        c = self.cfg
        self.dev = next(self.model.parameters()).device
        self.ratio = float(np.round(c.instance_size / c.exemplar_size))
        frame = FR.upload(img)
        self.init_frame = frame.clone() if isinstance(img, torch.Tensor) and img.is_cuda else frame        # update_template rotates THIS frame, every frame
        self.channel_average = frame.to(torch.float64).mean(dim=(0, 1)).cpu().numpy()                     # :138 (one read per sequence)
        self.host_syncs += 1
        self.frame_hw = (int(frame.shape[0]), int(frame.shape[1]))
        tr, sq, self.poly_shift_l, self.init_s_z = sequence_records(poly, rest[-1], self.channel_average, c)
        self.init_pos, self.init_size = tr[17:19].copy(), tr[14:16].copy()
        self.track = torch.from_numpy(tr).to(self.dev).view(1, -1)
        self.seq = torch.from_numpy(sq).to(self.dev).view(1, -1)
        assert self.seq.shape[1] == SEQ_DOUBLES
        self._dec = SimilarityDecoder(self.dev, c)
        self.state = self._dec.new_state(1)
        self._out = torch.zeros((1, OUT_DOUBLES), dtype=torch.float64, device=self.dev)
        self._zf_static, self._graph = None, None
        z_crop = FR.get_subwindow(frame, None, c.exemplar_size, None, None, params=self.track[:, 40:46], islog=1)    # :140-143
        self._template(z_crop, first=True)
        return z_crop

    # -------------------------------------------------------------------------------------------------- one frame
    def _body(self, frame):
        """Everything of a frame on the device; no host values, no data-dependent allocations.  -> out float64 [1, 20] (a buffer of this object)."""
        c, m, dec = self.cfg, self.model, self._dec
        x_crop = FR.get_subwindow(frame, None, c.instance_size, None, None, params=self.track[:, 24:30])                 # :189-192
        with torch.no_grad():
            o = m.track_new(x_crop)
        dec.translation(o["cls"], o["loc_c"], self.seq, self.state)                                                   # :193-233
        x_moved = FR.get_subwindow(frame, None, c.instance_size, None, None, params=self.state[:, 8:14])                # :236-238
        with torch.no_grad():
            o = m.track_new_lp(x_moved, [0, 0])
        dec.logpolar(o["cls_lp"], o["loc_lp"], self.seq, self.state)                                                  # :241-246
        with _lib.device_guard(self.dev):                                                                             # :213-227, :247-283 and the next :176-188
            _lib.check(_lib.load().hdn_simi_track_update_f64(_lib.ptr(self.state), _lib.ptr(self.track), _lib.ptr(self.seq), _lib.ptr(self._out), self.n,
                                                             self.frame_hw[1], self.frame_hw[0], self.scale_score_thresh, c.context_amount, self.ratio,
                                                             _lib.stream_ptr(self.dev)), "simi track update")
        rot_img = FR.warp_affine_cubic(self.init_frame, self.track[:, 32:38])                                          # update_template, :156-162
        z_crop = FR.get_subwindow(rot_img, None, c.exemplar_size, None, None, params=self.track[:, 40:46], islog=1)
        self._template(z_crop)
        return self._out

    def _snapshot(self):
        return [self.track.clone(), self.seq.clone(), self.state.clone()] + [t.clone() for f in self._zf_static for t in (f if isinstance(f, list) else [f])]

    def _restore(self, snap):
        live = [self.track, self.seq, self.state] + [t for f in self._zf_static for t in (f if isinstance(f, list) else [f])]
        for d, s in zip(live, snap):
            d.copy_(s)

    def _capture(self, frame_shape):
        """One hipGraph for the whole body.  The warm-up / capture runs advance the recurrences and refresh the template: everything they
        touched is put back afterwards (track record, seq, state, template features)."""
        self._static_frame = torch.empty(frame_shape, dtype=torch.uint8, device=self.dev)
        self._static_frame.copy_(self.init_frame)
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._body(self._static_frame)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._g_out = self._body(self._static_frame)
            self._graph = graph
        finally:
            torch.cuda.current_stream().wait_stream(side)
            self._restore(snap)

    def track_new(self, fr_idx, img, gt_box=None, gt_poly=None, gt_points=None, sync: bool = True):
        """-> {'bbox', 'bbox_aligned', 'best_score', 'rot', 'polygon'} as hdnTracker.track_new (:295-301); sync=False: the float64 device record
        [20] (include/hdn_hip.h: hdn_simi_track_update_f64's `out`) without any host read."""
        if tuple(img.shape[:2]) != self.frame_hw:
            raise ValueError(f"frames of a sequence have one size: init saw {self.frame_hw}, this one is {tuple(img.shape[:2])}")
        if self.use_graph:
            t = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
            if self._graph is None:
                try:
                    self._capture(tuple(t.shape))
                except RuntimeError as e:
                    import warnings
                    warnings.warn(f"hdn_amd: the per-frame body could not be captured as a hipGraph ({type(e).__name__}: {e}); running it eagerly")
                    self.use_graph, self._graph = False, None
                    return self.track_new(fr_idx, img, gt_box, gt_poly, gt_points, sync=sync)
            self._static_frame.copy_(t, non_blocking=True)
            self._graph.replay()
            out = self._g_out
        else:
            out = self._body(FR.upload(img))
        if not sync:
            return {"record": out.clone().view(-1)}
        h = out.view(-1).cpu().numpy()
        self.host_syncs += 1
        return {"bbox": [h[0], h[1], h[2], h[3]], "bbox_aligned": [h[4], h[5], h[6], h[7]], "best_score": np.float32(h[8]), "rot": h[9],
                "polygon": h[10:18].reshape(4, 2).copy()}

    def track(self, img):
        return self.track_new(None, img)

    def track_state(self) -> dict:
        """The recurrence state the reference keeps as host attributes (center_pos, size, rot, lp_shift, scale, v, lost_count, ...), read back on
        demand (one device->host read, counted)."""
        t = self.track.view(-1).cpu().numpy()
        self.host_syncs += 1
        return {"center_pos": t[0:2].copy(), "size": t[2:4].copy(), "rot": float(t[4]), "lp_shift": [0, float(t[5])], "scale": float(t[6]), "v": float(t[7]),
                "window_scale_factor": float(t[8]), "lost_count": int(t[9]), "last_lost": bool(t[10]), "rot_is_float32": bool(t[11]),
                "lp_shift_is_float32": bool(t[12]), "frames": int(t[13])}


class BatchedSimiTracker(SimiTracker):
    """n independent sequences of the similarity-only tracker in lock step on ONE GPU (what hdn_amd.batched_tracker is for the homography tracker; the
    reference: several videos at once by hand-split ranges, tools/test.py:91-103).  Every kernel of the frame body already takes a batch — the crops and
    the rotation of the resident first frames (blockIdx.y = sequence, per-sequence records read out of the [n, 48] arrays), the two decodes, the update
    kernel — and the model's own forward code is batch-general (n templates, refreshed in place every step).  One step = one upload of [n,H,W,3], one
    hipGraph replay, one host read of [n, 20].  Sequence b of a batch runs exactly the code it runs alone: tests/test_gpu_simi_tracker.py holds every
    sequence to its own B = 1 run and to the CPU loop."""

    def __init__(self, model, n: int, cfg: TrackerConfig = None, graph: bool = False, scale_score_thresh: float = 0.5, batch_template: bool = None):
        super().__init__(model, cfg=cfg, graph=graph, scale_score_thresh=scale_score_thresh, batch_template=batch_template)
        if n < 1:
            raise ValueError("n must be >= 1")
        self.n = int(n)
        self._staging = self._copy_done = None

    def _upload(self, imgs, into=None):
        from .batched_tracker import BatchedHomoTracker
        return BatchedHomoTracker._upload(self, imgs, into)          # (one pinned staging buffer, one asynchronous copy; the same checks)

    def init(self, imgs, bboxes, polys, *rest):
        """imgs n x BGR uint8 [H,W,3] (one size); bboxes, polys (cx, cy, w, h, theta) and first_points per sequence (hdnTracker.init's arguments; the
        launchers' extra gt_points list is accepted in front of first_points)."""
        if not rest:
            raise TypeError("init(imgs, bboxes, polys, first_points) or init(imgs, bboxes, polys, gt_points, first_points)")
        first_points, c, n = rest[-1], self.cfg, self.n
        if not (len(bboxes) == len(polys) == len(first_points) == n):
            raise ValueError(f"init takes {n} bboxes / polys / first_points")
        self.dev = next(self.model.parameters()).device
        self.ratio = float(np.round(c.instance_size / c.exemplar_size))
        frames = self._upload(imgs)
        self.init_frame = frames.clone()                              # update_template rotates THESE frames, every step
        self.channel_average = frames.to(torch.float64).mean(dim=(1, 2)).cpu().numpy()
        self.host_syncs += 1
        self.frame_hw = (int(frames.shape[1]), int(frames.shape[2]))
        recs = [sequence_records(polys[b], first_points[b], self.channel_average[b], c) for b in range(n)]
        self.track = torch.from_numpy(np.stack([r[0] for r in recs])).to(self.dev).contiguous()
        self.seq = torch.from_numpy(np.stack([r[1] for r in recs])).to(self.dev).contiguous()
        self.poly_shift_l, self.init_s_z = [r[2] for r in recs], [r[3] for r in recs]
        self._dec = SimilarityDecoder(self.dev, c)
        self.state = self._dec.new_state(n)
        self._out = torch.zeros((n, OUT_DOUBLES), dtype=torch.float64, device=self.dev)
        self._zf_static, self._graph = None, None
        z_crop = FR.get_subwindow(frames, None, c.exemplar_size, None, None, params=self.track[:, 40:46], islog=1)
        self._template(z_crop, first=True)
        return z_crop

    def track_new(self, fr_idx, imgs, gt_box=None, gt_poly=None, gt_points=None, sync: bool = True):
        """One frame of every sequence -> n result dictionaries with hdnTracker.track_new's keys (sync=True: one host read for all), or the float64 device
        records [n, 20] (sync=False)."""
        n = self.n
        shape = tuple(imgs.shape) if isinstance(imgs, torch.Tensor) else (len(imgs),) + tuple(np.asarray(imgs[0]).shape)
        if shape[0] != n or tuple(shape[1:3]) != self.frame_hw:
            raise ValueError(f"a step takes {n} frames of {self.frame_hw}, got {shape}")
        if self.use_graph:
            if self._graph is None:
                try:
                    self._capture(shape)
                except RuntimeError as e:
                    import warnings
                    warnings.warn(f"hdn_amd: the batched per-frame body could not be captured as a hipGraph ({type(e).__name__}: {e}); running it eagerly")
                    self.use_graph, self._graph = False, None
                    return self.track_new(fr_idx, imgs, sync=sync)
            self._upload(imgs, into=self._static_frame)
            self._graph.replay()
            out = self._g_out
        else:
            out = self._body(self._upload(imgs))
        if not sync:
            return {"record": out.clone()}
        h = out.cpu().numpy()
        self.host_syncs += 1
        return [{"bbox": list(h[b, 0:4]), "bbox_aligned": list(h[b, 4:8]), "best_score": np.float32(h[b, 8]), "rot": h[b, 9],
                 "polygon": h[b, 10:18].reshape(4, 2).copy()} for b in range(n)]

    def track_state(self) -> list:
        t = self.track.cpu().numpy()
        self.host_syncs += 1
        return [{"center_pos": r[0:2].copy(), "size": r[2:4].copy(), "rot": float(r[4]), "lp_shift": [0, float(r[5])], "scale": float(r[6]), "v": float(r[7]),
                 "lost_count": int(r[9]), "frames": int(r[13])} for r in t]


class DeviceTrackerSimi(SimiTracker):
    """Drop-in for hdnTracker behind build_tracker(model) (hdn/tracker/tracker_builder.py:18-19).  One hipGraph per frame, BatchNorm-folded
    backbone / necks and MIOpen find mode around this tracker's calls, exactly as hdn_amd.tracker.DeviceTrackerHomo."""

    def __init__(self, model, graph: bool = None, cfg: TrackerConfig = None, fold_backbone: bool = None, batch_template: bool = None):
        thresh = 0.5
        if cfg is None:
            cfg = TrackerConfig()
            try:
                from hdn.core.config import cfg as ref_cfg
                cfg = TrackerConfig.from_reference(ref_cfg)
                thresh = float(ref_cfg.TRACK.SCALE_SCORE_THRESH)
            except (ImportError, AttributeError):
                pass
        if graph is None:
            graph = os.environ.get("HDN_TRACKER_GRAPH", "1") not in ("", "0")
        model.eval()
        self.miopen_find = os.environ.get("HDN_MIOPEN_FIND", "1") not in ("", "0") and next(model.parameters()).is_cuda
        from . import backbone as BB
        self.folded = BB.optimize_similarity_model(model) if (BB.enabled() if fold_backbone is None else fold_backbone) else []
        super().__init__(model, cfg=cfg, graph=graph, scale_score_thresh=thresh, batch_template=batch_template)

    def _find_mode(self):
        import contextlib
        if not self.miopen_find:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def only_benchmark():
            before = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            try:
                yield
            finally:
                torch.backends.cudnn.benchmark = before
        return only_benchmark()

    def init(self, img, bbox, poly, *rest):
        with self._find_mode():
            return super().init(img, bbox, poly, *rest)

    def track_new(self, fr_idx, img, gt_box=None, gt_poly=None, gt_points=None, sync: bool = True):
        with self._find_mode():
            return super().track_new(fr_idx, img, gt_box, gt_poly, gt_points, sync=sync)
