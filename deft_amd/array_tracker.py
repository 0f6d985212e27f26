"""DEFT's tracking loop with the track state held in arrays -- `Tracker.update` + `STrack` (utils/tracker.py:140-1056) for every
configuration the reference runs: MOT17 / KITTI (2-D) and nuScenes (3-D, one tracker per class), Kalman or LSTM motion model.

Why arrays.  The reference keeps one Python object per track and walks the pool several times per frame (multi_predict, get_similarity,
fuse_motion, iou_distance, update / activate): at 100-200 live tracks that is milliseconds of interpreter time per frame -- more than the
network pass (profiles/r3_tracker_ops.json: 0.62 ms of 3.5 ms in per-track Python alone).  Here a frame is a fixed
sequence of array operations over the pool:

  pool state          one row per live track in `self.cols` (ids, flags, counters, last nodes, Kalman mean / covariance or the LSTM side's
                      last observation + running observation statistics, 3-D box, depth, ...)
  detections          one array per field, built once from the frame's results
  embeddings          model.AFE (AfeSeam): centres -> embeddings, new frame scored against the stored frames the pool can read
                      (deft_amd.tracker.FeatureRecorder, lazy blocks), tracks x detections similarity medians on the device
                      (deft_track_similarity) -- ONE device round trip per frame, after which `after_device_work` fires
  gates / costs /     ONE native host call per frame (per class for nuScenes): deft_associate_2d / deft_associate_ddd (csrc/assoc.hip) -- the motion
  assignment          gate, the fused costs, the 3-D and 2-D IoU matrices and the three / four assignments (own Jonker-Volgenant, fixed tie order)
                      on the similarity matrix as it lands in pinned memory; `native_assoc = False`: the same stages as numpy matrices
                      (association._maha2, bbox_overlaps, deft_iou3d_matrix, deft_lapjv), kept as the cross-check
  state update        Kalman predict / update of the pool's mean / covariance columns in place (deft_kf_predict / deft_kf_update) or ONE
                      deft_motion_step launch for every track touched in the frame (features + LSTM + future boxes; fetched lazily, when the
                      next frame first needs a prediction)
  ahead of update()   begin(): everything above that needs only the frame's detections and the track table of the previous frame, queued before
                      update() is called for it (Detector.run: the next frame of a finished lookahead pass; all per-class trackers of a nuScenes frame)

Semantics are the reference's, statement by statement (tests/test_mot_tracker.py replays scenes through the reference's own Tracker for
mot / kitti_tracking / nuscenes x Kalman / LSTM and requires identical ids, flags and boxes frame by frame):
  * nothing is ever marked Lost: `lost_stracks` stays empty, an unmatched track stays Tracked until `max_time_lost` frames have passed
    and is then removed in the IoU stage (tracker.py:1006-1010) -- for KITTI / nuScenes only while it is still an IoU candidate (seen
    within 6 / 3 frames, :982-990), i.e. never: such tracks simply stop matching and stay in the pool (kept as the reference does);
  * nuScenes, every class but pedestrian: a first association on 3-D IoU between detections and the tracks seen in the last 3 frames
    (:850-884, thresh 0.999), then the embedding association on the rest with the 3-D motion gate (fuse_motion_ddd), then a
    similarity-only association (:927-953), then a 2-D IoU stage with threshold 0 (:1001-1004: only identical boxes match);
  * KITTI: similarity-only second association (:954-980); MOT: straight to IoU;
  * LSTM: no Kalman predict; fuse_motion uses the "gaussian" distance, which is identically zero on the 2-D position-only path unless
    the track has >= 300 observations (then Mahalanobis on the predicted box with np.cov of its observations, matching.py:339-366);
    the IoU stage uses the LSTM's predicted box (prediction_at_frame_tlbr, float32 arithmetic like the reference's arrays);
  * ids come from the process-wide counter `kalman.TrackIds` (basetrack.py:18, 40-42).
"""
import ctypes as C

import numpy as np
import torch

from . import association as A
from . import tracker as DT
from .kalman import (NEW, TRACKED, LOST, REMOVED, TrackIds, Node, kf_initiate, kf_multi_predict, kf_multi_update,  # noqa: F401
                     tlbr_to_tlwh, tlwh_to_xyah, _F, _H, _SP, _SV)

TRACKED, REMOVED = 1, 3                         # basetrack.py:11-15 (New = 0 and Lost = 2 never occur in a pool)


class TrackView(object):
    """What `update()` returns and `tracked_stracks` lists: one track as the reference's STrack shows it to its callers (src/test.py:
    220-258: tlwh, track_id, score; nuScenes: org_ddd_box, classe, ddd_bbox, ddd_submission), a snapshot taken at the end of the frame."""
    __slots__ = ("track_id", "is_activated", "tracklet_len", "score", "tlwh", "frame_id", "start_frame", "state", "ddd_bbox", "depth",
                 "org_ddd_box", "ddd_submission", "classe")

    def __init__(self, track_id=None, is_activated=None, tracklet_len=None, score=None, tlwh=None, frame_id=None, start_frame=None,
                 ddd_bbox=None, depth=None, org_ddd_box=None, ddd_submission=None, classe=None):
        self.track_id, self.is_activated, self.tracklet_len, self.score, self.tlwh = track_id, is_activated, tracklet_len, score, tlwh
        self.frame_id, self.start_frame, self.state = frame_id, start_frame, TRACKED
        self.ddd_bbox, self.depth, self.org_ddd_box, self.ddd_submission, self.classe = ddd_bbox, depth, org_ddd_box, ddd_submission, classe

    end_frame = property(lambda self: self.frame_id)

    @property
    def tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r

    def __repr__(self):
        return "OT_{}_({}-{})".format(self.track_id, self.start_frame, self.frame_id)


class _Columns(object):
    """Structure of arrays with a common first dimension (the pool): append rows, keep a subset, index."""

    def __init__(self, spec):
        self.spec = spec                         # name -> (trailing shape, dtype)
        self.a = {k: np.zeros((0,) + shp, dt) for k, (shp, dt) in spec.items()}
        self.n = 0

    def __getitem__(self, k):
        return self.a[k]

    def append(self, m, **vals):
        for k, (shp, dt) in self.spec.items():
            v = vals.get(k)
            new = np.zeros((m,) + shp, dt) if v is None else np.asarray(v, dt).reshape((m,) + shp)
            self.a[k] = np.concatenate([self.a[k], new], 0)
        self.n += m

    def keep(self, idx):
        for k in self.a:
            self.a[k] = self.a[k][idx]
        self.n = len(self.a["tid"])


def _p(a):
    """Host pointer of a C-contiguous numpy array (the native helpers take raw pointers)."""
    assert a.flags["C_CONTIGUOUS"], "native helper needs a contiguous array"
    return C.c_void_p(a.ctypes.data)


def _finite_sim(sim):
    """The similarity matrix is where the embedding / affinity chain (AfePlan: the same split arithmetic as the backbone, no heat map in between)
    lands on the host: a non-finite entry means an operand left the two-fp16-piece range.  The native cascade would treat such pairs as gated --
    silent non-matches and identity switches; raise instead (Detector.run moves to the range-free arithmetic)."""
    if sim is not None and not np.isfinite(sim).all():
        raise FloatingPointError("non-finite track / detection similarity: an operand of the embedding / affinity chain left the range of the "
                                 "two-fp16-piece arithmetic (|x| >= 4094, csrc/common.h); run on the three-bf16-piece entry points (DEFT_ARITH=bf16x3)")


class ArrayTracker(object):
    lazy_blocks = True             # score the new frame only against the stored frames the pool's selected nodes live in
    native_assoc = True            # the association cascade of a frame in ONE host call (deft_associate_2d / deft_associate_ddd) and the Kalman filter in
    #                                two (deft_kf_predict / deft_kf_update); False = the numpy stages below, kept as the cross-check of the native calls
    after_device_work = None       # set by a caller (Detector.run's lookahead): called ONCE per update(), as soon as the frame's last
    #                                result-bearing device step has been read back; what follows is host work (plus one tiny motion launch)

    def __init__(self, opt, model, h=100, w=100, frame_rate=10):
        """opt: dataset ("mot" | "kitti_tracking" | "nuscenes"), lstm, track_buffer, max_object (+ load_model_traj for the LSTM).
        model: carries `.AFE` (deft_amd.integrate.AfeSeam); with opt.lstm optionally `.motion` (a deft_amd.tracker.MotionBank, or any
        object with alloc / free / step(slots, boxes, frame_id) -> (features, predictions [T, fut, dim])) -- built from
        deft_amd.integrate.KalmanFilterLSTM(opt) when absent.  h, w: the image size detection centres are normalised with
        (tracker.py:817-820; 100 until reset_tracking passes the real one)."""
        assert opt.dataset in ("mot", "kitti_tracking", "nuscenes"), opt.dataset
        self.opt, self.dataset, self.model = opt, opt.dataset, model
        self.img_height, self.img_width = h, w
        self.frame_id = 0
        self.max_time_lost = int(frame_rate / 30.0 * getattr(opt, "track_buffer", 30))          # tracker.py:648-649
        self.recorder = DT.FeatureRecorder(opt.dataset)
        self.det_thresh = 0.0
        self.use_lstm = bool(getattr(opt, "lstm", False))
        self.ddd = self.dataset == "nuscenes"
        self.mm = 2 if self.ddd else 4                                # STrack.get_similarity, tracker.py:233-236
        self.L = self.mm + 2                                          # nodes kept per track: enough to tell "more than mm + 1 young nodes"
        self.fut = 4 if self.ddd else 5                               # kalman_filter_lstm.py:60-63
        self.bank = None
        if self.use_lstm:
            self.bank = getattr(model, "motion", None)
            if self.bank is None:
                from . import integrate
                self.bank = DT.MotionBank(integrate.KalmanFilterLSTM(opt))
        od = 7 if self.ddd else 4
        spec = {"tid": ((), np.int64), "act": ((), bool), "score": ((), np.float64), "tlen": ((), np.int64), "fid": ((), np.int64),
                "start": ((), np.int64), "nf": ((self.L,), np.int64), "ni": ((self.L,), np.int64), "nn": ((), np.int64)}
        if self.use_lstm:
            spec.update({"tlwh": ((4,), np.float64), "slot": ((), np.int64), "nobs": ((), np.int64), "omean": ((4,), np.float64),
                         "om2": ((4, 4), np.float64)})
        else:
            spec.update({"mean": ((8,), np.float64), "cov": ((8, 8), np.float64)})
        if self.ddd:
            spec.update({"ddd": ((7,), np.float64), "depth": ((), np.float64), "org": ((), object), "sub": ((), object)})
        self.cols = _Columns(spec)
        self._od = od
        self.fut_arr = np.zeros((0, self.fut, od))          # LSTM: predictions of the pool rows [T, fut, dim] on the host ...
        self._pending = None                                  # ... and the motion step whose result has not been read back yet
        self.removed_ids = []
        self._begun = None                                    # begin(): the device half of the next frame, already queued
        self._prepared = None                                 # prepare(): embeddings + affinity blocks of the frame AFTER that one, already queued
        self.lost_stracks = []
        self.classe = None

    # ---- views ------------------------------------------------------------------------------------------------------------------
    def _tlwh_rows(self, idx):
        c = self.cols
        if self.use_lstm:
            return c["tlwh"][idx].copy()
        r = c["mean"][idx][:, :4].copy()
        r[:, 2] *= r[:, 3]
        r[:, :2] -= r[:, 2:] / 2
        return r

    def _views(self, idx, classe=None):
        c = self.cols
        idx = np.asarray(idx, dtype=int)
        if len(idx) == 0:
            return []
        tl = self._tlwh_rows(idx)
        tid, act, tlen = c["tid"][idx].tolist(), c["act"][idx].tolist(), c["tlen"][idx].tolist()
        fid, start = c["fid"][idx].tolist(), c["start"][idx].tolist()
        score = c["score"][idx] if self.ddd else c["score"][idx].astype(np.float32)      # (the 2-D datasets' rows are float32, tracker.py:790-803)
        if self.ddd:
            ddd, depth, org, sub = c["ddd"][idx], c["depth"][idx], c["org"][idx], c["sub"][idx]
        if not self.ddd:
            return [TrackView(tid[k], act[k], tlen[k], score[k], tl[k], fid[k], start[k]) for k in range(len(idx))]
        return [TrackView(tid[k], act[k], tlen[k], score[k], tl[k], fid[k], start[k], ddd[k].copy(), depth[k], org[k], sub[k], self.classe)
                for k in range(len(idx))]

    @property
    def tracked_stracks(self):
        return self._views(np.arange(self.cols.n))

    @property
    def removed_stracks(self):
        return self.removed_ids

    # ---- similarity -------------------------------------------------------------------------------------------------------------
    def _selected_nodes(self, fid):
        """For every pool row: (frames [T, L], ids [T, L], valid mask [T, L]) of the nodes STrack.get_similarity medians over
        (tracker.py:221-248), oldest first."""
        c = self.cols
        nf, ni, nn = c["nf"], c["ni"], c["nn"]
        T, L = nf.shape
        lib = self._track_nodes_lib()
        if lib is not None and T:                                      # the same rule in one host call (assoc.hip deft_track_nodes)
            sel = np.empty((T, L), np.uint8)
            rc = lib._fn["deft_track_nodes"](_p(nf), _p(ni), _p(nn), T, L, int(fid), self.mm, DT.max_track_node, _p(sel), None, None, None, None, 0,
                                             None, None, None, None)
            if rc != 0:
                raise RuntimeError("deft_track_nodes failed (%d): %s" % (rc, lib.last_error()))
            return nf, ni, sel.view(np.bool_)
        stored = np.minimum(nn, L)
        pos = np.arange(L)[None, :]
        have = pos >= (L - stored)[:, None]                            # nodes are right-aligned: the newest at column L - 1
        young = have & (fid - nf < DT.max_track_node)
        q = young.sum(1)                                               # young nodes form a suffix; with all L stored ones young there may be more
        nsel = np.where(q <= self.mm + 1, q, self.mm)
        sel = pos >= (L - nsel)[:, None]
        return nf, ni, sel

    def _track_nodes_lib(self):
        """The library when the node selection / gather table run natively (native_assoc, and model.AFE.plan.lib is a bound HipLib)."""
        if not self.native_assoc:
            return None
        lib = getattr(getattr(self.model.AFE, "plan", None), "lib", None)
        return lib if lib is not None and "deft_track_nodes" in getattr(lib, "_fn", ()) else None

    def _similarity(self, fid, rows_idx, nd, sel_all, defer=False):
        """deft_amd.tracker.get_similarity on the node arrays: float64 [len(rows_idx), nd + 1].  defer: queue the launch and the copy back, return
        a callable that waits for them -- update() does host work that does not need the matrix (Kalman prediction, the motion gate) in between."""
        if defer:
            got = self._similarity(fid, rows_idx, nd, sel_all, defer=None)
            return got if callable(got) else (lambda raw=False: got)
        T = len(rows_idx)
        if T == 0:
            return np.array([])
        nf, ni, sel = sel_all
        if T != nf.shape[0] or (T and (rows_idx[0] != 0 or rows_idx[-1] != T - 1)):     # (begin() asks for every pool row: no gather then)
            nf, ni, sel = nf[rows_idx], ni[rows_idx], sel[rows_idx]
        rec = self.recorder
        if rec._dev is None or rec._dev[0] != fid:
            if nd == 0 or not sel.any():
                return np.zeros((T, nd + 1))
            raise KeyError("no affinity blocks recorded for frame %r" % (fid,))
        _, sim, starts, index = rec._dev
        assert sim.shape[1] == nd + 1
        L = sel.shape[1]
        plan = self.model.AFE.plan
        dev = sim.device
        n1 = T * L
        need_in, need_out = 2 * n1 + T, T * (nd + 1)
        pin = None
        if dev.type == "cuda":                                         # pinned staging both ways: no pageable (= blocking) copies in the frame
            pin = getattr(self, "_pin", None)
            if pin is None or pin[0].numel() < need_in or pin[1].numel() < need_out:
                pin = self._pin = (torch.empty(max(2048, 2 * need_in), dtype=torch.int32).pin_memory(),
                                   torch.empty(max(32768, 2 * need_out), dtype=torch.float32).pin_memory())
        lib = self._track_nodes_lib()
        host = None
        if lib is not None and sel is sel_all[2]:                      # every pool row (begin()): selection + table in ONE host call, written in place
            arr = getattr(rec, "_dev_arrays", None)
            if arr is None or arr[0] is not index:                     # the block table as arrays, once per recorded frame
                st = np.asarray(starts, np.int64)
                blk = np.fromiter((b for b, _ in index.values()), np.int64, len(index))
                arr = rec._dev_arrays = (index, np.fromiter(index.keys(), np.int64, len(index)), np.ascontiguousarray(st[blk]),
                                         np.ascontiguousarray(st[blk + 1] - st[blk]), np.fromiter((d for _, d in index.values()), np.float32, len(index)))
            c = self.cols
            if pin is not None:
                base_h = pin[0].data_ptr()
            else:
                host = np.empty(need_in, np.int32)
                base_h = host.ctypes.data
            bad = C.c_longlong(0)
            rc = lib._fn["deft_track_nodes"](_p(nf), _p(ni), _p(c["nn"]), T, L, int(fid), self.mm, DT.max_track_node, None, _p(arr[1]), _p(arr[2]),
                                             _p(arr[3]), _p(arr[4]), len(index), C.c_void_p(base_h), C.c_void_p(base_h + 4 * n1),
                                             C.c_void_p(base_h + 8 * n1), C.byref(bad))
            if rc == -96:
                raise KeyError(int(bad.value))                         # KeyError like the reference for a frame without a block
            if rc == -97:
                raise IndexError("node id outside its frame")
            if rc != 0:
                raise RuntimeError("deft_track_nodes failed (%d): %s" % (rc, lib.last_error()))
            return self._similarity_launch(plan, sim, dev, pin, host, T, L, nd, n1, need_in, defer)
        rows = np.zeros((T, L), np.int32); scale = np.zeros((T, L), np.float32)
        cnt = sel.sum(1).astype(np.int32)
        if sel.any():
            f0, f1 = int(nf[sel].min()), int(nf[sel].max())
            # frame -> (first row of its block, its length, its decay): dense tables over the frames the selected nodes span
            st_t = np.zeros(f1 - f0 + 1, np.int64); ln_t = np.full(f1 - f0 + 1, -1, np.int64); dl_t = np.zeros(f1 - f0 + 1, np.float32)
            fr_k = np.fromiter(index.keys(), np.int64, len(index))
            blk_k = np.fromiter((b for b, _ in index.values()), np.int64, len(index))
            dl_k = np.fromiter((d for _, d in index.values()), np.float32, len(index))
            inr = (fr_k >= f0) & (fr_k <= f1)
            fr_k, blk_k = fr_k[inr] - f0, blk_k[inr]
            st_k = np.asarray(starts, np.int64)
            st_t[fr_k], ln_t[fr_k], dl_t[fr_k] = st_k[blk_k], st_k[blk_k + 1] - st_k[blk_k], dl_k[inr]
            fr = np.where(sel, nf - f0, 0)
            ln = ln_t[fr]
            if (sel & (ln < 0)).any():
                raise KeyError(int(nf[sel & (ln < 0)][0]))                 # KeyError like the reference for a frame without a block
            if (sel & ((ni < 0) | (ni >= ln))).any():
                raise IndexError("node id outside its frame")
            # the selected nodes are a suffix of every row (newest at column L - 1): reversed they are left-aligned, which is what the kernel
            # reads (node_row[t][0 .. cnt - 1]; a median does not care about the order)
            rows = np.where(sel, st_t[fr] + ni, 0)[:, ::-1].astype(np.int32)
            scale = np.where(sel, dl_t[fr], np.float32(0))[:, ::-1].astype(np.float32)
        host = np.concatenate([rows.reshape(-1), scale.view(np.int32).reshape(-1), cnt])
        if pin is not None:
            pin[0][:need_in] = torch.from_numpy(host)
        return self._similarity_launch(plan, sim, dev, pin, host, T, L, nd, n1, need_in, defer)

    def _similarity_launch(self, plan, sim, dev, pin, host, T, L, nd, n1, need_in, defer):
        """The gather table (pin[0][:need_in] on the device path, `host` otherwise: rows | scale | cnt) -> deft_track_similarity -> the copy back."""
        if pin is not None:
            pack = pin[0][:need_in].to(dev, non_blocking=True)
        else:
            pack = torch.from_numpy(host)
        out = torch.empty(T, nd + 1, dtype=torch.float32, device=dev)
        base = pack.data_ptr()
        plan.lib.call("deft_track_similarity", C.c_void_p(sim.data_ptr()), sim.shape[0], nd, C.c_void_p(base), C.c_void_p(base + 4 * n1),
                      C.c_void_p(base + 8 * n1), T, L, C.c_void_p(out.data_ptr()), plan._stream())
        if dev.type == "cuda":
            land = pin[1][:T * (nd + 1)].view(T, nd + 1)
            land.copy_(out, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))               # (an event, not the stream: the per-class trackers of a nuScenes frame queue their
            #                                                            chains back to back on one stream, each waits for its own copy only)

            def wait(raw=False):
                done.synchronize()                                     # THE device round trip of the frame
                return land.numpy() if raw else land.numpy().astype(np.float64)
            return wait if defer is None else wait()
        host = out.numpy()
        return (lambda raw=False: host if raw else host.astype(np.float64)) if defer is None else host.astype(np.float64)

    # ---- LSTM side --------------------------------------------------------------------------------------------------------------
    def _resolve(self):
        """Land the pending motion step (rows it covered, its result or a callable that waits for it) in `fut_arr`."""
        if self._pending is not None:
            rows, res = self._pending
            self._pending = None
            self.fut_arr[rows] = res() if callable(res) else res

    def close(self):
        """End of a video: land the pending motion step and hand this tracker's MotionBank slots back (the bank is shared -- model.motion --
        by every tracker built on the model; without this its h / c / last tensors double with every sequence of an evaluation)."""
        self._take_back()
        if self.use_lstm and self.bank is not None:
            try:
                self._resolve()
            except Exception:
                self._pending = None
            if self.cols.n:
                for s_ in self.cols["slot"][:self.cols.n].tolist():
                    self.bank.free(int(s_))
            self.cols.keep(np.zeros(self.cols.n, bool))
            self.fut_arr = self.fut_arr[:0]

    def _prediction_at(self, idx, fid):
        """STrack.prediction_at_frame (tracker.py:254-262): the predicted (cx, cy, a, h) / 3-D box of pool rows `idx` for frame `fid`."""
        self._resolve()
        delta = fid - self.cols["fid"][idx]
        k = np.where((delta >= 1) & (delta <= self.fut), delta, self.fut) - 1
        return self.fut_arr[idx, k]

    def _motion_step(self, rows, boxes, fid):
        """One deft_motion_step launch for the pool rows touched in this frame; its result is read when somebody asks for a prediction
        (the next frame's IoU stage), not here."""
        self._resolve()
        if self.fut_arr.shape[0] < self.cols.n:
            self.fut_arr = np.concatenate([self.fut_arr, np.zeros((self.cols.n - self.fut_arr.shape[0], self.fut, self._od))], 0)
        if len(rows):
            slots = self.cols["slot"][rows].tolist()
            step_async = getattr(self.bank, "step_async", None)
            res = step_async(slots, boxes, fid) if step_async is not None else self.bank.step(slots, boxes, fid)[1]
            self._pending = (np.asarray(rows), res)

    def _kf(self, entry, *args):
        """deft_kf_predict(T) / deft_kf_update(rows, measurements) on the pool's mean / cov columns, in place."""
        mean, cov = self.cols["mean"], self.cols["cov"]
        assert mean.flags.c_contiguous and cov.flags.c_contiguous and mean.dtype == cov.dtype == np.float64
        lib = self.model.AFE.plan.lib
        if entry == "deft_kf_predict":
            lib.call(entry, C.c_void_p(mean.ctypes.data), C.c_void_p(cov.ctypes.data), int(args[0]))
        else:
            rows, meas = args
            try:
                lib.call(entry, C.c_void_p(mean.ctypes.data), C.c_void_p(cov.ctypes.data), C.c_void_p(rows.ctypes.data), len(rows),
                         C.c_void_p(meas.ctypes.data))
            except Exception as e:
                if "not positive definite" in str(e):
                    raise np.linalg.LinAlgError(str(e))
                raise

    def _associate_2d(self, fid, T, N, sim_wait, xyah, tlbr):
        """tracker.py:886-1030 for mot / kitti_tracking through deft_associate_2d: the per-row inputs (gate centre, Cholesky factor of the 2 x 2
        position covariance, which rows the Mahalanobis gate applies to, which rows may enter the IoU stage, the box the IoU stage compares) are
        O(T) numpy; the O(T x N) matrices and the three assignments are the native call.  -> (matched rows, matched detections, IoU-stage rows left
        unmatched, detections left unmatched)."""
        c = self.cols
        lam = 0.9
        if T:
            if self.use_lstm:
                gated = c["nobs"][:T] >= 300                                # matching.py:342
                mean2 = np.zeros((T, 2)); cov2 = np.tile(np.eye(2), (T, 1, 1))
                pred = self._prediction_at(np.arange(T), fid).astype(np.float32)        # (float32 like the reference's arrays, tracker.py:254-280)
                if gated.any():
                    mean2[gated] = pred[gated].astype(np.float64)[:, :2]
                    cov2[gated] = (c["om2"][:T][gated] / (c["nobs"][:T][gated] - 1)[:, None, None])[:, :2, :2]
                p = pred.copy()
                p[:, 2] *= p[:, 3]
                p[:, :2] -= p[:, 2:] / 2
                p[:, 2:] += p[:, :2]
                a_tlbr = p.astype(np.float64)
            else:
                gated = np.ones(T, bool)
                mean2 = np.ascontiguousarray(c["mean"][:T, :2])
                cov2 = c["cov"][:T, :2, :2]
                a_tlbr = self._tlwh_rows(slice(0, T))
                a_tlbr[:, 2:] += a_tlbr[:, :2]
            with np.errstate(invalid="ignore", divide="ignore"):            # the 2 x 2 Cholesky factor of gating_distance (kalman_filter.py:266-275)
                l00 = np.sqrt(cov2[:, 0, 0])
                l10 = cov2[:, 1, 0] / l00
                l11 = np.sqrt(cov2[:, 1, 1] - l10 * l10)
            chol = np.stack([l00, l10, l11], 1)
            if N and not (np.isfinite(chol).all() and (l00 > 0).all() and (l11 > 0).all()):
                raise np.linalg.LinAlgError("Matrix is not positive definite")
            iou_ok = np.abs(fid - c["fid"][:T]) < 6 if self.dataset == "kitti_tracking" else np.ones(T, bool)
            gated = np.ascontiguousarray(gated, dtype=np.uint8); iou_ok = np.ascontiguousarray(iou_ok, dtype=np.uint8)
            a_tlbr = np.ascontiguousarray(a_tlbr, dtype=np.float64)
        else:
            mean2 = chol = a_tlbr = np.zeros((0, 4)); gated = iou_ok = np.zeros(0, np.uint8)
        meas2 = np.ascontiguousarray(xyah[:, :2], dtype=np.float64)
        d_tlbr = np.ascontiguousarray(tlbr, dtype=np.float64)
        k = min(T, N)
        out = np.empty(2 * k + T + N + 3, np.int32)
        mt, md, lost, new_d, cnt = out[:k], out[k:2 * k], out[2 * k:2 * k + T], out[2 * k + T:2 * k + T + N], out[2 * k + T + N:]
        sim = sim_wait(raw=True) if sim_wait is not None else None     # everything above is host work the device round trip hides behind
        self._device_done()
        _finite_sim(sim)
        if sim is not None:
            assert sim.dtype == np.float32 and sim.shape == (T, N + 1) and sim.flags.c_contiguous
        ptr = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)
        self.model.AFE.plan.lib.call(
            "deft_associate_2d", ptr(sim) if sim is not None else None, N + 1, T, N, ptr(mean2), ptr(chol), ptr(gated), ptr(meas2),
            C.c_double(5.0 * A.chi2inv95[2]), C.c_double(lam), C.c_double(0.05 * (1 - lam)), int(self.dataset == "kitti_tracking"),
            ptr(iou_ok), ptr(a_tlbr), ptr(d_tlbr), C.c_double(0.9), C.c_double(0.9), ptr(mt), ptr(md), C.c_void_p(cnt.ctypes.data),
            ptr(lost), C.c_void_p(cnt.ctypes.data + 4), ptr(new_d), C.c_void_p(cnt.ctypes.data + 8))
        return mt[:cnt[0]].astype(int), md[:cnt[0]].astype(int), lost[:cnt[1]].astype(int), new_d[:cnt[2]].astype(int)

    # ---- one frame --------------------------------------------------------------------------------------------------------------
    def _device_done(self):
        cb, self.after_device_work = self.after_device_work, None
        if cb is not None:
            cb()

    def _associate_ddd(self, fid, T, N, sim_wait, det_ddd, tlbr):
        """tracker.py:850-1030 for one class of a nuScenes frame through deft_associate_ddd (see _associate_2d)."""
        c = self.cols
        lam = 0.9
        if T:
            recent = np.ascontiguousarray(np.abs(c["fid"][:T] - fid) < 3, dtype=np.uint8)      # :852-856 and the IoU stage's filter (:999-1004)
            trk_ddd = np.ascontiguousarray(c["ddd"][:T], dtype=np.float64)
            depth = np.ascontiguousarray(c["depth"][:T], dtype=np.float64)
            a_tlbr = self._tlwh_rows(slice(0, T))
            a_tlbr[:, 2:] += a_tlbr[:, :2]
            a_tlbr = np.ascontiguousarray(a_tlbr, dtype=np.float64)
        else:
            recent = np.zeros(0, np.uint8); trk_ddd = depth = a_tlbr = np.zeros((0, 7))
        d_ddd = np.ascontiguousarray(det_ddd, dtype=np.float64)
        d_tlbr = np.ascontiguousarray(tlbr, dtype=np.float64)
        k = min(T, N)
        out = np.empty(2 * k + T + N + 3, np.int32)
        mt, md, lost, new_d, cnt = out[:k], out[k:2 * k], out[2 * k:2 * k + T], out[2 * k + T:2 * k + T + N], out[2 * k + T + N:]
        sim = sim_wait(raw=True) if sim_wait is not None else None
        self._device_done()
        _finite_sim(sim)
        if sim is not None:
            assert sim.dtype == np.float32 and sim.shape == (T, N + 1) and sim.flags.c_contiguous
        ptr = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)
        self.model.AFE.plan.lib.call(
            "deft_associate_ddd", ptr(sim) if sim is not None else None, N + 1, T, N, int(self.classe != "pedestrian"), ptr(recent), ptr(trk_ddd),
            ptr(d_ddd), ptr(depth), 0 if self.use_lstm else 1, C.c_double(5 if self.classe == "pedestrian" else 10), C.c_double(lam), C.c_double(0.001),
            ptr(recent), ptr(a_tlbr), ptr(d_tlbr), C.c_double(0.999), C.c_double(0.9), C.c_double(0.0), ptr(mt), ptr(md),
            C.c_void_p(cnt.ctypes.data), ptr(lost), C.c_void_p(cnt.ctypes.data + 4), ptr(new_d), C.c_void_p(cnt.ctypes.data + 8))
        return mt[:cnt[0]].astype(int), md[:cnt[0]].astype(int), lost[:cnt[1]].astype(int), new_d[:cnt[2]].astype(int)

    # ---- the device half of a frame, which may run ahead of update() --------------------------------------------------------------------
    def begin(self, results, FeatureMaps, ddd_boxes=None, depths_by_class=None, pre=None, feats=None):
        """The part of update(results, FeatureMaps) that depends only on the frame's detections, its feature maps and the track table as the previous
        update() left it -- detections as arrays, embedding extraction, the affinity blocks against the stored frames the pool reads, the
        similarity medians and their copy back (tracker.py:786-848, 663-688) -- queued NOW.  A caller that already holds the next frame's detections
        (Detector.run with a lookahead pass that has finished) calls this right behind update(k); update(k + 1) must then be given the SAME
        `results` object and finds its device round trip already under way (another object: the early work is taken back and redone).
        The seven per-class trackers of a nuScenes frame are begun one after the other before the first of them is updated: seven device round trips
        in flight at once instead of one at a time.  pre / feats: detections_as_arrays(results, ...) and the frame's embeddings [1, n, D] when the
        caller has them already (extract_together: one extraction for all classes)."""
        if self._begun is not None:
            self._undo(self._begun)
        snap = self._snapshot()
        try:
            self._begun = self._first_half(results, FeatureMaps, ddd_boxes, depths_by_class, pre, feats)
        except Exception:
            self._undo({"snap": snap})                 # the recorder may hold the frame already: the next update() must not find it half recorded
            self._begun = None
            raise
        self._begun.setdefault("snap", snap)           # (a frame that was prepare()d: the recorder as it was BEFORE that)

    def prepare(self, results, FeatureMaps, ddd_boxes=None, depths_by_class=None, pre=None, feats=None):
        """One frame further ahead than begin(): the part of a frame's device half that does not even need the track table -- detections as arrays,
        the embedding extraction and the affinity blocks (tracker.py:786-848; the pair MLP, by far the longest launch of a tracked frame) -- for the
        frame AFTER the one that is begun, queued while that one is still waiting for its update().  The blocks are scored against a superset of
        the stored frames the pool will read: the frames of the nodes today's table selects at that frame number, plus the begun frame itself (a
        track the pending update() does not match keeps exactly that selection; one it matches selects the new node and a suffix of it).  With nothing begun this is
        the first part of begin() for the next frame.  begin() / update() given the SAME `results` object continue from here; anything else takes
        the work back.  Frames are prepared in stream order, one at a time."""
        if self._prepared is not None:
            self._undo(self._prepared)
            self._prepared = None
        snap = self._snapshot()
        ahead = self._begun
        try:
            p = self._stage_a(self.frame_id + (2 if ahead is not None else 1), results, FeatureMaps, ddd_boxes, depths_by_class, pre, feats, ahead)
        except Exception:
            self._undo({"snap": snap})
            raise
        p["snap"] = snap
        self._prepared = p

    def _snapshot(self):
        rec = self.recorder
        return (rec.all_frame_index, dict(rec.all_features), dict(rec.all_boxes), dict(rec.all_similarity), rec._dev)

    def _undo(self, b):
        """Take back what begin() / prepare() left in the recorder.  Undoing the begun frame also drops a frame prepared behind it (the older
        snapshot knows neither)."""
        snap = b.get("snap")
        if snap is not None:
            rec = self.recorder
            rec.all_frame_index, rec.all_features, rec.all_boxes, rec.all_similarity, rec._dev = snap
        w = b.get("sim_wait")
        if w is not None:
            w(raw=True)                                                # let the queued launches finish with the staging buffers they read
        if b is not self._prepared:
            self._prepared = None

    def _take_back(self):
        if self._begun is not None:
            self._undo(self._begun)
            self._begun = None
        if self._prepared is not None:
            self._undo(self._prepared)
            self._prepared = None

    def detections_as_arrays(self, results, ddd_boxes=None, depths_by_class=None):
        """The frame's detections of this tracker as arrays (tracker.py:786-820): rows, boxes in the forms the stages read, and the embedding
        centres in [-1, 1] (convert_detection, image.py:391-412) as the [1, n, 1, 1, 2] tensor forward_feature_extracter takes."""
        det_ddd = det_depth = None
        if self.ddd:
            dets = np.array(results)
            det_ddd = np.array(ddd_boxes, dtype=np.float64).reshape(-1, 7) if len(dets) else np.zeros((0, 7))
            det_depth = np.array([d[0] for d in depths_by_class], dtype=np.float64) if len(dets) else np.zeros(0)
        elif hasattr(results, "arrays") and results.arrays() is not None and "bbox" in results.arrays():
            post = results.arrays()                                   # Detector.post_process' own arrays (postprocess.ResultList): no parsing back
            bb, sc = np.asarray(post["bbox"], np.float32).reshape(-1, 4), np.asarray(post["score"], np.float32)
            if self.dataset == "kitti_tracking":
                m = np.asarray(post["class"]) == 2                     # tracker.py:790-797
                bb, sc = bb[m], sc[m]
            dets = np.concatenate([bb, sc[:, None]], 1)
        elif self.dataset == "kitti_tracking":
            dets = np.array([np.asarray(d["bbox"]).tolist() + [d["score"]] for d in results if d["class"] == 2], np.float32)     # tracker.py:790-797
        else:
            dets = np.array([np.asarray(d["bbox"]).tolist() + [d["score"]] for d in results], np.float32)
        nd0 = len(dets)
        pre = {"results": results, "nd0": nd0, "det_ddd": det_ddd, "det_depth": det_depth, "centers": None, "org": None}
        if nd0 > 0:
            tl = dets[:, :4].copy()                                   # STrack.tlbr_to_tlwh in the dtype of the rows (float32 for the 2-D datasets)
            tl[:, 2:] -= tl[:, :2]
            tlwh = tl.astype(float)
            xyah = tlwh.copy()
            xyah[:, :2] += xyah[:, 2:] / 2
            xyah[:, 2] /= xyah[:, 3]
            tlbr = tlwh.copy()
            tlbr[:, 2:] += tlbr[:, :2]
            org = np.copy(dets[:, :4])
            d = np.array(org, dtype=np.float64)                           # convert_detection, image.py:391-412
            d[:, 2] -= d[:, 0]; d[:, 3] -= d[:, 1]
            d[:, 0] /= self.img_width; d[:, 2] /= self.img_width; d[:, 1] /= self.img_height; d[:, 3] /= self.img_height
            pre.update(tlwh=tlwh, xyah=xyah, tlbr=tlbr, dscore=dets[:, 4], org=org,
                       centers=torch.from_numpy(((2 * d[:, 0:2] + d[:, 2:4]) - 1.0).astype(float)).float().view(1, -1, 1, 1, 2))
        else:
            z = np.zeros((0, 4))
            pre.update(tlwh=z, xyah=z, tlbr=z, dscore=np.zeros(0, np.float32))
        return pre

    @staticmethod
    def extract_together(trackers, pres, FeatureMaps):
        """The embeddings of several trackers' detections of ONE frame (the per-class trackers of a nuScenes frame) in one forward_feature_extracter
        call when they share the extractor: [1, n_k, D] per tracker (None for a tracker without detections, or when they do not share it)."""
        afe = trackers[0].model.AFE
        counts = [p["nd0"] for p in pres]
        if sum(counts) == 0 or any(t.model.AFE is not afe for t in trackers):
            return [None] * len(trackers)
        if FeatureMaps[0].shape[0] == 2:                                  # flip-test pair: the un-flipped frame's maps (tracker.py:821-825)
            FeatureMaps = [fm[0].unsqueeze(0) for fm in FeatureMaps]
        feats = afe.forward_feature_extracter(FeatureMaps, torch.cat([p["centers"] for p in pres if p["nd0"]], 1))
        out, o = [], 0
        for n in counts:
            out.append(feats[:, o:o + n] if n else None)
            o += n
        return out

    def _stage_a(self, fid, results, FeatureMaps, ddd_boxes, depths_by_class, pre=None, feats=None, ahead=None):
        """Detections as arrays, embeddings, affinity blocks of frame `fid` (recorded).  ahead: the begun frame when `fid` is the one after it."""
        if pre is None or pre["results"] is not results:
            pre, feats = self.detections_as_arrays(results, ddd_boxes, depths_by_class), None
        sel_all = None
        if pre["nd0"] > 0:
            if feats is None:
                if FeatureMaps[0].shape[0] == 2:                          # flip-test pair: the un-flipped frame's maps (tracker.py:821-825)
                    FeatureMaps = [fm[0].unsqueeze(0) for fm in FeatureMaps]
                feats = self.model.AFE.forward_feature_extracter(FeatureMaps, pre["centers"])
            needed = None
            if self.lazy_blocks:
                if ahead is None:
                    sel_all = self._selected_nodes(fid)
                    needed = set(np.unique(sel_all[0][sel_all[2]]).tolist())
                else:
                    # what update(ahead) can leave selected at `fid`: a track it does not match keeps its nodes (selection at `fid` on today's
                    # table -- NOT the begun frame's selection: a node ageing out can take a track from "the last mm" back to "all mm + 1");
                    # a track it matches selects the new node and a suffix of that
                    sa = self._selected_nodes(fid)
                    needed = set(np.unique(sa[0][sa[2]]).tolist()) | {int(ahead["fid"])}
            self.recorder.update(self.model, fid, feats.data, pre["org"], needed=needed)
        return {"results": results, "fid": fid, "pre": pre, "sel_all": sel_all}

    def _first_half(self, results, FeatureMaps, ddd_boxes, depths_by_class, pre=None, feats=None):
        fid = self.frame_id + 1
        c = self.cols
        a, self._prepared = self._prepared, None
        if a is not None and (a["results"] is not results or a["fid"] != fid):
            self._undo(a)                                                 # prepared for a frame that is not the one that came
            a = None
        if a is None:
            a = self._stage_a(fid, results, FeatureMaps, ddd_boxes, depths_by_class, pre, feats)
        pre = a["pre"]
        nd0, det_ddd, det_depth = pre["nd0"], pre["det_ddd"], pre["det_depth"]
        tlwh, xyah, tlbr, dscore = pre["tlwh"], pre["xyah"], pre["tlbr"], pre["dscore"]
        sel_all = a["sel_all"] if a["sel_all"] is not None else self._selected_nodes(fid)
        T0 = c.n
        # the similarity of EVERY pool row to the frame's detections, queued now and read after the host work that does not depend on it (prediction, motion
        # gate; nuScenes: the 3-D IoU stage, which only decides which of these rows the embedding stage keeps)
        sim_wait = self._similarity(fid, np.arange(T0), nd0, sel_all, defer=True) if (T0 and nd0) else None
        b = {"results": results, "fid": fid, "nd0": nd0, "sel_all": sel_all, "tlwh": tlwh, "xyah": xyah, "tlbr": tlbr, "dscore": dscore, "T0": T0,
             "sim_wait": sim_wait, "det_ddd": det_ddd, "det_depth": det_depth}
        if "snap" in a:
            b["snap"] = a["snap"]
        return b

    def update(self, results, FeatureMaps, ddd_boxes=None, depths_by_class=None, ddd_org_boxes=None, submission=None, classe=None):
        """tracker.py:723-1056.  2-D: results = the frame's post-processed detections ({"bbox" tlbr, "score", "class"}).  nuScenes: results =
        rows (x1, y1, x2, y2, score) of this tracker's class with ddd_boxes (h, w, l, x, y, z, rot_y), depths_by_class, ddd_org_boxes,
        submission (detector.py:313-338).  Returns the tracks matched or started in this frame (TrackView)."""
        self.classe = classe if classe is not None else self.classe
        b, self._begun = self._begun, None
        if b is not None and b["results"] is not results:              # another frame than the one begin() was told about: take its traces back
            self._undo(b)
            b = None
        if b is None:
            b = self._first_half(results, FeatureMaps, ddd_boxes, depths_by_class)
        fid = self.frame_id = b["fid"]
        nd0, sel_all, tlwh, xyah, tlbr, dscore, T0, sim_wait = (b[k] for k in ("nd0", "sel_all", "tlwh", "xyah", "tlbr", "dscore", "T0", "sim_wait"))
        det_ddd, det_depth = b["det_ddd"], b["det_depth"]
        c = self.cols
        if not self.use_lstm and T0:                                   # STrack.multi_predict, tracker.py:193-207 (every pool track is Tracked)
            if self.native_assoc:
                self._kf("deft_kf_predict", T0)
            else:
                c.a["mean"], c.a["cov"] = kf_multi_predict(c["mean"], c["cov"])
        if self.native_assoc:                                          # the stages of the frame in one host call
            if self.ddd:
                mt, md, lost, new_d = self._associate_ddd(fid, T0, nd0, sim_wait, det_ddd, tlbr)
            else:
                mt, md, lost, new_d = self._associate_2d(fid, T0, nd0, sim_wait, xyah, tlbr)
            removed = lost[fid - c["fid"][lost] > self.max_time_lost].tolist()
        else:
            mt, md, removed, new_d = self._associate_stages(fid, T0, nd0, sim_wait, sel_all, xyah, tlbr, det_ddd)
        new_d = new_d[dscore[new_d] >= self.det_thresh] if len(new_d) else new_d
        return self._commit(fid, T0, mt, md, removed, new_d, dscore, tlwh, xyah, det_ddd, det_depth, ddd_org_boxes, submission)

    def _associate_stages(self, fid, T0, nd0, sim_wait, sel_all, xyah, tlbr, det_ddd):
        """The association stages in numpy (tracker.py:850-1030), every configuration: the cross-check of the native calls."""
        c = self.cols
        matched_t, matched_d = [], []                                  # pool row, detection index -- in the reference's output order
        pool = np.arange(T0)
        det_left = np.arange(nd0)
        # ---- nuScenes (not pedestrian): 3-D IoU association with the recently seen tracks (tracker.py:850-884) ----
        stage0 = self.ddd and self.classe != "pedestrian"
        if stage0:
            recent = np.abs(c["fid"] - fid) < 3
            new, old = pool[recent], pool[~recent]
            cost = A.iou_ddd_distance(c["ddd"][new], det_ddd)
            m, u_t, u_d = A.linear_assignment(cost, 0.999)
            if len(m):
                matched_t += new[m[:, 0]].tolist(); matched_d += m[:, 1].tolist()
            det_left = np.asarray(u_d, dtype=int)
            pool = np.concatenate([new[np.asarray(u_t, dtype=int)], old]).astype(int)
        # ---- embedding association fused with the motion gate (tracker.py:886-925) ----
        g_pre = None
        if sim_wait is not None and not self.use_lstm and not self.ddd:   # the Kalman gate of matching.fuse_motion (:330-338) while the device works
            g_pre = A._maha2(c["mean"][pool][:, :2], c["cov"][pool][:, :2, :2], xyah[det_left][:, :2])
        sim = None
        if sim_wait is not None:
            sim = sim_wait()
            sim = sim[pool] if len(pool) and len(det_left) else None   # (nuScenes: the rows the 3-D stage left, in its order)
        self._device_done()
        dists = np.zeros((len(pool), len(det_left)), dtype=float)
        if dists.size:
            dists = 1 - sim[:, :-1][:, det_left]
        if dists.size:
            lam = 0.9
            if self.ddd:                                               # matching.fuse_motion_ddd, :374-415
                dd = det_ddd[det_left][None, :, :] - c["ddd"][pool][:, None, :]
                if self.use_lstm:
                    g = np.sqrt(np.sum(dd[..., 3:-1] * dd[..., 3:-1], axis=2))          # kalman_filter_lstm.py:92-95
                else:
                    g = np.sum(dd * dd, axis=2)                                            # kalman_filter.py:271-273
                thr = np.maximum(0.2 * c["depth"][pool], 5 if self.classe == "pedestrian" else 10)
                dists[g > thr[:, None]] = np.inf
                dists = lam * dists + 0.001 * g
            elif not self.use_lstm:                                    # matching.fuse_motion, Kalman: :330-338
                g = g_pre if g_pre is not None else A._maha2(c["mean"][pool][:, :2], c["cov"][pool][:, :2, :2], xyah[det_left][:, :2])
                dists[g > 5.0 * A.chi2inv95[2]] = np.inf
                dists = lam * dists + 0.05 * (1 - lam) * g
            else:                                                      # LSTM: :339-366
                old_enough = c["nobs"][pool] >= 300
                g = np.zeros_like(dists)
                if old_enough.any():
                    rows = pool[old_enough]
                    pm = self._prediction_at(rows, fid).astype(np.float32).astype(np.float64)[:, :2]
                    cov = c["om2"][rows] / (c["nobs"][rows] - 1)[:, None, None]
                    gm = A._maha2(pm, cov[:, :2, :2], xyah[det_left][:, :2])
                    sub = dists[old_enough]
                    sub[gm > 5.0 * A.chi2inv95[2]] = np.inf
                    dists[old_enough] = lam * sub + 0.05 * (1 - lam) * gm
                if (~old_enough).any():
                    dists[~old_enough] = lam * dists[~old_enough] + 0.0005 * (1 - lam) * g[~old_enough]
        m, u_t, u_d2 = A.linear_assignment(dists, 0.9)
        if len(m):
            matched_t += pool[m[:, 0]].tolist(); matched_d += det_left[m[:, 1]].tolist()
        u_t = np.asarray(u_t, dtype=int); u_d2 = np.asarray(u_d2, dtype=int)
        # ---- similarity-only association of the leftovers (KITTI :954-980, nuScenes :927-953) ----
        cand = pool
        left2 = det_left[u_d2]
        if self.dataset in ("kitti_tracking", "nuscenes") and len(left2) > 0:
            r_tracked = pool[u_t]
            if len(r_tracked) and sim is not None:
                d2 = 1 - sim[u_t][:, :-1][:, left2]                    # (the rows of the matrix above: same tracks, same nodes, same frame)
                m, u_t, u_d = A.linear_assignment(d2, 0.9)
                if len(m):
                    matched_t += r_tracked[m[:, 0]].tolist(); matched_d += left2[m[:, 1]].tolist()
                left2 = left2[np.asarray(u_d, dtype=int)]
                cand = r_tracked
                u_t = np.asarray(u_t, dtype=int)
        # ---- IoU association of what is left (tracker.py:982-1030) ----
        rest = cand[u_t]
        if self.dataset in ("kitti_tracking", "nuscenes"):
            rest = rest[np.abs(fid - c["fid"][rest]) < (3 if self.ddd else 6)]
        if len(rest) and len(left2):
            if self.use_lstm and not self.ddd:                          # prediction_at_frame_tlbr (tracker.py:274-280), float32 like the reference's arrays
                p = self._prediction_at(rest, fid).astype(np.float32)
                p[:, 2] *= p[:, 3]
                p[:, :2] -= p[:, 2:] / 2
                p[:, 2:] += p[:, :2]
                a_tlbr = p.astype(np.float64)
            else:
                a_tlbr = self._tlwh_rows(rest)
                a_tlbr[:, 2:] += a_tlbr[:, :2]
            cost = 1 - A.bbox_overlaps(np.ascontiguousarray(a_tlbr), np.ascontiguousarray(tlbr[left2]))
        else:
            cost = np.zeros((len(rest), len(left2)), dtype=float)
        m, u_t, u_d = A.linear_assignment(cost, 0.0 if self.ddd else 0.9)
        if len(m):
            matched_t += rest[m[:, 0]].tolist(); matched_d += left2[m[:, 1]].tolist()
        removed = [int(g) for g in rest[np.asarray(u_t, dtype=int)] if fid - c["fid"][g] > self.max_time_lost]
        new_d = left2[np.asarray(u_d, dtype=int)]
        return np.asarray(matched_t, dtype=int), np.asarray(matched_d, dtype=int), removed, new_d

    def _commit(self, fid, T0, mt, md, removed, new_d, dscore, tlwh, xyah, det_ddd, det_depth, ddd_org_boxes, submission):
        """What a frame's association changes: STrack.update of the matched tracks, STrack.activate of the new ones, the motion step, the pool."""
        c = self.cols
        # ---- state update of the matched tracks: STrack.update (tracker.py:371-400), all at once ----
        if len(mt):
            c["fid"][mt] = fid
            c["tlen"][mt] += 1
            c["score"][mt] = dscore[md]
            c["act"][mt] = True
            c.a["nf"][mt] = np.concatenate([c["nf"][mt][:, 1:], np.full((len(mt), 1), fid)], 1)
            c.a["ni"][mt] = np.concatenate([c["ni"][mt][:, 1:], md[:, None]], 1)
            c["nn"][mt] += 1
            if self.ddd:
                c["ddd"][mt] = det_ddd[md]; c["depth"][mt] = det_depth[md]
                for g, j in zip(mt.tolist(), md.tolist()):
                    c["org"][g] = ddd_org_boxes[j]; c["sub"][g] = submission[j]
            if self.use_lstm:
                c["tlwh"][mt] = tlwh[md]
                if not self.ddd:
                    self._observe(mt, xyah[md])
            elif self.native_assoc:
                self._kf("deft_kf_update", np.ascontiguousarray(mt, dtype=np.int32), np.ascontiguousarray(xyah[md], dtype=np.float64))
            else:
                c["mean"][mt], c["cov"][mt] = kf_multi_update(c["mean"][mt], c["cov"][mt], xyah[md])
        # ---- new tracks: STrack.activate (tracker.py:285-311) ----
        nnew = len(new_d)
        if nnew:
            ids = [TrackIds.next_id() for _ in range(nnew)]
            vals = {"tid": ids, "act": np.full(nnew, fid == 1), "score": dscore[new_d], "tlen": np.zeros(nnew), "fid": np.full(nnew, fid),
                    "start": np.full(nnew, fid), "nn": np.ones(nnew),
                    "nf": np.concatenate([np.zeros((nnew, self.L - 1)), np.full((nnew, 1), fid)], 1),
                    "ni": np.concatenate([np.zeros((nnew, self.L - 1)), new_d[:, None]], 1)}
            if self.use_lstm:
                vals.update({"tlwh": tlwh[new_d], "slot": [self.bank.alloc() for _ in range(nnew)]})
            else:
                z = xyah[new_d]
                hh = z[:, 3]
                one = np.ones_like(hh)
                std = np.stack([2 * _SP * hh, 2 * _SP * hh, 1e-2 * one, 2 * _SP * hh, 10 * _SV * hh, 10 * _SV * hh, 1e-5 * one, 10 * _SV * hh], 1)
                cov = np.zeros((nnew, 8, 8)); ii = np.arange(8)
                cov[:, ii, ii] = np.square(std)                        # kalman_filter.py:53-88
                vals.update({"mean": np.concatenate([z, np.zeros((nnew, 4))], 1), "cov": cov})
            if self.ddd:
                o = np.empty(nnew, object); s_ = np.empty(nnew, object)
                for k, j in enumerate(new_d.tolist()):
                    o[k] = ddd_org_boxes[j]; s_[k] = submission[j]
                vals.update({"ddd": det_ddd[new_d], "depth": det_depth[new_d], "org": o, "sub": s_})
            c.append(nnew, **vals)
            if self.use_lstm and not self.ddd:
                self._observe(np.arange(T0, T0 + nnew), xyah[new_d])
        # ---- the LSTM motion update of everything touched in this frame: ONE launch (tracker.py:408-580) ----
        touched = np.concatenate([mt, np.arange(T0, T0 + nnew)]).astype(int)
        if self.use_lstm:
            src = det_ddd if self.ddd else tlwh
            self._motion_step(touched, np.concatenate([src[md], src[new_d]]).reshape(-1, self._od), fid)
        output = self._views(touched)
        # ---- pool for the next frame: tracked minus removed, new tracks at the end (tracker.py:1032-1054) ----
        if removed:
            keep = np.ones(c.n, bool); keep[removed] = False
            self.removed_ids += c["tid"][removed].tolist()
            if self.use_lstm:
                for s_ in c["slot"][removed].tolist():
                    self.bank.free(int(s_))
                self._resolve()
                self.fut_arr = self.fut_arr[keep]
            c.keep(keep)
        return output

    def _observe(self, rows, x):
        """np.cov of a 2-D LSTM track's (x, y, a, h) observations (tracker.py:410-412), kept as running mean / scatter (Welford, all
        touched tracks at once): only read for tracks with >= 300 observations (matching.py:342)."""
        c = self.cols
        c["nobs"][rows] += 1
        n = c["nobs"][rows][:, None]
        delta = x - c["omean"][rows]
        mean = c["omean"][rows] + delta / n
        c["om2"][rows] += delta[:, :, None] * (x - mean)[:, None, :]
        c["omean"][rows] = mean


class Tracker2D(ArrayTracker):
    """The 2-D datasets (MOT17 / KITTI): `Tracker(opt, model, h, w)` of the reference for BASELINE configs[1] / [2] / [3]."""

    def __init__(self, opt, model, h=100, w=100, frame_rate=10):
        assert opt.dataset in ("mot", "kitti_tracking"), "Tracker2D: mot / kitti_tracking (nuScenes: ArrayTracker, one per class)"
        super().__init__(opt, model, h, w, frame_rate)
