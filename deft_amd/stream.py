"""One video stream sharded over the GPUs of a node AT THE LEVEL THE REFERENCE'S TRACKER CONSUMES (SURVEY.md §8(e),
BASELINE configs[2]: 8 consecutive frames per step, one per GPU).

What is parallel and what is not (verified by reading every consumer):
  * detection, decode, post-process and the embedding extraction of a frame need nothing from other frames
    (`Detector.process` keeps no cross-frame state; tracker.py:776 / 826 read the frame's own FeatureMaps);
  * the affinity of frame t against the stored frames depends only on EMBEDDINGS (tracker.py:59-90 reads
    `all_features`), never on association decisions;
  * only `Tracker.update`'s association + track bookkeeping (tracker.py:836-1056) is sequential per video -- cheap host work.

So every step is:  (A) each rank runs detection + embedding extraction for its own frame(s);  (1) ONE all-gather of the
fixed-size per-frame records (detection rows + embeddings + counts);  (B) each rank scores ITS frames against the
replicated embedding history (the last <= 49 recorded frames, including earlier frames of the same step -- exactly the
set `FeatureRecorder.update` loops over);  (2) one all-gather of the affinity blocks;  (C) the association rank (0) replays
the frames in stream order through the reference's own `Tracker.update`, with `model.AFE` answered from the gathered
embeddings / blocks (`ReplayAFE`).  Same kernels on the same inputs in the same order as the single-process run, so the
tracks are identical (tests/test_stream_dist.py).  The collectives are torch.distributed: `nccl` (= RCCL over xGMI) on
GPUs, `gloo` in the CPU tests.  Message sizes are small (K x (6 + D) floats per frame, <= 2 MB of blocks per frame):
latency-bound, which is why both are single all-gathers of fixed-size buffers.
"""
import numpy as np
import torch
import torch.distributed as dist

MAX_RECORD = 50            # tracker.py:23 Max_record_frame


class DeviceFrame:
    """What a device-side `detect` hands to ShardedStream: everything of one frame's record, already on the device.
    rows [K, 6] = (bbox tlbr in image pixels, score, class) in score order, zero beyond n_res; emb [K, D]: the embeddings of the
    SELECTED rows (the detections `Tracker.update` builds, tracker.py:790-805) first, in their order; n_res / n_sel: 0-dim device
    tensors (float32) -- the host learns them from the one device->host copy per step, after the exchange."""

    def __init__(self, rows, emb, n_res, n_sel):
        self.rows, self.emb, self.n_res, self.n_sel = rows, emb, n_res, n_sel


class DeviceDetect:
    """The frame-local front half of `Detector.run` (detector.py:112-199, 553-583) + the embedding extraction the tracker asks for
    first (tracker.py:807-826), with NO host round trip: the fused launch list of engine.DlaSegPlan (one hipGraph replay per frame
    after the first), the output-grid -> image affine of `generic_post_process` (post_process.py:29-60), the `out_thresh` cut (or
    a fixed number of detections: random weights make a threshold meaningless), KITTI's class-2 filter (tracker.py:793-797),
    `convert_detection` (image.py:391-412, float64 like the reference's numpy) and engine.AfePlan.extract -- device tensors in,
    device tensors out."""

    def __init__(self, sd, H, W, dataset="mot", K=100, device="cuda", lib=None, img_h=None, img_w=None, out_thresh=0.0, first_n=None,
                 afe_plan=None, max_object=100, hip_graphs=True, center=None, scale=None):
        from . import engine, postprocess as PP
        self.device = torch.device(device)
        self.plan = engine.DlaSegPlan(sd, 1, H, W, dataset, K=K, device=device, lib=lib)
        self.afe = afe_plan if afe_plan is not None else engine.AfePlan(sd, max_object, device, lib)
        self.K, self.dataset = K, dataset
        self.img_h, self.img_w = float(img_h if img_h is not None else H), float(img_w if img_w is not None else W)
        c = np.array([W / 2.0, H / 2.0], np.float32) if center is None else center        # detector.py:364-367 (fix_res, frame = input size)
        sc = float(max(H, W)) if scale is None else scale
        tr = PP.inverse_affine(c, sc, W // 4, H // 4)                                       # [2, 3] float32
        self.A = torch.from_numpy(np.ascontiguousarray(tr[:, :2].T)).to(self.device)       # points @ A + t
        self.t = torch.from_numpy(np.ascontiguousarray(tr[:, 2])).to(self.device)
        self.thr = float(out_thresh)
        self.first_n = K if first_n is None else int(first_n)
        self.idx = torch.arange(K, device=self.device)
        self.graph = None
        self.hip_graphs = bool(hip_graphs) and self.device.type == "cuda"
        self._warm = False

    def _forward(self, frame):
        p = self.plan
        if not self.hip_graphs:
            p.forward(frame); return
        if not self._warm:
            p.forward(frame); self._warm = True; return          # first frame eager: kernel attributes, caches
        if self.graph is None:
            self.graph = p.capture_graph()
        p.image.copy_(frame, non_blocking=True)
        type(p).replay_graph(self.graph, p.device)          # (engine._Plan.replay_graph: never on the NULL stream)

    def __call__(self, frame):
        p, K = self.plan, self.K
        self._forward(frame.to(self.device, non_blocking=True))
        scores = p.scores[0]
        cls = p.clses[0].to(torch.float32) + 1.0
        bb = (p.bboxes[0].reshape(2 * K, 2) @ self.A + self.t).reshape(K, 4)               # post_process.py:52-55
        valid = (scores > self.thr) & (self.idx < self.first_n)                            # scores are sorted: a prefix
        sel = valid & (cls == 2.0) if self.dataset == "kitti_tracking" else valid
        vf = valid.to(torch.float32).unsqueeze(1)
        rows = torch.cat([bb, scores.unsqueeze(1), cls.unsqueeze(1)], 1) * vf
        order = torch.argsort((~sel).to(torch.int32), stable=True)                         # selected rows first, in score order
        d = bb[order].to(torch.float64)                                                    # convert_detection, image.py:391-412
        wv, hv = d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]
        cx = 2.0 * (d[:, 0] / self.img_w) + wv / self.img_w - 1.0
        cy = 2.0 * (d[:, 1] / self.img_h) + hv / self.img_h - 1.0
        centers = torch.stack([cx, cy], 1).to(torch.float32).unsqueeze(0)                  # [1, K, 2]
        emb = self.afe.extract(p.fmaps, centers)[0]
        return DeviceFrame(rows, emb, valid.sum().to(torch.float32), sel.sum().to(torch.float32))


def convert_detection(boxes, h, w):
    """image.py:391-412: tlbr boxes in image px -> box centres in [-1, 1] as [1,N,1,1,2] (float32), without the
    reference's unconditional `.cuda()`."""
    d = np.array(boxes, dtype=np.float64).copy()
    d[:, 2] -= d[:, 0]; d[:, 3] -= d[:, 1]
    d[:, 0] /= w; d[:, 2] /= w; d[:, 1] /= h; d[:, 3] /= h
    c = (2 * d[:, 0:2] + d[:, 2:4]) - 1.0
    return torch.from_numpy(c.astype(float)).float().view(1, -1, 1, 1, 2)


def select_2d(results, dataset):
    """The rows `Tracker.update` turns into detections (tracker.py:790-805): [bbox(4), score] float32; KITTI keeps class 2."""
    if dataset == "kitti_tracking":
        rows = [det["bbox"].tolist() + [det["score"]] for det in results if det["class"] == 2]
    else:
        rows = [det["bbox"].tolist() + [det["score"]] for det in results]
    return np.array(rows, np.float32).reshape(-1, 5)


class ReplayAFE:
    """`model.AFE` of the association rank: answers the tracker's two calls from the gathered records.  Carries `.plan`
    (an engine.AfePlan on this rank's device) because deft_amd.tracker.get_similarity launches its median kernel through it."""

    def __init__(self, plan=None, host_copy=True):
        self.plan, self.host_copy, self.last_device = plan, host_copy, None
        self._emb = self._block = self._starts = None
        self._calls = 0

    def load(self, emb, block, starts):
        self._emb, self._block, self._starts, self._calls = emb, block, starts, 0

    def forward_feature_extracter(self, FeatureMaps, centers):
        assert self._emb is not None and centers.shape[1] == self._emb.shape[1], "record / tracker detection count mismatch"
        return self._emb

    def affinity_many(self, hist, cur):
        assert len(hist) == len(self._starts) - 1 and self._block.shape[0] == self._starts[-1]
        self.last_device = (self._block, self._starts)
        if not self.host_copy:
            return [None] * len(hist)
        y = self._block.detach().cpu().numpy()
        return [y[self._starts[f]:self._starts[f + 1]] for f in range(len(hist))]

    def forward_stacker_features(self, xp, xn, fill_up_column=True):
        """The reference's own FeatureRecorder asks pair by pair, oldest stored frame first (tracker.py:76-90)."""
        k = self._calls
        self._calls += 1
        y = self._block[self._starts[k]:self._starts[k + 1]].detach().cpu().numpy()
        P, Q = xp.shape[1], xn.shape[1]
        assert y.shape == (P, Q + 1)
        if fill_up_column and P > 1:
            y = np.concatenate([y, np.repeat(y[:, Q:Q + 1], P - 1, axis=1)], axis=1)
        return y


class ShardedStream:
    """detect(frame [1,3,H,W]) -> (results, FeatureMaps): the frame-local front half of `Detector.run` (model, decode,
    post-process, score threshold) -- injected, so the exchange logic is testable with a stand-in.
    afe: this rank's `model.AFE` (deft_amd.integrate.AfeSeam): forward_feature_extracter + affinity_many.
    tracker: on rank 0 the (reference) `Tracker`; its `model.AFE` is replaced by a ReplayAFE.  img_h / img_w: what
    `Tracker` normalises centres with (tracker.py:817-820: the ORIGINAL image size set by reset_tracking)."""

    def __init__(self, detect, afe, D, tracker=None, dataset="mot", kmax=100, img_h=100, img_w=100, batch=1, device="cpu",
                 group=None, max_record=MAX_RECORD, force_collective=False, snapshot=None):
        """snapshot: applied to `Tracker.update`'s return value right away (the STrack objects it returns are mutated by the
        next frame's update, and one step replays several frames)."""
        assert dataset in ("mot", "kitti_tracking"), "ShardedStream drives the 2-D trackers (Tracker.update(results, FeatureMaps)); nuScenes runs one camera per GPU as replicas (FramePipeline(exchange=False))"
        self.snapshot = snapshot
        self.detect, self.afe, self.tracker, self.dataset = detect, afe, tracker, dataset
        self.kmax, self.D, self.batch, self.max_record = kmax, D, batch, max_record
        self.img_h, self.img_w = img_h, img_w
        self.device = torch.device(device)
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.collective = inited and (self.world > 1 or force_collective)
        self.RW = 6 + D                                            # bbox(4), score, class, embedding
        self.rec = torch.zeros(batch, kmax + 1, self.RW, dtype=torch.float32, device=self.device)
        self.all_rec = torch.zeros(self.world * batch, kmax + 1, self.RW, dtype=torch.float32, device=self.device)
        # affinity blocks: [frames, rows, kmax + 1] views of flat buffers, `rows` = the live history of the step (not 49 x kmax every step)
        self._blk_flat = torch.zeros(batch * (max_record - 1) * kmax * (kmax + 1), dtype=torch.float32, device=self.device)
        self._all_blk_flat = torch.zeros(self.world * batch * (max_record - 1) * kmax * (kmax + 1), dtype=torch.float32, device=self.device)
        self.blk = self.all_blk = None
        self.meta_host = torch.zeros(self.world * batch, kmax + 1, 6, dtype=torch.float32)
        if self.device.type == "cuda":
            self.meta_host = self.meta_host.pin_memory()
        self.timers = None                                         # set to {} to collect per-phase seconds (run_stream.py --bench)
        self.history = []                                          # replicated: [(global frame index, emb [n, D])] of the recorded frames
        self.frame0 = 0                                            # global index of the step's first frame
        if tracker is not None:
            self.replay = ReplayAFE(getattr(afe, "plan", None), getattr(afe, "host_copy", True))
            tracker.model.AFE = self.replay
            if hasattr(tracker, "lazy_blocks"):
                tracker.lazy_blocks = False          # (array_tracker.Tracker2D) the gathered blocks cover EVERY stored frame: nothing to skip
        self.bytes_gathered = 0

    # -- (A) frame-local: detection rows + embeddings of the detections the tracker will build ------------------------
    def _local_record(self, b, frame):
        got = self.detect(frame)
        if isinstance(got, DeviceFrame):              # device-side detect: three slice copies, nothing touches the host
            r = self.rec[b]
            K = got.rows.shape[0]
            assert K <= self.kmax
            r[:K, 0:6] = got.rows
            r[:K, 6:] = got.emb
            r[self.kmax, 0] = got.n_res; r[self.kmax, 1] = got.n_sel
            return
        results, fmaps = got
        assert len(results) <= self.kmax
        r = self.rec[b]
        r.zero_()
        sel = select_2d(results, self.dataset)
        for i, det in enumerate(results):
            r[i, 0:4] = torch.as_tensor(np.asarray(det["bbox"], np.float32))
            r[i, 4] = float(det["score"]); r[i, 5] = float(det["class"])
        r[self.kmax, 0] = len(results); r[self.kmax, 1] = sel.shape[0]
        if sel.shape[0] > 0:
            centers = convert_detection(np.copy(sel[:, :4]), self.img_h, self.img_w)
            emb = self.afe.forward_feature_extracter(fmaps, centers)                  # [1, n, D]
            r[:sel.shape[0], 6:] = emb[0].to(self.device)

    def _gather(self, out, inp):
        if self.collective:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)
            self.bytes_gathered += out.numel() * 4
            return out
        if self.world == 1 and out.shape == inp.shape and out.data_ptr() != inp.data_ptr():
            out.copy_(inp)                            # (one process: keep `all_*` what the collective would have produced)
            return out
        return inp

    def _tick(self, name, t0):
        if self.timers is not None:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            import time
            t1 = time.perf_counter()
            self.timers[name] = self.timers.get(name, 0.0) + (t1 - t0)
            return t1
        return t0

    def step(self, frames):
        """frames: this rank's `batch` frames ([1,3,H,W] each); the step covers global frames frame0 + rank*batch + b.
        Returns on the association rank [(global frame index, Tracker.update's return value)], elsewhere []."""
        import time
        assert len(frames) == self.batch
        t = time.perf_counter() if self.timers is not None else 0.0
        for b, f in enumerate(frames):
            self._local_record(b, f)
        t = self._tick("detect", t)
        all_rec = self._gather(self.all_rec, self.rec)                                # collective 1: records
        nfr = all_rec.shape[0]
        # ONE device -> host copy per step: the detection rows + counts of every frame of the step (the embeddings stay on the device)
        meta = self.meta_host[:nfr]
        meta.copy_(all_rec[:, :, :6], non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        meta_np = meta.numpy()
        n_res = [int(meta_np[g, self.kmax, 0]) for g in range(nfr)]
        n_sel = [int(meta_np[g, self.kmax, 1]) for g in range(nfr)]
        t = self._tick("gather1", t)
        # -- (B) affinity of MY frames against the replicated history, in stream order (earlier frames of this step count) --
        hist_len = [e.shape[0] for _, e in self.history]
        rows = 0                                                                      # the step's largest history, identical on every rank
        for g in range(nfr):
            if n_sel[g] == 0:
                continue
            rows = max(rows, sum(hist_len[-(self.max_record - 1):]))
            hist_len.append(n_sel[g])
        rows = max(rows, 1)
        kw = self.kmax + 1
        self.blk = self._blk_flat[: self.batch * rows * kw].view(self.batch, rows, kw)
        self.all_blk = self._all_blk_flat[: self.world * self.batch * rows * kw].view(self.world * self.batch, rows, kw)
        starts_of = {}
        for g in range(nfr):
            if n_sel[g] == 0:
                continue                                                              # tracker.py:829: an empty frame is not recorded
            emb = self._frame_emb(all_rec, g, n_sel[g])
            prev = self.history[-(self.max_record - 1):]
            starts = [0]
            for _, e in prev:
                starts.append(starts[-1] + e.shape[0])
            starts_of[g] = starts
            if prev and g // self.batch == self.rank:
                self.afe.affinity_many([e for _, e in prev], emb)
                out, st = self.afe.last_device
                assert list(st) == starts
                self.blk[g - self.rank * self.batch, :out.shape[0], :out.shape[1]] = out.to(self.device)
            self.history.append((self.frame0 + g, emb))
            del self.history[:-self.max_record]
        t = self._tick("affinity", t)
        all_blk = self._gather(self.all_blk, self.blk)                                # collective 2: affinity blocks (live history only)
        t = self._tick("gather2", t)
        # -- (C) association rank: the reference's Tracker, frame by frame ------------------------------------------------
        out = []
        if self.tracker is not None and self.rank == 0:
            for g in range(nfr):
                rws = meta_np[g]
                results = [{"bbox": rws[i, 0:4].copy(), "score": float(rws[i, 4]), "class": int(rws[i, 5])} for i in range(n_res[g])]
                if n_sel[g] > 0:
                    st = starts_of[g]
                    self.replay.load(self._frame_emb(all_rec, g, n_sel[g]).unsqueeze(0), all_blk[g, :st[-1], :n_sel[g] + 1].contiguous(), st)
                targets = self.tracker.update(results, [torch.zeros(1, 1, 1, 1)])
                out.append((self.frame0 + g, self.snapshot(targets) if self.snapshot is not None else targets))
        self._tick("tracker", t)
        self.frame0 += nfr
        return out

    def _frame_emb(self, all_rec, g, n):
        return all_rec[g, :n, 6:].contiguous()
