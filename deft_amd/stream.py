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
        self.blk = torch.zeros(batch, (max_record - 1) * kmax, kmax + 1, dtype=torch.float32, device=self.device)
        self.all_blk = torch.zeros(self.world * batch, (max_record - 1) * kmax, kmax + 1, dtype=torch.float32, device=self.device)
        self.history = []                                          # replicated: [(global frame index, emb [n, D])] of the recorded frames
        self.frame0 = 0                                            # global index of the step's first frame
        if tracker is not None:
            self.replay = ReplayAFE(getattr(afe, "plan", None), getattr(afe, "host_copy", True))
            tracker.model.AFE = self.replay
        self.bytes_gathered = 0

    # -- (A) frame-local: detection rows + embeddings of the detections the tracker will build ------------------------
    def _local_record(self, b, frame):
        results, fmaps = self.detect(frame)
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
        return inp

    def step(self, frames):
        """frames: this rank's `batch` frames ([1,3,H,W] each); the step covers global frames frame0 + rank*batch + b.
        Returns on the association rank [(global frame index, Tracker.update's return value)], elsewhere []."""
        assert len(frames) == self.batch
        for b, f in enumerate(frames):
            self._local_record(b, f)
        all_rec = self._gather(self.all_rec, self.rec)                                # collective 1: records
        nfr = all_rec.shape[0]
        n_res = [int(all_rec[g, self.kmax, 0]) for g in range(nfr)]
        n_sel = [int(all_rec[g, self.kmax, 1]) for g in range(nfr)]
        # -- (B) affinity of MY frames against the replicated history, in stream order (earlier frames of this step count) --
        self.blk.zero_()
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
        all_blk = self._gather(self.all_blk, self.blk)                                # collective 2: affinity blocks
        # -- (C) association rank: the reference's Tracker, frame by frame ------------------------------------------------
        out = []
        if self.tracker is not None and self.rank == 0:
            for g in range(nfr):
                results = []
                for i in range(n_res[g]):
                    row = all_rec[g, i]
                    results.append({"bbox": row[0:4].cpu().numpy().astype(np.float32), "score": float(row[4]), "class": int(row[5])})
                if n_sel[g] > 0:
                    st = starts_of[g]
                    self.replay.load(self._frame_emb(all_rec, g, n_sel[g]).unsqueeze(0), all_blk[g, :st[-1], :n_sel[g] + 1].contiguous(), st)
                targets = self.tracker.update(results, [torch.zeros(1, 1, 1, 1)])
                out.append((self.frame0 + g, self.snapshot(targets) if self.snapshot is not None else targets))
        self.frame0 += nfr
        return out

    def _frame_emb(self, all_rec, g, n):
        return all_rec[g, :n, 6:].contiguous()
