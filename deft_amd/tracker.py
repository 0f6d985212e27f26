"""Host-side mirror of the reference's `FeatureRecorder` (src/lib/utils/tracker.py:46-136) for the
kernel path: same attributes (`all_frame_index`, `all_features`, `all_boxes`, `all_similarity`),
same methods and the same values, but the affinity of the new frame against EVERY stored frame
(the reference's hottest loop in steady state: <= 49 `forward_stacker_features` calls per frame,
each with its own device->host copy, tracker.py:76-90) is ONE launch chain and ONE copy
(`AfeSeam.affinity_many`).  Drop-in: `tracker.recorder = deft_amd.tracker.FeatureRecorder(opt.dataset)`
after `Tracker.__init__` (tracker.py:651), nothing else in the reference's tracker changes.

Also here, each replacing a per-track Python loop of the reference by one launch per frame:
  * `get_similarity`: drop-in for `Tracker.get_similarity` (tracker.py:663-688, with STrack.get_similarity
    :219-252 folded in) -- `deft_track_similarity` gathers and medians on the device, the affinity blocks
    never travel to the host;
  * `MotionBank` + `install_batched_motion`: the LSTM motion update of all tracks matched in a frame
    (STrack.update_lstm_features / _ddd, tracker.py:408-580: feature builder, KalmanFilterLSTM.predict, future
    boxes, one D2H per track) as ONE `deft_motion_step` launch and one copy.

Association bookkeeping (STrack state machine, Tracker.update, matching) stays the reference's Python
(SURVEY.md §8(f) rank 1)."""
import ctypes as C
import weakref

import numpy as np
import torch

Max_record_frame = 50      # tracker.py:23
decay = 1.0                # tracker.py:24
decay2 = 0.01              # tracker.py:25


class FeatureRecorder:
    def __init__(self, dataset, max_record_frame=Max_record_frame):
        self.max_record_frame = max_record_frame
        self.all_frame_index = np.array([], dtype=int)
        self.all_features = {}
        self.all_boxes = {}
        self.all_similarity = {}
        self.dataset = dataset
        self._dev = None           # (frame, device tensor [sum P, Q+1], block row starts, {prev frame: (block, float32 delta)})

    def _m_frame(self):
        return {"kitti_tracking": 5, "nuscenes": 3}.get(self.dataset, 10)      # tracker.py:77-82

    def update(self, model, frame_index, features, boxes, needed=None):
        """tracker.py:59-90.  features [1,N,D] (device tensor from forward_feature_extracter).
        needed: the stored frames whose blocks this frame's association will read (a caller that knows its track pool: the frames of
        the pool's selected nodes) -- the reference scores the new frame against ALL (up to 49) stored frames, but `get_similarity` only
        ever reads the blocks of frames in which a live track has one of its last few nodes; the others are skipped, result-identical."""
        if frame_index in self.all_frame_index:
            return
        if len(self.all_frame_index) == self.max_record_frame:
            del_frame = self.all_frame_index[0]
            del self.all_features[del_frame]
            del self.all_boxes[del_frame]
            del self.all_similarity[del_frame]
            self.all_frame_index = self.all_frame_index[1:]
        self.all_frame_index = np.append(self.all_frame_index, frame_index)
        self.all_features[frame_index] = features
        self.all_boxes[frame_index] = boxes
        self.all_similarity[frame_index] = {}
        prev = [int(p) for p in self.all_frame_index[:-1]]
        if needed is not None:
            prev = [p for p in prev if p in needed]
        if not prev:
            self._dev = None
            return
        if not hasattr(model.AFE, "affinity_many"):
            raise TypeError("deft_amd.tracker.FeatureRecorder needs deft_amd.integrate.AfeSeam as model.AFE")
        sims = model.AFE.affinity_many([self.all_features[p][0] for p in prev], features[0])     # one chain, one D2H
        m_frame = self._m_frame()
        index = {}
        for k, (p, sim) in enumerate(zip(prev, sims)):
            gap = frame_index - p
            delta = pow(decay, gap / 3.0) if gap < m_frame else pow(decay2, gap / 3.0)
            if sim is not None:                       # AfeSeam.host_copy = False: device-only (get_similarity below)
                self.all_similarity[frame_index][p] = sim * delta
            index[p] = (k, np.float32(delta))         # numpy multiplies the float32 block by float32(delta)
        out, starts = model.AFE.last_device
        self._dev = (frame_index, out, starts, index)

    # ---- accessors with the reference's contract (tracker.py:92-136): None for an unknown frame, an empty
    #      frame or an index past the end ----
    def _stored(self, table, frame_index):
        if frame_index not in self.all_frame_index:
            return None
        entry = table[frame_index]
        return entry if len(entry) else None

    def _item(self, table, frame_index, k):
        entry = self._stored(table, frame_index)
        return entry[k] if entry is not None and k < len(entry) else None

    def get_features(self, frame_index):
        return self._stored(self.all_features, frame_index)

    def get_boxes(self, frame_index):
        return self._stored(self.all_boxes, frame_index)

    def get_feature(self, frame_index, detection_index):
        return self._item(self.all_features, frame_index, detection_index)

    def get_box(self, frame_index, detection_index):
        return self._item(self.all_boxes, frame_index, detection_index)


# -------------------------------------------------------------------------------------------------
# a8: tracks x detections similarity of the current frame
# -------------------------------------------------------------------------------------------------
max_track_node = 50        # tracker.py:26


def select_nodes(nodes, frame_index, dataset):
    """The rows STrack.get_similarity medians over (tracker.py:221-248): nodes younger than max_track_node
    frames; all of them while there are at most mm+1, else the last mm (mm = 2 nuScenes, 4 otherwise)."""
    mm = 2 if dataset == "nuscenes" else 4
    if len(nodes) > mm + 2 and all(frame_index - n.frame_index < max_track_node for n in nodes[-(mm + 2):]):
        # the last mm + 2 nodes are all young enough: more than mm + 1 qualify, so the answer is the last mm of the qualifying ones = the
        # last mm nodes -- without walking a list that holds the whole life of the track
        return nodes[-mm:]
    sel = [n for n in nodes if frame_index - n.frame_index < max_track_node]
    return sel if len(sel) <= mm + 1 else sel[len(sel) - mm:]


def get_similarity(self, frame_index, strack_pool, num_detections, selected=None):
    """Drop-in for `Tracker.get_similarity` (tracker.py:663-688): float64 [T, num_detections+1], row t = the
    column-wise median of track t's selected node rows of the frame's (decayed) affinity blocks, zeros for a
    track without usable nodes.  `self` needs `.recorder` (the FeatureRecorder above), `.dataset`, `.model.AFE`.
    Bind with `Tracker.get_similarity = deft_amd.tracker.get_similarity`."""
    T = len(strack_pool)
    if T == 0:
        return np.array([])
    rec = self.recorder
    if rec._dev is None or rec._dev[0] != frame_index:
        if num_detections == 0 or not any(select_nodes(t.nodes, frame_index, self.dataset) for t in strack_pool):
            return np.zeros((T, num_detections + 1))
        raise KeyError("no affinity blocks recorded for frame %r" % (frame_index,))
    _, sim, starts, index = rec._dev
    assert sim.shape[1] == num_detections + 1
    L = 5
    rows = np.zeros((T, L), np.int32); scale = np.zeros((T, L), np.float32); cnt = np.zeros(T, np.int32)
    for t, trk in enumerate(strack_pool):          # selected: {id(track): its select_nodes(...)} when the caller has them already
        for i, n in enumerate(selected[id(trk)] if selected is not None else select_nodes(trk.nodes, frame_index, self.dataset)):
            blk, delta = index[n.frame_index]                      # KeyError like the reference for an unknown frame
            if not 0 <= n.id < starts[blk + 1] - starts[blk]:
                raise IndexError("node id %d outside frame %d" % (n.id, n.frame_index))
            rows[t, i] = starts[blk] + n.id; scale[t, i] = delta
            cnt[t] = i + 1
    plan = self.model.AFE.plan
    dev = sim.device
    rows_d = torch.from_numpy(rows).to(dev); scale_d = torch.from_numpy(scale).to(dev); cnt_d = torch.from_numpy(cnt).to(dev)
    out = torch.empty(T, num_detections + 1, dtype=torch.float32, device=dev)
    plan.lib.call("deft_track_similarity", C.c_void_p(sim.data_ptr()), sim.shape[0], num_detections, C.c_void_p(rows_d.data_ptr()),
                  C.c_void_p(scale_d.data_ptr()), C.c_void_p(cnt_d.data_ptr()), T, L, C.c_void_p(out.data_ptr()), plan._stream())
    return out.cpu().numpy().astype(np.float64)


# -------------------------------------------------------------------------------------------------
# a9: LSTM motion update of all tracks touched in a frame, one launch
# -------------------------------------------------------------------------------------------------
class MotionBank:
    """Device-resident motion state of every live track -- (h, c) of the LSTM and the previous observation the
    feature deltas are taken against -- addressed by slot.  `step` = STrack.update_lstm_features(_ddd) for a
    whole frame's worth of tracks: one `deft_motion_step` launch, one device->host copy."""

    def __init__(self, kf, capacity=128):
        self.plan = kf.plan if hasattr(kf, "plan") else kf          # integrate.KalmanFilterLSTM or engine.LstmPlan
        self.dim = 7 if self.plan.nin == 18 else 4
        self.fut = self.plan.nout // 4
        dev = self.plan.device
        self.h = torch.zeros(capacity, 128, dtype=torch.float32, device=dev)
        self.c = torch.zeros(capacity, 128, dtype=torch.float32, device=dev)
        self.last = torch.zeros(capacity, 9, dtype=torch.float64, device=dev)
        self._free = list(range(capacity - 1, -1, -1))
        self.pending = []                # (track, slot, box float64[dim], frame_id) queued by the STrack adapter
        self.launches = 0

    def alloc(self):
        if not self._free:
            n = self.h.shape[0]
            grow = lambda t: torch.cat([t, torch.zeros_like(t)], 0)
            self.h, self.c, self.last = grow(self.h), grow(self.c), grow(self.last)
            self._free = list(range(2 * n - 1, n - 1, -1))
        s = self._free.pop()
        self.h[s].zero_(); self.c[s].zero_(); self.last[s].zero_()
        return s

    def free(self, slot):
        self._free.append(slot)

    def step(self, slots, boxes, frame_id):
        """slots [T] ints (distinct), boxes float64 [T, dim] (tlwh, or (h,w,l,x,y,z,rot_y)), all observed at
        `frame_id` -> (features float32 [T,nin], future boxes float64 [T,fut,dim]) as numpy."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, self.dim)
        assert len(set(slots)) == len(slots) == boxes.shape[0]
        dev = self.h.device
        st = torch.tensor(list(slots), dtype=torch.int32, device=dev)
        bt = torch.from_numpy(boxes).to(dev)
        feat, pred = self.plan.motion_step(st, bt, frame_id, self.h, self.c, self.last)
        self.launches += 1
        return feat.cpu().numpy(), pred.cpu().numpy()

    def step_async(self, slots, boxes, frame_id):
        """`step` without waiting: the launch and a non-blocking copy of the future boxes into pinned memory are queued; the returned
        callable waits for that copy (an event) and hands back the float64 [T, fut, dim] array -- the array tracker calls it when the
        NEXT frame first needs a prediction, so the motion update costs the frame that issues it no host time."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, self.dim)
        assert len(set(slots)) == len(slots) == boxes.shape[0]
        dev = self.h.device
        if dev.type != "cuda":
            pred = self.step(slots, boxes, frame_id)[1]
            return lambda: pred
        T = len(slots)
        # One pinned (in, out) pair PER CALL in flight: several trackers share a bank (the seven per-class nuScenes trackers, two 2-D
        # trackers on one model) and each reads its result a frame later -- a single pair would be rewritten by the next caller before the
        # first one has copied its predictions out (and its H2D source while that copy may still be queued).  The pair goes back to the
        # pool only when its own event has completed, i.e. when both copies are done; a closure that is dropped unread just frees it.
        pool = self.__dict__.setdefault("_pin_pool", [])
        pair = None
        for i, pr in enumerate(pool):
            if pr[0].shape[0] >= T:
                pair = pool.pop(i)
                break
        if pair is None:
            cap = max(128, 2 * T)
            pair = (torch.empty(cap, 1 + self.dim, dtype=torch.float64).pin_memory(),
                    torch.empty(cap, self.fut, self.dim, dtype=torch.float64).pin_memory())
        hin, hout = pair
        hin[:T, 0] = torch.as_tensor(list(slots), dtype=torch.float64)
        hin[:T, 1:] = torch.from_numpy(boxes)
        din = hin[:T].to(dev, non_blocking=True)
        st = din[:, 0].to(torch.int32).contiguous()
        bt = din[:, 1:].contiguous()
        _, pred = self.plan.motion_step(st, bt, frame_id, self.h, self.c, self.last)
        self.launches += 1
        hout[:T].copy_(pred, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        done = []

        def wait():
            if not done:
                ev.synchronize()
                done.append(hout[:T].numpy().copy())
                if len(pool) < 16:
                    pool.append(pair)
            return done[0]
        return wait

    # ---- deferred form used by the STrack adapter ----
    def enqueue(self, track, slot, box, frame_id):
        if any(p[1] == slot for p in self.pending):                # the same track twice before a read: keep order
            self.flush()
        self.pending.append((track, slot, np.array(box, dtype=np.float64), frame_id))

    def flush(self):
        todo, self.pending = self.pending, []
        for fid in sorted(set(p[3] for p in todo)):
            grp = [p for p in todo if p[3] == fid]
            _, pred = self.step([p[1] for p in grp], np.stack([p[2] for p in grp]), fid)
            for (trk, _, _, _), pr in zip(grp, pred):
                if self.dim == 4:
                    trk._future = {1 + i: pr[i].astype(np.float32) for i in range(self.fut)}     # exact: float32 values widened
                else:
                    trk._future = {1 + i: pr[i].copy() for i in range(self.fut)}


def install_batched_motion(STrack, bank):
    """Bind the bank to the reference's `STrack` class (utils/tracker.py:142): the two feature builders enqueue
    instead of running a batch-1 LSTM + copy per track, `future_predictions` becomes a property whose first read
    flushes the queue (Tracker.update reads predictions only in the NEXT association or in
    remove_duplicate_stracks, after all of the frame's updates).  Host bookkeeping of the builders
    (observation lists, np.cov) is kept as in the reference.  Returns a function that undoes the binding."""
    saved = {k: STrack.__dict__.get(k) for k in ("update_lstm_features", "update_lstm_features_ddd", "future_predictions")}

    def slot_of(trk):
        s = trk.__dict__.get("_motion_slot")
        if s is None:
            s = trk.__dict__["_motion_slot"] = bank.alloc()
            weakref.finalize(trk, bank.free, s)
        return s

    def update_lstm_features(self, tlwh):                                      # tracker.py:408-412, then deferred
        self.observations_tlwh.append(tlwh.copy())
        self.observations.append(self.tlwh_to_xyah(tlwh).tolist())
        self.covariance = np.cov(np.asarray(self.observations).copy().T)
        bank.enqueue(self, slot_of(self), tlwh, self.frame_id)

    def update_lstm_features_ddd(self, ddd_box):                               # tracker.py:482-485, then deferred
        self.observations_ddd_bboxes.append(ddd_box.copy())
        self.covariance = np.cov(np.asarray(self.observations_ddd_bboxes).copy().T)
        bank.enqueue(self, slot_of(self), ddd_box, self.frame_id)

    def get_future(self):
        if bank.pending:
            bank.flush()
        return self.__dict__.get("_future", {})

    def set_future(self, value):
        self.__dict__["_future"] = value

    STrack.update_lstm_features = update_lstm_features
    STrack.update_lstm_features_ddd = update_lstm_features_ddd
    STrack.future_predictions = property(get_future, set_future)

    def undo():
        for k, v in saved.items():
            if v is None:
                delattr(STrack, k)
            else:
                setattr(STrack, k, v)
    return undo


def accelerate(tracker_module, kf=None):
    """One call that binds every per-frame form in this package to the reference's `utils.tracker` module
    (pass the imported module): FeatureRecorder mirror, device-side `Tracker.get_similarity`, vectorised
    `matching.fuse_motion(_ddd)` / `linear_assignment` / IoU, and -- when a motion model `kf`
    (deft_amd.integrate.KalmanFilterLSTM) is given, i.e. `opt.lstm` -- the batched motion update.  The track
    state machine (`Tracker.update`, `STrack`) stays the reference's.  Returns undo()."""
    from . import association
    RT = tracker_module
    saved = (RT.Tracker.get_similarity, RT.FeatureRecorder)
    RT.Tracker.get_similarity = get_similarity
    RT.FeatureRecorder = FeatureRecorder
    undos = [association.bind(RT.matching)]
    if kf is not None:
        undos.append(install_batched_motion(RT.STrack, MotionBank(kf)))
        ref_kf = RT.KalmanFilterLSTM                    # tracker.py:144, 301, 661 build it by this name (gating_distance)
        RT.KalmanFilterLSTM = type(kf)
        undos.append(lambda: setattr(RT, "KalmanFilterLSTM", ref_kf))

    def undo():
        for u in undos:
            u()
        RT.Tracker.get_similarity, RT.FeatureRecorder = saved
    return undo
