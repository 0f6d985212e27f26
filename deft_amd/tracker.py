"""Host-side mirror of the reference's `FeatureRecorder` (src/lib/utils/tracker.py:46-136) for the
kernel path: same attributes (`all_frame_index`, `all_features`, `all_boxes`, `all_similarity`),
same methods and the same values, but the affinity of the new frame against EVERY stored frame
(the reference's hottest loop in steady state: <= 49 `forward_stacker_features` calls per frame,
each with its own device->host copy, tracker.py:76-90) is ONE launch chain and ONE copy
(`AfeSeam.affinity_many`).  Drop-in: `tracker.recorder = deft_amd.tracker.FeatureRecorder(opt.dataset)`
after `Tracker.__init__` (tracker.py:651), nothing else in the reference's tracker changes.

Association bookkeeping (STrack, Tracker.update, matching) stays the reference's Python
(SURVEY.md §8(f) rank 1 is the next row)."""
import numpy as np

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

    def _m_frame(self):
        return {"kitti_tracking": 5, "nuscenes": 3}.get(self.dataset, 10)      # tracker.py:77-82

    def update(self, model, frame_index, features, boxes):
        """tracker.py:59-90.  features [1,N,D] (device tensor from forward_feature_extracter)."""
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
        if not prev:
            return
        if not hasattr(model.AFE, "affinity_many"):
            raise TypeError("deft_amd.tracker.FeatureRecorder needs deft_amd.integrate.AfeSeam as model.AFE")
        sims = model.AFE.affinity_many([self.all_features[p][0] for p in prev], features[0])     # one chain, one D2H
        m_frame = self._m_frame()
        for p, sim in zip(prev, sims):
            gap = frame_index - p
            delta = pow(decay, gap / 3.0) if gap < m_frame else pow(decay2, gap / 3.0)
            self.all_similarity[frame_index][p] = sim * delta

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
