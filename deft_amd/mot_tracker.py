"""The 2-D tracking loop of DEFT (MOT17 / KITTI) on the accelerated path, frame in -> tracks out, without the reference tree:
`Tracker.update` + `STrack` + the DeepSORT Kalman filter (utils/tracker.py:626-1056, 140-420; utils/tracking_utils/kalman_filter.py)
re-expressed over this package's device forms --

  * detections -> box centres -> embeddings: `model.AFE.forward_feature_extracter` (AfeSeam; tracker.py:807-826);
  * affinity against every stored frame: `deft_amd.tracker.FeatureRecorder` (one launch chain per frame, tracker.py:59-90);
  * tracks x detections similarity: `deft_amd.tracker.get_similarity` (device-side decay + median, tracker.py:219-252, 663-688);
  * motion gating / assignment / IoU: `deft_amd.association` (matching.py:40-104, 311-371), all tracks at once;
  * the Kalman filter batched over the track pool (multi_predict / update as array operations instead of one scipy Cholesky per track).

SURVEY.md 8(f) rank 1: "once kernels are fast, the Python/numpy loops and lap dominate" -- this is the association step as arrays.
Semantics kept exactly as the reference runs them (checked frame by frame against the reference's own Tracker in
tests/test_mot_tracker.py, ids and boxes):
  * an unmatched track is never marked Lost: it stays in `tracked_stracks` (state Tracked) until it has not been seen for more than
    `max_time_lost` frames, then it is removed (tracker.py:1006-1010; `lost_stracks` stays empty);
  * `is_activated` only becomes true on frame 1 or at the first match (tracker.py:209-217, 263-277); every matched track and every
    new detection is returned (tracker.py:1012-1018), activated or not;
  * KITTI: a second, similarity-only association for the detections the first one left over, and tracks are kept as IoU
    candidates while seen within the last 6 frames (tracker.py:956-995); MOT: tracks in state Tracked;
  * track ids come from one process-wide counter that is never reset between videos (basetrack.py:18, 40-42: `BaseTrack._count`):
    `TrackIds` below, shared by every tracker of the process.
Scope (round 4): every configuration -- mot / kitti_tracking / nuscenes, Kalman or `--lstm` -- as `deft_amd.array_tracker.ArrayTracker`
(`Tracker2D` = its 2-D form); this module holds the batched Kalman filter.
"""
from .kalman import (NEW, TRACKED, LOST, REMOVED, TrackIds, Node, kf_initiate, kf_multi_predict, kf_multi_update,  # noqa: F401
                     tlbr_to_tlwh, tlwh_to_xyah, _F, _H, _SP, _SV)

# The tracking loop itself lives in deft_amd/array_tracker.py (track state as arrays; 2-D and 3-D, Kalman and LSTM); this module keeps the
# batched Kalman filter and the id counter it shares with every tracker of the process, and re-exports the tracker classes under their
# round-3 names.
from .array_tracker import ArrayTracker, Tracker2D, TrackView  # noqa: F401
