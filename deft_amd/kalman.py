"""The DeepSORT Kalman filter batched over the track pool (utils/tracking_utils/kalman_filter.py:24-275: initiate / multi_predict / update
as array operations instead of one scipy Cholesky per track) and the process-wide track id counter (basetrack.py:18, 40-42) that every
tracker of the process shares.  Checked against the reference's filter in tests/test_mot_tracker.py."""
import numpy as np


NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3          # basetrack.py:11-15


class TrackIds:
    """basetrack.py:18, 40-42."""
    count = 0

    @classmethod
    def next_id(cls):
        cls.count += 1
        return cls.count


class Node:
    """tracker.py:28-43: which detection of which frame."""
    __slots__ = ("frame_index", "id")

    def __init__(self, frame_index, id):
        self.frame_index, self.id = frame_index, id


# ---------------------------------------------------------------------------------------------------------------------
# DeepSORT Kalman filter, batched (utils/tracking_utils/kalman_filter.py:24-275)
# ---------------------------------------------------------------------------------------------------------------------
_F = np.eye(8)
for _i in range(4):
    _F[_i, 4 + _i] = 1.0                            # kalman_filter.py:41-44 (dt = 1)
_H = np.eye(4, 8)
_SP, _SV = 1.0 / 20, 1.0 / 160                      # kalman_filter.py:50-51


def kf_initiate(xyah):
    """kalman_filter.py:53-88."""
    mean = np.r_[xyah, np.zeros(4)]
    h = xyah[3]
    std = [2 * _SP * h, 2 * _SP * h, 1e-2, 2 * _SP * h, 10 * _SV * h, 10 * _SV * h, 1e-5, 10 * _SV * h]
    return mean, np.diag(np.square(std))


def kf_multi_predict(mean, cov):
    """kalman_filter.py:165-205: mean [T, 8], cov [T, 8, 8]."""
    h = mean[:, 3]
    one = np.ones_like(h)
    sqr = np.square(np.stack([_SP * h, _SP * h, 1e-2 * one, _SP * h, _SV * h, _SV * h, 1e-5 * one, _SV * h], 1))
    mean = np.dot(mean, _F.T)
    # F P F^T with F = [[I, I], [0, I]] (dt = 1) written out in 4 x 4 blocks: sums only (np.dot(F, P) adds exact products by 1 and 0;
    # the blocks below add the same terms in the same left-to-right order)
    cov = np.array(cov, dtype=np.float64, copy=True)
    cov[:, :4, :] += cov[:, 4:, :]                  # F P
    cov[:, :, :4] += cov[:, :, 4:]                  # (F P) F^T
    idx = np.arange(8)
    cov[:, idx, idx] += sqr
    return mean, cov


def kf_multi_update(mean, cov, meas):
    """kalman_filter.py:207-240 for T tracks at once: mean [T, 8], cov [T, 8, 8], meas [T, 4] (x, y, a, h)."""
    h = mean[:, 3]
    std = np.stack([_SP * h, _SP * h, 1e-1 * np.ones_like(h), _SP * h], 1)
    pm = mean[:, :4]                                                      # H . mean
    pc = cov[:, :4, :4].copy()                                            # H P H^T
    idx = np.arange(4)
    pc[:, idx, idx] += np.square(std)
    # K = P H^T S^-1  (the reference solves with a Cholesky factor of S; S is SPD, the solve below gives the same K to round-off)
    gain = np.linalg.solve(pc, cov[:, :4, :]).transpose(0, 2, 1)          # [T, 8, 4]
    innov = meas - pm
    new_mean = mean + np.matmul(gain, innov[:, :, None])[:, :, 0]
    # K S K^T as two batched products (the reference's multi_dot((K, S, K^T)) in the same association order)
    new_cov = cov - np.matmul(np.matmul(gain, pc), gain.transpose(0, 2, 1))
    return new_mean, new_cov


def tlbr_to_tlwh(tlbr):
    """STrack.tlbr_to_tlwh (tracker.py:600-604): in the dtype of its argument -- the tracker hands float32 rows, so the width / height
    are float32 differences (widened afterwards by STrack.__init__)."""
    r = np.asarray(tlbr).copy()
    r[2:] -= r[:2]
    return r


def tlwh_to_xyah(tlwh):
    r = np.asarray(tlwh, dtype=float).copy()
    r[:2] += r[2:] / 2
    r[2] /= r[3]
    return r


