"""Host side of the device pre-processing (detector.py:346-422, utils/image.py:42-72): the affine the
reference builds for a frame, the matrix cv2.warpAffine derives from it, the normalisation table -- and the feeder that keeps
uint8 frames flowing from pinned host memory on a copy stream.  The pixels are touched only by `deft_preprocess_u8`."""
import numpy as np
import torch

MEAN = np.array([0.40789654, 0.44719302, 0.47026115], dtype=np.float32)      # generic_dataset.py:67-72
STD = np.array([0.28863828, 0.27408164, 0.27809835], dtype=np.float32)


def input_affine(height, width, inp_h, inp_w):
    """`trans_input` of Detector.pre_process in fix_res mode (detector.py:363-367, 384): c = (w/2, h/2), s = max(h, w), rot 0 ->
    the 2x3 src -> dst matrix of utils/image.get_affine_transform (three point pairs; float64 solve for cv2.getAffineTransform)."""
    c = np.array([width / 2.0, height / 2.0], dtype=np.float32)
    s = np.float32(max(height, width) * 1.0)
    src = np.zeros((3, 2), np.float32); dst = np.zeros((3, 2), np.float32)
    src[0] = c
    src[1] = c + np.array([0.0, s * -0.5], np.float32)
    dst[0] = [inp_w * 0.5, inp_h * 0.5]
    dst[1] = np.array([inp_w * 0.5, inp_h * 0.5], np.float32) + np.array([0, inp_w * -0.5], np.float32)
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, dst.astype(np.float64)).T, c, s


def _affine(c, src_w, inp_h, inp_w):
    """utils/image.get_affine_transform(c, s, 0, [inp_w, inp_h]) with src_w = s[0] (image.py:42-72): three point pairs, float64 solve."""
    c = np.asarray(c, np.float32)
    src = np.zeros((3, 2), np.float32); dst = np.zeros((3, 2), np.float32)
    src[0] = c
    src[1] = c + np.array([0.0, np.float32(src_w) * -0.5], np.float32)
    dst[0] = [inp_w * 0.5, inp_h * 0.5]
    dst[1] = np.array([inp_w * 0.5, inp_h * 0.5], np.float32) + np.array([0, inp_w * -0.5], np.float32)
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def input_geometry(opt, height, width):
    """Detector._transform_scale + trans_input (detector.py:346-385) at test scale 1, for the three input modes of the reference:
      fix_short > 0   the short side becomes opt.fix_short, the long side follows the aspect ratio rounded up to a multiple of 64;
                      c = (w / 2, h / 2), s = (w, h);
      fix_res         (the default: opts.py `fix_res = not keep_res`) opt.input_h x opt.input_w, c = (w / 2, h / 2), s = max(h, w);
      keep_res        the frame's own size padded to a multiple of opt.pad + 1, c = (w // 2, h // 2), s = (inp_w, inp_h).
    -> (trans_input 2x3 float64, c float32[2], s (float32 scalar or [2], as the reference keeps it in `meta`), inp_h, inp_w)."""
    fix_short = int(getattr(opt, "fix_short", 0) or 0)
    if fix_short > 0:
        if height < width:
            inp_h, inp_w = fix_short, (int(width / height * fix_short) + 63) // 64 * 64
        else:
            inp_h, inp_w = (int(height / width * fix_short) + 63) // 64 * 64, fix_short
        c = np.array([width / 2, height / 2], dtype=np.float32)
        s = np.array([width, height], dtype=np.float32)
    elif getattr(opt, "fix_res", not getattr(opt, "keep_res", False)):
        inp_h, inp_w = int(getattr(opt, "input_h", 0)), int(getattr(opt, "input_w", 0))
        assert inp_h > 0 and inp_w > 0, "fix_res mode needs opt.input_h / opt.input_w"
        M, c, s = input_affine(height, width, inp_h, inp_w)
        return M, c, s, inp_h, inp_w
    else:
        pad = int(getattr(opt, "pad", 31))
        inp_h, inp_w = (height | pad) + 1, (width | pad) + 1
        c = np.array([width // 2, height // 2], dtype=np.float32)
        s = np.array([inp_w, inp_h], dtype=np.float32)
    return _affine(c, s[0], inp_h, inp_w), c, s, inp_h, inp_w


def invert_affine(M):
    """The dst -> src matrix cv2.warpAffine computes from a forward matrix (imgwarp.cpp: D = 1/det, b = -A^-1 t), float64 [6]."""
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = M[1, 1] * D, M[0, 0] * D
    i00, i01, i10, i11 = a11, -M[0, 1] * D, -M[1, 0] * D, a22
    return np.array([i00, i01, -i00 * M[0, 2] - i01 * M[1, 2], i10, i11, -i10 * M[0, 2] - i11 * M[1, 2]], np.float64)


def normalisation_table(mean=MEAN, std=STD):
    """((v / 255.0 - mean) / std).astype(float32) for v = 0..255 per channel -- float64 arithmetic like numpy's in detector.py:392."""
    v = np.arange(256, dtype=np.float64)[None, :] / 255.0
    return ((v - mean.astype(np.float32)[:, None]) / std.astype(np.float32)[:, None]).astype(np.float32)


class FrameFeeder:
    """Double-buffered uint8 frames: pinned host staging -> device on a copy stream, while the previous batch computes
    (replaces the reference's DataLoader(batch_size=1, num_workers=1) + cv2 pre-processing, src/test.py:106-112)."""

    def __init__(self, batch, sh, sw, device, depth=2):
        self.device = torch.device(device)
        self.host = [torch.empty(batch, sh, sw, 3, dtype=torch.uint8).pin_memory() if self.device.type == "cuda" else torch.empty(batch, sh, sw, 3, dtype=torch.uint8)
                     for _ in range(depth)]
        self.dev = [torch.empty(batch, sh, sw, 3, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.depth, self.i = depth, 0
        if self.device.type == "cuda":
            self.copy = torch.cuda.Stream(device=self.device)
            self.ready = [torch.cuda.Event() for _ in range(depth)]
            self.free = [torch.cuda.Event() for _ in range(depth)]
            for e in self.free:
                e.record(torch.cuda.current_stream(self.device))

    def push(self, frames_u8):
        """frames_u8 [batch, sh, sw, 3] uint8 (CPU): stage + start the H2D copy; returns the slot."""
        k = self.i % self.depth
        self.i += 1
        if self.device.type != "cuda":
            self.dev[k].copy_(frames_u8)
            return k
        src = frames_u8
        if not frames_u8.is_pinned():              # pageable source: stage it (a host memcpy; decoders should write into pinned memory)
            self.host[k].copy_(frames_u8)
            src = self.host[k]
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(self.free[k])
            self.dev[k].copy_(src, non_blocking=True)
            self.ready[k].record(self.copy)
        return k

    def take(self, k):
        """The device tensor of slot k, ordered after its copy on the current stream."""
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        return self.dev[k]

    def release(self, k):
        if self.device.type == "cuda":
            self.free[k].record(torch.cuda.current_stream(self.device))
