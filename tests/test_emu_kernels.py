"""CPU tests (-m "not gpu"): the UNMODIFIED HIP kernels, compiled against the SIMT
emulator in tests/hipemu, checked through the C ABI against the oracle.  These
validate indexing / fragment layouts / host packing before GPU time is spent; the
real-hardware parity tests are in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import parity_checks as pc
from parity_checks import T


@pytest.mark.parametrize("args", [
    (1, 8, 12, 16, 16, 3, 1, 1, T(128, 32)),
    (2, 9, 7, 3, 16, 7, 1, 3, T(128, 32)),          # image layer: Cin 3 padded to 4, 7x7
    (1, 10, 12, 32, 64, 3, 2, 1, T(64, 64)),
    (1, 10, 12, 64, 64, 3, 1, 1, T(128, 64)),
    (1, 6, 10, 448, 128, 1, 1, 0, T(128, 128)),     # Root over a 448-ch concat
    (1, 6, 10, 64, 200, 1, 1, 0, T(64, 128)),       # Cout not a tile multiple
    (1, 16, 16, 16, 40, 3, 1, 1, 0),                # auto tile
    (3, 4, 4, 128, 27, 3, 1, 1, 0),                 # offset/mask conv shape, M < BM
    (1, 10, 12, 64, 64, 3, 1, 1, T(128, 64) | (1 << 29)),    # 2-stage (1-barrier, double LDS) loop variant
    (1, 5, 9, 32, 48, 1, 1, 0, T(64, 64)),                   # single-chunk K (nk == 1)
    (1, 5, 9, 64, 48, 1, 1, 0, T(64, 64)),
    (2, 7, 9, 64, 27, 3, 1, 1, T(64, 32)),                   # intra-workgroup split-K, 2 waves per output tile
    (2, 7, 9, 64, 27, 3, 1, 1, T(32, 32)),                   # 4 waves per output tile
    (1, 6, 5, 16, 20, 3, 1, 1, T(32, 32) | (1 << 29)),       # split-K + 2-stage loop + Cin < 32
    (1, 11, 13, 128, 27, 3, 1, 1, 0),                        # auto tile -> split-K (few rows, Cout <= 32)                   # nk == 2
])
def test_conv(emu_lib, args):
    pc.check_conv(emu_lib, "cpu", *args, res=(args[4] % 3 == 1), relu=(args[3] != 448))


@pytest.mark.parametrize("Ci,k", [(3, 7), (16, 3), (8, 5)])
def test_conv_pixel_pair(emu_lib, Ci, k):
    pc.check_conv_pair(emu_lib, "cpu", Ci, k)


@pytest.mark.parametrize("Ci,k,kw", [(3, 7, {}), (16, 3, {}), (16, 3, {"N": 1, "H": 8, "W": 32, "Co": 12, "relu": False}), (3, 7, {"N": 1, "H": 19, "W": 70, "wide": True}),
                                     (16, 3, {"wide": True, "seed": 3}), (16, 3, {"stride": 2, "Co": 32}), (16, 3, {"stride": 2, "Co": 24, "H": 16, "W": 66, "N": 1, "relu": False}), (16, 3, {"stride": 2, "Co": 32, "H": 9, "W": 130, "wide": True})])
def test_conv_direct(emu_lib, Ci, k, kw):
    pc.check_conv_direct(emu_lib, "cpu", Ci, k, **kw)


def test_conv_direct_reads_the_image_planes(emu_lib):
    pc.check_conv_direct_planar(emu_lib, "cpu")


def test_concat_conv(emu_lib):
    pc.check_concat_conv(emu_lib, "cpu")


@pytest.mark.parametrize("args", [(1, 7, 9, 64, 64, 0), (2, 5, 6, 128, 64, 0), (1, 6, 6, 64, 160, T(64, 128)),
                                  (1, 9, 9, 64, 64, T(128, 64)), (1, 4, 5, 256, 128, T(128, 128)),
                                  (1, 7, 9, 64, 64, T(64, 64) | (1 << 29)), (1, 5, 5, 128, 64, T(128, 64) | (1 << 29))])
def test_dcn(emu_lib, args):
    pc.check_dcn(emu_lib, "cpu", *args[:5], tile=args[5])


@pytest.mark.parametrize("args", [(1, 7, 9, 64, 64, 0), (2, 5, 6, 128, 64, 0), (1, 6, 6, 64, 160, 0), (1, 9, 19, 64, 64, 0), (1, 17, 35, 32, 128, 0),
                                  (2, 3, 2, 64, 8, 0), (1, 8, 16, 64, 128, 64), (1, 11, 20, 32, 40, 128)])
def test_dcn_patch(emu_lib, args):
    """The patch form of the DCN main contraction (csrc/dcn.hip): partial tiles on every side, several images, Cout not a tile
    multiple, both tile widths, small and large offsets (large ones take the per-lane global fallback)."""
    for big in (False, True):
        pc.check_dcn(emu_lib, "cpu", *args[:5], tile=args[5], seed=3, big_offsets=big, patch=True)


@pytest.mark.parametrize("patch", [False, True])
@pytest.mark.parametrize("name", ["small", "borders", "wide"])
def test_dcn_golden_vectors(emu_lib, name, patch):
    pc.check_dcn_golden(emu_lib, "cpu", name, patch)


@pytest.mark.parametrize("args", [(1, 7, 9, 64), (2, 9, 19, 32), (1, 17, 35, 128)])
def test_dcn_producer_consumer_identical(emu_lib, args):
    """dcn_pc_kernel (8 waves: producers gather + blend + split into LDS, consumers multiply) == dcn_patch_kernel<2>, bit for bit."""
    for big in (False, True):
        pc.check_dcn_pc_identical(emu_lib, "cpu", *args, big_offsets=big, seed=5)


def test_dcn_patch_batch_invariance(emu_lib):
    pc.check_dcn_patch_batch_invariance(emu_lib, "cpu", 9, 18, 64, 64, N=3, reps=2)


def test_dcn_big_offsets(emu_lib):
    # offsets of several pixels: samples leave the image, all four zero-padding branches hit
    pc.check_dcn(emu_lib, "cpu", 1, 6, 8, 64, 64, big_offsets=True, seed=3)


def test_pool_upsample(emu_lib):
    pc.check_pool_upsample(emu_lib, "cpu")


def test_layout(emu_lib):
    pc.check_layout(emu_lib, "cpu")


@pytest.mark.parametrize("C,Co", [(16, 32), (64, 48), (128, 64), (512, 32)])
def test_embed_map(emu_lib, C, Co):
    pc.check_embed_map(emu_lib, "cpu", C, Co)


def test_embed_fused(emu_lib):
    pc.check_embed_fused(emu_lib, "cpu")


@pytest.mark.parametrize("tile", [0, T(128, 32), T(64, 32), T(32, 32), T(64, 64)])
def test_sparse_row_conv(emu_lib, tile):
    pc.check_sparse_conv(emu_lib, "cpu", tile)


def test_heads_at_peaks(emu_lib):
    pc.check_heads_at_peaks(emu_lib, "cpu")


def test_topk_edge_cases(emu_lib):
    pc.check_topk_edge_cases(emu_lib, "cpu")


@pytest.mark.parametrize("dataset", ["mot", "nuscenes"])
def test_lstm(emu_lib, dataset):
    pc.check_lstm(emu_lib, "cpu", dataset)


def test_pair_mlp_fused(emu_lib, monkeypatch):
    """The pair MLP as ONE launch, both arithmetics of the library; the persistent workgroups' walk over row tiles (grid capped at 2 for
    5 tiles: the weight stream runs on across tiles)."""
    monkeypatch.setenv("DEFT_PAIR_MLP_GRID", "2")
    pc.check_pair_mlp(emu_lib, "cpu", shapes=((5, 12, 1, 9), (100, 37, 20)), Q=(7, 8))
    pc.check_pair_mlp(emu_lib.twin(), "cpu", shapes=((3, 4),), Q=(5,), ring=False)


def test_affinity(emu_lib):
    import deft_oracle as O
    pc.check_affinity(emu_lib, "cpu", O.synth_state_dict("mot"), golden_tag="mot_128x160")


@pytest.mark.slow
def test_forward_embed_mot(emu_lib):
    """Whole path on a 32x64 frame (emulated): backbone+DCN neck+heads+decode+embedding."""
    import deft_oracle as O
    sd = O.synth_state_dict("mot")
    plan, rep, (ora_out, ora_maps) = pc.check_forward(emu_lib, "cpu", "mot", 32, 64, sd=sd)
    pc.check_embed(emu_lib, "cpu", plan, ora_maps, sd)


def test_forward_every_presplit_kernel_forced(emu_lib):
    """The same frame with the tile-count gates off (engine.P3_MIN_TILES = 0): the direct full-resolution kernel, the halo and im2col
    pre-split kernels and the folded heat-map head all run inside the model, against the oracle."""
    import deft_oracle as O
    from deft_amd import engine
    saved, engine.P3_MIN_TILES = engine.P3_MIN_TILES, 0
    saved_d = engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE, engine.OFFSET_FP32_MIN_HW
    engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE, engine.OFFSET_FP32_MIN_HW = 0, 1e9, 0            # ... and every DCN on the patch form (csrc/dcn.hip)
    try:
        sd = O.synth_state_dict("mot")
        plan, rep, _ = pc.check_forward(emu_lib, "cpu", "mot", 32, 128, sd=sd)
        kinds = [op[0] for op in plan.ops]
        assert kinds.count("deft_conv_direct") == 3 and kinds.count("deft_fold_finish") >= 1
        assert any(d.p3_kernel == 1 for _, _, d in plan._gemms) and any(d.x3 and not d.p3_kernel for _, _, d in plan._gemms)
        assert sum(d.p3_kernel == 2 for e, _, d in plan._gemms if e == "deft_dcn_v2_nhwc") == 16
        assert sum(d.p3_kernel == 3 for e, _, d in plan._gemms if e == "deft_conv2d_nhwc") == 16          # ... and their offset convs on the fp32-patch form
    finally:
        engine.P3_MIN_TILES = saved
        engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE, engine.OFFSET_FP32_MIN_HW = saved_d


@pytest.mark.parametrize("dataset", ["mot", "kitti_tracking"])
def test_device_detect_record(emu_lib, dataset):
    n_res, n_sel = pc.check_device_detect(emu_lib, "cpu", dataset)
    assert n_sel <= n_res and (dataset != "mot" or n_sel == n_res)


@pytest.mark.parametrize("mode", ["fix_res", "fix_short", "keep_res"])
def test_fused_detector_run_on_uint8_frames(emu_lib, mode):
    """mode: the three input modes of Detector._transform_scale (detector.py:346-376) on the device pre-processor."""
    pc.check_fused_run_u8(emu_lib, "cpu", **({} if mode == "fix_res" else {"sh": 40, "sw": 70, "H": 64, "K": 8, "mode": mode}))


def test_seam_dcn_module(emu_lib):
    pc.check_seam_dcn(emu_lib, "cpu")


def test_seam_lstm(emu_lib):
    pc.check_seam_lstm(emu_lib, "cpu", "mot")
    pc.check_seam_lstm(emu_lib, "cpu", "nuscenes")


@pytest.mark.slow
def test_seam_model_afe_decode(emu_lib):
    pc.check_seam_model(emu_lib, "cpu", "mot", 32, 64)


def test_track_similarity(emu_lib):
    pc.check_track_similarity(emu_lib, "cpu")


@pytest.mark.parametrize("dataset", ["mot", "nuscenes"])
def test_motion_step(emu_lib, dataset):
    pc.check_motion(emu_lib, "cpu", dataset)


@pytest.mark.parametrize("kw", [
    dict(bm=64, bn=64, S=4),
    dict(bm=64, bn=64, S=3, k=3, Ci=32),                      # S does not divide the 9 chunks evenly
    dict(bm=128, bn=64, S=2, two_stage=True),                  # LDS-DMA loop form
    dict(bm=32, bn=32, S=4, Co=27),                            # intra-workgroup (4 waves) + cross-workgroup split
    dict(bm=64, bn=32, S=2, Co=20, two_stage=True),
    dict(bm=64, bn=64, S=4, k=1, Ci=320),                      # 1x1: the cursor is the flat k
    dict(bm=64, bn=64, S=5, korder=1),                         # (channel block, tap) K order
    dict(bm=128, bn=128, S=2, k=3, Ci=16),                     # Cin < 32: per-lane taps
])
def test_conv_splitk(emu_lib, kw):
    pc.check_conv_splitk(emu_lib, "cpu", **kw)


def test_splitk_auto(emu_lib):
    pc.check_splitk_auto(emu_lib, "cpu")


def test_conv_random_shapes(emu_lib):
    """Seeded sweep over geometry the fixed cases do not list: odd sizes, strides, paddings, channel counts that
    are not tile multiples, every tile (incl. both loop forms), library-chosen split-K."""
    import random
    rnd = random.Random(20260926)
    tiles = [pc.T(128, 128), pc.T(128, 64), pc.T(128, 32), pc.T(64, 64), pc.T(64, 128), pc.T(64, 32), pc.T(32, 32), 0]
    for case in range(14):
        k = rnd.choice([1, 3, 3, 5])
        Ci = rnd.choice([4, 8, 16, 32, 64, 128]) if k > 1 else rnd.choice([4, 12, 36, 64, 100, 256])
        Co = rnd.choice([1, 5, 16, 27, 33, 64, 96, 130])
        stride = rnd.choice([1, 1, 2])
        pad = rnd.choice([0, k // 2])
        H = rnd.randint(max(k, 3), 13); W = rnd.randint(max(k, 3), 15)
        N = rnd.randint(1, 3)
        tile = rnd.choice(tiles)
        if tile and rnd.random() < 0.4:
            tile |= 1 << 29
        if Ci < 32 and k == 5 and ((Ci * 25 + 31) // 32 * 32) // Ci > 64:
            continue                                     # Cin < 32 supports at most 64 taps incl. K padding
        pc.check_conv(emu_lib, "cpu", N, H, W, Ci, Co, k, stride, pad, tile, res=bool(case & 1), relu=bool(case & 2), seed=case)


def test_dcn_random_shapes(emu_lib):
    import random
    rnd = random.Random(7)
    for case in range(8):
        Ci = rnd.choice([32, 64, 128])
        Co = rnd.choice([8, 27, 40, 64, 100, 130])
        N, H, W = rnd.randint(1, 2), rnd.randint(2, 9), rnd.randint(2, 11)
        tile = rnd.choice([0, pc.T(64, 64), pc.T(64, 128), pc.T(128, 64), pc.T(64, 64) | (1 << 29)])
        pc.check_dcn(emu_lib, "cpu", N, H, W, Ci, Co, tile=tile, seed=case, big_offsets=bool(case & 1))


def test_embed_align_corners_switch(emu_lib):
    """grid_sample(align_corners=True): the torch 1.2 behaviour of the authors' environment."""
    pc.check_embed_map(emu_lib, "cpu", 64, 48, align_corners=True)
    pc.check_embed_fused(emu_lib, "cpu", align_corners=True)


@pytest.mark.parametrize("prec", [0, 1])
def test_both_contraction_arithmetics(emu_lib, prec):
    """DeftGemmDesc.prec: 0 = fp32 MFMA (k-ordered fmaf chain), 1 = six bf16 MFMA products per fp32 product with fp32
    accumulation (the default).  The same conv / DCN / pair-MLP checks against the oracle with either."""
    import deft_oracle as O
    from deft_amd import engine
    saved, engine.PREC = engine.PREC, prec
    try:
        pc.check_conv(emu_lib, "cpu", 1, 10, 12, 64, 64, 3, 1, 1, pc.T(128, 64), res=True)
        pc.check_conv(emu_lib, "cpu", 2, 6, 10, 448, 128, 1, 1, 0, pc.T(128, 128))
        pc.check_conv(emu_lib, "cpu", 1, 9, 7, 16, 40, 3, 2, 1, pc.T(64, 64))          # Cin < 32: per-lane taps
        pc.check_conv_splitk(emu_lib, "cpu", bm=64, bn=64, S=4)
        pc.check_dcn(emu_lib, "cpu", 1, 9, 11, 64, 64)
        pc.check_dcn(emu_lib, "cpu", 2, 5, 6, 128, 130, tile=pc.T(64, 128), big_offsets=True)
        pc.check_affinity(emu_lib, "cpu", O.synth_state_dict("mot"))
    finally:
        engine.PREC = saved


def test_split_is_exact_identity_conv(emu_lib):
    """prec = 1: the operand pieces carry fp32 values through the matrix cores -- exactly (three bf16 pieces) or to half an fp32 ulp inside
    the fp16 range and loudly beyond it (two fp16 pieces).  See parity_checks.check_split_identity."""
    from deft_amd import engine
    assert engine.PREC == 1
    pc.check_split_identity(emu_lib, "cpu")


def test_track_similarity_random_sweep(emu_lib):
    """Random node lists (0..9 nodes per track, random frames / rows / decay factors) for both node-selection rules:
    the device medians must equal numpy's on the decayed blocks bit for bit."""
    import deft_oracle as O
    from deft_amd import tracker as DT
    rnd = np.random.RandomState(42)
    for trial in range(6):
        frame = 60 + trial
        prev = sorted(rnd.choice(np.arange(frame - 55, frame), size=rnd.randint(3, 12), replace=False).tolist())
        ndet = int(rnd.randint(1, 9))
        raw = {p: rnd.rand(int(rnd.randint(1, 7)), ndet + 1).astype(np.float32) for p in prev}
        deltas = {p: float(rnd.choice([1.0, 0.3, pow(0.01, (frame - p) / 3.0)])) for p in prev}
        tracks_nodes = []
        for _ in range(int(rnd.randint(1, 9))):
            n = int(rnd.randint(0, 10))
            fr = sorted(rnd.choice(prev, size=min(n, len(prev)), replace=False).tolist())
            tracks_nodes.append([(f, int(rnd.randint(raw[f].shape[0]))) for f in fr])
        pool = [pc.SimpleNamespaceNodes(nodes) for nodes in tracks_nodes]
        for ds in ("mot", "nuscenes"):
            me = pc._similarity_harness(emu_lib, "cpu", raw, deltas, ds)
            me.recorder._dev = (frame,) + me.recorder._pack
            got = DT.get_similarity(me, frame, pool, ndet)
            want = O.track_similarity({p: raw[p] * deltas[p] for p in prev}, tracks_nodes, frame, ndet, ds)
            assert np.array_equal(got, want), (trial, ds)


# pre-split operands (igemm3.hip): bit-identical to the in-loop split of igemm.hip, P3 epilogue output exact
@pytest.mark.parametrize("args", [
    (1, 9, 11, 64, 64, 128, 3, 1, T(128, 64), T(128, 128)),
    (2, 10, 12, 32, 64, 64, 3, 2, T(64, 64), T(256, 64)),
    (1, 7, 9, 96, 128, 64, 1, 1, T(256, 128), T(128, 64) | (1 << 29)),        # 1x1 with Cin not a power of two; 3 LDS stages
    (1, 7, 9, 64, 128, 256, 3, 1, T(128, 128) | (1 << 29), T(128, 256)),
    (1, 5, 6, 64, 64, 64, 3, 1, T(64, 64) | (1 << 29), 0),                    # nk = 18 / auto tile
    (1, 4, 5, 32, 64, 64, 1, 1, 0, T(64, 64)),
    (2, 9, 11, 64, 128, 128, 3, 1, T(128, 128) | (1 << 30), T(128, 64) | (1 << 30)),     # one LDS stage, several workgroups per CU
    (2, 7, 9, 64, 128, 64, 3, 2, T(64, 128) | (1 << 30), T(64, 64) | (1 << 30)),                                # single-chunk K (nk == 1) feeding a 3x3
])
def test_conv_presplit(emu_lib, args):
    pc.check_conv_p3(emu_lib, "cpu", *args)


def test_conv_presplit_splitk(emu_lib):
    pc.check_conv_p3(emu_lib, "cpu", 1, 7, 9, 64, 64, 128, 3, 1, T(128, 64), T(128, 128), splitk=3)
    pc.check_conv_p3(emu_lib, "cpu", 1, 7, 9, 128, 128, 64, 3, 1, T(256, 128), T(64, 64) | (1 << 29), splitk=4)


def test_peaked_heatmap_ordered_topk(emu_lib):
    """Small-map version of the GPU test: a peaked heat map must give the oracle's ordered top-K outright."""
    pc.check_peaked_heatmap(emu_lib, "cpu", 64, 96, K=5, nblobs=6)


# halo-tile form of the 3x3 / stride 1 convs (DeftGemmDesc.p3_kernel = 1): ragged map edges, every tile shape
@pytest.mark.parametrize("args", [
    (1, 8, 32, 32, 64, 3, 1, 1, 0), (2, 9, 37, 64, 128, 3, 1, 1, 0), (1, 5, 70, 64, 64, 3, 1, 1, T(4, 64)),
    (1, 6, 33, 32, 32, 3, 1, 1, T(4, 32)), (1, 10, 40, 64, 128, 3, 1, 1, T(8, 128)), (1, 9, 31, 128, 64, 3, 1, 1, T(8, 64)),
    (1, 4, 20, 64, 200, 3, 1, 1, T(4, 128)),
    (1, 9, 24, 64, 128, 3, 1, 1, T(8, 128) | (1 << 28)), (2, 16, 16, 32, 64, 3, 1, 1, T(8, 64) | (1 << 28)), (1, 11, 37, 64, 32, 3, 1, 1, T(8, 32) | (1 << 28)),   # 8 x 16 pixel tiles
    (2, 7, 40, 64, 32, 3, 1, 1, T(4, 32) | (1 << 29)),        # narrow tile, one tap per interval (default: a filter row per interval)
])
def test_conv_halo(emu_lib, args):
    pc.check_conv(emu_lib, "cpu", *args, res=True, relu=True, p3="halo")


# the heat-map head's 1x1 conv folded into the 3x3 conv's epilogue (DeftGemmDesc.fold_w): halo and im2col kernels, 1 and 2 n-tiles, ragged edges
@pytest.mark.parametrize("args", [(1, 9, 37, 64, 256, 1, "halo", 0), (2, 5, 33, 64, 256, 3, "halo", T(4, 128)), (1, 6, 40, 64, 128, 10, "halo", T(8, 64)),
                                  (1, 7, 19, 64, 256, 2, "im2col", T(64, 128) | (1 << 30)), (1, 6, 21, 64, 72, 1, "im2col", T(128, 64) | (1 << 30))])
def test_conv_fold(emu_lib, args):
    pc.check_conv_fold(emu_lib, "cpu", *args)


@pytest.mark.parametrize("kw", [{}, {"k": 1, "stride": 1, "Ci": 96, "Cm": 128, "Co": 64}, {"Cm": 256, "Ci": 64, "H": 9, "W": 11}])
def test_conv_inloop_piece_output(emu_lib, kw):
    pc.check_conv_inloop_y3(emu_lib, "cpu", **kw)


def test_weight_dma_identical(emu_lib):
    pc.check_weight_dma_identical(emu_lib, "cpu")


@pytest.mark.slow
@pytest.mark.parametrize("tag", ["mot", "mot_lstm"])
def test_composed_dropin_replays_reference_trace(emu_lib, tag):
    """The reference's own Detector.run, traced (tests/golden/detector_trace_*.npz), replayed through the composed HIP path."""
    print(pc.check_detector_trace(emu_lib, "cpu", tag))


@pytest.mark.slow
def test_nuscenes_run_replays_reference_trace(emu_lib):
    """BASELINE configs[4]: the reference's nuScenes Detector.run, traced, replayed up to the per-class tracker calls."""
    print(pc.check_detector_trace_nuscenes(emu_lib, "cpu"))


def test_fused_run_with_lookahead(emu_lib):
    pc.check_fused_run_prefetch(emu_lib, "cpu", sh=30, sw=50, H=32, W=64, K=8, T=3, hook=True)


def test_fused_run_with_pair_lookahead(emu_lib):
    """Detector.lookahead_frames = 2: two frames per lookahead pass, handed out over two calls."""
    pc.check_fused_run_prefetch(emu_lib, "cpu", sh=30, sw=50, H=32, W=64, K=8, T=6, hook=True, pairs=True)


def test_preprocess_u8(emu_lib):
    pc.check_preprocess_u8(emu_lib, "cpu")


# ---- the same kernels with LDS-DMA pieces delivered as LATE as the hardware may (tests/hipemu: late_dma): a read of a DMA stage
# ---- that is not ordered behind the issuing wave's vmcnt wait + a barrier sees stale LDS, a piece in flight at wave exit aborts
@pytest.fixture
def late_dma(emu_lib):
    emu_lib.cdll.hipemu_set_late_dma(1)
    yield emu_lib
    emu_lib.cdll.hipemu_set_late_dma(0)


@pytest.mark.parametrize("args", [(1, 9, 11, 64, 64, 64, 3, 1, T(128, 128), T(128, 128)), (1, 7, 9, 64, 128, 64, 3, 1, T(64, 128) | (1 << 30), T(128, 64) | (1 << 30)),
                                  (1, 6, 10, 32, 64, 128, 3, 1, T(128, 64) | (1 << 29), T(128, 128) | (1 << 29))])
def test_late_dma_presplit_im2col(late_dma, args):
    pc.check_conv_p3(late_dma, "cpu", *args)


@pytest.mark.parametrize("args", [(2, 9, 37, 64, 128, 3, 1, 1, 0), (1, 6, 33, 32, 32, 3, 1, 1, T(4, 32)), (1, 9, 24, 64, 128, 3, 1, 1, T(8, 128) | (1 << 28)),
                                  (1, 11, 37, 64, 32, 3, 1, 1, T(8, 32) | (1 << 28))])
def test_late_dma_halo(late_dma, args):
    pc.check_conv(late_dma, "cpu", *args, res=True, relu=True, p3="halo")


def test_late_dma_inloop_weight_dma(late_dma):
    pc.check_weight_dma_identical(late_dma, "cpu")


@pytest.mark.parametrize("args", [(1, 7, 9, 64, 64), (1, 9, 19, 64, 160), (2, 5, 6, 128, 64)])
def test_late_dma_dcn_patch(late_dma, args):
    """The patch form's LDS-DMA pipeline (patch per 16-channel block, double-buffered weight chunks) under late delivery."""
    for big in (False, True):
        pc.check_dcn(late_dma, "cpu", *args, big_offsets=big, patch=True)


def test_late_dma_pair_mlp(late_dma):
    pc.check_pair_mlp(late_dma, "cpu", shapes=((40, 31),), Q=(20,), ring=False)          # 6 row tiles, the weight ring under late delivery


def test_late_dma_dcn_producer_consumer(late_dma):
    for big in (False, True):
        pc.check_dcn_pc_identical(late_dma, "cpu", 2, 9, 19, 64, big_offsets=big, seed=6)


def test_late_dma_forward_every_presplit_kernel(late_dma):
    import deft_oracle as O
    from deft_amd import engine
    saved, engine.P3_MIN_TILES = engine.P3_MIN_TILES, 0
    saved_b = engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE
    engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE = 0, 1e9              # every DCN on the patch form too (three weight stages, two patch buffers)
    try:
        pc.check_forward(late_dma, "cpu", "mot", 32, 128, sd=O.synth_state_dict("mot"))
    finally:
        engine.P3_MIN_TILES = saved
        engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE = saved_b


def test_dataflow_schedule_is_order_independent(emu_lib):
    """_Plan.dependencies / build_schedule: executing the launch list in any order the dependencies allow gives the same bits.  The
    emulator runs launches synchronously, so the multi-stream schedule is replayed as two adversarial topological orders (always the
    LAST ready op in program order; the scheduler's own issue order) against program order."""
    import torch
    from deft_amd import engine, synth
    sd = synth.synth_state_dict("mot")
    x = torch.randn(1, 3, 32, 64, generator=torch.Generator().manual_seed(5))
    plan = engine.DlaSegPlan(sd, 1, 32, 64, "mot", K=20, device="cpu", lib=emu_lib)
    outs = lambda: [t.clone() for t in (plan.scores, plan.inds, plan.bboxes, plan.head_vals)] + [fm.buf.clone() for fm in plan.fmaps]
    plan.forward(x)
    ref = outs()
    deps = plan.dependencies()
    n = len(plan.ops)
    assert all(j < i for i in range(n) for j in deps[i])
    # the structure the schedule exploits: a level's project branch is independent of its first conv, DLAUp's projections of backbone maps
    # do not wait for the previous IDA stage
    name = {o[1]: i for i, o in enumerate(plan.ops)}
    reach = [set(d) for d in deps]
    for i in range(n):
        for j in list(reach[i]):
            reach[i] |= reach[j]
    assert name["base.level3.tree1.project"] not in reach[name["base.level3.tree1.tree1.conv1"]]
    assert name["base.level3.tree1.tree1.conv1"] not in reach[name["base.level3.tree1.project"]]
    assert name["dla_up.ida_0.node_1.dcn"] not in reach[name["dla_up.ida_1.proj_1.dcn"]]
    assert name["dla_up.ida_1.proj_1.offset"] in reach[name["dla_up.ida_1.proj_1.dcn"]]
    assert name["dla_up.ida_0.node_1.dcn"] in reach[name["dla_up.ida_1.proj_2.offset"]]
    serial, par = plan.build_schedule(3)
    sc = plan.sched
    assert par < serial and sorted(sc["order"]) == list(range(n)) and any(sc["where"])
    pos = {i: k for k, i in enumerate(sc["order"])}
    assert all(pos[j] < pos[i] for i in range(n) for j in deps[i])
    for i in range(n):            # every cross-stream dependency is covered by an event wait of this op or of an earlier op on its stream
        for j in deps[i]:
            if sc["where"][j] != sc["where"][i]:
                cover = [w for k in sc["order"][:pos[i] + 1] if sc["where"][k] == sc["where"][i] for w in sc["waits"][k]
                         if sc["where"][w] == sc["where"][j] and pos[w] >= pos[j]]
                assert cover, (plan.ops[i][1], plan.ops[j][1])
    last_first, done, left = [], set(), set(range(n))
    while left:
        i = max(k for k in left if deps[k] <= done)
        last_first.append(i); done.add(i); left.remove(i)
    assert last_first != list(range(n))
    x2 = torch.randn(1, 3, 32, 64, generator=torch.Generator().manual_seed(6))
    for order in (last_first, sc["order"]):
        plan.forward(x2)                   # every buffer now holds ANOTHER frame's values: an op that ran too early would read those
        assert not torch.equal(outs()[-1], ref[-1])
        plan.image.copy_(x)
        for i in order:
            plan.ops[i][2]()
        for a, b in zip(ref, outs()):
            assert torch.equal(a, b)
    # the check has teeth: drop one dependency and the reordered run reads stale data
    broken = [set(d) for d in deps]
    victim = name["dla_up.ida_1.proj_2.offset"]
    broken[victim] = set()
    order, done, left = [], set(), set(range(n))
    while left:
        i = max(k for k in left if broken[k] <= done)
        order.append(i); done.add(i); left.remove(i)
    plan.forward(x2)
    plan.image.copy_(x)
    for i in order:
        plan.ops[i][2]()
    assert not all(torch.equal(a, b) for a, b in zip(ref, outs()))


@pytest.mark.parametrize("dataset,lstm", [("kitti_tracking", True), ("nuscenes", True), ("nuscenes", False)])
def test_fused_run_array_tracker(emu_lib, dataset, lstm):
    pc.check_fused_run_array_tracker(emu_lib, "cpu", dataset, lstm, T=3, pairs=(dataset == "kitti_tracking"))


@pytest.mark.parametrize("tag", ["mot", "mot_lstm", "nuscenes"])
def test_tracks_against_reference_trace(emu_lib, tag):
    pc.check_tracks_against_reference_trace(emu_lib, "cpu", tag)


def test_out_of_range_activation_is_rerun_on_the_range_free_arithmetic(emu_lib, monkeypatch):
    from deft_amd import hiplib
    monkeypatch.setattr(hiplib, "_lib", emu_lib)
    pc.check_out_of_range_fallback(emu_lib, -1)


def test_non_finite_similarity_is_an_error():
    """the embedding / affinity chain has no heat map behind it -- its overflow shows in the similarity matrix, which the
    native cascade would read as gated pairs.  array_tracker checks it where it lands on the host."""
    from deft_amd import array_tracker as AT
    AT._finite_sim(None)
    AT._finite_sim(np.ones((3, 4), np.float32))
    bad = np.ones((3, 4), np.float32); bad[1, 2] = np.nan
    with pytest.raises(FloatingPointError, match="similarity"):
        AT._finite_sim(bad)
