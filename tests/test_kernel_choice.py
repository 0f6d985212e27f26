"""Which kernel a conv runs on is decided per layer shape by engine.p3_choice from measurements on MI355X (tools/bench_p3.py,
profiles/r2_bench_p3.log).  This pins the decisions for the config-B layer shapes (16 frames per launch) and for one
frame per launch, so that a change of the rules shows up as a diff here and is re-measured, not discovered in a profile."""
from deft_amd import engine

W16 = engine.P3H_W16


def choice(Cin, Cout, H, W, N=16, k=3, stride=1):
    return engine.p3_choice(k, k, stride, k // 2, Cin, Cout, H, W, N * (H // stride) * (W // stride), 1)


def test_config_b_throughput_shapes():
    assert choice(64, 256, 152, 272) == ("halo", (8 << 16) | 128 | W16)          # hm head: 8 x 16 tiles pad W = 272 by 0 %
    assert choice(64, 64, 152, 272) == ("halo", (8 << 16) | 64 | W16)            # 64-column convs only as 8 x 16
    assert choice(64, 32, 152, 272) == ("halo", (8 << 16) | 32 | W16)            # offset/mask conv (27 -> 32 columns)
    assert choice(128, 128, 76, 136) == ("halo", (8 << 16) | 128 | W16)          # 11 % padding instead of 18 %
    assert choice(128, 32, 76, 136) == ("halo", (8 << 16) | 32 | W16)
    assert choice(256, 32, 38, 68) == ("halo", (8 << 16) | 32 | W16)             # narrow convs: up to 30 % padding, from 384 tiles
    kind, tile = choice(256, 256, 38, 68)                                        # 24 % halo padding: the im2col loop wins; round 6: 646 tiles of 128 x 128 remain
    assert kind == "im2col" and tile == ((128 << 16) | 128)                      # -> the two-stage 128 x 128 tile (+1.1 % of the step, profiles/r6_im2col_tile_ab.log)
    kind, tile = choice(512, 512, 19, 34)                                        # 324 tiles of 128 x 128: too few -> the one-stage 64 x 128 tile
    assert kind == "im2col" and tile == ((64 << 16) | 128 | engine.P3_1STAGE)
    kind, tile = choice(512, 512, 19, 34, N=32)                                  # the 32-frame sub-batch plans of bench.py's default step: 648 tiles
    assert kind == "im2col" and tile == ((128 << 16) | 128)
    assert choice(512, 32, 19, 34) is None                                       # 144 tiles: the fp32-instruction split-K tile is faster
    assert choice(64, 128, 152, 272, stride=2) is None                           # stride 2 and 1x1: no gain from 6-byte pieces
    assert choice(448, 128, 76, 136, k=1) is None


def test_tile_shape_follows_the_padding():
    assert choice(64, 128, 128, 128) == ("halo", 0)                              # W % 32 == 0: 4 x 32 tiles (automatic)
    assert choice(64, 128, 96, 320)[0] == "halo" and choice(64, 128, 96, 320)[1] == 0
    assert choice(64, 128, 112, 200) == ("halo", (8 << 16) | 128 | W16)          # nuScenes 800x448 / 4: 200 = 8 mod 16


def test_one_frame_per_launch_stays_on_the_in_loop_kernel():
    """Latency mode: launches with fewer than 512 tiles keep igemm.hip and its cross-workgroup split-K (2.39 vs 2.88 ms per frame)."""
    assert choice(128, 128, 76, 136, N=1) is None
    assert choice(256, 256, 38, 68, N=1) is None
    assert choice(64, 32, 152, 272, N=1) is None                                  # 323 tiles < 384
    assert choice(64, 256, 152, 272, N=1)[0] == "halo"                            # 646 tiles: the head still fills the chip
