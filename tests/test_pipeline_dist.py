"""N>1 path on CPU: 2 gloo ranks run FramePipeline with a stand-in compute object and must
reproduce, frame for frame, what a single rank computes over the same global stream order
(frames sharded rank-major inside each step; one all-gather per step; history ring carried
across steps).  Exercises the exchange/scheduling logic only -- kernels are covered elsewhere."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deft_amd.pipeline import FramePipeline  # noqa: E402

K, D, HIST = 3, 8, 2


class FakeCompute:
    """Embedding = per-frame signature; 'affinity' = (history signatures, current signature)."""

    def detect_embed(self, images):
        sig = images.reshape(images.shape[0], -1).sum(1)
        return (sig.view(-1, 1, 1) + torch.arange(K * D, dtype=torch.float32).view(1, K, D)).contiguous()

    def affinity(self, hist, cur):
        return torch.stack([h[0, 0] for h in hist] + [cur[0, 0]])


class FakeRingCompute(FakeCompute):
    """Also offers the batched steady-state entry the product compute has (HipCompute.affinity_ring)."""

    def affinity_ring(self, ring, g0, Bc, hist):
        assert ring.is_contiguous() and g0 - hist >= 0 and g0 + Bc <= ring.shape[0]
        return torch.stack([torch.stack([ring[t, 0, 0] for t in range(g0 + c - hist, g0 + c + 1)]) for c in range(Bc)])


def _frames(nsteps, world, batch):
    g = torch.Generator().manual_seed(0)
    return torch.randn(nsteps, world * batch, 1, 2, 2, generator=g)


def _run(rank, world, batch, nsteps, port, q, ring=False):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _frames(nsteps, world, batch)
    pipe = FramePipeline(FakeRingCompute() if ring else FakeCompute(), batch, K, D, history=HIST, device="cpu")
    outs = []
    for s in range(nsteps):
        res = pipe.step(frames[s, rank * batch:(rank + 1) * batch])
        outs.append([None if r is None else [float(v) for v in r] for r in res])   # plain lists: picklable
    q.put((rank, outs))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _collect(world, batch, nsteps, port, ring=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_run, args=(r, world, batch, nsteps, port, q, ring)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("batch", [1, 2])
def test_two_ranks_match_single_rank(batch):
    nsteps, world = 4, 2
    two = _collect(world, batch, nsteps, 29613 + batch)
    one = _collect(1, world * batch, nsteps, 0)[0]           # same global frames, one rank
    frames = _frames(nsteps, world, batch)
    sig = frames.reshape(nsteps, world * batch, -1).sum(2)
    for s in range(nsteps):
        for g in range(world * batch):
            r, b = divmod(g, batch)
            a, ref = two[r][s][b], one[s][g]
            gi = s * world * batch + g                        # global frame index in the stream
            if gi == 0:
                assert a is None and ref is None
                continue
            assert a == ref
            want = [float(sig.reshape(-1)[t]) for t in range(max(0, gi - HIST), gi + 1)]
            assert [round(float(v), 4) for v in a] == [round(v, 4) for v in want]


def test_two_ranks_ring_path_matches_per_frame_path():
    """The batched ring entry (taken once every local frame has `history` predecessors) must give
    every rank exactly what the per-frame path gives a single rank."""
    nsteps, world, batch = 4, 2, 3
    two = _collect(world, batch, nsteps, 29655, ring=True)
    one = _collect(1, world * batch, nsteps, 0, ring=False)[0]
    for s in range(nsteps):
        for g in range(world * batch):
            r, b = divmod(g, batch)
            assert two[r][s][b] == one[s][g], (s, g)


# ---------------------------------------------------------------------------------------
# the same sharding with the REAL affinity kernels (SIMT-emulator build of the HIP sources):
# embeddings are synthetic, the affinity blocks go through AfePlan.affinity / affinity_ring
# ---------------------------------------------------------------------------------------
class EmuAffinityCompute:
    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import subprocess
        import deft_oracle as O
        from deft_amd import engine, hiplib
        so = os.path.join(ROOT, "tests", "hipemu", "_build", "libdeft_emu.so")
        if not os.path.exists(so):
            subprocess.check_call([os.path.join(ROOT, "tests", "hipemu", "build_emu.sh")])
        self.sd = O.synth_state_dict("mot")
        self.afe = engine.AfePlan(self.sd, 100, "cpu", hiplib.HipLib(so))
        self.D = self.afe.D

    def detect_embed(self, images):                     # [batch, ...] -> [batch, K2, D]: deterministic stand-in embeddings
        out = []
        for f in images:                                  # per-frame seed: the embedding depends on the frame alone
            g = torch.Generator().manual_seed(int(f.reshape(-1).sum().abs() * 1000) % 100000)
            out.append(torch.rand(K2, self.D, generator=g) * 3)
        return torch.stack(out).contiguous()

    def affinity(self, hist, cur):
        return self.afe.affinity(hist, cur)[0]

    def affinity_ring(self, ring, g0, Bc, hist):
        return self.afe.affinity_ring(ring, g0, Bc, hist).clone()


K2 = 5


def _run_emu(rank, world, batch, nsteps, port, q):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _frames(nsteps, world, batch)
    comp = EmuAffinityCompute()
    pipe = FramePipeline(comp, batch, K2, comp.D, history=HIST, device="cpu")
    outs = []
    for s in range(nsteps):
        res = pipe.step(frames[s, rank * batch:(rank + 1) * batch])
        outs.append([None if r is None else r.reshape(-1).tolist() for r in res])
    q.put((rank, outs))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_real_affinity_kernels():
    nsteps, world, batch = 3, 2, 2
    ctx = mp.get_context("spawn")

    def collect(w, b, port):
        q = ctx.Queue()
        ps = [ctx.Process(target=_run_emu, args=(r, w, b, nsteps, port, q)) for r in range(w)]
        for p in ps:
            p.start()
        got = dict(q.get(timeout=300) for _ in range(w))
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        return got
    two = collect(world, batch, 29677)
    one = collect(1, world * batch, 0)[0]
    for s in range(nsteps):
        for g in range(world * batch):
            r, b = divmod(g, batch)
            a, ref = two[r][s][b], one[s][g]
            if ref is None:
                assert a is None
                continue
            assert len(a) == len(ref) and max(abs(x - y) for x, y in zip(a, ref)) <= 1e-6, (s, g)
