"""Frame pipeline: detect -> embed -> (exchange) -> affinity, one batch of frames per step.

This is the data-parallel form of `Detector.run` + `FeatureRecorder.update`
(detector.py:112-344, tracker.py:59-90): whole frames are independent through
detection, decode and embedding extraction, and the affinity of frame t against its
history depends only on embeddings (tracker.py:76-90 reads `all_features` only).
With world_size > 1 consecutive frames of the stream are sharded over the ranks and
ONE all-gather of fixed-size embedding records per step (RCCL over xGMI) gives every
rank the history its own frames need; there is no other cross-GPU state.

`compute` is injected so the CPU/gloo tests can drive the exchange logic with a
stand-in; the product default (HipCompute) calls the HIP library and nothing else.
"""
import torch
import torch.distributed as dist

from . import engine


class HipCompute:
    """detect+embed and affinity on the local GPU through libdeft_hip.so.

    `streams` > 1 splits the step's frames into that many independent sub-batches, each with
    its own plan (own buffers) on its own HIP stream: frames share no state before the
    embedding exchange, so the hardware overlaps one sub-batch's kernel tails (partially
    filled last wave of workgroups) and barrier stalls with the other's workgroups."""

    def __init__(self, sd, batch, H, W, dataset="mot", K=100, max_object=100, device="cuda", lib=None, streams=1, ndet=None, overlap=False):
        """ndet: embeddings are extracted for the first `ndet` (<= K) decoded detections of every frame (default K).
        overlap: with ONE sub-batch, still run detection on a side stream, so that step k+1's detection overlaps step k's affinity
        chain on the caller's stream.  Off by default: measured SLOWER at one frame per step (2.07 -> 2.11 ms: the cross-stream
        hand-overs cost more than the chain's idle compute units give back, tools/probe/dataflow_ab.sh)."""
        assert batch % streams == 0
        ndet = K if ndet is None else ndet
        self.device = torch.device(device)
        self.nstream, self.sub = streams, batch // streams
        self.plans = [engine.DlaSegPlan(sd, self.sub, H, W, dataset, K=K, device=device, lib=lib) for _ in range(streams)]
        self.plan = self.plans[0]
        self.afe = engine.AfePlan(sd, max_object, device, lib)
        self.D = self.afe.D
        self.K, self.ndet = K, ndet
        self.emb = torch.zeros(batch, ndet, self.D, dtype=torch.float32, device=self.device)
        self.overlap = streams > 1 or (bool(overlap) and self.device.type == "cuda")
        if self.overlap:
            self.side = [torch.cuda.Stream(device=self.device) for _ in range(streams)]
            self.ev_main = torch.cuda.Event()
            self.ev_in = torch.cuda.Event()
            self.ev_side = [torch.cuda.Event() for _ in range(streams)]

    def use_u8(self, sh, sw):
        """Feed the plans uint8 HWC frames [batch, sh, sw, 3]: warp + normalise + layout on the device (deft_preprocess_u8,
        detector.py:377-395) instead of fp32 NCHW tensors pre-processed by the host."""
        for p in self.plans:
            p.use_u8_input(sh, sw)

    def autotune(self, images, verbose=False):
        """One-off per-layer tile search (engine._Plan.autotune) on real activations: run the
        sub-batch plans once on `images` [batch,3,H,W], then time the candidates."""
        for s_, p in enumerate(self.plans):
            p.forward(images[s_ * self.sub:(s_ + 1) * self.sub])
            torch.cuda.synchronize(self.device)
            p.autotune(verbose=verbose and s_ == 0)

    def tune_dataflow(self, images, nstreams=None):
        """Spread every sub-batch plan's launch list over several HIP streams along its data dependencies (engine._Plan.tune_schedule;
        decided by engine.DATAFLOW / DATAFLOW_MAX_N unless `nstreams` is given).  Returns the modelled (serial, scheduled) ms per plan."""
        out = []
        for s_, p in enumerate(self.plans):
            (p.image_u8 if images.dtype == torch.uint8 else p.image).copy_(images[s_ * self.sub:(s_ + 1) * self.sub])
            out.append(p.tune_schedule(nstreams))
        return out

    graphs = None          # per sub-batch hipGraph of (plan launches + embedding extraction), see capture()

    def capture(self, images):
        """Capture each sub-batch's launch list (backbone, neck, heads, decode, embedding: ~170
        launches) into a hipGraph and replay it per step: the per-launch host cost (Python +
        hipLaunchKernel, ~10-20 us each) disappears from the critical path, which matters when the
        sub-batch is small (latency mode).  Buffers are static (plan-owned), so replays are valid."""
        assert self.device.type == "cuda"
        side = self.side if self.overlap else [torch.cuda.Stream(device=self.device)]
        self.graphs = []
        for s_, p in enumerate(self.plans):
            sl = slice(s_ * self.sub, (s_ + 1) * self.sub)
            p.forward(images[sl]); self.afe.extract(p.fmaps, p.centers if self.ndet == self.K else p.centers[:, :self.ndet], out=self.emb[sl])      # warm-up: attributes, caches
            torch.cuda.synchronize(self.device)
            if engine.DATAFLOW > 1 and p.sched is None:
                p.tune_schedule()                  # one frame per GPU: independent branches of the launch list become parallel graph branches
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side[s_ % len(side)]):
                p.run()
                self.afe.extract(p.fmaps, p.centers if self.ndet == self.K else p.centers[:, :self.ndet], out=self.emb[sl])
            self.graphs.append(g)
        torch.cuda.synchronize(self.device)

    def _run_plan(self, s_, images):
        p = self.plans[s_]
        sl = slice(s_ * self.sub, (s_ + 1) * self.sub)
        if self.graphs is not None:
            (p.image_u8 if images.dtype == torch.uint8 else p.image).copy_(images[sl], non_blocking=True)
            engine._Plan.replay_graph(self.graphs[s_], self.device)
        else:
            p.forward_u8(images[sl]) if images.dtype == torch.uint8 else p.forward(images[sl])
            self.afe.extract(p.fmaps, p.centers if self.ndet == self.K else p.centers[:, :self.ndet], out=self.emb[sl])

    serialize = False      # profiling aid: run the sub-batch plans one after the other on the current stream

    def detect_embed(self, images):
        if self.serialize or not self.overlap:
            for s in range(len(self.plans)):
                self._run_plan(s, images)
            return self.emb                                                     # [batch, K, D]
        main = torch.cuda.current_stream(self.device)
        if not self.emb_released:                       # nobody told us when emb was consumed: wait for everything
            self.ev_main.record(main)                   # queued on the main stream so far
        self.emb_released = False
        # the step's INPUT: whatever the caller ordered on the main stream before this call (e.g. FrameFeeder.take(): the H2D copy of
        # these frames) must be visible to the side streams too -- a fresh event every step, independent of the emb hand-over above
        self.ev_in.record(main)
        for s, (p, st) in enumerate(zip(self.plans, self.side)):
            st.wait_event(self.ev_main)                 # the previous step's readers of emb are done
            st.wait_event(self.ev_in)                   # this step's frames are on the device
            with torch.cuda.stream(st):
                self._run_plan(s, images)
                self.ev_side[s].record(st)
        for ev in self.ev_side:
            main.wait_event(ev)
        return self.emb

    emb_released = False

    def release_emb(self):
        """Called by the pipeline once the step's embeddings have been copied out of `emb` (into the
        history ring): the NEXT step's detection may start on the side streams while this step's
        affinity chain is still running on the main stream (cross-step overlap)."""
        if self.overlap:
            self.ev_main.record(torch.cuda.current_stream(self.device))
            self.emb_released = True

    def affinity(self, hist, cur):
        return self.afe.affinity(hist, cur)[0]

    def affinity_ring(self, ring, g0, Bc, hist):
        return self.afe.affinity_ring(ring, g0, Bc, hist)


class FramePipeline:
    def __init__(self, compute, batch, K, D, history=5, device="cuda", group=None, exchange=True):
        """exchange=False: every rank is an independent replica with its own stream (BASELINE configs[4]: one nuScenes camera per
        GPU, convert_nuScenes.py:173-175) -- no collective at all, even inside a process group."""
        self.c, self.batch, self.K, self.D, self.history = compute, batch, K, D, history
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if exchange and dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # ONE flat device buffer holds the stream's recent embeddings in GLOBAL frame order: the up-to-`history` frames before the
        # step at [pos - tail_valid, pos), the step's world * batch frames at [pos, pos + world * batch) -- the all-gather writes them
        # there directly (one rank: one copy out of the compute's static buffer).  The window slides by a step per step and is moved
        # back to the front of the buffer when it reaches the end: no per-step allocation, no torch.cat of the ring.
        wb = self.world * batch
        slots = -(-history // wb) * (1 if wb >= history else 4)
        self.cap = history + wb * slots
        self.buf = torch.zeros(self.cap, K, D, dtype=torch.float32, device=self.device)
        self.pos = history
        self.tail_valid = 0
        # test hook: run the collective even in a 1-rank group (exercises the RCCL path on a 1-GPU box)
        self.force_gather = bool(exchange and dist.is_available() and dist.is_initialized() and self.world == 1)
        self.collectives = 0
        self.bytes_gathered = 0

    def step(self, images):
        """images [batch,3,H,W]: this rank's frames  (global frame index within the step =
        rank*batch + b).  Returns the list (one per local frame) of affinity blocks
        [sum_f P_f, Q+1] against the up-to-`history` preceding frames of the stream."""
        emb = self.c.detect_embed(images)                                   # [batch,K,D]
        wb = self.world * self.batch
        n0 = self.pos
        dst = self.buf[n0:n0 + wb]
        if self.world > 1 or self.force_gather:
            dist.all_gather_into_tensor(dst, emb.contiguous(), group=self.group)
            self.collectives += 1
            self.bytes_gathered += dst.numel() * 4
        else:
            dst.copy_(emb)
        if hasattr(self.c, "release_emb"):
            self.c.release_emb()                                            # emb has been read: the next step's detection may start
        ring = self.buf[n0 - self.tail_valid:n0 + wb]                       # (a view)
        base = self.tail_valid + self.rank * self.batch
        outs = []
        if base >= self.history and hasattr(self.c, "affinity_ring"):
            # steady state: every local frame has `history` predecessors -> one batched chain
            # only the frames this rank scores and their history: the layer-1 products U'/V' are not
            # computed for the other ranks' frames of the step
            own = ring[base - self.history: base + self.batch]              # contiguous slice of the buffer: no copy
            blk = self.c.affinity_ring(own, self.history, self.batch, self.history)
            outs = [blk[b] for b in range(self.batch)]
        for b in range(self.batch if not outs else 0):
            g = base + b
            lo = max(0, g - self.history)
            if g == lo:
                outs.append(None)                                           # very first frame: no history
                continue
            hist = [ring[t] for t in range(lo, g)]
            outs.append(self.c.affinity(hist, ring[g]))
        self.tail_valid = min(self.history, self.tail_valid + wb)
        self.pos = n0 + wb
        if self.pos + wb > self.cap:                                        # window at the end of the buffer: move the history to the front
            tv = self.tail_valid                                            # (source starts at pos - tv >= history: the ranges are disjoint)
            self.buf[self.history - tv:self.history].copy_(self.buf[self.pos - tv:self.pos])
            self.pos = self.history
        return outs
