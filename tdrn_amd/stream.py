"""Streamed inference: frames arrive from the HOST, detections go back to the host.

The reference's drivers feed one frame at a time through the host (test_video.py:98-115: cv2 frame -> BaseTransform on
the CPU -> .cuda() -> net -> Detect -> .cpu(); evaluate.py:452-461 the same from the dataset).  Here a batch of uint8
BGR frames travels H2D as uint8 (4x less than the fp32 tensor the reference uploads), is resized / mean-subtracted on the
device (tdrn_preprocess), runs net + Detect, and only the (B, C, top_k, 5) detections travel back.

Three queues, chained by events, `slots` (>= 3) buffers used cyclically:
    copy-in stream :  H2D of batch k+2's pinned frames          (waits for the step that last read that slot's device buffer)
    caller's stream:  hipGraph[slot] = tdrn_preprocess -> net (~60 launches on 4 lanes) -> Detect      (waits for its H2D)
                      (round 5: given a LIST of engines, slot s runs on pipeline s % len(engines) -- the caller's stream and
                      one extra stream per further engine: that many steps in flight, tdrn_amd/engine.py InFlight)
    copy-out stream:  D2H of batch k's detections               (waits for step k; step k + slots waits for it)
Measured (bench.py `stream`, profiles/r03_experiments.md): 0.97 of the resident rate, given copy streams that do not share a
hardware queue with a busy lane (see _pick_streams: chosen by a calibration at construction -- 16 stream pairs x 12 pipeline
steps + the four best x 32: ~320 steps on whatever the slots' buffers hold, ~1 s at batch 32; `calibrate=False` skips it).  What did NOT work: the D2H on the
step's own stream (a 2.7-MB device-to-pinned copy there is a blit kernel that queues behind the step and costs 0.54 ms:
0.85 of resident -- that alone was the whole shortfall of the first versions); the H2D as a node of the step's graph (the
executor ran it ~2.4 ms into the step, in series with Detect: 0.86-0.91); copying only one batch ahead (the copy then has
to start and finish inside one step).
Round 5, options: graph=False launches the slots' steps eagerly (two hipGraph replays on two streams do not overlap on ROCm 7.2; eager
pipelines do) on pipeline streams picked by a calibration of their own (_pick_pipeline_streams); zero_copy_out=True lets Detect write
into the slot's pinned host buffer (no D2H, no copy-out stream); copy_in="own" issues batch k + 2's H2D on the stream of the pipeline that
will run it, and the detections' D2H behind that pipeline's step (no copy stream at all; the pipeline pauses ~0.4 ms per step for its
copies while the other one keeps the chip busy).  Measured with two eager pipelines: copy streams 10.3-11.0k frames/s, copy_in="own"
10.9-11.3k (0.89 of the resident 12.45k on that box), zero-copy out -5 %: profiles/r05_experiments.md.
Protocol: batches 0 and 1 go into pinned_in(0), pinned_in(1), then prime(); before every run() -- which launches the
oldest batch not yet run -- the producer writes the batch TWO ahead of it into pinned_in(next_in()); result(slot) is
that run's output.  run() returns the slot as a TICKET (an int that also carries the step number): a consumer that lags
by `slots` or more steps and hands a stale ticket to result() gets a RuntimeError instead of a newer batch's (or a
half-overwritten) result.
"""
import torch

from .data import base_transform_u8

AHEAD = 2        # batches copied in ahead of the one being computed


class Ticket(int):
    """the slot a run() used (an int, usable as before) + the pipeline step it belongs to"""
    def __new__(cls, slot, step):
        t = int.__new__(cls, slot)
        t.step = step
        return t


class FrameStream(object):
    """slots x (pinned input, device input, captured step, device output, pinned output), used cyclically."""

    def __init__(self, engine, detect, priors, batch, frame_hw=(375, 500), mean=(104.0, 117.0, 123.0), scale=None, slots=3, calibrate=True, graph=True, zero_copy_out=False, copy_in="stream"):
        if slots < AHEAD + 1:
            raise ValueError("FrameStream needs at least %d slots (it copies %d batches ahead)" % (AHEAD + 1, AHEAD))
        # `engine`: one NetEngine, or a list of them (engine.clone(): own workspace and lanes, shared weights) = that many STEPS IN
        # FLIGHT: slot s runs on pipeline s % len(engines), pipeline 0 on the caller's stream, the others on streams of their
        # own, so the latency-bound tail of one step (small layers, deformable heads, Detect) runs under the next step's trunk
        engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
        NP = len(engines)
        if slots % NP:
            raise ValueError("FrameStream: %d slots cannot be dealt evenly to %d pipelines" % (slots, NP))
        import copy
        detects = [detect] + [copy.copy(detect) for _ in range(NP - 1)]
        for d in detects[1:]:
            d._ws = None                                 # (a Detect owns its device workspace: one per pipeline)
        dev = engines[0].device
        self.dev, self.B, self.slots, self.pipelines = dev, batch, slots, NP
        H0, W0 = frame_hw
        size = engines[0].cfg.size
        scale = scale if scale is not None else [float(W0), float(H0), float(W0), float(H0)]

        # zero_copy_out: Detect writes a slot's detections straight into the slot's PINNED host buffer (its rows are written once,
        # coalesced, never read back: layers/functions/detection.py `out=`), so there is no device-to-host copy, no copy-out stream and
        # no event chain behind it -- a D2H that waits for step k's end in a stream of its own blocks whatever shares its hardware
        # queue, the next H2D included (scripts/dev/stream_timeline.py).  result(slot) then waits for the step's own event.
        self.zero_copy_out = bool(zero_copy_out)
        # copy_in="own": batch k + 2's frames go to the device on the stream of the pipeline that will run them, right behind step k of
        # that pipeline (AHEAD is a multiple of the pipeline count) -- no copy-in stream, no events between it and the steps; the
        # pipeline pauses for the copy (0.35 ms at batch 32) while the other pipeline keeps the chip busy
        # copy_in="prio" (round 6): copy streams as with "stream", but created with HIGH PRIORITY and not calibrated -- HIP keeps the
        # hardware queues of a priority level apart, so the copies cannot sit behind a compute lane's kernels in a shared in-order queue
        # (the reason "stream" needs its calibration), and a stream that only carries DMA takes nothing from the compute lanes
        if copy_in not in ("stream", "own", "prio") or (copy_in == "own" and AHEAD % NP):
            raise ValueError("copy_in must be 'stream', 'prio' or 'own' (own: the pipeline count has to divide %d)" % AHEAD)
        self.copy_priority = copy_in == "prio"
        self.copy_in = "stream" if copy_in == "prio" else copy_in

        def make_step(eng, det):
            def one_step(u8, out=None):
                # the frame stays uint8 until the first conv's loader reads it: resize to uint8 planes, a quarter of the
                # fp32 tensor; the mean is subtracted inside the net (tdrn_net_io.reserved[3])
                x = base_transform_u8(u8, size, mean)
                r = eng.forward(x)
                return det.forward(r["odm_loc"], r["conf"], priors, arm_loc_data=r["arm_loc"], scale=scale, out=out)
            return one_step
        steps = [make_step(e, d) for e, d in zip(engines, detects)]
        self._fn = steps[0]
        self._steps = steps
        # graph=False: the step's launches are issued eagerly on the pipeline's stream (the reference's drivers launch eagerly too).  On
        # ROCm 7.2 two hipGraph replays on two streams hardly overlap (scripts/dev/inflight_timeline.py) while eagerly launched
        # pipelines do; the host has to keep up (~70 launches per step)
        self.graph = bool(graph)
        self._extra_streams = [torch.cuda.Stream(dev) for _ in range(NP - 1)]
        self._pipe_streams = None                         # eager mode: the pipelines' streams picked by calibration (_pick_pipeline_streams)
        self.host_in = [torch.empty((batch, H0, W0, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.dev_in = [torch.zeros((batch, H0, W0, 3), dtype=torch.uint8, device=dev) for _ in range(slots)]
        # lazily created resources (lanes, LDS attributes, workspaces) before any capture
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for st in steps:
                for _ in range(2):
                    probe = st(self.dev_in[0])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.host_out = [torch.empty(tuple(probe.shape), dtype=probe.dtype).pin_memory() for _ in range(slots)]
        self.graphs, self.dev_out = [], []
        for s in range(slots if self.graph else 0):
            g = torch.cuda.CUDAGraph()
            # one private pool per PIPELINE: a pipeline's steps run one after another on its stream, so their intermediates may
            # share memory (each slot's OUTPUT stays live and is never aliased); steps of different pipelines run concurrently
            with torch.cuda.graph(g, pool=self.graphs[s % NP].pool() if s >= NP else None):
                out = steps[s % NP](self.dev_in[s], out=self.host_out[s] if self.zero_copy_out else None)
            self.graphs.append(g)
            self.dev_out.append(out)
        if not self.graph:
            self.dev_out = [torch.empty_like(probe) for _ in range(slots)]     # (the step's result is copied here: 2.7 MB on the device)
        self._in_stream = torch.cuda.Stream(dev, priority=-1) if self.copy_priority else torch.cuda.Stream(dev)
        self._out_stream = torch.cuda.Stream(dev, priority=-1) if self.copy_priority else torch.cuda.Stream(dev)
        self.ev_in = [torch.cuda.Event() for _ in range(slots)]        # slot's frames are on the device
        self.ev_step = [torch.cuda.Event() for _ in range(slots)]      # slot's step has run (its device input is free again)
        self.ev_out = [torch.cuda.Event() for _ in range(slots)]       # slot's detections are on the host (its device output is free)
        cur = torch.cuda.current_stream(dev)
        for e in self.ev_in + self.ev_step + self.ev_out:
            e.record(cur)
        self._k = 0
        self._step_of = [-1] * slots                                   # the pipeline step whose result a slot holds (or will hold)
        self.calibration = None
        self.pipeline_calibration = None
        if calibrate:
            if not self.graph and NP == 2:
                self._pick_pipeline_streams()
            if self.copy_in != "own" and not self.copy_priority:     # (own: there are no copy streams to place; prio: placed by construction)
                self._pick_streams()

    def _pick_pipeline_streams(self, candidates=4, steps=6):
        """Eager mode, two pipelines: which hardware queues the two main lanes sit on decides how much of a step overlaps the next
        (engine.InFlight.pick_streams: +7 % between the worst and the best pairing); every ordered pair of a few candidate streams is
        timed over `steps` pipeline steps, the best kept."""
        import time
        cands = [torch.cuda.Stream(self.dev) for _ in range(candidates)]

        def timed(i, j, n):
            torch.cuda.synchronize(self.dev)
            self._pipe_streams = [cands[i], cands[j]]
            self.prime()
            for _ in range(2):
                self.run()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(n):
                self.run()
            torch.cuda.synchronize(self.dev)
            return (time.perf_counter() - t0) / n
        timings = {(i, j): timed(i, j, steps) for i in range(candidates) for j in range(candidates) if i != j}
        for k in sorted(timings, key=timings.get)[:3]:
            timings[k] = timed(k[0], k[1], 3 * steps)
        best = min(timings, key=timings.get)
        torch.cuda.synchronize(self.dev)
        self._pipe_streams = [cands[best[0]], cands[best[1]]]
        self.pipeline_calibration = {"ms_per_step": {"%d,%d" % k: round(v * 1e3, 3) for k, v in timings.items()}, "picked": "%d,%d" % best}
        self.prime()

    def _pick_streams(self, candidates=4, steps=10):
        """ROCm maps HIP streams onto a few in-order hardware queues (4 by default) in creation order, and HIP does not say
        which: a copy stream that shares the queue of the step's stream, or of a busy side lane of the net, has its event
        waits -- and with them the copy -- held behind that lane's kernels (0.85 of the resident rate), one that shares an
        idle lane's queue costs nothing (0.97); giving every stream a queue of its own (GPU_MAX_HW_QUEUES=8 / 16) is worse
        still (0.65 / 0.52: the copies' blit kernels then really do run beside the convolutions).  So: a few candidate streams,
        one short timed run of the pipeline per (copy-in, copy-out) pair, keep the best pair."""
        import time
        cands = [torch.cuda.Stream(self.dev) for _ in range(candidates)]

        def timed(i, j, n):
            self._in_stream, self._out_stream = cands[i], cands[j]
            self.prime()
            for _ in range(2):
                self.run()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(n):
                self.run()
            torch.cuda.synchronize(self.dev)
            return (time.perf_counter() - t0) / n
        timings = {(i, j): timed(i, j, steps) for i in range(candidates) for j in range(candidates)}
        # second pass over the four best pairs with three times the steps: the first pass only separates the bad pairs
        for k in sorted(timings, key=timings.get)[:4]:
            timings[k] = timed(k[0], k[1], 3 * steps)
        best = min(timings, key=timings.get)
        self._in_stream, self._out_stream = cands[best[0]], cands[best[1]]
        self.calibration = {"ms_per_step": {"%d,%d" % k: round(v * 1e3, 3) for k, v in timings.items()}, "picked": "%d,%d" % best}
        self.prime()

    def eager(self, frames_u8_dev):
        """the same step without capture, slots or copies (for the bit-identity check)"""
        return self._fn(frames_u8_dev)

    def pinned_in(self, slot):
        """the pinned (B,H,W,3) uint8 buffer a producer (decoder, camera) fills for `slot`; blocks until the last copy-in
        that read it has finished"""
        self.ev_in[slot].synchronize()
        return self.host_in[slot]

    def next_in(self):
        """the slot the producer has to fill before the next run(): the batch AHEAD of the one that run() will launch"""
        return (self._k + AHEAD) % self.slots

    def _pipe_stream(self, p):
        if self._pipe_streams is not None:
            return self._pipe_streams[p]
        return torch.cuda.current_stream(self.dev) if p == 0 else self._extra_streams[p - 1]

    def _copy_in(self, slot):
        if self.copy_in == "own":
            cur = self._pipe_stream(slot % self.pipelines)
            with torch.cuda.stream(cur):
                cur.wait_event(self.ev_out[slot])        # (stream order covers the step that last read the buffer)
                self.dev_in[slot].copy_(self.host_in[slot], non_blocking=True)
                self.ev_in[slot].record(cur)
            return
        with torch.cuda.stream(self._in_stream):
            self._in_stream.wait_event(self.ev_step[slot])      # the step that last read this device buffer has run
            self._in_stream.wait_event(self.ev_out[slot])       # ... and its detections have left the slot's output buffer, so
            self.dev_in[slot].copy_(self.host_in[slot], non_blocking=True)      # ev_in is all the slot's next step waits for
            self.ev_in[slot].record(self._in_stream)

    def prime(self, frames=None):
        """Before the first run(): the first AHEAD batches (already in pinned_in(0..AHEAD-1), or copied there from the list
        `frames`) go to the device."""
        torch.cuda.synchronize(self.dev)
        self._k = 0
        self._step_of = [-1] * self.slots
        for s in range(AHEAD):
            if frames is not None:
                self.host_in[s].copy_(frames[s])
            self._copy_in(s)

    def run(self):
        """Queue: the copy-in of the batch in pinned_in(next_in()), the step of the oldest batch not yet run, the copy-out of
        its detections.  Returns that step's slot; never blocks the host."""
        s = self._k % self.slots
        if self.copy_in != "own":
            self._copy_in((self._k + AHEAD) % self.slots)
        p = s % self.pipelines
        cur = self._pipe_stream(p)
        with torch.cuda.stream(cur):
            cur.wait_event(self.ev_in[s])
            if self.graph:
                self.graphs[s].replay()
            elif self.zero_copy_out:
                self._steps[p](self.dev_in[s], out=self.host_out[s])
            else:
                cur.wait_event(self.ev_out[s])           # (the slot's previous detections have left its output buffer)
                self.dev_out[s].copy_(self._steps[p](self.dev_in[s]))
            self.ev_step[s].record(cur)
            if self.zero_copy_out:
                self.ev_out[s].record(cur)               # (the detections are on the host when the step is done)
        if not self.zero_copy_out:
            if self.copy_in == "own":                    # ... and the detections leave on the pipeline's stream too: no copy stream at all
                with torch.cuda.stream(cur):
                    self.host_out[s].copy_(self.dev_out[s], non_blocking=True)
                    self.ev_out[s].record(cur)
            else:
                with torch.cuda.stream(self._out_stream):
                    self._out_stream.wait_event(self.ev_step[s])
                    self.host_out[s].copy_(self.dev_out[s], non_blocking=True)
                    self.ev_out[s].record(self._out_stream)
        if self.copy_in == "own":
            self._copy_in((self._k + AHEAD) % self.slots)     # behind this step, on this pipeline's stream
        self._step_of[s] = self._k
        self._k += 1
        return Ticket(s, self._k - 1)

    def result(self, slot):
        """Detections of the batch last run in `slot`, on the host (blocks until they have arrived).  Given the Ticket run()
        returned, raises if the slot has been re-issued since (the consumer lagged by `slots` or more steps: the buffer holds,
        or is being overwritten with, a newer batch); a plain int asks for whatever the slot holds."""
        step = getattr(slot, "step", None)
        if step is not None and self._step_of[int(slot)] != step:
            raise RuntimeError("FrameStream.result: slot %d was re-issued for step %d; the result of step %d is gone (read results "
                               "within %d steps, or use more slots)" % (int(slot), self._step_of[int(slot)], step, self.slots - 1))
        self.ev_out[int(slot)].synchronize()
        return self.host_out[int(slot)]

    def drain(self):
        torch.cuda.synchronize(self.dev)
