"""Streamed inference: frames arrive from the HOST, detections go back to the host.

The reference's drivers feed one frame at a time through the host (test_video.py:98-115: cv2 frame -> BaseTransform on
the CPU -> .cuda() -> net -> Detect -> .cpu(); evaluate.py:452-461 the same from the dataset).  Here a batch of uint8
BGR frames travels H2D as uint8 (4x less than the fp32 tensor the reference uploads), is resized / mean-subtracted on the
device (tdrn_preprocess), runs net + Detect, and only the (B, C, top_k, 5) detections travel back.

One captured hipGraph per slot holds the WHOLE turn of the pipeline:
    graph[s] = {  H2D of slot s+1's pinned frames -> its device buffer      (a branch of its own: the copy engine)
               || tdrn_preprocess(slot s) -> net (~60 launches, 4 streams) -> Detect -> D2H of slot s's detections }
so the copy-in of the NEXT batch runs under THIS batch's convolutions with the dependencies inside the graph.  (First
version: copies on separate HIP streams chained to the step graphs by events -- measured with rocprofv3's memory-copy
trace: the copy-in started 2.6 ms into a 3.1-ms step, whatever the host order, more streams or more slots, and the next
step waited for it: 85-91 % of the resident rate.)
Protocol: write batch k+1 into `pinned_in(next slot)` BEFORE `run()` launches batch k; `result(slot)` is batch k's output.
"""
import torch

from .data import base_transform


class FrameStream(object):
    """slots x (pinned input, device input, captured turn, device output, pinned output), used cyclically."""

    def __init__(self, engine, detect, priors, batch, frame_hw=(375, 500), mean=(104.0, 117.0, 123.0), scale=None, slots=2):
        dev = engine.device
        self.dev, self.B, self.slots = dev, batch, slots
        H0, W0 = frame_hw
        size = engine.cfg.size
        scale = scale if scale is not None else [float(W0), float(H0), float(W0), float(H0)]

        def one_step(u8):
            x = base_transform(u8, size, mean)
            r = engine.forward(x)
            return detect.forward(r["odm_loc"], r["conf"], priors, arm_loc_data=r["arm_loc"], scale=scale)
        self._fn = one_step
        self.host_in = [torch.empty((batch, H0, W0, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.dev_in = [torch.zeros((batch, H0, W0, 3), dtype=torch.uint8, device=dev) for _ in range(slots)]
        # lazily created resources (lanes, LDS attributes, workspaces) before any capture
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                probe = one_step(self.dev_in[0])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.host_out = [torch.empty(tuple(probe.shape), dtype=probe.dtype).pin_memory() for _ in range(slots)]
        self.graphs, self.dev_out = [], []
        self._copy_stream = torch.cuda.Stream(dev)
        for s in range(slots):
            nxt = (s + 1) % slots
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cur = torch.cuda.current_stream(dev)
                self._copy_stream.wait_stream(cur)                      # fork: the next slot's frames
                with torch.cuda.stream(self._copy_stream):
                    self.dev_in[nxt].copy_(self.host_in[nxt], non_blocking=True)
                out = one_step(self.dev_in[s])
                self.host_out[s].copy_(out, non_blocking=True)
                cur.wait_stream(self._copy_stream)                      # join
            self.graphs.append(g)
            self.dev_out.append(out)
        self.ev_done = [torch.cuda.Event() for _ in range(slots)]
        self._k = 0

    def eager(self, frames_u8_dev):
        """the same step without capture, slots or copies (for the bit-identity check)"""
        return self._fn(frames_u8_dev)

    def pinned_in(self, slot):
        """the pinned (B,H,W,3) uint8 buffer a producer (decoder, camera) fills for `slot`"""
        return self.host_in[slot]

    def prime(self, frames=None):
        """Before the first run(): slot 0's frames (already in pinned_in(0), or copied there from `frames`) go to the device."""
        if frames is not None:
            self.host_in[0].copy_(frames)
        self.dev_in[0].copy_(self.host_in[0], non_blocking=True)
        self._k = 0

    def run(self):
        """Launch the turn of the current slot (its frames were copied in by the previous turn, or by prime()); the NEXT
        slot's pinned buffer must already hold the next batch.  Returns the slot; never blocks the host."""
        s = self._k % self.slots
        self.graphs[s].replay()
        self.ev_done[s].record(torch.cuda.current_stream(self.dev))
        self._k += 1
        return s

    def result(self, slot):
        """Detections of the batch last run in `slot`, on the host (blocks until its turn has finished)."""
        self.ev_done[slot].synchronize()
        return self.host_out[slot]

    def drain(self):
        torch.cuda.synchronize(self.dev)
