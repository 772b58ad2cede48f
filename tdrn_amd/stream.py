"""Streamed inference: frames arrive from the HOST, detections go back to the host.

The reference's drivers feed one frame at a time through the host (test_video.py:98-115: cv2 frame -> BaseTransform on
the CPU -> .cuda() -> net -> Detect -> .cpu(); evaluate.py:452-461 the same from the dataset).  Here a batch of uint8
BGR frames travels H2D as uint8 (4x less than the fp32 tensor the reference uploads), is resized / mean-subtracted on the
device (tdrn_preprocess), runs net + Detect, and only the (B, C, top_k, 5) detections travel back.  Two slots are in
flight: while slot k computes (one captured hipGraph per slot: preprocess + ~60 launches on 4 streams), slot k+1's frames
are being copied in on a copy stream and slot k-1's detections copied out on another -- so the feed costs the step
nothing as long as a copy is shorter than a step (18 MB over PCIe Gen5 is ~0.4 ms against ~3 ms).
"""
import torch

from .data import base_transform
from .engine import GraphedCall


class FrameStream(object):
    """slots x (pinned input, device input, captured step, device output, pinned output)."""

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
        self.s_in, self.s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        self.s_run = torch.cuda.current_stream(dev)
        self.host_in = [torch.empty((batch, H0, W0, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)]
        example = torch.zeros((batch, H0, W0, 3), dtype=torch.uint8, device=dev)
        self.steps = [GraphedCall(one_step, example) for _ in range(slots)]
        self.dev_in = [g.inputs[0] for g in self.steps]
        self.host_out = [torch.empty(tuple(g.outputs.shape), dtype=g.outputs.dtype).pin_memory() for g in self.steps]
        self.ev_in = [torch.cuda.Event() for _ in range(slots)]       # slot's frames are on the device
        self.ev_run = [torch.cuda.Event() for _ in range(slots)]      # slot's step is done (its input may be overwritten)
        self.ev_out = [torch.cuda.Event() for _ in range(slots)]      # slot's detections are on the host (its output may be overwritten)
        self._k = 0

    def eager(self, frames_u8_dev):
        """the same step without capture, slots or copies (for the bit-identity check)"""
        return self._fn(frames_u8_dev)

    def submit(self, frames_host=None):
        """Queue one batch: H2D of `frames_host` (a pinned (B,H,W,3) uint8 tensor; None = re-send the slot's own pinned
        buffer), the step, D2H of the detections.  Returns the slot; nothing here blocks the host."""
        s = self._k % self.slots
        src = frames_host if frames_host is not None else self.host_in[s]
        if self._k >= self.slots:
            self.s_in.wait_event(self.ev_run[s])          # the step that last read this input buffer
            self.s_run.wait_event(self.ev_out[s])         # the copy that last read this output buffer
        with torch.cuda.stream(self.s_in):
            self.dev_in[s].copy_(src, non_blocking=True)
            self.ev_in[s].record(self.s_in)
        self.s_run.wait_event(self.ev_in[s])
        self.steps[s].graph.replay()
        self.ev_run[s].record(self.s_run)
        self.s_out.wait_event(self.ev_run[s])
        with torch.cuda.stream(self.s_out):
            self.host_out[s].copy_(self.steps[s].outputs, non_blocking=True)
            self.ev_out[s].record(self.s_out)
        self._k += 1
        return s

    def result(self, slot):
        """Detections of the batch last submitted to `slot` (blocks until its D2H has finished)."""
        self.ev_out[slot].synchronize()
        return self.host_out[slot]

    def drain(self):
        torch.cuda.synchronize(self.dev)
