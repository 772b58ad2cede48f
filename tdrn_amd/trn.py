"""Key-frame driver of the temporal refinement network (TRN): the per-frame protocol of
evaluate_trn.py:438-467 / test_video_trn.py:83-103 around the static and temporal SSD4Scale nets.

Every `interval` frames (or on a new video) the static net produces refined anchors (its loc output,
scaled by `loose`) and the raw loc maps; the temporal net turns the loc maps into deformable offsets
once and reuses them until the next key frame.  State per stream = (anchors, loc maps / offsets), so a
clip must stay on one rank (tdrn_amd.dist shards by clip)."""


class TRNDriver(object):
    def __init__(self, static_net, temporal_net, detector, priors, interval=4, loose=1.0, deform=True):
        self.static_net, self.net, self.detector, self.priors = static_net, temporal_net, detector, priors
        self.interval, self.loose, self.deform = int(interval), float(loose), bool(deform)
        self.reset()

    def reset(self):
        self.pre_video_name, self.current_i = None, 0
        self.static_out, self.ref_loc, self.offset_list = None, [], []
        self.key_frames = 0

    def step(self, x, video_name=None, scale=None):
        """x: (1,3,S,S) preprocessed frame.  Returns Detect output (1, C, top_k, 5)."""
        new_video = video_name != self.pre_video_name
        if self.static_out is None or new_video or self.current_i % self.interval == 0:
            self.static_out = list(self.static_net(x, ret_loc=self.deform))
            self.static_out[0] = self.static_out[0] * self.loose
            self.key_frames += 1
            if self.deform:
                self.ref_loc, self.offset_list = self.static_out[2], []
            if new_video:
                self.pre_video_name, self.current_i = video_name, 0
        need_off = self.deform and not self.offset_list
        out = self.net(x, ref_loc=self.ref_loc, offset_list=self.offset_list, ret_off=need_off)
        if len(out) == 3:
            self.offset_list, self.ref_loc = out[2], []
        dets = self.detector.forward(out[0], out[1], self.priors, arm_loc_data=self.static_out[0], scale=scale)
        self.current_i += 1
        return dets

    def clips(self, frames, scale=None, side_stream=None):
        """Offline evaluation of whole intervals: `frames` (F, Bk, 3, S, S) = the F <= interval consecutive frames of Bk clips,
        FRAME-major ([0] = the key frames).  One static forward over the Bk key frames, ONE temporal forward over all F * Bk frames
        (tdrn_net_io.reserved[1]: frame i reads the offsets of key frame i % Bk) and one Detect call on the static anchors --
        the same detections, bit for bit, as `step` frame by frame (within an interval a frame depends on the key frame only
        through the cached offsets and anchors, evaluate_trn.py:452-462), at the throughput of one batch of F * Bk frames.
        `side_stream` (a torch.cuda.Stream): the static net runs THERE, beside the temporal net's trunk, and the temporal forward
        waits for its event right before its first read of the loc maps (tdrn_net_io.reserved[2]) -- the static net's launches over
        a few key frames leave most of the chip idle.  (Inside a stream capture the static net needs its one-stream plan,
        `set_plan_flags(PLAN_ONE_STREAM)`: see tests/test_gpu_net.py::test_trn_static_net_beside_the_temporal_trunk.)
        Returns the Detect output (F * Bk, C, top_k, 5), frame-major; the per-stream state of `step` is not touched."""
        if frames.dim() != 5 or frames.size(0) > self.interval:
            raise ValueError("frames must be (F <= interval, clips, 3, S, S), got %r" % (tuple(frames.shape),))
        F, Bk = int(frames.size(0)), int(frames.size(1))
        allf = frames.reshape(F * Bk, *frames.shape[2:])
        if side_stream is not None and self.deform:
            import torch
            main = torch.cuda.current_stream(frames.device)
            if getattr(self, "_maps_ready", None) is None:
                self._maps_ready = torch.cuda.Event()          # (one event for the driver's life: never destroyed inside a capture)
            side_stream.wait_stream(main)
            with torch.cuda.stream(side_stream):
                static_out = list(self.static_net(frames[0], ret_loc=True))
                anchors = static_out[0] * self.loose
                self._maps_ready.record(side_stream)
            out = self.net(allf, ref_loc=static_out[2], ref_event=self._maps_ready)
            main.wait_stream(side_stream)
            for t in [anchors] + list(static_out[2]):
                t.record_stream(main)
        else:
            static_out = list(self.static_net(frames[0], ret_loc=self.deform))
            anchors = static_out[0] * self.loose
            out = self.net(allf, ref_loc=static_out[2]) if self.deform else self.net(allf)
        self.key_frames += Bk
        return self.detector.forward(out[0], out[1], self.priors, arm_loc_data=anchors.repeat(F, 1, 1), scale=scale)
