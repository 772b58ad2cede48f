"""NetEngine -- thin host-side owner of one tdrn_net handle (include/tdrn_hip.h section iii).

PyTorch is plumbing here: it allocates the weight blob, the activation workspace and the output
tensors on the GPU and supplies the HIP stream; every kernel is launched by libtdrn_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import NetConfig, NetIO, KernelStat, check, ptr


class NetEngine(object):
    def __init__(self, model, size, num_classes=21, c7_channel=1024, def_groups=1, bn=True,
                 multihead=False, deform=False, test_phase=True, dtype="fp32", use_refine=False, plan_flags=0):
        self.lib = _lib.lib()
        self.dtype_name = dtype
        self._ctor = dict(model=model, size=size, num_classes=num_classes, c7_channel=c7_channel, def_groups=def_groups, bn=bn,
                          multihead=multihead, deform=deform, test_phase=test_phase, dtype=dtype, use_refine=use_refine, plan_flags=plan_flags)
        cfg = NetConfig(model=model, size=size, num_classes=num_classes, c7_channel=c7_channel,
                        def_groups=def_groups, bn=int(bool(bn)), multihead=int(bool(multihead)),
                        deform=int(bool(deform)), test_phase=int(bool(test_phase)),
                        dtype=_lib.DTYPES[dtype], use_refine=int(bool(use_refine)), plan_flags=int(plan_flags))
        self.cfg = cfg
        h = C.c_void_p()
        check(self.lib.tdrn_net_create(C.byref(cfg), C.byref(h)), "tdrn_net_create")
        self.handle = h
        self.num_classes = num_classes
        self.num_priors = self.lib.tdrn_net_num_priors(h)
        self.fm = [size // 8, size // 16, size // 32, size // 64]
        self.weights = None          # device blob (torch.uint8)
        self._ws = None
        self.device = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tdrn_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- parameters -----------------------------------------------------------------------
    def param_specs(self):
        out = []
        for i in range(self.lib.tdrn_net_param_count(self.handle)):
            name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            check(self.lib.tdrn_net_param_info(self.handle, i, C.byref(name), C.byref(shape), C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def load(self, state_dict, device):
        """state_dict: {name: tensor/ndarray} in the reference's key layout.  Entries the plan does
        not use (num_batches_tracked) are ignored; a missing or mis-shaped one raises."""
        for name, shape in self.param_specs():
            if name not in state_dict:
                raise KeyError("state_dict is missing %r" % name)
            v = state_dict[name]
            if isinstance(v, torch.Tensor):
                v = v.detach().to("cpu", torch.float32).contiguous().numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(v.shape) != shape:
                raise ValueError("size mismatch for %s: expected %r, got %r" % (name, shape, tuple(v.shape)))
            check(self.lib.tdrn_net_set_param(self.handle, name.encode(), v.ctypes.data_as(C.c_void_p), v.size), name)
        self._alloc_weights(device)
        check(self.lib.tdrn_net_pack_weights(self.handle, ptr(self.weights), self.weights.numel(),
                                             _lib.current_stream(self.device)), "pack_weights")

    def _alloc_weights(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise NotImplementedError("tdrn_amd runs on the GPU only (no CPU path)")
        n = self.lib.tdrn_net_weight_bytes(self.handle)
        self.weights = torch.zeros(n, dtype=torch.uint8, device=self.device)

    def share_weights(self, other):
        """Use another engine's packed blob (same model/dtype): several engines -- e.g. one per HIP stream --
        can run concurrently on one GPU from a single copy of the weights."""
        if other.weights is None:
            raise RuntimeError("the other engine has no packed weights")
        if self.lib.tdrn_net_weight_bytes(self.handle) != other.weights.numel():
            raise ValueError("engines differ in model or dtype")
        self.weights, self.device = other.weights, other.device
        check(self.lib.tdrn_net_adopt_weights(self.handle))

    def clone(self):
        """A second handle of the same plan over the SAME packed blob, with its own workspace and side lanes: what a second
        step in flight runs on (InFlight below, FrameStream with several engines)."""
        e = NetEngine(**self._ctor)
        e.share_weights(self)
        return e

    def broadcast_weights(self, src=0):
        """One RCCL broadcast of the packed blob over xGMI; every other rank adopts it (no
        per-frame collective follows).  Call after load() on rank `src`, instead of load() elsewhere."""
        import torch.distributed as dist
        if self.weights is None:
            self._alloc_weights(torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(self.weights, src=src)
        check(self.lib.tdrn_net_adopt_weights(self.handle))

    # ---- forward --------------------------------------------------------------------------
    def workspace(self, batch):
        need = self.lib.tdrn_net_workspace_bytes(self.handle, batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def write_tensor(self, index, values):
        """analysis only: fp32 (B,C,H,W) values -> tensor `index` of the workspace, rounded once to the net dtype"""
        v = values.to(self.device, torch.float32).contiguous()
        check(self.lib.tdrn_net_write_tensor(self.handle, ptr(self._ws), v.size(0), index, ptr(v), _lib.current_stream(self.device)))

    def forward(self, x, want_offsets=False, ref_loc=None, want_loc_maps=False, out=None, reuse_offsets_token=None, first_op=0,
                ref_event=None):
        u8 = None
        if hasattr(x, "planes"):                        # tdrn_amd.data.U8Frames: the batch stays uint8 (tdrn_net_io.reserved[3])
            u8, x = x, x.planes
        _lib.require_cuda(x, "input")
        if self.weights is None:
            raise RuntimeError("weights were never packed (load() / broadcast_weights())")
        if x.dim() != 4 or x.size(1) != 3 or x.size(2) != self.cfg.size or x.size(3) != self.cfg.size:
            raise ValueError("expected input (B,3,%d,%d), got %r" % (self.cfg.size, self.cfg.size, tuple(x.shape)))
        x = x.contiguous() if u8 is not None else x.contiguous().float()
        B, P, Cn = x.size(0), self.num_priors, self.num_classes
        dev = x.device
        ws = self.workspace(B)
        ssd = self.cfg.model in (_lib.SSD4SCALE_MOBILE, _lib.SSD4SCALE_VGG)
        o = out or {}
        arm_loc = o.get("arm_loc")
        has_arm = not (self.cfg.model == _lib.REFINEDET_VGG and not self.cfg.use_refine)
        if arm_loc is None and has_arm:
            arm_loc = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
        odm_loc = None
        if not ssd:
            odm_loc = o.get("odm_loc")
            if odm_loc is None:
                odm_loc = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
        conf = o.get("conf")
        if conf is None:
            conf = torch.empty((B * P, Cn), dtype=torch.float32, device=dev)
        io = NetIO()
        if u8 is None:
            io.x = x.data_ptr()
        else:
            frames = _lib.U8FramesABI(planes=x.data_ptr(), mean=(C.c_float * 3)(*u8.mean))
            io.reserved[3] = C.addressof(frames)        # (read during the call below; `frames` lives until this function returns)
        io.batch = B
        io.arm_loc = arm_loc.data_ptr() if arm_loc is not None else None
        io.odm_loc = odm_loc.data_ptr() if odm_loc is not None else None
        io.conf = conf.data_ptr()
        offsets, loc_maps, keep = None, None, [x]
        # key-frame broadcast (TRN clips, tdrn_net_io.reserved[1]): ref_loc maps of Bk key frames for a batch of B = n * Bk frames in
        # frame-major order -- frame i uses the offsets of key frame i % Bk
        Bk = B
        if ref_loc is not None and len(ref_loc) and ref_loc[0].size(0) != B:
            Bk = int(ref_loc[0].size(0))
            if not self.cfg.deform or Bk < 1 or B % Bk:
                raise ValueError("ref_loc holds %d key frames for a batch of %d frames (needs a deform=True ssd4scale net and B %% Bk == 0)" % (Bk, B))
        # TRN temporal nets: the offsets computed by an earlier forward of THIS engine in the same workspace at the same batch are
        # reused when the caller presents the token that forward handed out (tdrn_net_io.reserved[0])
        reuse = (reuse_offsets_token is not None and reuse_offsets_token is getattr(self, "_offs_token", None)
                 and self._offs_key[:2] == (ws.data_ptr(), B))
        if reuse:
            io.reserved[0] = 1
            Bk = self._offs_key[2]
            ref_loc = None
        elif self.cfg.deform and ref_loc is not None:
            self._offs_token, self._offs_key = object(), (ws.data_ptr(), B, Bk)
        if Bk != B:
            io.reserved[1] = Bk
        if want_offsets:
            # (sized AFTER the reuse case has resolved Bk: a reused key-frame forward fills Bk samples, not B)
            g18 = (8 if self.cfg.deform else self.cfg.def_groups) * 18
            offsets = [torch.empty((Bk if self.cfg.deform else B, g18, f, f), dtype=torch.float32, device=dev) for f in self.fm]
            for i, t in enumerate(offsets):
                io.offsets[i] = t.data_ptr()
        if ref_loc is not None:
            for i, t in enumerate(ref_loc):
                t = t.contiguous().float()
                keep.append(t)
                io.ref_loc[i] = t.data_ptr()
            if ref_event is not None:
                # a torch.cuda.Event recorded on the stream that is still producing `ref_loc` (tdrn_net_io.reserved[2])
                io.reserved[2] = int(ref_event.cuda_event)
        if want_loc_maps:
            loc_maps = [torch.empty((B, 12, f, f), dtype=torch.float32, device=dev) for f in self.fm]
            for i, t in enumerate(loc_maps):
                io.loc_maps[i] = t.data_ptr()
        if first_op:
            check(self.lib.tdrn_net_forward_from(self.handle, ptr(self.weights), ptr(ws), ws.numel(), C.byref(io), int(first_op),
                                                 _lib.current_stream(dev)), "tdrn_net_forward_from")
        else:
            check(self.lib.tdrn_net_forward(self.handle, ptr(self.weights), ptr(ws), ws.numel(), C.byref(io),
                                            _lib.current_stream(dev)), "tdrn_net_forward")
        res = {"arm_loc": arm_loc, "odm_loc": odm_loc, "conf": conf, "offsets": offsets, "loc_maps": loc_maps,
               "offsets_token": getattr(self, "_offs_token", None)}
        return res

    def check(self):
        """tdrn_net_check: raises TdrnError(TDRN_E_DEVICE) when a forward enqueued since the last check reported a device-side
        hand-off failure (bounded poll ran out).  Does not synchronise: call it behind the synchronisation that ends the
        forward (or the hipGraph replay) in question."""
        detail = C.c_uint(0)
        rc = self.lib.tdrn_net_check(self.handle, C.byref(detail))
        check(rc, "device status 0x%x (1: chained split of conv3x3_pp, 2: chain launch)" % detail.value)

    # ---- debug / test access to internal activations ------------------------------------------
    def tensor_infos(self):
        out = []
        for i in range(self.lib.tdrn_net_tensor_count(self.handle)):
            lab, c, h, w = C.c_char_p(), C.c_int(), C.c_int(), C.c_int()
            check(self.lib.tdrn_net_tensor_info(self.handle, i, C.byref(lab), C.byref(c), C.byref(h), C.byref(w)))
            out.append((lab.value.decode(), c.value, h.value, w.value))   # '' = not materialised (fused away)
        return out

    def op_infos(self):
        """The plan's ops (tdrn_net_op_info) as dicts; tensor fields are indices into tensor_infos()."""
        out = []
        for i in range(self.lib.tdrn_net_op_count(self.handle)):
            oi = _lib.OpInfo()
            check(self.lib.tdrn_net_op_info(self.handle, i, C.byref(oi)))
            d = {f[0]: getattr(oi, f[0]) for f in _lib.OpInfo._fields_}
            d["in"] = d.pop("in_")
            d["kind"] = _lib.OP_KINDS[d["kind"]]
            d["off_c0"] = list(d["off_c0"])
            for k in ("w", "b", "bn", "w2", "b2"):
                d[k] = d[k].decode()
            out.append(d)
        return out

    def read_tensor(self, index, batch):
        lab, c, h, w = self.tensor_infos()[index]
        out = torch.empty((batch, c, h, w), dtype=torch.float32, device=self.device)
        check(self.lib.tdrn_net_read_tensor(self.handle, ptr(self._ws), batch, index, ptr(out),
                                            _lib.current_stream(self.device)))
        return out

    # ---- accounting for bench.py -------------------------------------------------------------
    def set_profile(self, mode):
        """0 / False: off; 1 / True: per-launch events, single stream; 2: per-launch events on the production lanes."""
        check(self.lib.tdrn_net_profile(self.handle, int(mode)))

    def kernel_stats(self):
        arr = (KernelStat * 16)()
        n = self.lib.tdrn_net_kernel_stats(self.handle, arr, 16)
        if n < 0:
            check(n)
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, flops=arr[i].flops,
                     bytes=arr[i].bytes, ms=arr[i].ms) for i in range(n)]

    def op_stats(self):
        arr = (KernelStat * 128)()
        n = self.lib.tdrn_net_op_stats(self.handle, arr, 128)
        if n < 0:
            check(n)
        return [dict(name=arr[i].name.decode(), flops=arr[i].flops, bytes=arr[i].bytes, ms=arr[i].ms) for i in range(n)]

    def op_timeline(self):
        """op_stats() plus start / end (ms after the first launch's start) and stream lane of every launch"""
        import ctypes as C
        ops = self.op_stats()
        st, en, ln = (C.c_float * 128)(), (C.c_float * 128)(), (C.c_int * 128)()
        n = self.lib.tdrn_net_op_timeline(self.handle, st, en, ln, 128)
        if n < 0:
            check(n)
        for i in range(min(n, len(ops))):
            ops[i].update(start=st[i], end=en[i], lane=ln[i])
        return ops


class GraphedCall(object):
    """One hipGraph per (callable, input shapes): `fn(*inputs)` -- e.g. net forward + Detect of one batch size -- is
    captured once and replayed as a single graph launch.  Nothing in the hot calls of libtdrn_hip.so allocates or
    synchronises (include/tdrn_hip.h), the side-lane fork/join of tdrn_net_forward is made of events, so the whole
    step, ~55 launches on 4 streams, captures as is.  Inputs are copied into the captured input buffers before each
    replay (or pass the tensors returned by `.inputs` and fill them in place); the outputs are the captured output
    tensors, overwritten by every replay -- a caller that keeps a result across steps asks for `clone_outputs=True`
    (or copies it itself, on the stream that replays).
    What capture freezes: every host-side value `fn` reads.  Detect.forward with a CPU tensor or a list as `scale`
    turns it into kernel arguments at capture, so a later change of that value is NOT seen by the replays; pass
    per-call values as device tensors among the inputs.  Shapes and dtypes are checked on every call."""

    def __init__(self, fn, *example_inputs, warmup=2, clone_outputs=False):
        self.clone_outputs = clone_outputs
        self.inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream(self.inputs[0].device)
        side.wait_stream(torch.cuda.current_stream(self.inputs[0].device))
        with torch.cuda.stream(side):                      # lazily created resources (lanes, LDS attributes, workspaces)
            for _ in range(max(1, warmup)):
                fn(*self.inputs)
        torch.cuda.current_stream(self.inputs[0].device).wait_stream(side)
        torch.cuda.synchronize(self.inputs[0].device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(*self.inputs)

    def __call__(self, *inputs):
        if len(inputs) != len(self.inputs):
            raise ValueError("GraphedCall captured %d inputs, called with %d" % (len(self.inputs), len(inputs)))
        for i, (dst, src) in enumerate(zip(self.inputs, inputs)):
            if src.shape != dst.shape or src.dtype != dst.dtype or src.device != dst.device:
                raise ValueError("GraphedCall input %d: captured %s %s on %s, got %s %s on %s (one graph per shape: build another)"
                                 % (i, tuple(dst.shape), dst.dtype, dst.device, tuple(src.shape), src.dtype, src.device))
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if not self.clone_outputs:
            return self.outputs
        return _clone_tree(self.outputs)


def _clone_tree(o):
    if isinstance(o, torch.Tensor):
        return o.clone()
    if isinstance(o, dict):
        return {k: _clone_tree(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_clone_tree(v) for v in o)
    return o


def _record_tree(o, stream):
    if isinstance(o, torch.Tensor):
        if o.is_cuda:
            o.record_stream(stream)
    elif isinstance(o, dict):
        for v in o.values():
            _record_tree(v, stream)
    elif isinstance(o, (list, tuple)):
        for v in o:
            _record_tree(v, stream)


class InFlight(object):
    """N whole steps in flight (round 5; the default schedule of bench.py and FrameStream is N = 2).

    graph=True: one hipGraph replay per step on the pipeline's stream.  graph=False: the step's launches are issued eagerly on it -- on
    ROCm 7.2 two replays on two streams hardly overlap (step k + 1 starts when step k ends: scripts/dev/inflight_timeline.py) while eager
    pipelines do; call pick_streams() once to put the pipelines' main lanes on hardware queues that do not serialise them (+8 % frames/s
    over the replays at batch 32; bench.py --launch auto times both and runs the faster).

    A step -- forward + Detect of one batch -- is a ~2-ms trunk of convolutions that fill the chip, followed by ~1 ms in which ~40
    small dependent launches, the deformable heads and Detect leave most CUs idle.  With two pipelines (each its own engine handle,
    workspace, HIP stream and hipGraph; ONE packed weight blob) the tail of step k runs under the trunk of step k + 1:
    +3.5...4.5 % frames/s at batch 32, every pipeline's results bit-identical to the same step run alone (soak: 10 000 two-pipeline
    and 9 000 three-pipeline replays, profiles/r05_attribution/soak*.txt).  Resident batch j is captured on pipeline j % N, so
    n_batches should be a multiple of N.

        def make_step(eng):                     # called ONCE PER PIPELINE: everything a step writes besides the engine's own
            detect = Detect(21, 0, cfg, ...)    # workspace must be created in here -- a Detect owns ONE device scratch (_ws), so a
            return lambda x: detect(eng.forward(x), priors)   # Detect shared by two pipelines would be written by two steps at once
        fl = InFlight(make_step, engine, batches, n=2)
        fl.launch(k)            # step k: replays batch k % len(batches) on its pipeline's stream; never blocks the host
        fl.output(j)            # output of batch j: valid after fl.sync()

    Reading an output from ANOTHER stream without fl.sync(): graph=True outputs are static buffers, an event wait on fl.stream_of(j)
    is enough.  graph=False outputs are fresh allocations of pipeline p's stream and the next launch of batch j returns the previous
    ones to that stream's allocator pool -- a consumer stream that merely waited on an event could read recycled memory: ask for them
    with fl.output(j, consumer=stream), which records the stream on every tensor of the output (torch keeps the blocks until that
    stream's work behind the call has finished).
    """

    def __init__(self, make_step, engine, batches, n=2, graph=True, steps=None, engines=None):
        # (steps / engines: pipelines built by the caller -- one step callable per pipeline and the engines to check() -- for steps that
        # are more than one engine's forward, e.g. the TRN clips' static + temporal nets: bench.py --config 5)
        dev = engine.device
        self.n = max(1, int(n)) if steps is None else len(steps)
        self.engines = ([engine] + [engine.clone() for _ in range(self.n - 1)]) if steps is None else list(engines or [engine])
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.n)]
        self.steps = [make_step(e) for e in self.engines] if steps is None else list(steps)
        self.batches = list(batches)
        self.graph = bool(graph)
        self.calls = []
        self.dev = dev
        for j, xb in enumerate(self.batches):
            p = j % self.n
            if self.graph:
                with torch.cuda.stream(self.streams[p]):
                    self.calls.append(GraphedCall(self.steps[p], xb))
                torch.cuda.synchronize(dev)
            else:
                self.calls.append(None)
        self._eager_out = [None] * len(self.batches)
        self.stream_calibration = None

    def pick_streams(self, candidates=4, steps=6):
        """Eager pipelines only: choose the HIP streams the pipelines launch on by a short calibration.

        ROCm maps HIP streams onto a few in-order hardware queues (4 by default) in creation order and does not say which.  Two
        pipelines whose main lanes share a queue run one after the other whatever the streams say (a queue is a FIFO: step k + 1's
        first kernel sits behind every launch of step k's main lane); main lanes that share queues with the OTHER pipeline's side
        lanes lose less; the best pairing measured +7 % over the worst (profiles/r05_experiments.md "Hardware queues").  So: a few
        candidate streams (consecutive streams of torch's pool sit on consecutive queues), every ordered pair timed over `steps`
        steps, the best pair kept.  ~16 x steps steps on whatever the resident batches hold; results are unaffected (the same
        launches in the same per-pipeline order).  Returns {"i,j": ms per step} and keeps the best pair."""
        import itertools
        import os
        import time
        if self.graph or self.n < 2 or self.n > candidates:      # (replays launched on picked streams were tried: 11.3k frames/s whatever the pair)
            return None
        # (round 6, measured and rejected as deterministic replacements of this calibration -- profiles/r06_experiments.md: the pipelines on
        # streams of different PRIORITY 10.5k frames/s, on dedicated hardware queues (hipExtStreamCreateWithCUMask, all CUs) 10.6k, on
        # the streams InFlight creates 11.2k, calibrated 12.4-12.5k: what the good pairs have in common is not "different queues")
        if os.environ.get("TDRN_INFLIGHT_PICK") == "none":
            return None
        cands = [torch.cuda.Stream(self.dev) for _ in range(candidates)]
        nb = len(self.batches)

        def timed(sel, n):
            torch.cuda.synchronize(self.dev)
            self.streams = [cands[i] for i in sel]
            for k in range(self.n):
                self.launch(k)
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for k in range(n):
                self.launch(k % nb)
            torch.cuda.synchronize(self.dev)
            return (time.perf_counter() - t0) / n
        timings = {sel: timed(sel, steps) for sel in itertools.permutations(range(candidates), self.n)}
        for key in sorted(timings, key=timings.get)[:3]:
            timings[key] = timed(key, 3 * steps)
        best = min(timings, key=timings.get)
        torch.cuda.synchronize(self.dev)
        self.streams = [cands[i] for i in best]
        self.stream_calibration = {"ms_per_step": {",".join(map(str, k)): round(v * 1e3, 3) for k, v in timings.items()}, "picked": ",".join(map(str, best))}
        return self.stream_calibration

    def launch(self, k):
        j = k % len(self.batches)
        p = j % self.n
        with torch.cuda.stream(self.streams[p]):
            if self.graph:
                self.calls[j].graph.replay()
            else:
                self._eager_out[j] = self.steps[p](self.batches[j])
        return j

    def stream_of(self, j):
        return self.streams[j % self.n]

    def output(self, j, consumer=None):
        out = self.calls[j].outputs if self.graph else self._eager_out[j]
        if consumer is not None and not self.graph:
            _record_tree(out, consumer)
        return out

    def sync(self):
        torch.cuda.synchronize(self.dev)

    def check(self):
        for e in self.engines:
            e.check()
