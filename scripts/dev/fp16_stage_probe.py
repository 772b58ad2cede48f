"""Where do the few multi-ulp fp16 outliers of the exact-input stage test come from?  Dumps the worst elements of conv2_1
(fp16, default plan, batch 2) and tests hypotheses about the matrix cores' fp16 arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_pin16 as T
from tdrn_amd.utils import synth
torch.set_num_threads(16)
dtype = "fp16"
net, sd = T._build(T.VGG[0], T.VGG[1], phase="train", dtype=dtype)
x = torch.from_numpy(synth.synth_frames(2, 320, seed=41)).to(T.DEV)
net(x); torch.cuda.synchronize()
eng = net._engine
ops = eng.op_infos()
op = [o for o in ops if o["w"] == "backbone.7"][0]
xin = eng.read_tensor(op["in"], 2).cpu()[0:1].double()
got = eng.read_tensor(op["out"], 2).cpu()[0:1].double()
wf, bf = T._fold(sd, op)
w16 = T._round16(wf, dtype)
y = F.conv2d(xin, w16, bf.double(), padding=1).clamp(min=0)
S = F.conv2d(xin.abs(), w16.abs(), bf.double().abs(), padding=1)
err = (got - y).abs()
ulp = T._ulp16(y.abs(), dtype)
ratio = err / (ulp + 2e-6 * S)
idx = torch.topk(ratio.flatten(), 12).indices
print("input: max %.3f, nonzero frac %.3f, subnormal-nonzero count %d, min nonzero %.3g" % (float(xin.max()), float((xin > 0).double().mean()),
      int(((xin > 0) & (xin < 2.0 ** -14)).sum()), float(xin[xin > 0].min())))
print("weights: subnormal count %d of %d" % (int(((w16.abs() > 0) & (w16.abs() < 2.0 ** -14)).sum()), w16.numel()))
for i in idx.tolist():
    c, yy, xx = np.unravel_index(i, y.shape[1:])
    print("c %3d y %3d x %3d  got %.6f ref %.6f err %.3g ulp %.3g S %.2f err/S %.3g pre-relu-ref %.6f" % (c, yy, xx, float(got[0, c, yy, xx]), float(y[0, c, yy, xx]),
          float(err[0, c, yy, xx]), float(ulp[0, c, yy, xx]), float(S[0, c, yy, xx]), float(err[0, c, yy, xx] / S[0, c, yy, xx]),
          float(F.conv2d(xin, w16, bf.double(), padding=1)[0, c, yy, xx])))
bad = ratio > 1
print("bad elements %d; per-channel histogram (top): %r" % (int(bad.sum()), torch.topk(bad[0].sum((1, 2)), 5)))
print("bad by row (top):", torch.topk(bad[0].sum((0, 2)), 5), "bad by col (top):", torch.topk(bad[0].sum((0, 1)), 5))
# H1: subnormal operands flushed
def flush(t):
    return t * (t.abs() >= 2.0 ** -14)
y1 = F.conv2d(flush(xin), flush(w16), bf.double(), padding=1).clamp(min=0)
print("H1 flushed-subnormal reference: bad %d" % int((((got - y1).abs()) / (T._ulp16(y1.abs(), dtype) + 2e-6 * S) > 1).sum()))
# H2: inputs above some magnitude? relation of bad outputs to large inputs in the window
xmax = F.max_pool2d(xin.abs().amax(1, keepdim=True), 3, 1, 1)
print("H2 window input max at bad outputs: %r ; overall quantiles %r" % (xmax.expand_as(bad)[bad][:10].tolist(), torch.quantile(xmax.flatten(), torch.tensor([0.5, 0.99, 1.0], dtype=torch.float64)).tolist()))
# H3: fp32 sequential accumulation in the kernel's K order is not the issue (bf16 passes); try products rounded to fp16?
# H4: is the OUTPUT an exact fp16 of something near? distance of got to the two nearest fp16 neighbours of ref
print("err in units of ulp at the worst:", (err.flatten()[idx] / ulp.flatten()[idx]).tolist())
