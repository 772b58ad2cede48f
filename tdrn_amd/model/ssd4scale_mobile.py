"""4-scale SSD on MobileNet-v1: drop-in for model/ssd4scale_mobile.py (SSD4Scale_MobNet :9-84,
forward(x, ref_loc, offset_list, ret_loc, ret_off) :86-140, build_net :151-156).  With
deform=True it is the TRN temporal net (8 deformable groups, offsets from the static net's
loc maps)."""
import torch.nn as nn

from .. import _lib
from ..layers.modules.l2norm import L2Norm
from ._base import EngineModule
from .dualrefinedet_mobilenet import extras_block
from .networks import ConvOffset2d, mobilenet_backbone


class SSD4Scale_MobNet(EngineModule):
    def __init__(self, size, num_classes=21, phase='train', c7_channel=1024, deform=False):
        super(SSD4Scale_MobNet, self).__init__()
        self.num_classes, self.size, self.phase, self.deform = num_classes, size, phase, deform
        nb = 3
        self.backbone = mobilenet_backbone(c7_channel)
        self.L2Norm_4_3 = L2Norm(512, 10)
        self.L2Norm_5_3 = L2Norm(1024, 8)
        self.extras = nn.ModuleList([extras_block(c7_channel), extras_block(512)])
        chans = [512, c7_channel, 512, 512]
        if deform:
            g = 8
            self.offset = nn.ModuleList([nn.Conv2d(nb * 4, g * 18, 1) for _ in range(4)])
            mk = lambda cout: nn.ModuleList([ConvOffset2d(c, cout, 3, 1, 1, num_deformable_groups=g) for c in chans])
        else:
            mk = lambda cout: nn.ModuleList([nn.Conv2d(c, cout, 3, 1, 1) for c in chans])
        self.arm_loc, self.arm_conf = mk(nb * 4), mk(nb * num_classes)
        if phase == 'test':
            self.softmax = nn.Softmax(dim=1)
        self._engine_init(model=_lib.SSD4SCALE_MOBILE, size=size, num_classes=num_classes, c7_channel=c7_channel,
                          deform=deform, test_phase=(phase == 'test'))

    def forward(self, x, ref_loc=list(), offset_list=list(), ret_loc=False, ret_off=False, ref_event=None):
        if self.deform and offset_list:
            # the reference's precedence (ssd4scale_mobile.py:87-93): a non-empty offset_list wins over ref_loc.  Its offsets are a pure
            # function of the loc maps it was computed from, which it carries along: THOSE are the source of truth, whether or
            # not the engine's reuse token is still current (the token only decides whether they are recomputed or still there)
            cached = getattr(offset_list, "ref_loc", None)
            if cached is None:
                raise ValueError("offset_list was not returned by this net (it carries no ref_loc to recompute the offsets from)")
            ref_loc = cached
        elif self.deform and not ref_loc:
            raise ValueError("deform=True needs ref_loc (or an offset_list returned by this net)")
        # (the frames between two key frames hand back the offset_list of the key frame: its offsets are still in the engine's
        # workspace -- the token says so -- and are not recomputed, as in evaluate_trn.py:459-462)
        r = self.engine_for(x).forward(x, want_offsets=bool(ret_off and self.deform),
                                          ref_loc=ref_loc if self.deform else None,
                                          want_loc_maps=bool(ret_loc and not self.deform),
                                          reuse_offsets_token=getattr(offset_list, "token", None) if (self.deform and not ret_off) else None,
                                          ref_event=ref_event)
        conf = r["conf"] if self.phase == 'test' else r["conf"].view(x.size(0), -1, self.num_classes)
        out = [r["arm_loc"], conf]
        if ret_loc:
            out.append(r["loc_maps"])
        if ret_off:
            offs = _OffsetList(r["offsets"] or [])
            offs.ref_loc = list(ref_loc)
            offs.token = r["offsets_token"]
            out.append(offs)
        return tuple(out)


class _OffsetList(list):
    """arm_offset_list that remembers the loc maps it was computed from (and the engine forward that still holds them)."""
    ref_loc = None
    token = None


def build_net(phase, size=320, num_classes=21, c7_channel=1024, deform=False):
    if size not in [320, 512]:
        print("Error: Sorry only SSD320 and SSD512 is supported currently!")
        return
    return SSD4Scale_MobNet(size, num_classes=num_classes, phase=phase, c7_channel=c7_channel, deform=deform)
