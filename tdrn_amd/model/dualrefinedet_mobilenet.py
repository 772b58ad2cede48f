"""Dual-refinement detector on MobileNet-v1: drop-in for model/dualrefinedet_mobilenet.py
(RefineSSD :8-125, forward :127-199, build_net :210-214).  All head / TCB convs are bias-free."""
import torch.nn as nn

from .. import _lib
from ..layers.modules.l2norm import L2Norm
from ._base import EngineModule
from .dualrefinedet_vggbn import add_refine_head
from .networks import conv_dw, mobilenet_backbone


def extras_block(cin):
    return nn.Sequential(nn.Conv2d(cin, 256, 1), nn.BatchNorm2d(256), nn.ReLU(inplace=True), conv_dw(256, 512, 2))


class RefineSSD(EngineModule):
    def __init__(self, size, num_classes=21, phase='train', def_groups=1, multihead=False):
        super(RefineSSD, self).__init__()
        self.num_classes, self.size, self.phase = num_classes, size, phase
        self.def_groups, self.multihead = def_groups, multihead
        self.backbone = mobilenet_backbone(1024)
        self.L2Norm_4_3 = L2Norm(512, 20)
        self.L2Norm_5_3 = L2Norm(1024, 8)
        self.extras = nn.ModuleList([extras_block(1024), extras_block(512)])
        add_refine_head(self, [512, 1024, 512, 512], num_classes, def_groups, multihead, bias=False)
        if phase == 'test':
            self.softmax = nn.Softmax(dim=1)
        self._engine_init(model=_lib.DRN_MOBILENET, size=size, num_classes=num_classes, def_groups=def_groups,
                          multihead=multihead, test_phase=(phase == 'test'))

    def forward(self, x):
        r = self.engine_for(x).forward(x)
        conf = r["conf"] if self.phase == 'test' else r["conf"].view(x.size(0), -1, self.num_classes)
        return (r["arm_loc"], None, r["odm_loc"], conf)


def build_net(phase, size=320, num_classes=21, def_groups=1, multihead=False):
    if size not in [320, 512]:
        print("Error: Sorry only SSD320 and SSD512 is supported currently!")
        return
    return RefineSSD(size, num_classes=num_classes, phase=phase, def_groups=def_groups, multihead=multihead)
