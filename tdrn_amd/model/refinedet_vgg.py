"""RefineDet on VGG16 (plain, non-deformable ODM heads): drop-in for model/refinedet_vgg.py
(RefineSSD :27-110, forward :112-219, build_net :230-235).  The TRN drivers' 'FPN' branch and
BASELINE config #5 name this model."""
import torch.nn as nn

from .. import _lib
from ..layers.modules.l2norm import L2Norm
from ._base import EngineModule
from .networks import vgg, vgg_base


def _c(cin, cout, k):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2)


class RefineSSD(EngineModule):
    def __init__(self, size, num_classes=21, use_refine=False, phase='train', c7_channel=1024, bn=False, multihead=False):
        super(RefineSSD, self).__init__()
        self.num_classes, self.size, self.use_refine, self.phase = num_classes, size, use_refine, phase
        self.bn, self.multihead = bn, multihead
        nb = 3
        self.backbone = nn.ModuleList(vgg(vgg_base['320'], 3, batch_norm=bn, pool5_ds=True, c7_channel=c7_channel))
        self.L2Norm_4_3 = L2Norm(512, 10)
        self.L2Norm_5_3 = L2Norm(512, 8)
        self.last_layer_trans = nn.Sequential(_c(512, 256, 3), nn.ReLU(inplace=True), _c(256, 256, 3), _c(256, 256, 3))
        if bn:
            self.extras = nn.Sequential(nn.Conv2d(c7_channel, 256, 1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                                        nn.Conv2d(256, 512, 3, 2, 1), nn.BatchNorm2d(512), nn.ReLU(inplace=True))
        else:
            self.extras = nn.Sequential(nn.Conv2d(c7_channel, 256, 1), nn.ReLU(inplace=True),
                                        nn.Conv2d(256, 512, 3, 2, 1), nn.ReLU(inplace=True))
        chans = [512, 512, c7_channel, 512]
        if use_refine:
            self.arm_loc = nn.ModuleList([_c(c, nb * 4, 3) for c in chans])
        self.odm_loc = nn.ModuleList([_c(256, nb * 4, 3) for _ in range(4)])
        self.odm_conf = nn.ModuleList([_c(256, nb * num_classes, 3) for _ in range(4)])
        if multihead:
            self.odm_loc_2 = nn.ModuleList([_c(256, nb * 4, 5) for _ in range(4)])
            self.odm_conf_2 = nn.ModuleList([_c(256, nb * num_classes, 5) for _ in range(4)])
        self.trans_layers = nn.ModuleList([nn.Sequential(_c(c, 256, 3), nn.ReLU(inplace=True), _c(256, 256, 3)) for c in chans[:3]])
        self.up_layers = nn.ModuleList([nn.ConvTranspose2d(256, 256, 2, 2, 0) for _ in range(3)])
        self.latent_layers = nn.ModuleList([_c(256, 256, 3) for _ in range(3)])
        if phase == 'test':
            self.softmax = nn.Softmax(dim=1)
        self._engine_init(model=_lib.REFINEDET_VGG, size=size, num_classes=num_classes, c7_channel=c7_channel, bn=bn,
                          multihead=multihead, use_refine=use_refine, test_phase=(phase == 'test'))

    def forward(self, x):
        r = self.engine_for(x).forward(x)
        conf = r["conf"] if self.phase == 'test' else r["conf"].view(x.size(0), -1, self.num_classes)
        if self.use_refine:
            return (r["arm_loc"], None, r["odm_loc"], conf)
        return (r["odm_loc"], conf)


def build_net(phase, size=320, num_classes=21, use_refine=False, c7_channel=1024, bn=False, multihead=False):
    if size not in [320, 512]:
        print("Error: Sorry only SSD300 and SSD512 is supported currently!")
        return
    return RefineSSD(size, num_classes=num_classes, use_refine=use_refine, phase=phase, c7_channel=c7_channel, bn=bn,
                     multihead=multihead)
