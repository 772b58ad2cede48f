"""Dual-refinement detector on VGG16(-BN): drop-in for model/dualrefinedet_vggbn.py of the
reference (RefineSSD.__init__ :10-117, forward :119-206, build_net :217-222).  Same
state_dict keys, same `net(x)` 4-tuple, same prior ordering; the arithmetic runs in
libtdrn_hip.so (MFMA implicit-GEMM convs, fused deformable heads)."""
import torch.nn as nn

from .. import _lib
from ..layers.modules.l2norm import L2Norm
from ._base import EngineModule
from .networks import ConvOffset2d, vgg, vgg_base


def _c3(cin, cout, **kw):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, **kw)


def add_refine_head(m, arm_channels, num_classes, def_groups, multihead, bias=True):
    """ARM loc / offset convs, TCB-FPN and deformable ODM heads shared by the VGG and MobileNet
    variants (dualrefinedet_vggbn.py:30-34,47-114; dualrefinedet_mobilenet.py:50-122)."""
    nb = 3
    m.last_layer_trans = nn.Sequential(_c3(arm_channels[3], 256, bias=bias), nn.ReLU(inplace=True),
                                       _c3(256, 256, bias=bias), _c3(256, 256, bias=bias))
    m.arm_loc = nn.ModuleList([_c3(c, nb * 4, bias=bias) for c in arm_channels])
    m.offset = nn.ModuleList([nn.Conv2d(nb * 4, def_groups * 18, 1, bias=bias) for _ in range(4)])
    dc = lambda cout, k: nn.ModuleList([ConvOffset2d(256, cout, k, 1, k // 2, num_deformable_groups=def_groups)
                                        for _ in range(4)])
    m.odm_loc, m.odm_conf = dc(nb * 4, 3), dc(nb * num_classes, 3)
    if multihead:
        m.offset2 = nn.ModuleList([nn.Conv2d(nb * 4, def_groups * 50, 1, bias=bias) for _ in range(4)])
        m.odm_loc_2, m.odm_conf_2 = dc(nb * 4, 5), dc(nb * num_classes, 5)
    m.trans_layers = nn.ModuleList([nn.Sequential(_c3(c, 256, bias=bias), nn.ReLU(inplace=True), _c3(256, 256, bias=bias))
                                    for c in arm_channels[:3]])
    m.up_layers = nn.ModuleList([nn.ConvTranspose2d(256, 256, 2, 2, 0, bias=bias) for _ in range(3)])
    m.latent_layers = nn.ModuleList([_c3(256, 256, bias=bias) for _ in range(3)])


class RefineSSD(EngineModule):
    def __init__(self, size, num_classes=21, phase='train', c7_channel=1024, def_groups=1, bn=True,
                 multihead=False, return_feature=False, device='cuda'):
        super(RefineSSD, self).__init__()
        self.num_classes, self.size, self.phase = num_classes, size, phase
        self.def_groups, self.bn, self.multihead = def_groups, bn, multihead
        self.return_feature, self.device = return_feature, device   # kwarg kept; its norm map was never returned
        self.backbone = nn.ModuleList(vgg(vgg_base['320'], 3, batch_norm=bn, pool5_ds=True, c7_channel=c7_channel))
        self.L2Norm_4_3 = L2Norm(512, 10)
        self.L2Norm_5_3 = L2Norm(512, 8)
        if bn:
            self.extras = nn.Sequential(nn.Conv2d(c7_channel, 256, 1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                                        nn.Conv2d(256, 512, 3, 2, 1), nn.BatchNorm2d(512), nn.ReLU(inplace=True))
        else:
            self.extras = nn.Sequential(nn.Conv2d(c7_channel, 256, 1), nn.ReLU(inplace=True),
                                        nn.Conv2d(256, 512, 3, 2, 1), nn.ReLU(inplace=True))
        add_refine_head(self, [512, 512, c7_channel, 512], num_classes, def_groups, multihead, bias=True)
        if phase == 'test':
            self.softmax = nn.Softmax(dim=1)
        self._engine_init(model=_lib.DRN_VGGBN, size=size, num_classes=num_classes, c7_channel=c7_channel,
                          def_groups=def_groups, bn=bn, multihead=multihead, test_phase=(phase == 'test'))

    def forward(self, x):
        r = self.engine_for(x).forward(x, want_offsets=True)
        conf = r["conf"] if self.phase == 'test' else r["conf"].view(x.size(0), -1, self.num_classes)
        return (r["arm_loc"], r["offsets"] if self.phase == 'test' else None, r["odm_loc"], conf)


def build_net(phase, size=320, num_classes=21, c7_channel=1024, def_groups=1, bn=True, multihead=False,
              return_feature=False):
    if size not in [320, 512]:
        print("Error: Sorry only SSD320 and SSD512 is supported currently!")
        return
    return RefineSSD(size, num_classes=num_classes, phase=phase, c7_channel=c7_channel, def_groups=def_groups,
                     bn=bn, multihead=multihead, return_feature=return_feature)
