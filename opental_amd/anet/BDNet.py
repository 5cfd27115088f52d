"""OpenTAL / AFSD ActivityNet1.3 detector on MI355X (configs/anet_opental.yaml, BASELINE config 4).

Same constructor, forward signature, output dict and state-dict keys as the reference's AFSD/anet/BDNet.py
(CoarsePyramid :120-391, BDNet :394-501).  What differs from the THUMOS14 model (opental_amd/thumos14/BDNet.py, whose
HIP-backed pieces are reused unchanged):

  * 768-frame clips; ONE projection -- Unit3D [1,3,3] on Mixed_5c -> (B,512,96) -- followed by five stride-2 levels
    (96,48,24,12,6,3: 189 anchors), anet/BDNet.py:130-155,:284-290;
  * `loc = exp(scale_i * conv) * fpn_stride_i`, i.e. in frames (:307-311);
  * priors are (189,2): centre and level id (:262-269), the latter drives the per-level regression bounds of the
    ActivityNet MultiSegmentLoss;
  * reset_params re-draws the tower / head / proposal-branch Conv1d weights from N(0, 0.01) (:435-451).
"""
import torch
import torch.nn as nn

from ..prop_pooling.boundary_pooling_op import BoundaryMaxPooling
from ..thumos14 import BDNet as _thumos
from ..thumos14.BDNet import DirichletLayer, I3D_BackBone, ProposalBranch, ScaleExp  # noqa: F401  (reference names)

layer_num = 6
conv_channels = 512
fpn_strides = [4, 8, 16, 32, 64, 128]
DEFAULT_MODEL_CFG = dict(num_classes=151, freeze_bn=True, freeze_bn_affine=True, evidence='exp', os_head=True,
                         backbone_model='models/i3d_models/rgb_imagenet.pt')


def model_cfg_from(config=None):
    """The keys anet/BDNet.py:11-18 reads from the global config at import time."""
    cfg = dict(DEFAULT_MODEL_CFG)
    if config is not None:
        cfg['num_classes'] = config['dataset']['num_classes']
        for k in ('freeze_bn', 'freeze_bn_affine', 'evidence', 'os_head', 'backbone_model'):
            if k in config['model']:
                cfg[k] = config['model'][k]
    return cfg


class CoarsePyramid(_thumos.CoarsePyramid):
    def __init__(self, feat_channels=(832, 1024), num_cls=2, frame_num=768, os_head=True):
        super(CoarsePyramid, self).__init__([feat_channels[1]], num_cls, frame_num=frame_num, os_head=os_head,
                                            projections=(('Mixed_5c', [1, 3, 3]),), first_level_t=frame_num // 8,
                                            fpn_strides=tuple(fpn_strides))

    def forward(self, feat_dict, ssl=False):
        outs = super(CoarsePyramid, self).forward(feat_dict, ssl=ssl)
        return outs if ssl else outs[:14] + outs[16:]   # the reference returns 14 tensors (anet/BDNet.py:384-391) [+ the extras dict]


class BDNet(_thumos.BDNet):
    def __init__(self, in_channels=3, backbone_model=None, training=True, frame_num=768, use_edl=False, cfg=None):
        nn.Module.__init__(self)
        if cfg is None:
            try:
                from ..common import config as _c
                cfg = model_cfg_from(_c._config) if _c._config is not None else dict(DEFAULT_MODEL_CFG)
            except Exception:
                cfg = dict(DEFAULT_MODEL_CFG)
        self.cfg = cfg
        self.os_head = cfg['os_head']
        self.num_classes = cfg['num_classes'] - 1 if self.os_head else cfg['num_classes']
        self.coarse_pyramid_detection = CoarsePyramid(frame_num=frame_num, num_cls=self.num_classes, os_head=self.os_head)
        self.reset_params()
        self.boundary_max_pooling = BoundaryMaxPooling()
        self.backbone = I3D_BackBone(in_channels=in_channels, freeze_bn=cfg['freeze_bn'],
                                     freeze_bn_affine=cfg['freeze_bn_affine'])
        self._training = training
        if self._training:
            self.backbone.load_pretrained_weight(cfg['backbone_model'] if backbone_model is None else backbone_model)
        self.scales = [1, 4, 4]
        self.use_edl = use_edl
        self.evidence = cfg['evidence']
        if self.use_edl:
            self.out_layer = DirichletLayer(self.evidence, dim=-1)
        self.coarse_pyramid_detection.dirichlet_exp = bool(self.use_edl and self.evidence == 'exp')
        self.use_rpl = False

    @staticmethod
    def weight_init(m):
        _thumos.BDNet.weight_init(m)
        if isinstance(m, nn.GroupNorm):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)

    def reset_params(self):
        for m in self.modules():
            self.weight_init(m)
        cpd = self.coarse_pyramid_detection
        for modules in [cpd.loc_tower, cpd.conf_tower, cpd.loc_head, cpd.conf_head, cpd.loc_proposal_branch,
                        cpd.conf_proposal_branch, cpd.prop_loc_head, cpd.prop_conf_head, cpd.center_head]:
            for layer in modules.modules():
                if isinstance(layer, nn.Conv1d):
                    torch.nn.init.normal_(layer.weight, mean=0, std=0.01)
                    torch.nn.init.constant_(layer.bias, 0)

    def forward(self, x, proposals=None, ssl=False):
        return super(BDNet, self).forward(x, proposals=proposals, ssl=ssl)
