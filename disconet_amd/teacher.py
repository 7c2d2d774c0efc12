"""Distillation teacher on the HIP path.

Surface of upstream:coperception/models/det/TeacherNet.py :: TeacherNet (built by
`train_codet.py --kd_flag 1 --resume_teacher <ckpt>`, /root/reference/README.md:58-59):

    TeacherNet(config)
    forward(bevs [N, 1, H, W, Z]) -> (x_8, x_7, x_6, x_5, x_3, x_2)     NCHW-shaped views

the early-fusion teacher: holistic-view voxels through the MotionNet backbone (encoder +
decoder of one `stpn` module), no communication.  The same conv engine and packed-weight plan
as DiscoNet, minus the fusion block; inference only (the teacher is frozen during KD).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .model import _DEC_CONVS, _ENC_CONVS, _ClsHeadParams, _Conv3DParams, _ConvLayer, _RegHeadParams, _bn_name


class _BackboneParams(nn.Module):
    """upstream Backbone.py :: STPN_KD parameter names (one module holds encoder and decoder)"""

    def __init__(self, in_channels):
        super().__init__()
        for name, cin, cout, stride in _ENC_CONVS:
            setattr(self, name, nn.Conv2d(cin or in_channels, cout, 3, stride, 1))
            setattr(self, _bn_name(name), nn.BatchNorm2d(cout))
        self.conv3d_1 = _Conv3DParams(64)
        self.conv3d_2 = _Conv3DParams(128)
        for name, cin, cout in _DEC_CONVS:
            setattr(self, name, nn.Conv2d(cin, cout, 3, 1, 1))
            setattr(self, _bn_name(name), nn.BatchNorm2d(cout))


class TeacherNet(nn.Module):
    def __init__(self, config, in_channels=13):
        super().__init__()
        self.stpn = _BackboneParams(in_channels)
        self.classification = _ClsHeadParams(config)
        self.regression = _RegHeadParams(config, 1 if config.only_det else config.pred_len)
        self.conv_math = os.environ.get("DISCONET_CONV_MATH", "sp")
        self._plan, self._plan_sig = None, None

    def load_state_dict(self, state_dict, strict=True, **kw):
        cleaned = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        self._plan = None
        return super().load_state_dict(cleaned, strict=strict, **kw)

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("TeacherNet is the frozen distillation teacher: eval() only")
        return super().train(mode)

    def _get_plan(self):
        sig = (self.conv_math,) + tuple((t.data_ptr(), t._version) for t in
                                        list(self.parameters()) + list(self.buffers()))
        if self._plan is None or sig != self._plan_sig:
            math = ops.MATH_MODES[self.conv_math]
            s, P = self.stpn, {}
            for name, _, _, stride in _ENC_CONVS:
                conv = getattr(s, name)
                P[name] = _ConvLayer(name, conv.weight, conv.bias, getattr(s, _bn_name(name)), 3, stride, math=math)
            for name in ("conv3d_1", "conv3d_2"):
                m = getattr(s, name)
                P[name] = _ConvLayer(name, m.conv3d.weight, m.conv3d.bias, m.bn3d, 1, math=math)
            for name, _, _ in _DEC_CONVS:
                conv = getattr(s, name)
                P[name] = _ConvLayer(name, conv.weight, conv.bias, getattr(s, _bn_name(name)), 3, math=math)
            self._plan, self._plan_sig = P, sig
        return self._plan

    def forward_nhwc(self, bevs):
        """-> (x8, x7, x6, x5, x3, x2) as dense NHWC tensors (what the KD kernel reads)"""
        if self.training:
            raise NotImplementedError("TeacherNet: eval() only")
        if not bevs.is_cuda:
            raise ops._lib.DnError("TeacherNet.forward needs GPU tensors; there is no CPU path")
        P = self._get_plan()
        n = bevs.shape[0] * bevs.shape[1]
        x = bevs.reshape(n, bevs.shape[2], bevs.shape[3], bevs.shape[4])
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        x0 = P["conv_pre_2"].run(P["conv_pre_1"].run(x))
        x1 = P["conv3d_1"].run(P["conv1_2"].run(P["conv1_1"].run(x0)))
        x2 = P["conv3d_2"].run(P["conv2_2"].run(P["conv2_1"].run(x1)))
        x3 = P["conv3_2"].run(P["conv3_1"].run(x2))
        x4 = P["conv4_2"].run(P["conv4_1"].run(x3))
        x5 = P["conv5_2"].run(P["conv5_1"].run(x4, x3, up0=True))
        x6 = P["conv6_2"].run(P["conv6_1"].run(x5, x2, up0=True))
        x7 = P["conv7_2"].run(P["conv7_1"].run(x6, x1, up0=True))
        x8 = P["conv8_2"].run(P["conv8_1"].run(x7, x0, up0=True))
        return tuple(ops.as_nhwc(t) for t in (x8, x7, x6, x5, x3, x2))

    def forward(self, bevs):
        with torch.no_grad():
            return tuple(t.permute(0, 3, 1, 2) for t in self.forward_nhwc(bevs))
