"""torch-CPU restatement of the distillation teacher and the KD loss term.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no source / vectors in /root/reference).

Follows, from recollection (SURVEY.md §2 row 6, §3 call stack, §8(f) next #2):
  upstream:coperception/models/det/TeacherNet.py :: TeacherNet -- the early-fusion teacher:
      the holistic-view voxels (all agents' points in each agent's frame) through the same
      MotionNet backbone (upstream Backbone.py :: STPN_KD), NO communication; forward returns
      the decoder pyramid and two encoder maps (x_8, x_7, x_6, x_5, x_3, x_2).
  upstream:coperception/utils/CoDetModule.py :: get_kd_loss -- for x_5, x_6, x_7 and the
      student's fused layer-3 map against the teacher's x_3:
      KLDivLoss(size_average=True, reduce=True)(log_softmax(student_rows, 1), softmax(teacher_rows, 1))
      on [N*H*W, C] rows, summed and multiplied by kd_weight (default 1e5, SURVEY Appx A.1);
      enabled by `--kd_flag 1 --resume_teacher ...` (/root/reference/README.md:58-59).
Parameter names (`stpn.*`) are recalled, not verified.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .disconet_ref import Backbone, ClassificationHead, SingleRegressionHead


class STPN_KD(Backbone):
    """one Backbone instance doing encode + decode, returning the KD pyramid"""

    def forward(self, x):
        x0, x1, x2, x3, x4 = self.encode(x)
        x8, x7, x6, x5 = self.decode(x0, x1, x2, x3, x4, x.shape[0], kd_flag=True)
        return x8, x7, x6, x5, x3, x2


class TeacherNetRef(nn.Module):
    def __init__(self, config, in_channels=13):
        super().__init__()
        self.stpn = STPN_KD(in_channels)
        self.classification = ClassificationHead(config)
        self.regression = SingleRegressionHead(config)

    def forward(self, bevs):
        bevs = bevs.permute(0, 1, 4, 2, 3)          # (N, seq, z, h, w)
        return self.stpn(bevs)


def kd_loss(student, teacher, kd_weight):
    """student = (x5, x6, x7, fused), teacher = (x5, x6, x7, x3): NCHW maps -> scalar"""
    total = 0.0
    for s, t in zip(student, teacher):
        c = s.shape[1]
        s_rows = s.permute(0, 2, 3, 1).reshape(-1, c)
        t_rows = t.permute(0, 2, 3, 1).reshape(-1, c)
        total = total + F.kl_div(F.log_softmax(s_rows, dim=1), F.softmax(t_rows, dim=1), reduction="mean")
    return kd_weight * total


def build_teacher(config, seed=3, init="kaiming"):
    from .disconet_ref import kaiming_reinit, randomize_bn_stats
    torch.manual_seed(seed)
    t = TeacherNetRef(config)
    if init == "kaiming":
        kaiming_reinit(t, seed + 1)
    randomize_bn_stats(t, seed + 2)
    return t.eval()
