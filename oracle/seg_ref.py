"""CPU oracle of the segmentation variant of `--com disco` (SURVEY.md §2.1 #9, §8(f) #4;
BASELINE.json configs[3]).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.

PARITY UNPINNED.  /root/reference holds no source (empty `coperception` submodule,
/root/reference/.gitmodules:1-3); the only mounted evidence of this task is
/root/reference/README.md:15 ("New tasks including segmentation ... are included") and :37, :49.
What follows restates, from recollection, upstream:coperception/models/seg/{SegModelBase,DiscoNet}.py
and upstream:coperception/utils/SegModule.py:

  * SegModelBase = the classic bilinear UNet: inc = DoubleConv(n_channels, 64); down1..down4 =
    MaxPool2d(2) + DoubleConv to 128 / 256 / 512 / 512 (1024 // 2); up1..up4 = Upsample(x2, bilinear,
    align_corners=True) of the deeper map, cat([skip, upsampled], channel), DoubleConv(in, out, in // 2)
    to 256 / 128 / 64 / 64; outc = Conv2d(64, n_classes, 1).  DoubleConv = (Conv3x3 pad 1 -> BatchNorm
    -> ReLU) twice.
  * seg DiscoNet: the DiscoGraph fusion of the det model (same two-pass pose warp, same
    PixelWeightedFusionSoftmax on cat[ego, neighbour], exp / sum softmax over the agents, weighted
    sum), applied to x4 = down3's output (512 channels at H/8 x W/8 = 32 x 32) before down4.
  * forward(bevs [A*B, H, W, 13] permuted to NCHW by the caller, trans_matrices [B, A, A, 4, 4],
    num_agent_tensor [B, A]) -> logits [A*B, n_classes, H, W]; kd_flag == 1 additionally returns
    (x9, x8, x7, x6, x5, fused x4) for the distillation loss.
  * SegModule.step: per-pixel cross entropy over n_classes = 8 (mean over pixels), Adam.

The warp / attention pieces are imported from oracle/disconet_ref.py (same upstream base code).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .disconet_ref import PixelWeightedFusionSoftmax, feature_transformation, kaiming_reinit, randomize_bn_stats

N_CLASSES = 8


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid_channels = mid_channels or out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1), nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class SegDiscoNetRef(nn.Module):
    """upstream:coperception/models/seg/DiscoNet.py :: DiscoNet on SegModelBase"""

    def __init__(self, n_channels=13, n_classes=N_CLASSES, num_agent=5, kd_flag=False, compress_level=0,
                 only_v2i=False):
        super().__init__()
        self.n_channels, self.n_classes = n_channels, n_classes
        self.agent_num, self.kd_flag, self.only_v2i = num_agent, kd_flag, only_v2i
        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 512)
        self.up1 = Up(1024, 256)
        self.up2 = Up(512, 128)
        self.up3 = Up(256, 64)
        self.up4 = Up(128, 64)
        self.outc = OutConv(64, n_classes)
        self.pixel_weighted_fusion = PixelWeightedFusionSoftmax(512)

    def build_local_communication_matrix(self, feat_maps, batch_size):
        return torch.cat([feat_maps[batch_size * i: batch_size * (i + 1)].unsqueeze(1)
                          for i in range(self.agent_num)], 1)           # [B, A, C, H, W]

    def agents_to_batch(self, feats):
        return torch.cat([feats[:, i] for i in range(self.agent_num)], 0)

    def fuse(self, x4, trans_matrices, num_agent_tensor, batch_size):
        size = (1,) + tuple(x4.shape[1:])
        com = self.build_local_communication_matrix(x4, batch_size)
        out = com.clone()
        for b in range(batch_size):
            n = int(num_agent_tensor[b, 0])
            for i in range(n):
                ego = com[b, i]
                nbrs = [ego]
                for j in range(n):
                    if j != i and not (self.only_v2i and i != 0 and j != 0):
                        nbrs.append(feature_transformation(b, j, com, trans_matrices[b, i], size))
                e = [torch.exp(torch.squeeze(self.pixel_weighted_fusion(torch.cat([ego, nb], 0).unsqueeze(0))))
                     for nb in nbrs]
                total = 0
                for ek in e:
                    total = total + ek
                acc = 0
                for ek, nb in zip(e, nbrs):
                    acc = acc + torch.div(ek, total) * nb
                out[b, i] = acc
        return self.agents_to_batch(out)

    def forward(self, x, trans_matrices, num_agent_tensor, batch_size=None):
        """x: [A*B, n_channels, H, W] (NCHW, what the reference's SegModule feeds after its permute)"""
        batch_size = x.shape[0] // self.agent_num if batch_size is None else batch_size
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x4 = self.fuse(x4, trans_matrices, num_agent_tensor, batch_size)
        x5 = self.down4(x4)
        x6 = self.up1(x5, x4)
        x7 = self.up2(x6, x3)
        x8 = self.up3(x7, x2)
        x9 = self.up4(x8, x1)
        logits = self.outc(x9)
        if self.kd_flag:
            return logits, x9, x8, x7, x6, x5, x4
        return logits


def seg_loss(logits, labels):
    """upstream SegModule: nn.CrossEntropyLoss() on [N, classes, H, W] logits and [N, H, W] int labels"""
    return F.cross_entropy(logits, labels.long())


def seg_train_loss(logits, labels, bev):
    """upstream SegModule.step with `com` on: pred / labels of the images whose BEV is empty (torch.sum(bev[i]) <= 1e-4:
    padded agent slots) are dropped, image by image, before the criterion"""
    keep = [i for i in range(bev.shape[0]) if float(torch.sum(bev[i])) > 1e-4]
    if len(keep) != bev.shape[0]:
        idx = torch.tensor(keep, dtype=torch.long)
        logits, labels = logits[idx], labels[idx]
    return seg_loss(logits, labels)


def build_seg_ref(seed=0, init="kaiming", **kw):
    torch.manual_seed(seed)
    m = SegDiscoNetRef(**kw)
    if init == "kaiming":
        kaiming_reinit(m)
    randomize_bn_stats(m)
    return m.eval()
