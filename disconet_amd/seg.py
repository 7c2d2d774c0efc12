"""Segmentation variant of `--com disco` on the MI355X path (BASELINE.json
configs[3]: "DiscoNet seg head, 5-agent, 256x256 BEV").

Surface of upstream:coperception/models/seg/DiscoNet.py :: DiscoNet on SegModelBase (recollection:
the source is not in the mount -- /root/reference/coperception is an empty submodule directory; the
only mounted mention of the task is /root/reference/README.md:15):

    SegDiscoNet(n_channels=13, n_classes=8, num_agent=5, kd_flag=False, compress_level=0, only_v2i=False)
    forward(bevs [A*B, n_channels, H, W] (NCHW, as the reference's SegModule feeds it),
            trans_matrices [B, A, A, 4, 4], num_agent_tensor [B, A])
        -> logits [A*B, n_classes, H, W]      (kd_flag: + x9, x8, x7, x6, x5, fused x4)

a bilinear UNet (DoubleConv / Down / Up / OutConv under the reference's state_dict names) with the
DiscoGraph fusion of the det model at the 512-channel bottleneck.  Parameters live in ordinary
torch modules that are never called; the eval forward packs them once and runs

    3x3 convs (18) + outc     dn_spconv2d on split-planar activations (csrc/conv_sp.hip); the skip
                              concat of the Up blocks is the conv's two-source operand gather
    MaxPool2d(2)              dn_sp_maxpool2          (csrc/seg_ops.hip)
    Upsample x2 bilinear      dn_sp_upsample2_bilinear
    fusion at x4              dn_warp_neighbors + attention MLP + dn_disco_fuse_tail (C = 512)
    cross entropy (SegModule) dn_seg_ce_loss: value + d/d(logits)

There is no torch / CPU fallback.  Training: SegModule.step (seg_train.py: the detector's training engine + the
UNet's max-pool / bilinear-upsample backward kernels).
"""
import torch
import torch.nn as nn

from . import ops
from .model import _ConvLayer, _FusionParams
from .profiling import region


class _DoubleConv(nn.Module):
    def __init__(self, cin, cout, mid=None):
        super().__init__()
        mid = mid or cout
        self.double_conv = nn.Sequential(nn.Conv2d(cin, mid, 3, padding=1), nn.BatchNorm2d(mid), nn.ReLU(),
                                         nn.Conv2d(mid, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU())


class _Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _DoubleConv(cin, cout))


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = _DoubleConv(cin, cout, cin // 2)


class _OutConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)


class SegDiscoNet(nn.Module):
    FUSE_CHANNELS = 512

    def __init__(self, n_channels=13, n_classes=8, num_agent=5, kd_flag=False, compress_level=0,
                 only_v2i=False):
        super().__init__()
        if compress_level:
            raise NotImplementedError("seg variant: compress_level > 0 is not built")
        self.n_channels, self.n_classes = n_channels, n_classes
        self.agent_num, self.kd_flag, self.only_v2i = num_agent, kd_flag, only_v2i
        self.inc = _DoubleConv(n_channels, 64)
        self.down1, self.down2, self.down3, self.down4 = _Down(64, 128), _Down(128, 256), _Down(256, 512), _Down(512, 512)
        self.up1, self.up2, self.up3, self.up4 = _Up(1024, 256), _Up(512, 128), _Up(256, 64), _Up(128, 64)
        self.outc = _OutConv(64, n_classes)
        self.pixel_weighted_fusion = _FusionParams(self.FUSE_CHANNELS)
        self._plan, self._plan_sig = None, None

    def _replicate_for_data_parallel(self):
        # nn.DataParallel over ONE device never replicates (it calls self.module directly): the reference tools'
        # wrapper works unchanged there.  Over several devices it would clone this module per call and per thread --
        # replicas sharing one packed-weight plan and one stream: refused instead of undefined behaviour.
        raise RuntimeError(
            "disconet_amd: nn.DataParallel over more than one device is not supported (its per-call replicas would "
            "share one packed-weight plan and one HIP stream).  Keep nn.DataParallel(model, device_ids=[k]) for one "
            "device, or launch one process per GPU: python -m torch.distributed.run --nproc-per-node N ... "
            "(bench.py --gpus N, disconet_amd.sharded)")

    def load_state_dict(self, state_dict, strict=True, **kw):
        cleaned = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        self._plan = None
        return super().load_state_dict(cleaned, strict=strict, **kw)

    def train(self, mode=True):
        """train(): SegModule.step / SegTrainStep run the explicit HIP training graph (seg_train.py); the module's
        own forward() stays the eval plan -- a train()-mode forward() through autograd is not provided."""
        self._plan = None
        return super().train(mode)

    # ------------------------------------------------------------------
    def _get_plan(self):
        sig = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if self._plan is not None and sig == self._plan_sig:
            return self._plan
        P = {}

        def double(name, dc):
            seq = dc.double_conv
            P[name + "a"] = _ConvLayer(name + "a", seq[0].weight, seq[0].bias, seq[1], 3, math=2)
            P[name + "b"] = _ConvLayer(name + "b", seq[3].weight, seq[3].bias, seq[4], 3, math=2)

        double("inc", self.inc)
        for k in (1, 2, 3, 4):
            double("down%d" % k, getattr(self, "down%d" % k).maxpool_conv[1])
            double("up%d" % k, getattr(self, "up%d" % k).conv)
        P["outc"] = _ConvLayer("outc", self.outc.conv.weight, self.outc.conv.bias, None, 1, relu=False, math=2)
        # attention MLP at C = 512: layer 1 split W1 = [W_ego | W_nbr] on the NHWC engine + the tail kernel
        f, C = self.pixel_weighted_fusion, self.FUSE_CHANNELS
        w1 = f.conv1_1.weight.detach().reshape(128, 2 * C)
        dev = w1.device
        P["mlp_g"] = _ConvLayer("mlp_g", torch.cat([w1[:, :C], w1[:, C:]], 0).contiguous().reshape(256, C, 1, 1),
                                None, None, 1, relu=False, math=1,
                                scale_shift=(torch.ones(256, device=dev),
                                             torch.cat([f.conv1_1.bias.detach().float(),
                                                        torch.zeros(128, device=dev)]).contiguous()))
        P["mlp_f"] = _ConvLayer("mlp_f", w1[:, C:].contiguous().reshape(128, C, 1, 1), None, None, 1, relu=False,
                                math=1, scale_shift=(torch.ones(128, device=dev), torch.zeros(128, device=dev)))
        bn1_scale, bn1_shift = ops.fold_bn(None, f.bn1_1, 128)
        s2, t2 = ops.fold_bn(f.conv1_2.bias, f.bn1_2, 32)
        s3, t3 = ops.fold_bn(f.conv1_3.bias, f.bn1_3, 8)
        tail = {"bn1_scale": bn1_scale, "bn1_shift": bn1_shift,
                "w2": f.conv1_2.weight.detach().reshape(32, 128).float().contiguous(), "s2": s2, "t2": t2,
                "w3": f.conv1_3.weight.detach().reshape(8, 32).float().contiguous(), "s3": s3, "t3": t3,
                "w4": f.conv1_4.weight.detach().reshape(8).float().contiguous(),
                "b4": f.conv1_4.bias.detach().float().contiguous()}
        P["_tail_tensors"], P["_tail"] = tail, ops.make_tail_params(tail)
        self._plan, self._plan_sig = P, sig
        return P

    def fuse(self, x4, trans, num_agent, batch_size, P):
        """DiscoGraph fusion of the bottleneck maps: x4 SpTensor / NHWC [A*B, h, w, 512] -> NHWC"""
        A, B = self.agent_num, batch_size
        feat = ops.as_nhwc(x4)
        n, h, w, c = feat.shape
        pairs = B * A * (A - 1)
        map_bytes = 4.0 * h * w * c
        warped = torch.empty((B, A, max(A - 1, 0), h, w, c), dtype=torch.float32, device=feat.device)
        with region("warp", "warp_neighbors_kernel", 0.0, map_bytes * (n + pairs)):
            ops.warp_neighbors(feat, trans, num_agent, B, A, self.only_v2i, 0, A, out=warped)
        g = P["mlp_g"].run(feat)
        fw = P["mlp_f"].run(warped.view(pairs, h, w, c)) if A > 1 else None
        fused = torch.empty((A * B, h, w, c), dtype=torch.float32, device=feat.device)
        with region("fuse_tail", "disco_fuse_tail_kernel", 0.0, map_bytes * (2 * A * B + pairs)):
            return ops.disco_fuse_tail(feat, warped, g, fw, num_agent, P["_tail"], B, A, self.only_v2i,
                                       False, 0, A, out=fused)

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=None):
        if self.training:
            raise NotImplementedError("SegDiscoNet.forward in train() mode: use SegModule.step (disconet_amd/seg_train.py), "
                                      "the explicit HIP training step; forward() is the eval plan")
        if isinstance(bevs, ops.SpTensor):
            x, dev = bevs, bevs.device
        else:
            if not bevs.is_cuda:
                raise ops._lib.DnError("SegDiscoNet.forward needs GPU tensors; there is no CPU path")
            dev = bevs.device
            # [A*B, C, H, W] -> channels-last rows (a no-copy view when the caller permuted an NHWC
            # voxel batch, as the reference's SegModule does)
            x = bevs.permute(0, 2, 3, 1)
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.float().contiguous()
        A = self.agent_num
        n = x.shape[0]
        B = n // A if batch_size is None else batch_size
        if n != A * B:
            raise ValueError("bevs has %d images, expected num_agent*batch_size = %d" % (n, A * B))
        trans = trans_matrices.to(device=dev, dtype=torch.float32).contiguous()
        num_agent = num_agent_tensor[:, 0].to(device=dev, dtype=torch.int32).contiguous()
        P = self._get_plan()

        def double(name, src0, src1=None):
            return P[name + "b"].run(P[name + "a"].run(src0, src1))

        x1 = double("inc", x)
        x2 = double("down1", ops.sp_maxpool2(x1))
        x3 = double("down2", ops.sp_maxpool2(x2))
        x4 = double("down3", ops.sp_maxpool2(x3))
        fused = self.fuse(x4, trans, num_agent, B, P)
        x4f = ops.as_sp(fused)
        x5 = double("down4", ops.sp_maxpool2(x4f))
        # Up: cat([skip, upsampled], channel) -> DoubleConv: the concat is the conv's two-source gather
        x6 = double("up1", x4f, ops.sp_upsample2_bilinear(x5))
        x7 = double("up2", x3, ops.sp_upsample2_bilinear(x6))
        x8 = double("up3", x2, ops.sp_upsample2_bilinear(x7))
        x9 = double("up4", x1, ops.sp_upsample2_bilinear(x8))
        logits = P["outc"].run(x9).nhwc().permute(0, 3, 1, 2)         # NCHW-shaped view of NHWC rows
        if self.kd_flag:
            nchw = lambda t: ops.as_nhwc(t).permute(0, 3, 1, 2)
            return logits, nchw(x9), nchw(x8), nchw(x7), nchw(x6), nchw(x5), nchw(fused)
        return logits


class SegModule:
    """upstream:coperception/utils/SegModule.py :: SegModule, the evaluation half: forward + the
    per-pixel cross entropy on the HIP path (value and gradient w.r.t. the logits)."""

    def __init__(self, model, optimizer=None, lr=1e-3):
        self.model = model
        self._optimizer, self._lr, self._trainer = optimizer, lr, None

    # the training engine behind step() (seg_train.SegTrainEngine; None until the first step built it)
    engine = property(lambda self: self._trainer.engine if self._trainer is not None else None)

    def step(self, data, batch_size=None):
        """upstream SegModule.step: one training step (train-mode forward with batch statistics, cross entropy,
        explicit HIP reverse pass, Adam) -> {"loss": float}.  The training engine is built on first use (its flat
        parameter buffer re-points the module's Parameters: build after the model is on the GPU)."""
        if self._trainer is None:
            from .seg_train import SegTrainStep
            self._trainer = SegTrainStep(self.model, self._optimizer, self._lr)
        return self._trainer.step(data, batch_size)

    def loss(self, logits, labels, want_grad=True):
        """logits [N, classes, H, W] (the model's NCHW-shaped, channels-last view), labels [N, H, W]
        -> (loss float, dlogits NCHW-shaped or None)"""
        z = logits.permute(0, 2, 3, 1)
        if not z.is_contiguous():
            z = z.contiguous()
        loss, grad = ops.seg_ce_loss(z, labels, want_grad)
        return float(loss), (grad.permute(0, 3, 1, 2) if grad is not None else None)

    def evaluate(self, data, batch_size):
        with torch.no_grad():
            out = self.model(data["bev_seq"], data["trans_matrices"], data["num_agent"], batch_size)
        logits = out[0] if isinstance(out, tuple) else out
        loss, _ = self.loss(logits, data["labels"], want_grad=False)
        return {"loss": loss, "pred": logits.argmax(1)}
