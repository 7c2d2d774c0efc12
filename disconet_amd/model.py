"""`--com disco` detector with the reference's class surface, computed by the
gfx950 kernels of libdisconet_hip.so.

Surface kept (upstream:coperception/models/det/DiscoNet.py,
selected by `--com disco` at /root/reference/README.md:56,70):

    DiscoNet(config, layer=3, in_channels=13, kd_flag=True, num_agent=5,
             compress_level=0, only_v2i=False)
    forward(bevs [A*B,1,H,W,Z] f32 agent-major, trans_matrices [B,A,A,4,4] f32,
            num_agent_tensor [B,A] int, batch_size=1)
        -> {"loc": [A*B,H,W,6,1,6], "cls": [A*B,H*W*6,2]}
        or (result, x8, x7, x6, x5, fused) when kd_flag == 1

Parameters live in ordinary torch modules under the reference's names
(u_encoder.conv1_1.weight, decoder.bn5_1.running_mean,
pixel_weighted_fusion.conv1_1.weight, classification.conv2.bias,
regression.box_prediction.3.weight, ...; SURVEY.md Appx A.3) so a reference
checkpoint's model_state_dict loads; those modules are never *called*.  The
forward path packs them once into the kernels' tile-major layouts and runs
NHWC (channels-last) end to end.  There is no torch/CPU fallback: tensors must
be on the GPU and the shared object must be built.
"""
import os
import re

import torch
import torch.nn as nn

from . import ops
from .profiling import region

LAYER_CHANNEL = {4: 512, 3: 256, 2: 128, 1: 64, 0: 32}

_ENC_CONVS = [  # name, c_in, c_out, stride
    ("conv_pre_1", None, 32, 1), ("conv_pre_2", 32, 32, 1),
    ("conv1_1", 32, 64, 2), ("conv1_2", 64, 64, 1),
    ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 1),
    ("conv3_1", 128, 256, 2), ("conv3_2", 256, 256, 1),
    ("conv4_1", 256, 512, 2), ("conv4_2", 512, 512, 1),
]
_DEC_CONVS = [  # name, c_in, c_out
    ("conv5_1", 512 + 256, 256), ("conv5_2", 256, 256),
    ("conv6_1", 256 + 128, 128), ("conv6_2", 128, 128),
    ("conv7_1", 128 + 64, 64), ("conv7_2", 64, 64),
    ("conv8_1", 64 + 32, 32), ("conv8_2", 32, 32),
]


def _bn_name(conv_name):
    return "bn" + conv_name[len("conv"):]


class _Conv3DParams(nn.Module):
    """upstream Backbone.py :: Conv3D parameter names (conv3d, bn3d)."""

    def __init__(self, ch):
        super().__init__()
        self.conv3d = nn.Conv3d(ch, ch, kernel_size=(1, 1, 1), stride=1, padding=(0, 0, 0))
        self.bn3d = nn.BatchNorm3d(ch)


class _EncoderParams(nn.Module):
    def __init__(self, in_channels, compress_level):
        super().__init__()
        for name, cin, cout, stride in _ENC_CONVS:
            setattr(self, name, nn.Conv2d(cin or in_channels, cout, 3, stride, 1))
            setattr(self, _bn_name(name), nn.BatchNorm2d(cout))
        self.conv3d_1 = _Conv3DParams(64)
        self.conv3d_2 = _Conv3DParams(128)
        self.compress_level = compress_level
        if compress_level > 0:
            cc = 256 // (2 ** compress_level)
            self.com_compresser = nn.Conv2d(256, cc, 1, 1)
            self.bn_compress = nn.BatchNorm2d(cc)
            self.com_decompresser = nn.Conv2d(cc, 256, 1, 1)
            self.bn_decompress = nn.BatchNorm2d(256)


class _DecoderParams(nn.Module):
    def __init__(self):
        super().__init__()
        for name, cin, cout in _DEC_CONVS:
            setattr(self, name, nn.Conv2d(cin, cout, 3, 1, 1))
            setattr(self, _bn_name(name), nn.BatchNorm2d(cout))


class _FusionParams(nn.Module):
    """PixelWeightedFusionSoftmax parameter names."""

    def __init__(self, channel):
        super().__init__()
        self.conv1_1 = nn.Conv2d(channel * 2, 128, 1)
        self.bn1_1 = nn.BatchNorm2d(128)
        self.conv1_2 = nn.Conv2d(128, 32, 1)
        self.bn1_2 = nn.BatchNorm2d(32)
        self.conv1_3 = nn.Conv2d(32, 8, 1)
        self.bn1_3 = nn.BatchNorm2d(8)
        self.conv1_4 = nn.Conv2d(8, 1, 1)


class _ClsHeadParams(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.conv1 = nn.Conv2d(32, 32, 3, 1, 1)
        self.conv2 = nn.Conv2d(32, config.category_num * len(config.anchor_size), 1)
        self.bn1 = nn.BatchNorm2d(32)


class _RegHeadParams(nn.Module):
    def __init__(self, config, out_seq_len):
        super().__init__()
        self.box_prediction = nn.Sequential(
            nn.Conv2d(32, 32, 3, 1, 1), nn.BatchNorm2d(32), nn.ReLU(),
            nn.Conv2d(32, len(config.anchor_size) * config.box_code_size * out_seq_len, 1))


# The reference's encoder and decoder both instantiate the WHOLE Backbone, so its checkpoints carry the decoder's
# layers under u_encoder.* and the encoder's under decoder.* -- parameters that never run.  Only these names
# are dropped on load; any other unknown key reaches torch's strict check.
_TENSORS = r"(weight|bias|running_mean|running_var|num_batches_tracked)"
_DUPLICATE_BACKBONE_KEY = re.compile(
    r"^(u_encoder\.(conv|bn)[5-8]_[12]\." + _TENSORS +
    r"|decoder\.((conv|bn)_pre_[12]|(conv|bn)[1-4]_[12]|conv3d_[12]\.(conv3d|bn3d)"
    r"|com_compresser|bn_compress|com_decompresser|bn_decompress)\." + _TENSORS + r")$")


# K-sliced layers of the detector (SURVEY.md 8 a3 / a8 names).  DN_SP_KSLICES=all slices every deep layer (A/B runs);
# the default is the one layer where it pays in both regimes: conv5_1, 48 chunks of K on 640 tiles
_KSLICES_ALL = {"conv3_2": 4, "conv4_1": 4, "conv4_2": 4, "conv5_1": 4, "conv5_2": 4, "conv6_1": 4}
_KSLICES = {"conv5_1": 4}


def _ks_enabled():
    return os.environ.get("DN_SP_KSLICES", "1") != "0"


class _ConvLayer:
    """One packed conv of the plan: weights in tile-major layout + folded affine."""

    __slots__ = ("name", "packed", "scale", "shift", "c_in", "c_out", "ksize", "stride", "relu",
                 "math", "affine", "weight", "packed_sig", "kslices")

    def __init__(self, name, weight, bias, bn, ksize, stride=1, relu=True, scale_shift=None,
                 math=0, up_split=None):
        """up_split = (c0, c1): the layer always runs as cat([up2(src0), src1]) with these channel counts
        (the decoder's *_1 convs) -- its SP weight image is the tap-merged one and is packed here, once."""
        self.name = name
        self.math = math
        # K slices of the layer (SP engine; include/disconet_hip.h :: dn_spconv2d_ks): a property of the LAYER, never of
        # the batch -- the deep layers, whose launches are a round and a quarter on the chip at the BASELINE batch and
        # fewer tiles than CUs on an agent-sharded rank
        table = _KSLICES_ALL if os.environ.get("DN_SP_KSLICES", "1") == "all" else _KSLICES
        self.kslices = table.get(name, 1) if (math == 2 and _ks_enabled()) else 1
        c_out = weight.shape[0]
        c_in = weight.shape[1]
        if up_split is not None and math == 2:
            assert sum(up_split) == c_in
            d = ops.conv_desc(1, 8, 8, up_split[0], c_out, ksize, stride, relu, c1=up_split[1], up0=True, math=math)
        else:
            up_split = None
            d = ops.conv_desc(1, 8, 8, c_in, c_out, ksize, stride, relu, math=math)
        if scale_shift is not None:
            self.scale, self.shift = scale_shift
        else:
            self.scale, self.shift = ops.fold_bn(bias, bn, c_out)
        self.affine = (self.scale, self.shift)      # the layer's own y = acc * scale + shift
        self.weight, self.packed_sig = None, None
        if math == 2:     # SP engine: weights lifted by a power of two, undone in the scale
            self.packed, wmul = ops.sp_pack_conv_weights(d, weight)
            self.scale = (self.scale / wmul).contiguous()
            # the packed image of a layer whose first source is upsampled depends on the source split
            # (merged taps, csrc/conv_sp.hip UPM / conv_spq.hip): packed above when the plan knows the
            # split (even map sizes assumed), else repacked on the first run that shows it (a host
            # sync + allocation: not inside a hipGraph capture)
            self.weight = weight.detach()
            self.packed_sig = (up_split[0], up_split[1], 1) if up_split is not None else (c_in, 0, 0)
        elif math == 1:   # split-f16 on the NHWC engine: the same power-of-two lift (the lo halves of |w| ~ 1e-4
            wmul = ops._pow2_lift(weight)      # would otherwise be f16 subnormals: 6e-4 relative error)
            self.packed = ops.pack_conv_weights(d, weight.detach() * wmul)
            self.scale = (self.scale / wmul).contiguous()
        else:
            self.packed = ops.pack_conv_weights(d, weight)
        self.c_in, self.c_out, self.ksize, self.stride, self.relu = c_in, c_out, ksize, stride, relu

    def run(self, src0, src1=None, up0=False, nhwc_copy=False):
        """nhwc_copy (SP engine, not on an upsampled source): -> (SpTensor, float32 NHWC copy from the same launch)"""
        n, h0, w0, c0 = src0.shape
        h_in, w_in = (h0 * 2, w0 * 2) if up0 else (h0, w0)
        c1 = src1.shape[3] if src1 is not None else 0
        assert c0 + c1 == self.c_in, (c0, c1, self.c_in)
        d = ops.conv_desc(n, h_in, w_in, c0, self.c_out, self.ksize, self.stride, self.relu,
                          c1=c1, up0=up0, math=self.math)
        ho, wo = ops.conv_out_hw(d)
        # algorithmic work of this launch: true (unpadded) channel counts
        flops = 2.0 * n * ho * wo * self.c_out * self.c_in * self.ksize * self.ksize
        nbytes = 4.0 * (src0.numel() + (src1.numel() if src1 is not None else 0)
                        + n * ho * wo * self.c_out + self.c_out * self.c_in * self.ksize ** 2)
        if self.math == 2:
            sig = (c0, c1, 1 + (h_in % 2) + 2 * (w_in % 2)) if up0 else self.packed_sig
            if sig != self.packed_sig:
                self.packed, _ = ops.sp_pack_conv_weights(d, self.weight)      # same wmul: same weights
                self.packed_sig = sig
            # split-planar engine: NHWC inputs (the voxel grid, the fused map) are split once here
            src0, src1 = ops.as_sp(src0), (ops.as_sp(src1) if src1 is not None else None)
            # executed MFMA work: 3 products per MAC (hi*hi, hi*lo, lo*hi); the tap-merged kernel (conv_spq.hip: 3x3 over a
            # nearest-upsampled first source) runs 4 merged taps instead of 9 on that source's chunks
            taps0 = {2: 4, 1: 6}.get(ops.sp_upmode(), 9) if (up0 and self.ksize == 3 and self.stride == 1) else self.ksize ** 2
            xflops = 3.0 * 2.0 * n * ho * wo * self.c_out * (taps0 * c0 + self.ksize ** 2 * c1)
            with region(self.name, "conv_sp_kernel", flops, nbytes, exec_flops=xflops):
                dual = nhwc_copy and not up0 and self.c_out % 4 == 0
                return ops.sp_conv2d(d, src0, self.packed, self.scale, self.shift, src1=src1, nhwc_copy=dual,
                                     kslices=self.kslices)
        src0, src1 = ops.as_nhwc(src0), (ops.as_nhwc(src1) if src1 is not None else None)
        out = torch.empty((n, ho, wo, self.c_out), dtype=torch.float32, device=src0.device)
        with region(self.name, "conv_mfma_kernel", flops, nbytes):
            return ops.conv2d(d, src0, self.packed, self.scale, self.shift, src1=src1, out=out)


class _ConvPostLayer:
    """A 3x3 conv (64 channels) + affine + ReLU fused with a following 1x1 stage
    (dn_conv2d_post1x1): split-f16 math only.  out_a gets columns [0, split) of the
    1x1 stage, out_b the rest (two-headed use) -- or everything goes to out_a."""

    def __init__(self, name, weight, scale, shift, w2, scale2, shift2, split, relu2, math=1,
                 block_diag=False):
        self.name = name
        self.math = math
        # the 1x1 stage is two independent heads: outputs [0, split) read hidden channels 0..31, the
        # rest channels 32..63 (SP engine: one 32-channel head per workgroup, weights LDS-resident)
        self.block_diag = block_diag and math == 2 and split < w2.shape[0]
        self.c_out, self.c_in = weight.shape[0], weight.shape[1]
        assert self.c_out == 64
        d = ops.conv_desc(1, 8, 8, self.c_in, 64, 3, 1, True, math=math)
        self.c_out2, self.split, self.relu2 = w2.shape[0], split, relu2
        if math == 2:
            self.packed, wmul = ops.sp_pack_conv_weights(d, weight)
            self.packed2, wmul2 = (ops.sp_pack_heads_weights(w2, split) if self.block_diag
                                   else ops.sp_pack_post1x1_weights(w2))
            # `scale`/`shift` may belong to another layer of the plan: scale copies, never in place
            self.scale, self.scale2 = (scale / wmul).contiguous(), (scale2 / wmul2).contiguous()
        else:
            wmul, wmul2 = ops._pow2_lift(weight), ops._pow2_lift(w2)
            self.packed = ops.pack_conv_weights(d, weight.detach() * wmul)
            self.packed2 = ops.pack_post1x1_weights(w2.detach() * wmul2)
            self.scale, self.scale2 = (scale / wmul).contiguous(), (scale2 / wmul2).contiguous()
        self.shift, self.shift2 = shift.contiguous(), shift2.contiguous()

    def run(self, src0):
        n, h, w, c0 = src0.shape
        assert c0 == self.c_in
        flops = 2.0 * n * h * w * (64 * self.c_in * 9 + self.c_out2 * 64)
        nbytes = 4.0 * (src0.numel() + n * h * w * self.c_out2)
        if self.math == 2:
            d = ops.conv_desc(n, h, w, c0, 64, 3, 1, True, math=2)
            src0 = ops.as_sp(src0)
            if self.split < self.c_out2:      # two-headed fp32 NHWC output (the detection heads)
                out_a = torch.empty((n, h, w, self.split), dtype=torch.float32, device=src0.device)
                out_b = torch.empty((n, h, w, self.c_out2 - self.split), dtype=torch.float32,
                                    device=src0.device)
            else:                             # one SP output (conv*_2 + the 1x1x1 Conv3D)
                out_a, out_b = ops.SpTensor(n, h, w, self.c_out2, device=src0.device), None
            with region(self.name, "conv_sp_kernel", flops, nbytes):
                ops.sp_conv2d_post1x1(d, src0, self.packed, self.scale, self.shift, self.packed2,
                                      self.scale2, self.shift2, self.c_out2, self.split, self.relu2,
                                      out_a, out_b, block_diag=self.block_diag)
            return out_a, out_b
        src0 = ops.as_nhwc(src0)
        d = ops.conv_desc(n, h, w, c0, 64, 3, 1, True, math=1)
        out_a = torch.empty((n, h, w, self.split), dtype=torch.float32, device=src0.device)
        out_b = None
        if self.split < self.c_out2:
            out_b = torch.empty((n, h, w, self.c_out2 - self.split), dtype=torch.float32,
                                device=src0.device)
        with region(self.name, "conv_mfma_kernel", flops, nbytes):
            ops.conv2d_post1x1(d, src0, self.packed, self.scale, self.shift, self.packed2,
                               self.scale2, self.shift2, self.c_out2, self.split, self.relu2,
                               out_a, out_b)
        return out_a, out_b


class DiscoNet(nn.Module):
    def __init__(self, config, layer=3, in_channels=13, kd_flag=True, num_agent=5,
                 compress_level=0, only_v2i=False):
        super().__init__()
        self.kd_flag = kd_flag
        self.layer = layer
        self.agent_num = num_agent
        self.only_v2i = only_v2i
        self.in_channels = in_channels
        self.category_num = config.category_num
        self.anchor_num_per_loc = len(config.anchor_size)
        self.box_code_size = config.box_code_size
        self.out_seq_len = 1 if config.only_det else config.pred_len

        self.u_encoder = _EncoderParams(in_channels, compress_level)
        self.decoder = _DecoderParams()
        self.classification = _ClsHeadParams(config)
        self.regression = _RegHeadParams(config, self.out_seq_len)
        self.pixel_weighted_fusion = _FusionParams(LAYER_CHANNEL[layer])

        self._plan = None
        self._plan_sig = None
        # conv arithmetic: "f16x3" (default) = split-f16: every fp32 operand as hi + lo halves,
        # three f16 MFMAs per product, fp32 accumulate -- max abs deviation from the fp32 oracle
        # ~1e-5 over the whole network (the 1e-4 parity bar; same size as the exact mode's own
        # summation-order noise); needs |activations|, |weights| < 65504.  "f32" = exact-fp32
        # MFMA at half the throughput.  Not a constructor argument so the reference's signature
        # is untouched: set model.conv_math or DISCONET_CONV_MATH before the first forward.
        # "sp" (the default) = the same split-f16 arithmetic with the activations kept pre-split in HBM
        # (ops.SpTensor, csrc/conv_sp.hip): no conversion work in the conv loop.
        self.conv_math = os.environ.get("DISCONET_CONV_MATH", "sp")
        # fold the 1x1 layers that follow a 64-channel 3x3 conv into that conv's launch
        self.fuse_1x1 = os.environ.get("DISCONET_FUSE_1X1", "1") != "0"
        # one-launch attention MLP + softmax + weighted sum (csrc/fuse_mlp.hip) instead of
        # two 1x1 conv launches + the tail kernel (split-f16 engines, C in {64, 128, 256})
        self.fuse_mlp = os.environ.get("DISCONET_FUSE_MLP", "1") != "0"
        # run the encoder levels above the exchanged one beside the fusion block on a second HIP stream:
        # REFUSED unless DISCONET_UNSAFE_OVERLAP=1 (the property below).  A kernel that shares a SIMD with
        # the split-f16 conv kernels has been observed to compute with corrupted VGPR lanes (DESIGN.md
        # 3.6 (B), profiles/r02_hazard_repro.txt), so the library runs strictly in stream order;
        # bench.py --in-flight N > 1 sets the override and checksums every replay.
        self._overlap_streams = False
        self.overlap_streams = os.environ.get("DISCONET_OVERLAP", "0") == "1"
        self._side = {}

    @property
    def overlap_streams(self):
        return self._overlap_streams

    @overlap_streams.setter
    def overlap_streams(self, value):
        self._overlap_streams = ops.check_overlap_request(value, "DiscoNet.overlap_streams")

    # ------------------------------------------------------------------
    # checkpoint compatibility
    # ------------------------------------------------------------------
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
        """Accepts a reference model_state_dict: strips the DataParallel
        `module.` prefix and ignores the duplicate, unused parameters the
        reference keeps because its encoder and decoder both instantiate the
        whole Backbone (SURVEY.md Appx A.3)."""
        own = set(super().state_dict().keys())
        cleaned, dropped, unknown = {}, [], []
        for k, v in state_dict.items():
            k = k[len("module."):] if k.startswith("module.") else k
            if k not in own and _DUPLICATE_BACKBONE_KEY.match(k):
                dropped.append(k)      # the half of the shared Backbone definition this side never runs
            elif k not in own and (k.startswith("u_encoder.") or k.startswith("decoder.")):
                # The duplicate-key list above is written from recollection (the reference source is not in
                # the mount).  A Backbone key it does not know is dropped WITH a warning rather than failing
                # the load of a real checkpoint: every parameter this model runs is in `own`, and a missing
                # one still fails torch's strict check below.
                unknown.append(k)
            else:
                cleaned[k] = v         # anything else: torch reports it when it is unexpected (strict)
        if unknown:
            import warnings
            warnings.warn("DiscoNet.load_state_dict: dropped %d Backbone keys this model does not run and the "
                          "duplicate-Backbone list does not name: %s" % (len(unknown), ", ".join(sorted(unknown)[:8])))
        self._plan = None
        return super().load_state_dict(cleaned, strict=strict, **kw)

    def _side_stream(self, device):
        key = str(device)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def train(self, mode=True):
        """train(): forward() runs the training-mode graph of disconet_amd/train.py (batch
        statistics, explicit HIP backward); eval(): the fused inference plan."""
        self._plan = None
        return super().train(mode)

    # ------------------------------------------------------------------
    # plan: packed weights + folded BN, rebuilt when any parameter changes
    # ------------------------------------------------------------------
    def _signature(self):
        if self.conv_math not in ops.MATH_MODES:
            raise ValueError("conv_math must be one of %s" % sorted(ops.MATH_MODES))
        return (self.conv_math, self.fuse_1x1, self.fuse_mlp) + tuple((t.data_ptr(), t._version) for t in
                                         list(self.parameters()) + list(self.buffers()))

    def _build_plan(self):
        enc, dec = self.u_encoder, self.decoder
        P = {}
        math = ops.MATH_MODES[self.conv_math]

        def _Layer(*args, **kw):          # every conv of the plan uses the model's math mode
            return _ConvLayer(*args, math=math, **kw)

        for name, _, _, stride in _ENC_CONVS:
            conv = getattr(enc, name)
            P[name] = _Layer(name, conv.weight, conv.bias, getattr(enc, _bn_name(name)), 3, stride)
        for name in ("conv3d_1", "conv3d_2"):
            m = getattr(enc, name)
            P[name] = _Layer(name, m.conv3d.weight, m.conv3d.bias, m.bn3d, 1)
        if enc.compress_level > 0:
            P["compress"] = _Layer("compress", enc.com_compresser.weight, enc.com_compresser.bias,
                                   enc.bn_compress, 1)
            P["decompress"] = _Layer("decompress", enc.com_decompresser.weight, enc.com_decompresser.bias,
                                     enc.bn_decompress, 1)
        for name, cin, cout in _DEC_CONVS:
            conv = getattr(dec, name)
            # conv5_1..conv8_1 read cat([up2(deeper level), skip]): the deeper level has 2 * cout channels
            split = (2 * cout, cin - 2 * cout) if name.endswith("_1") else None
            P[name] = _Layer(name, conv.weight, conv.bias, getattr(dec, _bn_name(name)), 3, up_split=split)
        cls, reg = self.classification, self.regression.box_prediction
        P["cls1"] = _Layer("cls1", cls.conv1.weight, cls.conv1.bias, cls.bn1, 3)
        P["cls2"] = _Layer("cls2", cls.conv2.weight, cls.conv2.bias, None, 1, relu=False)
        P["reg1"] = _Layer("reg1", reg[0].weight, reg[0].bias, reg[1], 3)
        P["reg2"] = _Layer("reg2", reg[3].weight, reg[3].bias, None, 1, relu=False)

        if math in (1, 2) and self.fuse_1x1:
            # split-f16 only: 1x1 layers ride in the epilogue of the 3x3 conv before them
            dev = cls.conv1.weight.device
            n_cls, n_reg = cls.conv2.weight.shape[0], reg[3].weight.shape[0]
            if n_cls % 4 == 0 and n_reg % 4 == 0 and n_cls + n_reg <= 64:
                w1 = torch.cat([cls.conv1.weight, reg[0].weight], 0).detach()
                w2 = torch.zeros(n_cls + n_reg, 64, device=dev)
                w2[:n_cls, :32] = cls.conv2.weight.detach().reshape(n_cls, 32)
                w2[n_cls:, 32:] = reg[3].weight.detach().reshape(n_reg, 32)
                P["heads_fused"] = _ConvPostLayer(
                    "heads", w1, torch.cat([P["cls1"].affine[0], P["reg1"].affine[0]]),
                    torch.cat([P["cls1"].affine[1], P["reg1"].affine[1]]), w2,
                    torch.ones(n_cls + n_reg, device=dev),
                    torch.cat([cls.conv2.bias, reg[3].bias]).detach().float(), n_cls, False, math=math,
                    block_diag=True)
            c3 = enc.conv3d_1
            s3, t3 = ops.fold_bn(c3.conv3d.bias, c3.bn3d, 64)
            P["conv1_2_3d"] = _ConvPostLayer(
                "conv1_2+3d", enc.conv1_2.weight.detach(), P["conv1_2"].affine[0], P["conv1_2"].affine[1],
                c3.conv3d.weight.detach().reshape(64, 64), s3, t3, 64, True, math=math)

        # attention MLP: layer 1 split W1 = [W1_ego | W1_nbr] (see fuse_tail.hip)
        f = self.pixel_weighted_fusion
        C = LAYER_CHANNEL[self.layer]
        w1 = f.conv1_1.weight.detach().reshape(128, 2 * C)
        dev = w1.device
        w_cat = torch.cat([w1[:, :C], w1[:, C:]], 0).contiguous()          # [256, C]
        ones256 = torch.ones(256, device=dev)
        shift_g = torch.cat([f.conv1_1.bias.detach().float(), torch.zeros(128, device=dev)])
        # the fusion block's kernels read fp32 NHWC maps: its two 1x1 launches stay on the NHWC engine
        nmath = ops.nhwc_math(math)
        P["mlp_g"] = _ConvLayer("mlp_g", w_cat.reshape(256, C, 1, 1), None, None, 1, relu=False,
                                scale_shift=(ones256, shift_g.contiguous()), math=nmath)
        P["mlp_f"] = _ConvLayer("mlp_f", w1[:, C:].contiguous().reshape(128, C, 1, 1), None, None, 1,
                                relu=False, scale_shift=(torch.ones(128, device=dev),
                                                         torch.zeros(128, device=dev)), math=nmath)
        bn1_scale, bn1_shift = ops.fold_bn(None, f.bn1_1, 128)
        s2, t2 = ops.fold_bn(f.conv1_2.bias, f.bn1_2, 32)
        s3, t3 = ops.fold_bn(f.conv1_3.bias, f.bn1_3, 8)
        tail = {
            "bn1_scale": bn1_scale, "bn1_shift": bn1_shift,
            "w2": f.conv1_2.weight.detach().reshape(32, 128).float().contiguous(),
            "s2": s2, "t2": t2,
            "w3": f.conv1_3.weight.detach().reshape(8, 32).float().contiguous(),
            "s3": s3, "t3": t3,
            "w4": f.conv1_4.weight.detach().reshape(8).float().contiguous(),
            "b4": f.conv1_4.bias.detach().float().contiguous(),
        }
        P["_tail_tensors"] = tail                 # keep the storage alive
        P["_tail"] = ops.make_tail_params(tail)
        if math != 0 and self.fuse_mlp and ops.fuse_mlp_supported(C):
            # split-f16 engines: the whole attention MLP + agent softmax + weighted sum in one launch
            P["_fuse_mlp"], P["_fuse_mlp_tensors"] = ops.make_fuse_mlp_params(
                w1, f.conv1_1.bias, (bn1_scale, bn1_shift),
                f.conv1_2.weight.reshape(32, 128), f.conv1_2.bias, ops.fold_bn(None, f.bn1_2, 32),
                f.conv1_3.weight.reshape(8, 32), f.conv1_3.bias, ops.fold_bn(None, f.bn1_3, 8),
                f.conv1_4.weight, f.conv1_4.bias, C)
        return P

    def _get_plan(self):
        sig = self._signature()
        if self._plan is None or sig != self._plan_sig:
            self._check_finite_parameters()
            self._plan = self._build_plan()
            self._plan_sig = sig
        return self._plan

    def _check_finite_parameters(self):
        """NaN / Inf in a weight, bias or BatchNorm statistic is refused HERE, once per plan (one reduction + one host
        read): inside the step a ReLU or the clamp of the hi/lo split would turn the resulting NaN into a finite number
        and nothing downstream could tell.  (Testing for NaN in the conv epilogues cost ~1 % of the step:
        profiles/r04_nancheck_ab.txt; overflow at run time is the range guard's clamp bit, a NaN out of the attention
        softmax its NaN bit.)"""
        bad = [n for n, t in list(self.named_parameters()) + list(self.named_buffers())
               if t.is_floating_point() and t.numel() and not bool(torch.isfinite(t).all())]
        if bad:
            raise ops._lib.DnError("DiscoNet: non-finite values in %s%s: refusing to pack a plan whose outputs would hold "
                                   "plausible garbage" % (", ".join(bad[:4]), " ..." if len(bad) > 4 else ""))

    # ------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------
    def _enc_input(self, bevs):
        if isinstance(bevs, ops.SpTensor):      # ops.scatter_dense_sp: already in the engine's layout
            return bevs
        n = bevs.shape[0] * bevs.shape[1]
        h, w, z = bevs.shape[2], bevs.shape[3], bevs.shape[4]
        # [A*B, 1, H, W, Z] is already NHWC with Z as the channel: the reference's
        # permute(0, 1, 4, 2, 3) to NCHW is a no-op for a channels-last engine.
        x = bevs.reshape(n, h, w, z)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        return x

    def _stem_pair(self, x, P):
        """conv_pre_1 -> conv_pre_2 as ONE launch (dn_spconv2d_pre_pair: the intermediate map stays in LDS) when the input is
        an occupancy bit grid and the two layers have the stem's shape; else None.  DN_STEM_PAIR=0: the two launches (A/B)."""
        l1, l2 = P["conv_pre_1"], P["conv_pre_2"]
        if not (isinstance(x, ops.SpTensor) and x.bits and l1.math == 2 and l2.math == 2
                and os.environ.get("DN_STEM_PAIR", "1") != "0"):
            return None
        n, h, w, c = x.shape
        d1 = ops.conv_desc(n, h, w, c, l1.c_out, l1.ksize, l1.stride, l1.relu, math="sp")
        d2 = ops.conv_desc(n, h, w, l2.c_in, l2.c_out, l2.ksize, l2.stride, l2.relu, math="sp")
        if c != l1.c_in or not ops.sp_conv2d_pre_pair_supported(d1, d2):
            return None
        flops = 2.0 * n * h * w * 9 * (l1.c_out * l1.c_in + l2.c_out * l2.c_in)
        nbytes = 4.0 * (n * h * w + n * h * w * l2.c_out)
        # executed: conv_pre_1 reads a hi-only (0 / 1) operand -> 2 MFMAs per product, conv_pre_2 3 (halo recompute not counted)
        xflops = 2.0 * n * h * w * 9 * (2 * l1.c_out * l1.c_in + 3 * l2.c_out * l2.c_in)
        with region("conv_pre_1+2", "conv_sp_kernel", flops, nbytes, exec_flops=xflops):      # (the SP engine's family name: bench.py sums it with the conv launches)
            return ops.sp_conv2d_pre_pair(d1, d2, x, l1.packed, l1.scale, l1.shift, l2.packed, l2.scale, l2.shift)

    def _enc_group(self, k, x, P, nhwc_copy=False):
        """encoder group k: its last layer's output is the pyramid level e[k].  nhwc_copy (SP engine): the last
        layer also writes the level as fp32 NHWC from its epilogue (the exchanged level: no dn_sp_to_nhwc pass);
        returns (level, nhwc copy or None)."""
        dual = nhwc_copy and self.conv_math == "sp"
        unpack = lambda r: r if isinstance(r, tuple) else (r, None)
        if k == 0:
            pair = self._stem_pair(x, P) if not dual else None
            if pair is not None:
                return pair, None
            return unpack(P["conv_pre_2"].run(P["conv_pre_1"].run(x), nhwc_copy=dual))
        if k == 1:
            if "conv1_2_3d" in P:
                return P["conv1_2_3d"].run(P["conv1_1"].run(x))[0], None
            return unpack(P["conv3d_1"].run(P["conv1_2"].run(P["conv1_1"].run(x)), nhwc_copy=dual))
        if k == 2:
            return unpack(P["conv3d_2"].run(P["conv2_2"].run(P["conv2_1"].run(x)), nhwc_copy=dual))
        return unpack(P["conv%d_2" % k].run(P["conv%d_1" % k].run(x), nhwc_copy=dual))

    def encode(self, bevs, P):
        x = self._enc_input(bevs)
        enc = []
        flat = None
        for k in range(5):
            x, copy = self._enc_group(k, x, P, nhwc_copy=(k == self.layer and "compress" not in P))
            flat = copy if k == self.layer else flat
            enc.append(x)
        if "compress" in P:
            enc[3] = P["decompress"].run(P["compress"].run(enc[3]))
        # the exchanged level leaves the conv engine (fusion kernels, the agent all-gather): fp32 NHWC, written by the
        # producing conv's own epilogue where that form exists, else converted
        enc[self.layer] = flat if flat is not None else ops.as_nhwc(enc[self.layer])
        return enc

    def fuse(self, feat, trans_matrices, num_agent, batch_size, P, want_weights=False,
             ego_first=0, ego_count=None, sp_out=False):
        """DiscoGraph fusion of the layer-`layer` maps.  `feat` holds the maps of
        ALL agents (agent-major NHWC); the result covers the egos
        [ego_first, ego_first + ego_count) -- every agent on one GPU, this rank's
        agents when a scene is sharded one agent per GPU (sharded.py)."""
        A = self.agent_num
        E = A if ego_count is None else ego_count
        feat = ops.as_nhwc(feat)
        n, h, w, c = feat.shape
        B = batch_size
        map_bytes = 4.0 * h * w * c
        pairs = B * E * (A - 1)
        warped = torch.empty((B, E, max(A - 1, 0), h, w, c), dtype=torch.float32,
                             device=feat.device)
        # the one-launch attention kernel takes the warped maps in its own (fragment-major) read order when the shape
        # allows: an opaque intermediate between the two launches (include/disconet_hip.h :: dn_warp_neighbors_fm)
        fm = ("_fuse_mlp" in P and os.environ.get("DN_FUSE_FM", "1") != "0" and A > 1 and ops.warp_fm_supported(h, w, c))
        with region("warp", "warp_neighbors_kernel", 0.0, map_bytes * (n + pairs)):
            ops.warp_neighbors(feat, trans_matrices, num_agent, B, A, self.only_v2i,
                               ego_first, E, out=warped, fm=fm)
        if "_fuse_mlp" in P:
            flops = 2.0 * B * E * h * w * (128.0 * c * (2 + (A - 1)) + A * (128 * 32 + 32 * 8 + 8))
            with region("fuse_mlp", "disco_fuse_mlp_kernel", flops, map_bytes * (3 * E * B + 2 * pairs)):
                return ops.disco_fuse_mlp(feat, warped, num_agent, P["_fuse_mlp"], B, A, self.only_v2i,
                                          want_weights, ego_first, E, sp_out=sp_out, fm=fm)
        g = P["mlp_g"].run(feat[ego_first * B:(ego_first + E) * B])
        fw = None
        if A > 1:
            fw = P["mlp_f"].run(warped.view(pairs, h, w, c))
        fused = torch.empty((E * B, h, w, c), dtype=torch.float32, device=feat.device)
        with region("fuse_tail", "disco_fuse_tail_kernel", 0.0, map_bytes * (2 * E * B + pairs)):
            return ops.disco_fuse_tail(feat, warped, g, fw, num_agent, P["_tail"], B, A,
                                       self.only_v2i, want_weights, ego_first, E, out=fused)

    def decode(self, enc, P):
        x0, x1, x2, x3, x4 = enc
        x5 = P["conv5_2"].run(P["conv5_1"].run(x4, x3, up0=True))
        x6 = P["conv6_2"].run(P["conv6_1"].run(x5, x2, up0=True))
        x7 = P["conv7_2"].run(P["conv7_1"].run(x6, x1, up0=True))
        x8 = P["conv8_2"].run(P["conv8_1"].run(x7, x0, up0=True))
        return x8, x7, x6, x5

    def heads(self, x8, P):
        if "heads_fused" in P:                          # conv1 of both heads + both conv2, one launch
            cls, loc = P["heads_fused"].run(x8)
        else:
            cls = ops.as_nhwc(P["cls2"].run(P["cls1"].run(x8)))   # [N, H, W, A_loc*cat]  (NHWC: the
            loc = ops.as_nhwc(P["reg2"].run(P["reg1"].run(x8)))   #  reference's permute(0,2,3,1) is free)
        n, h, w = cls.shape[0], cls.shape[1], cls.shape[2]
        cls_preds = cls.view(n, -1, self.category_num)
        loc_preds = loc.view(n, h, w, self.anchor_num_per_loc, self.out_seq_len, self.box_code_size)
        return {"loc": loc_preds, "cls": cls_preds}

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        if not bevs.is_cuda:
            raise ops._lib.DnError("DiscoNet.forward needs GPU tensors; there is no CPU path")
        if self.training:
            from .train import train_forward
            if isinstance(bevs, ops.SpTensor):
                n, h, w, z = bevs.shape
                bevs = bevs.nhwc().view(n, 1, h, w, z)
            return train_forward(self, bevs, trans_matrices, num_agent_tensor, batch_size)
        P = self._get_plan()
        A = self.agent_num
        if bevs.shape[0] != A * batch_size:
            raise ValueError("bevs has %d images, expected num_agent*batch_size = %d"
                             % (bevs.shape[0], A * batch_size))
        trans = trans_matrices.to(device=bevs.device, dtype=torch.float32).contiguous()
        num_agent = ops.live_agent_counts(num_agent_tensor, bevs.device)

        if self.overlap_streams and self.layer < 4:
            # The encoder groups above the exchanged level do not depend on the fusion: they run on a
            # second HIP stream next to the warp / attention kernels (small-grid, MFMA-bound convs
            # beside L2/HBM-bound gathers) and join before the decoder.  Inside a hipGraph capture the
            # fork / join become parallel branches of the graph.
            x = self._enc_input(bevs)
            enc = []
            for k in range(self.layer + 1):
                x, _ = self._enc_group(k, x, P)
                enc.append(x)
            main = torch.cuda.current_stream()
            side = self._side_stream(bevs.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                up = x
                for k in range(self.layer + 1, 5):
                    up, _ = self._enc_group(k, up, P)
                    enc.append(up)
            feat = ops.as_nhwc(enc[self.layer])
            if "compress" in P and self.layer != 3:
                with torch.cuda.stream(side):                       # x3 is on the side branch then
                    enc[3] = P["decompress"].run(P["compress"].run(enc[3]))
            elif "compress" in P:
                feat = P["decompress"].run(P["compress"].run(feat))
            fused = self.fuse(feat, trans, num_agent, batch_size, P, sp_out=self.conv_math == "sp")
            main.wait_stream(side)
            for t in enc[self.layer + 1:]:
                t.record_stream(main)
            enc[self.layer] = fused
        else:
            enc = self.encode(bevs, P)
            fused = self.fuse(enc[self.layer], trans, num_agent, batch_size, P, sp_out=self.conv_math == "sp")
            enc[self.layer] = fused
        x8, x7, x6, x5 = self.decode(enc, P)

        result = self.heads(x8, P)
        ops.check_sp_range("DiscoNet.forward")      # asynchronous range guard: a clamp / NaN of THIS forward raises at the next one
        if self.kd_flag == 1:
            # NCHW-shaped, channels-last-strided views of the NHWC buffers
            nchw = lambda t: ops.as_nhwc(t).permute(0, 3, 1, 2)
            return (result, nchw(x8), nchw(x7), nchw(x6), nchw(x5), nchw(fused))
        return result
