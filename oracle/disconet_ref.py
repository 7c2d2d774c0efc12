"""torch-CPU restatement of the coperception `--com disco` detector forward.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (disconet_amd/).

PARITY UNPINNED.  The reference mount has no source for this path
(/root/reference/coperception/ is an empty submodule directory,
/root/reference/.gitmodules:1-3), so every function cites the *upstream* file
it restates (no line numbers: the file cannot be opened here) and the
behavioural spec it follows.  The only mounted call
sites are the shell lines /root/reference/README.md:54-63 (train_codet.py
--com disco) and README.md:68-75 (test_codet.py --com disco).

Written on torch-CPU primitives whose default semantics are pinned in
tests/test_oracle_primitives.py (affine_grid / grid_sample align_corners=False,
zeros padding, bilinear; F.interpolate nearest).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class RefConfig:
    """Values of upstream:coperception/configs/Config.py used on the path
    (SURVEY.md Appx A.1).  Attribute names follow the reference."""

    def __init__(self, map_hw=256):
        self.voxel_size = (0.25, 0.25, 0.4)
        half = map_hw * 0.25 / 2.0
        self.area_extents = np.array([[-half, half], [-half, half], [-3.0, 2.0]])
        self.map_dims = [map_hw, map_hw, 13]
        self.category_num = 2
        self.box_code_size = 6
        self.anchor_size = np.asarray(
            [[2.0, 4.0, 0.0], [2.0, 4.0, math.pi / 2.0], [2.0, 4.0, -math.pi / 4.0],
             [3.0, 12.0, 0.0], [3.0, 12.0, math.pi / 2.0], [3.0, 12.0, -math.pi / 4.0]])
        self.only_det = True
        self.binary = True
        self.use_map = False
        self.use_vis = False
        self.motion_state = False
        self.pred_len = 1


# ---------------------------------------------------------------------------
# upstream:coperception/models/det/backbone/Backbone.py  (SURVEY.md Appx A.6)
# ---------------------------------------------------------------------------
class Conv3D(nn.Module):
    """upstream Backbone.py :: Conv3D -- Conv3d(1,1,1) + BatchNorm3d + ReLU over
    a sequence dim of length 1."""

    def __init__(self, in_channel, out_channel, kernel_size, stride, padding):
        super().__init__()
        self.conv3d = nn.Conv3d(in_channel, out_channel, kernel_size=kernel_size,
                                stride=stride, padding=padding)
        self.bn3d = nn.BatchNorm3d(out_channel)

    def forward(self, x):
        # x: (batch, seq, c, h, w)
        x = x.permute(0, 2, 1, 3, 4).contiguous()
        x = F.relu(self.bn3d(self.conv3d(x)))
        x = x.permute(0, 2, 1, 3, 4).contiguous()
        return x


class Backbone(nn.Module):
    """upstream Backbone.py :: Backbone (encode + decode).  The reference's
    LidarEncoder / LidarDecoder both subclass this and each instantiate the
    full parameter set; that duplication is reproduced so state_dict keys
    match (SURVEY.md Appx A.3)."""

    def __init__(self, height_feat_size, compress_level=0):
        super().__init__()
        self.conv_pre_1 = nn.Conv2d(height_feat_size, 32, 3, 1, 1)
        self.conv_pre_2 = nn.Conv2d(32, 32, 3, 1, 1)
        self.bn_pre_1 = nn.BatchNorm2d(32)
        self.bn_pre_2 = nn.BatchNorm2d(32)

        self.conv3d_1 = Conv3D(64, 64, (1, 1, 1), 1, (0, 0, 0))
        self.conv3d_2 = Conv3D(128, 128, (1, 1, 1), 1, (0, 0, 0))

        self.conv1_1 = nn.Conv2d(32, 64, 3, 2, 1)
        self.conv1_2 = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv2_1 = nn.Conv2d(64, 128, 3, 2, 1)
        self.conv2_2 = nn.Conv2d(128, 128, 3, 1, 1)
        self.conv3_1 = nn.Conv2d(128, 256, 3, 2, 1)
        self.conv3_2 = nn.Conv2d(256, 256, 3, 1, 1)
        self.conv4_1 = nn.Conv2d(256, 512, 3, 2, 1)
        self.conv4_2 = nn.Conv2d(512, 512, 3, 1, 1)

        self.conv5_1 = nn.Conv2d(512 + 256, 256, 3, 1, 1)
        self.conv5_2 = nn.Conv2d(256, 256, 3, 1, 1)
        self.conv6_1 = nn.Conv2d(256 + 128, 128, 3, 1, 1)
        self.conv6_2 = nn.Conv2d(128, 128, 3, 1, 1)
        self.conv7_1 = nn.Conv2d(128 + 64, 64, 3, 1, 1)
        self.conv7_2 = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv8_1 = nn.Conv2d(64 + 32, 32, 3, 1, 1)
        self.conv8_2 = nn.Conv2d(32, 32, 3, 1, 1)

        for name, ch in (("1_1", 64), ("1_2", 64), ("2_1", 128), ("2_2", 128),
                         ("3_1", 256), ("3_2", 256), ("4_1", 512), ("4_2", 512),
                         ("5_1", 256), ("5_2", 256), ("6_1", 128), ("6_2", 128),
                         ("7_1", 64), ("7_2", 64), ("8_1", 32), ("8_2", 32)):
            setattr(self, "bn" + name, nn.BatchNorm2d(ch))

        self.compress_level = compress_level
        if compress_level > 0:
            compress_channel_num = 256 // (2 ** compress_level)
            self.com_compresser = nn.Conv2d(256, compress_channel_num, 1, 1)
            self.bn_compress = nn.BatchNorm2d(compress_channel_num)
            self.com_decompresser = nn.Conv2d(compress_channel_num, 256, 1, 1)
            self.bn_decompress = nn.BatchNorm2d(256)

    def encode(self, x):
        # x: [batch, seq, z, h, w]
        batch, seq, z, h, w = x.size()
        x = x.reshape(-1, x.size(-3), x.size(-2), x.size(-1)).to(torch.float)
        x = F.relu(self.bn_pre_1(self.conv_pre_1(x)))
        x = F.relu(self.bn_pre_2(self.conv_pre_2(x)))

        x_1 = F.relu(self.bn1_1(self.conv1_1(x)))
        x_1 = F.relu(self.bn1_2(self.conv1_2(x_1)))
        x_1 = x_1.view(batch, -1, x_1.size(1), x_1.size(2), x_1.size(3)).contiguous()
        x_1 = self.conv3d_1(x_1)
        x_1 = x_1.view(-1, x_1.size(2), x_1.size(3), x_1.size(4)).contiguous()

        x_2 = F.relu(self.bn2_1(self.conv2_1(x_1)))
        x_2 = F.relu(self.bn2_2(self.conv2_2(x_2)))
        x_2 = x_2.view(batch, -1, x_2.size(1), x_2.size(2), x_2.size(3)).contiguous()
        x_2 = self.conv3d_2(x_2)
        x_2 = x_2.view(-1, x_2.size(2), x_2.size(3), x_2.size(4)).contiguous()

        x_3 = F.relu(self.bn3_1(self.conv3_1(x_2)))
        x_3 = F.relu(self.bn3_2(self.conv3_2(x_3)))

        x_4 = F.relu(self.bn4_1(self.conv4_1(x_3)))
        x_4 = F.relu(self.bn4_2(self.conv4_2(x_4)))

        if self.compress_level > 0:
            x_3 = F.relu(self.bn_compress(self.com_compresser(x_3)))
            x_3 = F.relu(self.bn_decompress(self.com_decompresser(x_3)))
        return [x, x_1, x_2, x_3, x_4]

    def decode(self, x, x_1, x_2, x_3, x_4, batch, kd_flag=False):
        x_5 = F.relu(self.bn5_1(self.conv5_1(
            torch.cat((F.interpolate(x_4, scale_factor=(2, 2)), x_3), dim=1))))
        x_5 = F.relu(self.bn5_2(self.conv5_2(x_5)))
        # the reference's view/permute/adaptive_max_pool3d block on x_2 and x_1
        # is the identity for a sequence length of 1 (flag off) -- Appx A.6.
        x_6 = F.relu(self.bn6_1(self.conv6_1(
            torch.cat((F.interpolate(x_5, scale_factor=(2, 2)), x_2), dim=1))))
        x_6 = F.relu(self.bn6_2(self.conv6_2(x_6)))
        x_7 = F.relu(self.bn7_1(self.conv7_1(
            torch.cat((F.interpolate(x_6, scale_factor=(2, 2)), x_1), dim=1))))
        x_7 = F.relu(self.bn7_2(self.conv7_2(x_7)))
        x_8 = F.relu(self.bn8_1(self.conv8_1(
            torch.cat((F.interpolate(x_7, scale_factor=(2, 2)), x), dim=1))))
        x_8 = F.relu(self.bn8_2(self.conv8_2(x_8)))
        if kd_flag:
            return [x_8, x_7, x_6, x_5]
        return [x_8]


class LidarEncoder(Backbone):
    def forward(self, x):
        return super().encode(x)


class LidarDecoder(Backbone):
    def forward(self, x, x_1, x_2, x_3, x_4, batch, kd_flag=False):
        return super().decode(x, x_1, x_2, x_3, x_4, batch, kd_flag)


# ---------------------------------------------------------------------------
# upstream:coperception/models/det/base/*  heads  
# ---------------------------------------------------------------------------
class ClassificationHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        channel = 32
        anchor_num_per_loc = len(config.anchor_size)
        self.conv1 = nn.Conv2d(channel, channel, 3, 1, 1)
        self.conv2 = nn.Conv2d(channel, config.category_num * anchor_num_per_loc, 1, 1, 0)
        self.bn1 = nn.BatchNorm2d(channel)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        return self.conv2(x)


class SingleRegressionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        channel = 32
        anchor_num_per_loc = len(config.anchor_size)
        out_seq_len = 1 if config.only_det else config.pred_len
        self.box_prediction = nn.Sequential(
            nn.Conv2d(channel, channel, 3, 1, 1),
            nn.BatchNorm2d(channel),
            nn.ReLU(),
            nn.Conv2d(channel, anchor_num_per_loc * config.box_code_size * out_seq_len, 1, 1, 0))

    def forward(self, x):
        return self.box_prediction(x)


# ---------------------------------------------------------------------------
# upstream:coperception/models/det/DiscoNet.py  (SURVEY.md Appx A.5)
# ---------------------------------------------------------------------------
class PixelWeightedFusionSoftmax(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.conv1_1 = nn.Conv2d(channel * 2, 128, 1, 1, 0)
        self.bn1_1 = nn.BatchNorm2d(128)
        self.conv1_2 = nn.Conv2d(128, 32, 1, 1, 0)
        self.bn1_2 = nn.BatchNorm2d(32)
        self.conv1_3 = nn.Conv2d(32, 8, 1, 1, 0)
        self.bn1_3 = nn.BatchNorm2d(8)
        self.conv1_4 = nn.Conv2d(8, 1, 1, 1, 0)

    def forward(self, x):
        x = x.view(-1, x.size(-3), x.size(-2), x.size(-1))
        x_1 = F.relu(self.bn1_1(self.conv1_1(x)))
        x_1 = F.relu(self.bn1_2(self.conv1_2(x_1)))
        x_1 = F.relu(self.bn1_3(self.conv1_3(x_1)))
        x_1 = F.relu(self.conv1_4(x_1))
        return x_1


def feature_transformation(b, nb_agent_idx, local_com_mat, all_warp, size):
    """upstream base/IntermediateModelBase :: feature_transformation
    (SURVEY.md Appx A.4): rotate, zero-pad, then translate; two bilinear
    resamples; theta detached from the pose tensor."""
    nb_agent = torch.unsqueeze(local_com_mat[b, nb_agent_idx], 0)
    nb_warp = all_warp[nb_agent_idx]
    x_trans = (4 * nb_warp[0, 3]) / 128
    y_trans = -(4 * nb_warp[1, 3]) / 128

    theta_rot = torch.tensor([[nb_warp[0, 0], nb_warp[0, 1], 0.0],
                              [nb_warp[1, 0], nb_warp[1, 1], 0.0]]).type(dtype=torch.float)
    theta_rot = torch.unsqueeze(theta_rot, 0)
    grid_rot = F.affine_grid(theta_rot, size=torch.Size(size), align_corners=False)

    theta_trans = torch.tensor([[1.0, 0.0, x_trans], [0.0, 1.0, y_trans]]).type(dtype=torch.float)
    theta_trans = torch.unsqueeze(theta_trans, 0)
    grid_trans = F.affine_grid(theta_trans, size=torch.Size(size), align_corners=False)

    warp_feat_rot = F.grid_sample(nb_agent, grid_rot, mode="bilinear",
                                  padding_mode="zeros", align_corners=False)
    warp_feat_trans = F.grid_sample(warp_feat_rot, grid_trans, mode="bilinear",
                                    padding_mode="zeros", align_corners=False)
    return torch.squeeze(warp_feat_trans, 0)


LAYER_CHANNEL = {4: 512, 3: 256, 2: 128, 1: 64, 0: 32}


class DiscoNetRef(nn.Module):
    """upstream:coperception/models/det/DiscoNet.py :: DiscoNet (+ the
    DetModelBase / IntermediateModelBase plumbing it inherits).  Constructor
    and forward signatures per SURVEY.md §8(b)."""

    def __init__(self, config, layer=3, in_channels=13, kd_flag=True, num_agent=5,
                 compress_level=0, only_v2i=False):
        super().__init__()
        self.kd_flag = kd_flag
        self.layer = layer
        self.agent_num = num_agent
        self.only_v2i = only_v2i
        self.category_num = config.category_num
        self.anchor_num_per_loc = len(config.anchor_size)
        self.box_code_size = config.box_code_size
        self.out_seq_len = 1 if config.only_det else config.pred_len
        self.map_hw = config.map_dims[0]

        self.u_encoder = LidarEncoder(in_channels, compress_level)
        self.decoder = LidarDecoder(in_channels)
        self.classification = ClassificationHead(config)
        self.regression = SingleRegressionHead(config)
        self.pixel_weighted_fusion = PixelWeightedFusionSoftmax(LAYER_CHANNEL[layer])

    # -- re-indexing helpers ------------------------------
    def build_local_communication_matrix(self, feat_maps, batch_size):
        feats = [torch.unsqueeze(feat_maps[batch_size * i: batch_size * (i + 1)], 1)
                 for i in range(self.agent_num)]
        return torch.cat(tuple(feats), 1)          # [B, A, C, H, W]

    def agents_to_batch(self, feats):
        return torch.cat([feats[:, i] for i in range(self.agent_num)], 0)

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        bevs = bevs.permute(0, 1, 4, 2, 3)          # (A*B, seq, z, h, w)
        encoded = self.u_encoder(bevs)
        feat_maps = encoded[self.layer]
        down = 2 ** self.layer
        size = (1, LAYER_CHANNEL[self.layer], self.map_hw // down, self.map_hw // down)

        local_com_mat = self.build_local_communication_matrix(feat_maps, batch_size)
        local_com_mat_update = self.build_local_communication_matrix(feat_maps, batch_size).clone()

        for b in range(batch_size):
            num_agent = int(num_agent_tensor[b, 0])
            for i in range(num_agent):
                tg_agent = local_com_mat[b, i]
                neighbor_feat_list = [tg_agent]
                all_warp = trans_matrices[b, i]
                for j in range(num_agent):
                    if j != i:
                        if self.only_v2i and i != 0 and j != 0:
                            continue
                        neighbor_feat_list.append(
                            feature_transformation(b, j, local_com_mat, all_warp, size))

                tmp_agent_weight_list = []
                sum_weight = 0
                for nb in neighbor_feat_list:
                    cat_feat = torch.cat([tg_agent, nb], dim=0).unsqueeze(0)
                    agent_weight = torch.squeeze(self.pixel_weighted_fusion(cat_feat))
                    tmp_agent_weight_list.append(torch.exp(agent_weight))
                    sum_weight = sum_weight + torch.exp(agent_weight)

                agent_wise_weight_feat = 0
                for k, nb in enumerate(neighbor_feat_list):
                    agent_weight = torch.div(tmp_agent_weight_list[k], sum_weight)
                    agent_wise_weight_feat = agent_wise_weight_feat + agent_weight * nb
                local_com_mat_update[b, i] = agent_wise_weight_feat

        feat_fuse_mat = self.agents_to_batch(local_com_mat_update)
        encoded = list(encoded)
        encoded[self.layer] = feat_fuse_mat
        decoded = self.decoder(*encoded, batch_size, kd_flag=self.kd_flag)
        x = decoded[0]

        cls_preds = self.classification(x).permute(0, 2, 3, 1).contiguous()
        cls_preds = cls_preds.view(cls_preds.shape[0], -1, self.category_num)
        loc_preds = self.regression(x).permute(0, 2, 3, 1).contiguous()
        loc_preds = loc_preds.view(-1, loc_preds.size(1), loc_preds.size(2),
                                   self.anchor_num_per_loc, self.out_seq_len, self.box_code_size)
        result = {"loc": loc_preds, "cls": cls_preds}
        if self.kd_flag == 1:
            return (result, *decoded, feat_fuse_mat)
        return result


def randomize_bn_stats(model, seed=7):
    """BN running stats mildly randomised so eval-mode BN is
    non-trivial: mean ~ N(0,0.1), var ~ U(0.5,1.5), gamma ~ U(0.8,1.2),
    beta ~ N(0,0.05)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            n = m.num_features
            m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            with torch.no_grad():
                m.weight.copy_(torch.rand(n, generator=g) * 0.4 + 0.8)
                m.bias.copy_(torch.randn(n, generator=g) * 0.05)


def kaiming_reinit(model, seed=5):
    """Test-only re-initialisation that keeps activations O(1) through the 24
    conv layers (torch's default init lets them shrink towards the biases, which
    would make an absolute 1e-4 parity bound nearly vacuous)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            fan_in = m.weight[0].numel()
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def build_ref_model(config=None, seed=0, init="torch", **kw):
    """init="torch": default nn init under manual_seed(seed);
    init="kaiming": additionally kaiming_reinit (used by the parity tests)."""
    torch.manual_seed(seed)
    config = config or RefConfig()
    model = DiscoNetRef(config, **kw)
    if init == "kaiming":
        kaiming_reinit(model)
    randomize_bn_stats(model)
    model.eval()
    return model
