"""TEST-ONLY compute engine for disconet_amd.sharded: the CPU oracle cut at the
same three seams (encode / fuse own egos / decode+heads) so the N>1 host logic
(sharding, agent-major all-gather, ego ranges) can run under gloo without a GPU."""
import torch

from oracle.disconet_ref import LAYER_CHANNEL, feature_transformation


class OracleEngine:
    def __init__(self, ref):
        self.ref = ref
        self.layer = ref.layer
        self.agent_num = ref.agent_num

    def encode(self, bevs_local):
        with torch.no_grad():
            return self.ref.u_encoder(bevs_local.permute(0, 1, 4, 2, 3))

    def fuse(self, feat_all, trans, num_agent, batch_size, ego_first, ego_count):
        ref, A, B = self.ref, self.agent_num, batch_size
        com = ref.build_local_communication_matrix(feat_all, B)           # [B, A, C, H, W]
        size = (1,) + tuple(feat_all.shape[1:])
        out = torch.empty((ego_count * B,) + tuple(feat_all.shape[1:]))
        with torch.no_grad():
            for b in range(B):
                n = int(num_agent[b])
                for il in range(ego_count):
                    i = ego_first + il
                    if i >= n:
                        out[il * B + b] = com[b, i]
                        continue
                    nbrs = [com[b, i]]
                    for j in range(n):
                        if j != i and not (ref.only_v2i and i != 0 and j != 0):
                            nbrs.append(feature_transformation(b, j, com, trans[b, i], size))
                    e = [torch.exp(torch.squeeze(ref.pixel_weighted_fusion(
                        torch.cat([com[b, i], nb], 0).unsqueeze(0)))) for nb in nbrs]
                    s = sum(e)
                    out[il * B + b] = sum((ek / s) * nb for ek, nb in zip(e, nbrs))
        return out

    def decode_heads(self, enc_local):
        ref = self.ref
        with torch.no_grad():
            x = ref.decoder(*enc_local, 1, kd_flag=False)[0]
            cls = ref.classification(x).permute(0, 2, 3, 1).contiguous()
            loc = ref.regression(x).permute(0, 2, 3, 1).contiguous()
        return {"cls": cls.view(cls.shape[0], -1, ref.category_num),
                "loc": loc.view(-1, loc.size(1), loc.size(2), ref.anchor_num_per_loc,
                                ref.out_seq_len, ref.box_code_size)}
