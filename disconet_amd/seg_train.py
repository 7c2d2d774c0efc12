"""Training step of the segmentation variant of `--com disco` on the HIP path (BASELINE.json
configs[3]: "DiscoNet seg head, 5-agent, 256x256 BEV ... exercises decoder path + per-pixel CE").

Mirrors upstream:coperception/utils/SegModule.py :: SegModule.step (recollection; the source is not in the mount,
/root/reference/coperception is an empty submodule directory -- README.md:15, :37, :49 are the mounted mentions):

    model.train();  pred = model(bev, trans_matrices, num_agent)
    pred, labels = the images whose BEV is not empty (torch.sum(bev[i]) > 1e-4; padded agent slots are dropped)
    loss = nn.CrossEntropyLoss()(pred, labels);  optimizer.zero_grad();  loss.backward();  optimizer.step()

Built on the detector's training engine (train.py): the same conv forward / BatchNorm-statistics / data-gradient /
weight-gradient / fusion / Adam kernels, plus the four UNet ops of csrc/seg_ops.hip on fp32 NHWC maps (max-pool and
bilinear-upsample forward / backward).  No autograd, no ATen math.  The reverse pass walks the static UNet graph:

    outc <- up4 <- up3 <- up2 <- up1 <- down4 <- [fusion at x4] <- down3 <- down2 <- down1 <- inc

A skip map (x1..x3, the fused x4) has two consumers -- the max-pool below it and the Up block's concat -- and both
gradients enter the BatchNorm backward of the layer that produced it (its `dy_b` input): no add pass.
"""
import torch

from . import ops, train_ops as T
from .train import TrainEngine, _Layer, _EPS, _MOMENTUM

_DOUBLES = ("inc", "down1", "down2", "down3", "down4", "up1", "up2", "up3", "up4")


class SegTrainEngine(TrainEngine):
    FUSE_LEVEL_CHANNELS = 512
    _FWD_SP_OK = False      # the UNet's convs are fed by max-pool / bilinear-upsample kernels that write fp32 rows only: NHWC engine

    def _param_order(self, model):
        return list(model.parameters())

    def _math(self):
        # forward convs of the training step: split-f16 on the NHWC engine (the eval plan's arithmetic; "f32" =
        # exact-fp32 MFMA, set model.train_math); data and weight gradients are always exact fp32 (train.py)
        return ops.nhwc_math(getattr(self.model, "train_math", "f16x3"))

    def _graph(self):
        m = self.model
        L = {}

        def double(name, dc):
            seq = dc.double_conv
            L[name + "a"] = _Layer(name + "a", seq[0].weight, seq[0].bias, seq[1], 3)
            L[name + "b"] = _Layer(name + "b", seq[3].weight, seq[3].bias, seq[4], 3)

        double("inc", m.inc)
        for k in (1, 2, 3, 4):
            double("down%d" % k, getattr(m, "down%d" % k).maxpool_conv[1])
            double("up%d" % k, getattr(m, "up%d" % k).conv)
        f = m.pixel_weighted_fusion
        for i, (cname, bname) in enumerate((("conv1_2", "bn1_2"), ("conv1_3", "bn1_3")), 2):
            conv = getattr(f, cname)
            L["mlp%d" % i] = _Layer("mlp%d" % i, conv.weight, conv.bias, getattr(f, bname), 1)
        self.L = L

    # ------------------------------------------------------------------
    def forward(self, x, trans_matrices, num_agent_tensor, batch_size):
        """x: [A*B, H, W, n_channels] float32 NHWC (dense).  -> logits [A*B, H, W, n_classes] NHWC"""
        self.check_aliasing()
        self.generation += 1
        self._pack_multi()
        m, L = self.model, self.L
        A, B = m.agent_num, batch_size
        n, H, W = x.shape[0], x.shape[1], x.shape[2]
        if n != A * B:
            raise ValueError("bevs has %d images, expected num_agent*batch_size = %d" % (n, A * B))
        if H % 16 or W % 16:
            raise ValueError("seg training needs H, W multiples of 16 (got %d x %d): the Up blocks' F.pad is not built" % (H, W))
        dev = x.device
        trans = trans_matrices.to(device=dev, dtype=torch.float32).contiguous()
        F = self._fusion_lists(trans, num_agent_tensor[:, 0].cpu(), B, dev)
        self.F = F

        def double(name, src0, src1=None, into=None):
            a = self._layer_fwd(L[name + "a"], src0, src1)
            return self._layer_fwd(L[name + "b"], a, y_out=into)

        x1 = double("inc", x)
        x2 = double("down1", T.maxpool2(x1))
        x3 = double("down2", T.maxpool2(x2))
        NI, NW = n, F["n_warps"]
        C = self.FUSE_LEVEL_CHANNELS
        maps = torch.empty((NI + NW, H // 8, W // 8, C), dtype=torch.float32, device=dev)   # [own maps | warped maps]
        double("down3", T.maxpool2(x3), into=maps[:NI])
        fused = self._fusion_fwd(maps, NI, NW, F)
        x5 = double("down4", T.maxpool2(fused))
        # Up: cat([skip, upsampled], channel) -> DoubleConv; the concat is the conv's two-source gather
        u1 = T.upsample2_bilinear(x5)
        x6 = double("up1", fused, u1)
        u2 = T.upsample2_bilinear(x6)
        x7 = double("up2", x3, u2)
        u3 = T.upsample2_bilinear(x7)
        x8 = double("up3", x2, u3)
        u4 = T.upsample2_bilinear(x8)
        x9 = double("up4", x1, u4)
        logits, d_out = self._conv(m.outc.conv.weight, m.outc.conv.bias, x9, ksize=1)
        self.sctx = dict(x1=x1, x2=x2, x3=x3, fused=fused, x9=x9, d_out=d_out)

        self._tracked = []
        for lay in L.values():          # running statistics, the reference's momentum update (inside the statistics' launch where
            if lay.name.startswith("mlp"):      # _layer_fwd could fuse it: `running_done`)
                continue
            c = lay.ctx
            self._update_running(lay.bn, c["mean"], c["var"], c["z"].numel() // c["z"].shape[-1], done=c.get("running_done", False))
        if self._tracked:
            torch._foreach_add_(self._tracked, 1)
        self._tracked = []
        self.outs = dict(x9=x9, x8=x8, x7=x7, x6=x6, x5=x5, fused=fused)
        self.last_logits = logits
        return logits

    # ------------------------------------------------------------------
    def _backward_pass(self, dlogits, G=None):
        """dlogits [A*B, H, W, n_classes] NHWC -> every parameter's gradient (into the flat buffer G); TrainEngine.backward
        runs it, polls the split-f16 range guard and re-runs it on the fp32 kernels when a gradient map was clamped"""
        m, L, c = self.model, self.L, self.sctx
        G = self.flat_g if G is None else G
        if not dlogits.is_contiguous():
            dlogits = dlogits.contiguous()
        dx9 = self._conv_bwd(c["d_out"], m.outc.conv.weight, c["x9"], None, dlogits, self.g(m.outc.conv.weight, G),
                             self.g(m.outc.conv.bias, G))

        def double_bwd(name, dy, dy_b=None, need_dx=True):
            d = self._layer_bwd(L[name + "b"], dy, G, dy_b=dy_b)
            return self._layer_bwd(L[name + "a"], d, G, need_dx=need_dx)

        def up_bwd(name, dy, c_skip):
            dcat = double_bwd(name, dy)                                   # [.., c_skip (skip) | rest (upsampled)]
            return dcat[..., :c_skip], T.upsample2_bilinear_backward(dcat[..., c_skip:])

        dskip1, dx8 = up_bwd("up4", dx9, 64)
        dskip2, dx7 = up_bwd("up3", dx8, 128)
        dskip3, dx6 = up_bwd("up2", dx7, 256)
        dskip4, dx5 = up_bwd("up1", dx6, 512)
        dp4 = double_bwd("down4", dx5)
        # the fused map feeds the max-pool of down4 and the skip of up1
        dfused = T.add_rows(T.maxpool2_backward(c["fused"], dp4), dskip4)
        dx4 = self._fusion_bwd(dfused, G)                                 # gradient w.r.t. the agents' own x4 maps
        dp3 = double_bwd("down3", dx4)
        dp2 = double_bwd("down2", T.maxpool2_backward(c["x3"], dp3), dy_b=dskip3)
        dp1 = double_bwd("down1", T.maxpool2_backward(c["x2"], dp2), dy_b=dskip2)
        double_bwd("inc", T.maxpool2_backward(c["x1"], dp1), dy_b=dskip1, need_dx=False)
        return G


class SegTrainStep:
    """SegModule.step on the HIP path: forward (train mode) + cross entropy + reverse pass + Adam."""

    def __init__(self, model, optimizer=None, lr=1e-3):
        kw = {}
        self.optimizer = optimizer      # its param_groups[0]["lr"] is re-read every step: an lr scheduler on it is honoured
        if optimizer is not None:
            grp = optimizer.param_groups[0]
            lr = grp["lr"]
            kw = {"betas": tuple(grp.get("betas", (0.9, 0.999))), "eps": grp.get("eps", 1e-8),
                  "weight_decay": grp.get("weight_decay", 0.0)}
        self.model = model
        self.engine = SegTrainEngine(model, lr=lr, **kw)
        model.__dict__["_train_engine"] = self.engine

    def step(self, data, batch_size=None):
        """data: bev_seq [A*B, C, H, W] (NCHW, as the reference feeds it) or [A*B, H, W, C], trans_matrices,
        num_agent, labels [A*B, H, W] int.  -> {"loss": float}"""
        eng, m = self.engine, self.model
        m.training = True
        for mod in m.modules():
            mod.training = True
        bev = data["bev_seq"]
        if not bev.is_cuda:
            raise ops._lib.DnError("SegModule.step needs GPU tensors; there is no CPU path")
        x = bev.permute(0, 2, 3, 1) if bev.shape[1] == m.n_channels and bev.shape[-1] != m.n_channels else bev
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        B = x.shape[0] // m.agent_num if batch_size is None else batch_size
        if self.optimizer is not None:
            eng.lr = float(self.optimizer.param_groups[0]["lr"])
        with torch.no_grad():
            # the reference drops the images whose BEV is empty (sum <= 1e-4: the padded slots of scenes with fewer
            # live agents than num_agent) from pred and labels before the criterion.  Same loss, divisor and
            # gradients with their labels set to the ignore index.  The reference casts with labels.long() first, so
            # uint8 label maps are legal input: convert BEFORE the ignore index (-100) goes in.
            empty = x.reshape(x.shape[0], -1).sum(1) <= 1e-4
            if bool(empty.all()):
                # no live image at all: the criterion's divisor is zero (the reference would produce a NaN loss and NaN
                # gradients out of an empty batch) -- skip the step, leave the parameters and the optimizer state alone
                return {"loss": float("nan"), "skipped": True}
            logits = eng.forward(x, data["trans_matrices"], data["num_agent"], B)
            labels = data["labels"].to(device=x.device, dtype=torch.int64)
            labels = torch.where(empty.view(-1, 1, 1), torch.full_like(labels, -100), labels)
            loss, dlogits = ops.seg_ce_loss(logits, labels, want_grad=True, check_labels=False)
            eng.backward(dlogits)
            eng.allreduce_grads()
            eng.optimizer_step()
        return {"loss": float(loss)}

    def scheduler_step(self, lr=None):
        """lr for the following steps (the detector's CoDetModule.scheduler_step): an explicit value, or the passed
        optimizer's current one (already re-read every step)."""
        if lr is not None:
            self.engine.lr = float(lr)
        elif self.optimizer is not None:
            self.engine.lr = float(self.optimizer.param_groups[0]["lr"])
