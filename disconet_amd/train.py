"""Training step of the `--com disco` detector on the HIP path.

Mirrors upstream:coperception/utils/CoDetModule.py :: CoDetModule.step for the no-KD
configuration (`python train_codet.py ... --com disco`, /root/reference/README.md:54-63):

    model.train();  result = model(bev_seq, trans_matrices, num_agent, batch_size)
    loss = focal(cls, labels) / N + smooth_l1(loc, reg_targets)[reg_loss_mask] / N
    optimizer.zero_grad();  loss.backward();  optimizer.step()

Nothing here runs through autograd or ATen math: the forward is the conv engine with
batch-statistics BatchNorm kernels, the backward is an explicit reverse pass over the
(static) layer graph -- data gradients through the same MFMA conv engine with flipped /
transposed weights, weight gradients on the fp32 MFMA split-K kernel, BN / ReLU / upsample /
concat / warp / softmax backward kernels -- and Adam is one launch over a flat parameter
buffer (the module's nn.Parameters are views into it, so state_dict() / checkpoints keep
working and a DDP-style gradient all-reduce is ONE RCCL collective over 31.5 MB).

Two ways in:
  * `CoDetModule(model, ...).step(data, batch_size)` -- native loss + backward + Adam.
  * `model.train(); out = model(...); loss.backward()` -- the reference's own step code:
    DiscoNet.forward returns tensors attached to one autograd node whose backward is the
    explicit reverse pass (torch only routes d(loss)/d(cls, loc) in and parameter grads out).
"""
import os

import torch

from . import ops, train_ops as T
from .model import LAYER_CHANNEL, _bn_name

_ENC_GROUPS = (("conv_pre_1", "conv_pre_2"), ("conv1_1", "conv1_2", "conv3d_1"),
               ("conv2_1", "conv2_2", "conv3d_2"), ("conv3_1", "conv3_2"), ("conv4_1", "conv4_2"))
_EPS = 1e-5          # nn.BatchNorm default
_DGRAD_MATH_DEFAULT = "sp"        # measured (round 5): per-tensor gradient error vs the float64 oracle equal to the fp32 form's to 3 digits
_WGRAD_MATH_DEFAULT = "sp"        # the weight gradients of the layers dn_conv_wgrad_sp takes on the f16 MFMA with split operands
_WGRAD_X_LIFT = 16.0              # power-of-two lift of the activations in that kernel (post-BatchNorm maps: |x| << 4094)
_FWD_MATH_DEFAULT = "sp"          # round 6: the training forward's 3x3 / 1x1 convs on the inference engine's split-f16 LDS-DMA kernels
_MOMENTUM = 0.1


class _Layer:
    """conv (+ BatchNorm in training mode + ReLU) of the graph and what its backward needs"""

    def __init__(self, name, conv_w, conv_b, bn, ksize, stride=1):
        self.name, self.w, self.b, self.bn = name, conv_w, conv_b, bn
        self.ksize, self.stride = ksize, stride
        self.c_out, self.c_in = conv_w.shape[0], conv_w.shape[1]
        self.ctx = None


def _desc_key(d):
    """a conv descriptor's fields as a tuple (part of the key of a packed weight form: the image follows the layer's shape)"""
    return tuple(getattr(d, f) for f, _ in d._fields_)


def _desc_copy(d):
    """a private copy of a conv descriptor (launch wrappers write into theirs: math, leading dimensions)"""
    return type(d).from_buffer_copy(d)


def _param_order(model):
    """flat-buffer order: the two heads' first convs / BNs adjacent (they run as one 64-channel
    layer), then everything else in module order"""
    cls, reg = model.classification, model.regression.box_prediction
    first = [cls.conv1.weight, reg[0].weight, cls.conv1.bias, reg[0].bias,
             cls.bn1.weight, reg[1].weight, cls.bn1.bias, reg[1].bias]
    seen = {id(p) for p in first}
    return first + [p for p in model.parameters() if id(p) not in seen]


def fusion_call_counts(agents, only_v2i, num_agent_cpu, B):
    """counts[b][i] = maps in ego i's list in scene b (the ego itself + its listed neighbours; 0 for a padded agent) = calls
    of the attention MLP the reference makes for that ego.  Host-side; what the ranks of an agent-parallel step use to put
    their per-call BatchNorm statistics back into the reference's call order."""
    out = []
    for b in range(B):
        n = int(num_agent_cpu[b])
        row = []
        for i in range(agents):
            if i >= n:
                row.append(0)
            else:
                row.append(1 + sum(1 for j in range(n) if j != i and not (only_v2i and i != 0 and j != 0)))
        out.append(row)
    return out


_LIST_CACHE = {}      # (agents, only_v2i, live counts, B, device, ego range) -> the index tensors of fusion_lists (they do not depend on the poses)


class _DeviceFlag:
    """A boolean computed on the device whose host copy is read when it is first asked for: the copy into pinned memory and an
    event are queued behind the kernels that compute it, bool() waits for THAT event -- not for whatever the stream holds by
    then (an .item() where the flag is used would drain the whole stream; an .item() where it is computed stalls the launch
    queue at the start of the step, when the GPU is idle and waiting for its first conv: profiles/r06_train_gap_sites.txt)."""

    def __init__(self, flag):
        self._host = torch.empty(1, dtype=torch.bool, pin_memory=True)
        self._host.copy_(flag.reshape(1), non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._value = None

    def __bool__(self):
        if self._value is None:
            self._event.synchronize()
            self._value = bool(self._host[0])
        return self._value


def fusion_poses(out, trans):
    """adds the warps' poses (gathered from trans [B, A, A, 4, 4]) and their rigidity to the lists of fusion_lists(poses=False)"""
    nw, bi, dev = out["n_warps"], out.pop("_poses_idx"), trans.device
    if nw:
        poses = trans[bi[:, 0], bi[:, 1], bi[:, 2]].contiguous()
    else:
        poses = torch.zeros((0, 4, 4), dtype=torch.float32, device=dev)
    # rigid poses (rotation + translation: what V2X agents' relative poses are) take the deterministic
    # gather form of the warp backward
    rigid = True
    if nw:
        R = poses[:, :2, :2].double()
        eye = torch.eye(2, dtype=torch.float64, device=R.device)
        flag = (R @ R.transpose(1, 2) - eye).abs().max() < 1e-3
        rigid = _DeviceFlag(flag) if flag.is_cuda else bool(flag.item())
    out["poses"], out["rigid"] = poses, rigid
    return out


def fusion_lists(agents, only_v2i, trans, num_agent_cpu, B, dev, ego_first=0, ego_count=None, poses=True):
    """Index lists of the DiscoGraph fusion for one batch, in the reference's loop order
    (upstream DiscoNet.forward: for b, for ego i < n_b: [ego] + [warp(j -> i) for j < n_b, j != i],
    honouring only_v2i).  Images are agent-major (agent * B + b); maps / pairs = own maps then warps.
    Host-side logic only (runs on CPU tensors too).

    ego_first / ego_count (agent-parallel training: a rank fuses ITS egos against every agent's map): the lists cover the
    egos [ego_first, ego_first + ego_count) only; map / pair indices still address the buffer [all A*B maps | this rank's
    warps], `ego_out` is the LOCAL image index (i - ego_first) * B + b of the fused output.

    The index tensors depend on the live-agent counts only, not on the poses: they are built (a dozen small host -> device
    copies) once per distinct (counts, B, ego range) and cached (round 6: the step's start was the GPU's longest idle gap);
    every call gathers the poses and tests them for rigidity -- poses=False leaves that to a later fusion_poses(lists, trans)
    (the training forward queues its encoder first: the GPU is idle until the first conv arrives)."""
    key = (agents, bool(only_v2i), tuple(int(v) for v in num_agent_cpu[:B]), B, str(dev), ego_first, ego_count)
    hit = _LIST_CACHE.get(key)
    if hit is None:
        if len(_LIST_CACHE) > 64:
            _LIST_CACHE.clear()
        hit = _fusion_index_lists(agents, only_v2i, num_agent_cpu, B, dev, ego_first, ego_count)
        _LIST_CACHE[key] = hit
    out = dict(hit)
    return fusion_poses(out, trans) if poses else out


def _fusion_index_lists(agents, only_v2i, num_agent_cpu, B, dev, ego_first=0, ego_count=None):
    A = agents
    E = A if ego_count is None else ego_count
    NI = A * B
    img = lambda a, b: a * B + b
    src_image, poses_idx, warp_ego = [], [], []
    first, pair_index, map_image, ego_out = [0], [], [], []
    order = []
    for b in range(B):
        n = int(num_agent_cpu[b])
        for i in range(ego_first, ego_first + E):
            ego_out.append((i - ego_first) * B + b)
            if i >= n:
                pair_index.append(-1)
                map_image.append(img(i, b))
                first.append(len(pair_index))
                continue
            pair_index.append(img(i, b))
            map_image.append(img(i, b))
            order.append(img(i, b))
            for j in range(n):
                if j == i or (only_v2i and i != 0 and j != 0):
                    continue
                wi = len(src_image)
                src_image.append(img(j, b))
                poses_idx.append((b, i, j))
                warp_ego.append(img(i, b))
                pair_index.append(NI + wi)
                map_image.append(NI + wi)
                order.append(NI + wi)
            first.append(len(pair_index))
    nw = len(src_image)
    longest = max([first[k + 1] - first[k] for k in range(len(first) - 1)] or [0])
    if longest > 16:      # kMaxNbr of csrc/train_ops.hip :: fuse_combine kernels (per-lane weight arrays)
        raise ops._lib.DnError("training fusion: %d maps in one ego's list; the combine kernels hold at most 16" % longest)
    ego_image = list(range(NI)) + warp_ego
    # pairs per ego image, for dE = sum over the ego's pairs
    per = [[] for _ in range(NI)]
    for p, e in enumerate(ego_image):
        per[e].append(p)
    efirst = [0]
    for lst in per:
        efirst.append(efirst[-1] + len(lst))
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    bi = torch.tensor(poses_idx, dtype=torch.long, device=dev) if nw else None
    return dict(n_warps=nw, src_image=i32(src_image), first=i32(first), _poses_idx=bi,
                pair_index=i32(pair_index), map_image=i32(map_image), ego_out=i32(ego_out),
                ego_image=i32(ego_image), efirst=i32(efirst), epairs=i32([p for l in per for p in l]),
                order=i32(order), n_calls=len(order))


class TrainEngine:
    # a second stream beside the conv kernels is refused unless DISCONET_UNSAFE_OVERLAP=1 (ops.check_overlap_request)
    overlap_streams = property(lambda self: self._overlap_streams,
                               lambda self, v: setattr(self, "_overlap_streams",
                                                       ops.check_overlap_request(v, "TrainEngine.overlap_streams")))

    # the training forward may run its convs on the SP engine (fwd_math = "sp"): subclasses whose graph feeds the convs from
    # kernels that do not write SP copies (the segmentation variant's pooling / bilinear upsampling) switch it off
    _FWD_SP_OK = True

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, shard=None, dgrad_math=None,
                 wgrad_math=None, fwd_math=None):
        """fwd_math: where the training FORWARD's convs run while model.conv_math is a split-f16 mode -- "sp" (round 6: the
        inference engine's LDS-DMA kernels, dn_spconv2d_nhwc: every BatchNorm apply writes its y a second time as an SP tensor
        and the next conv reads that, writing z as fp32 rows for the statistics) or "nhwc" (rounds 2-5: the fp32-NHWC engine,
        which splits the rows on the VALU while staging them); same arithmetic (hi + lo operands, three products, fp32
        accumulation) in another summation order.  None: DISCONET_FWD_MATH, default _FWD_MATH_DEFAULT.
        wgrad_math: arithmetic of the weight gradients -- "f32" (the exact fp32 MFMA kernels) or "sp" (dn_conv_wgrad_sp where
        it takes the layer: f16 hi + lo operands split while staging, dz lifted by the same measured power of two as below; the
        other layers stay fp32); None: DISCONET_WGRAD_MATH, default _WGRAD_MATH_DEFAULT.
        dgrad_math: arithmetic of the 3x3 data gradients -- "f32" (the exact fp32 MFMA) or "sp" (the inference engine's
        split-f16 LDS-DMA kernels on a dz the BatchNorm backward writes pre-split and lifted, see _dz_sp_plan; a stride-2 layer
        as ONE launch over its four parity classes where the next BatchNorm backward can read the space-to-depth result);
        None: DISCONET_DGRAD_MATH, default _DGRAD_MATH_DEFAULT.
        shard: a sharded.AgentShard -- this process trains the agents [shard.first, shard.first + shard.count) of every
        scene (agent-parallel training): forward() / backward() then take the LOCAL agent-major images and
        exchange BatchNorm sums (all-reduce), the layer-`layer` maps (all-gather), the gradient of those maps (reduce-scatter)
        and the parameter gradients (all-reduce, summed) through it.  None: every agent lives here."""
        self.model = model
        self.shard = shard
        self.dgrad_math = dgrad_math if dgrad_math is not None else os.environ.get("DISCONET_DGRAD_MATH", _DGRAD_MATH_DEFAULT)
        if self.dgrad_math not in ("f32", "sp"):
            raise ValueError("dgrad_math must be 'f32' or 'sp' (got %r)" % (self.dgrad_math,))
        self.wgrad_math = wgrad_math if wgrad_math is not None else os.environ.get("DISCONET_WGRAD_MATH", _WGRAD_MATH_DEFAULT)
        if self.wgrad_math not in ("f32", "sp"):
            raise ValueError("wgrad_math must be 'f32' or 'sp' (got %r)" % (self.wgrad_math,))
        self.fwd_math = fwd_math if fwd_math is not None else os.environ.get("DISCONET_FWD_MATH", _FWD_MATH_DEFAULT)
        if self.fwd_math not in ("nhwc", "sp"):
            raise ValueError("fwd_math must be 'nhwc' or 'sp' (got %r)" % (self.fwd_math,))
        self._dz_lift = {}           # layer name -> (power-of-two lift of its dz, step it was measured at)
        self.f32_fallback_steps = 0  # backward passes that were re-run on the fp32 kernels after a clamped dz (backward())
        self.last_fallback_step = None
        self._force_range_flags = []  # tests: flag words OR-ed into the next range polls
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.generation = 0          # bumped by every forward(): the saved activations belong to it
        self._overlap_streams = False
        params = self._param_order(model)
        dev = params[0].device
        if dev.type != "cuda":
            raise ops._lib.DnError("TrainEngine needs the model on the GPU; there is no CPU path")
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]      # 16-byte aligned views
        total = sum(sizes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_of = {}
        off = 0
        for p, n in zip(params, sizes):
            view = self.flat_p[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.grad_of[id(p)] = (off, p.numel(), p.shape)
            off += n
        self.params = params
        self._graph()

    # ------------------------------------------------------------------
    def _param_order(self, model):      # subclasses (the segmentation variant) order their own parameters
        return _param_order(model)

    def _side_stream(self, dev):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def g(self, p, flat=None):
        off, n, shape = self.grad_of[id(p)]
        return (self.flat_g if flat is None else flat)[off:off + n].view(shape)

    def _graph(self):
        m = self.model
        enc, dec = m.u_encoder, m.decoder
        L = {}
        for name in ("conv_pre_1", "conv_pre_2", "conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1",
                     "conv3_2", "conv4_1", "conv4_2"):
            conv = getattr(enc, name)
            L[name] = _Layer(name, conv.weight, conv.bias, getattr(enc, _bn_name(name)), 3, conv.stride[0])
        for name in ("conv3d_1", "conv3d_2"):
            mod = getattr(enc, name)
            L[name] = _Layer(name, mod.conv3d.weight, mod.conv3d.bias, mod.bn3d, 1)
        for name in ("conv5_1", "conv5_2", "conv6_1", "conv6_2", "conv7_1", "conv7_2", "conv8_1", "conv8_2"):
            conv = getattr(dec, name)
            L[name] = _Layer(name, conv.weight, conv.bias, getattr(dec, _bn_name(name)), 3)
        if enc.compress_level > 0:       # 1x1 compress / decompress of the exchanged map
            L["compress"] = _Layer("compress", enc.com_compresser.weight, enc.com_compresser.bias,
                                   enc.bn_compress, 1)
            L["decompress"] = _Layer("decompress", enc.com_decompresser.weight, enc.com_decompresser.bias,
                                     enc.bn_decompress, 1)
        f = m.pixel_weighted_fusion
        for i, (cname, bname) in enumerate((("conv1_2", "bn1_2"), ("conv1_3", "bn1_3")), 2):
            conv = getattr(f, cname)
            L["mlp%d" % i] = _Layer("mlp%d" % i, conv.weight, conv.bias, getattr(f, bname), 1)
        self.L = L

    # ------------------------------------------------------------------
    # one conv + BN(train) + ReLU
    # ------------------------------------------------------------------
    def _math(self):
        return ops.nhwc_math(self.model.conv_math)      # the training kernels take fp32 NHWC

    def _conv(self, w, bias, src0, src1=None, up0=0, stride=1, ksize=3, out=None, h_in=None, w_in=None, lift_of=None, cut=None):
        """raw conv + bias through the forward engine (weights packed on the fly).  lift_of: the Parameter whose
        power-of-two lift applies when `w` is a temporary cut out of it; cut = (Parameter, first column) with w = None: the 1x1
        conv over the columns cut[1] .. + c0 of that weight (the attention MLP's W1 halves) -- its pack joins the step's
        one-launch packs, which read the columns in place"""
        n, h0, w0 = src0.shape[0], src0.shape[1], src0.shape[2]
        c0 = src0.shape[3]
        if h_in is None:
            h_in, w_in = (h0 * 2, w0 * 2) if up0 else (h0, w0)
        c1 = src1.shape[3] if src1 is not None else 0
        if cut is not None and w is None:       # the columns cut[1] .. + c0 of a 1x1 weight, copied out only if a single pack needs them
            c_out, lift_of = cut[0].shape[0], cut[0]
        else:
            c_out = w.shape[0]
        d = ops.conv_desc(n, h_in, w_in, c0, c_out, ksize, stride, False, c1=c1, up0=up0,
                          ld0=src0.stride(2), ld1=src1.stride(2) if src1 is not None else None,
                          ldo=out.stride(2) if out is not None else None, math=self._math())
        dev = src0.device
        sp0, sp1 = self._sp_copy(src0), (self._sp_copy(src1) if src1 is not None else None)
        if (self._sp_forward() and d.math == 1 and sp0 is not None and (src1 is None or sp1 is not None) and lift_of is None
                and c_out % 4 == 0 and (c1 == 0 or c0 % 16 == 0)):
            # the inference engine's kernels on the SP copies the BatchNorm applies wrote (d stays the fp32 tensors' descriptor:
            # the backward's weight / data gradients read those)
            ds = ops.conv_desc(n, h_in, w_in, c0, c_out, ksize, stride, False, c1=c1, up0=up0)
            wmul = self._wmul_of(w)
            packed = self._packed_form(w, 0, ds, 0, 0, lambda: ops.sp_pack_conv_weights(ds, w, wmul)[0], lift=w)
            if out is None:
                ho, wo = ops.conv_out_hw(ds)
                out = torch.empty((n, ho, wo, c_out), dtype=torch.float32, device=dev)
            ops.sp_conv2d_nhwc(ds, sp0, packed, self._const(dev, c_out, 1.0 / wmul),
                               bias if bias is not None else self._const(dev, c_out, 0.0), out, src1=sp1)
            return out, d
        wmul = self._wmul_of(w if lift_of is None else lift_of) if d.math == 1 else 1.0
        def single():
            wc = w if w is not None else cut[0].detach().reshape(c_out, -1)[:, cut[1]:cut[1] + c0].contiguous().view(c_out, c0, 1, 1)
            return ops.pack_conv_weights(d, wc if wmul == 1.0 else wc.detach() * wmul)
        if cut is not None:       # w = cut[0][:, cut[1] : cut[1] + c_in]: the one-launch pack reads the columns in place
            packed = self._packed_form(cut[0], 0, d, cut[1], 0, single, engine="nhwc", lift=lift_of if d.math == 1 else None)
        else:
            packed = self._packed_form(w, 0, d, 0, 0, single, engine="nhwc", lift=w if d.math == 1 else None)
        one = self._const(dev, c_out, 1.0 / wmul)
        shift = bias if bias is not None else self._const(dev, c_out, 0.0)
        if out is None:
            ho, wo = ops.conv_out_hw(d)
            out = torch.empty((n, ho, wo, c_out), dtype=torch.float32, device=dev)
        ops.conv2d(d, src0, packed, one, shift, src1=src1, out=out)
        return out, d

    def _sp_forward(self):
        return self._FWD_SP_OK and self.fwd_math == "sp"

    @staticmethod
    def _sp_copy(t):
        """the SP twin of an fp32 NHWC activation, where a BatchNorm apply (or _attach_sp) left one"""
        return getattr(t, "_dn_sp", None)

    def _attach_sp(self, t):
        """an activation that no BatchNorm apply of this engine wrote (the input voxels, the fused map): one dn_sp_from_nhwc pass"""
        if self._sp_forward() and self._math() == 1 and self._sp_copy(t) is None:
            t._dn_sp = ops.SpTensor.from_nhwc(t)
        return t

    def _wmul_of(self, w):
        """power-of-two lift of a layer's weights for the split-f16 forward (ops._pow2_lift: keeps the lo halves
        of small weights out of the f16 subnormal range; 1 / wmul rides in the conv's scale vector).  The max |w|
        behind it is read back from the device (a host sync): cached per PARAMETER -- its offset in the flat buffer, a
        key that survives allocator address reuse; temporaries cut out of a parameter pass the parent (`lift_of`) --
        and refreshed when the step count has moved 64 either way (a resumed run may step backwards); load_state_dict
        drops the cache."""
        ent = self.grad_of.get(id(w))
        key = ("p", ent[0]) if ent is not None else ("t", w.data_ptr(), tuple(w.shape), w._version)
        cache = self.__dict__.setdefault("_wmul_cache", {})
        hit = cache.get(key)
        if hit is None or abs(self.step_count - hit[1]) >= 64:
            if ent is None and len(cache) > 256:      # temporaries without a parent: do not let the table grow
                cache.clear()
            hit = (ops._pow2_lift(w), self.step_count)
            cache[key] = hit
        return hit[0]

    def _const(self, dev, n, val, cache={}):
        key = (str(dev), n, val)
        if key not in cache:
            cache[key] = torch.full((n,), val, dtype=torch.float32, device=dev)
        return cache[key]

    def _layer_fwd(self, lay, src0, src1=None, up0=0, groups=1, w=None, b=None, gamma=None, beta=None,
                   y_out=None, want_sp=True):
        """want_sp: a conv of this engine reads y (write its SP twin with the BatchNorm apply when the forward runs on the SP
        engine); False where only other kernels do (the heads' hidden layer feeds two 1x1 convs on channel slices)"""
        w = lay.w if w is None else w
        b = lay.b if b is None else b
        z, d = self._conv(w, b, src0, src1, up0, lay.stride, lay.ksize)
        # one process, one group, the layer's own BatchNorm: the running-statistics update rides in the statistics' finish launch
        # (the momentum updates of different BatchNorms touch different buffers: their order is free)
        fused_running = (lay.bn is not None and groups == 1 and (self.shard is None or self.shard.world == 1)
                         and os.environ.get("DN_BN_FUSED_RUNNING", "1") != "0")
        mean, var = T.bn_stats(z, groups, running=(lay.bn.running_mean, lay.bn.running_var, _MOMENTUM) if fused_running else None,
                               **self._bn_sync(z, groups))
        gamma = lay.bn.weight if gamma is None else gamma
        beta = lay.bn.bias if beta is None else beta
        # the backward's ReLU gate as one byte per four channels: its two passes then do not read y (1/16 of the bytes)
        mask = torch.empty(z.numel() // 4, dtype=torch.uint8, device=z.device) if z.shape[-1] % 4 == 0 else None
        y_sp = None
        if (want_sp and self._sp_forward() and self._math() == 1 and mask is not None and T.bn_apply_sp_supported(z, groups)):
            y_sp = ops.SpTensor(z.shape[0], z.shape[1], z.shape[2], z.shape[3], device=z.device)
        y = T.bn_apply(z, mean, var, gamma, beta, _EPS, relu=True, out=y_out, relu_mask=mask, sp_out=y_sp)
        if y_sp is not None:
            y._dn_sp = y_sp
        lay.ctx = dict(src0=src0, src1=src1, up0=up0, z=z, y=y, mean=mean, var=var, desc=d,
                       groups=groups, w=w, gamma=gamma, mask=mask, running_done=fused_running)
        return y

    def _bn_sync(self, z, groups):
        """keyword arguments that make a BatchNorm reduction span every rank's images (agent-parallel training): the batch of
        a layer's BatchNorm is all A*B images, of which this rank holds count*B -- its sums are all-reduced and normalised by
        the global row count.  Grouped statistics (the attention MLP: one (ego, neighbour) pair per call) are rank-local."""
        if self.shard is None or self.shard.world == 1 or groups != 1:
            return {}
        return {"sync": self.shard.sum_, "norm_rows": (z.numel() // z.shape[-1]) * self.shard.world}

    def _update_running(self, bn, mean, var, rows, order=None, calls=1, done=False):
        """done: the momentum update already ran inside the statistics' finish launch (_layer_fwd); only the call counter is left"""
        if not done:
            if self.shard is not None and order is None:
                rows = rows * self.shard.world            # the unbiased variance's n / (n - 1) is the global batch's
            T.bn_update_running(mean, var, rows, bn.running_mean, bn.running_var, _MOMENTUM, order)
        if calls == 1 and getattr(self, "_tracked", None) is not None:
            self._tracked.append(bn.num_batches_tracked)      # forward() adds 1 to all of them in one launch
        else:
            bn.num_batches_tracked.add_(calls)

    def _layer_bwd(self, lay, dy_a, G, dy_b=None, up_a=False, need_dx=True, gw=None, gb=None,
                   ggamma=None, gbeta=None, s2d_ok=False):
        """-> gradient w.r.t. the layer's (concatenated) input, [n, h_in, w_in, c_in].
        s2d_ok: the caller hands the result to another _layer_bwd as dy_a and to nothing else -- a stride-2 layer may then
        return its data gradient as the SPACE-TO-DEPTH image [n, h_in / 2, w_in / 2, 4 c_in] (marked `_dn_s2d`; one split-f16
        launch over the four parity classes, _dgrad), which the next BatchNorm backward reads in place (up_a = 2)."""
        c = lay.ctx
        if getattr(dy_a, "_dn_s2d", False):
            assert not up_a, "a space-to-depth gradient is not also an upsampled one"
            up_a = 2
        gw = self.g(lay.w, G) if gw is None else gw
        gb = self.g(lay.b, G) if gb is None else gb
        ggamma = self.g(lay.bn.weight, G) if ggamma is None else ggamma
        gbeta = self.g(lay.bn.bias, G) if gbeta is None else gbeta
        sp, lift = self._dz_sp_plan(lay, c, need_dx, s2d_ok)
        wsp = self._wgrad_sp_layer(c)
        ent = self._dz_lift.get(lay.name)
        # the bias gradient (sum of dz per channel) from the launch that writes dz, where its one-group fast form runs
        fused_bias = gb is not None and T.bn_backward_bias_supported(c["z"], c["groups"])
        # the fp32 copy of dz is written only where something reads it: with the data gradient and the weight gradient on the SP
        # copy (dn_conv_wgrad_sp_z), the bias gradient out of the same launch and no re-measurement of the lift due this step,
        # nothing does -- a quarter of the launch's bytes (DN_TRAIN_DZ_SP_ONLY=0: always written)
        sp_only = (sp is not None and fused_bias and wsp and ent is not None and abs(self.step_count - ent[1]) < 64
                   and os.environ.get("DN_TRAIN_DZ_SP_ONLY", "1") != "0")
        dz = T.bn_backward(dy_a, c["y"], c["z"], c["mean"], c["var"], c["gamma"], _EPS, ggamma, gbeta,
                           relu=True, dy_b=dy_b, up_a=up_a, sp_out=sp, sp_lift=lift, relu_mask=c.get("mask"),
                           dbias=gb if fused_bias else None, folds=self.__dict__.get("_folds_active") if fused_bias else None,
                           want_dz=not sp_only, **self._bn_sync(c["z"], c["groups"]))
        if (lift is not None or wsp) and dz is not None:
            self._dz_lift_refresh(lay, dz)
        return self._conv_bwd(c["desc"], c["w"], c["src0"], c["src1"], dz, gw, None if fused_bias else gb, need_dx,
                              dz_sp=sp, dz_lift=lift, wgrad_lift=ent[0] if (wsp and ent is not None) else None)

    def _wgrad_sp_layer(self, c):
        """does this layer's weight gradient run on the split-f16 kernel (wgrad_math = "sp")?  Like the data gradient it needs the
        layer's measured lift: the first step (and a layer whose lift was dropped) runs the fp32 kernel and measures."""
        return self.wgrad_math == "sp" and c["groups"] == 1 and T.conv_wgrad_sp_supported(c["desc"])

    def _dz_sp_plan(self, lay, c, need_dx, s2d_ok=False):
        """-> (SpTensor that shall receive dz * lift, lift) when this layer's data gradient runs on the split-f16 engine, else
        (None, None) / (None, 1.0) while the lift is still unknown.  The engine's operands are f16 hi + lo pairs: 2^-22 relative
        only while 2^-3 <= |x| <= 65504, and a gradient's magnitude is anything -- so dz is LIFTED by a power of two that puts its
        largest element near 2^8 (19 binades of full precision below it, 256 x of head room above; include/disconet_train.h ::
        dn_bn_train_backward_finish_sp).  The lift of a layer is measured (max |dz|, a host read) on the first step, which runs
        that layer's data gradient in fp32, and again every 64 steps; a dz that outgrows it is clamped AND flagged (the range
        guard polled at the end of the pass: backward() then drops the lifts and repeats the pass on the fp32 kernels)."""
        d = c["desc"]
        if (self.dgrad_math != "sp" or not need_dx or d.ksize != 3 or c["groups"] != 1
                or d.c_out % 16 != 0 or (d.c0 + d.c1) % 4 != 0):
            return None, None
        if d.stride != 1 and not (s2d_ok and d.stride == 2 and d.c1 == 0 and not d.up0 and d.h_in % 2 == 0 and d.w_in % 2 == 0
                                  and os.environ.get("DN_DGRAD_S2D", "1") != "0"):
            return None, None       # (a stride-2 layer: only as the one-launch space-to-depth form, where the caller can take it)
        ent = self._dz_lift.get(lay.name)
        if ent is None:
            return None, 1.0            # not measured yet: fp32 this step, _dz_lift_refresh measures
        z = c["z"]
        return ops.SpTensor(z.shape[0], z.shape[1], z.shape[2], z.shape[3], device=z.device), ent[0]

    def _dz_lift_refresh(self, lay, dz):
        ent = self._dz_lift.get(lay.name)
        if ent is not None and abs(self.step_count - ent[1]) < 64:
            return
        m = float(dz.abs().max())        # host read: first step and every 64th only
        if self.shard is not None and self.shard.world > 1:
            t = torch.tensor([m], dtype=torch.float64, device=dz.device)
            m = float(self.shard.max_(t)[0])         # one lift for all ranks: the replicas must stay bit-identical
        if not (m > 0.0) or m != m or m == float("inf"):
            self._dz_lift.pop(lay.name, None)
            return
        import math
        self._dz_lift[lay.name] = (float(2.0 ** max(-100, min(100, 8 - math.floor(math.log2(m))))), self.step_count)

    def _conv_bwd(self, d, w, src0, src1, dz, gw, gb, need_dx=True, dw_cin_total=0, w_ci_first=0,
                  w_c_in=None, dx_out=None, dz_sp=None, dz_lift=None, wgrad_lift=None):
        # (the weight gradient reads dz from its SP copy where there is one with the same lift: the same bits, less staging work)
        T.conv_wgrad(d, src0, src1, dz, gw, dw_cin_total=dw_cin_total, sp_lift=wgrad_lift, x_lift=_WGRAD_X_LIFT,
                     dz_sp=dz_sp if (wgrad_lift is not None and dz_sp is not None and wgrad_lift == dz_lift and d.c_out % 16 == 0
                                     and os.environ.get("DN_TRAIN_WGRAD_ZSP", "1") != "0") or dz is None else None)
        if gb is not None:
            T.channel_sum(dz, gb, folds=self.__dict__.get("_folds_active"))
        if not need_dx:
            return None
        return self._dgrad(d, w, dz, w_ci_first, w_c_in, dx_out, dz_sp, dz_lift)

    def _dgrad(self, d, w, dz, ci_first=0, c_in=None, dx_out=None, dz_sp=None, dz_lift=None):
        """data gradient = a forward engine on dz with flipped / transposed weights.
        fp32 form (dgrad_math = "f32", and always: 1x1 layers, the parity-phase stride-2 form, a layer whose lift is not
        measured yet): the exact-fp32 MFMA.  This backward amplifies relative error by ~1e5 (BatchNorm's mean subtraction,
        tests/test_gpu_train_step.py), and the NHWC engine's split-f16 mode, which splits dz as it stages it WITHOUT a lift,
        measured 3.6 % error on conv5_1.weight's gradient against fp32's 1 % (rounds 2-4: most of a small gradient's lo halves
        are f16 subnormals there).
        split-f16 form (dz_sp given: dz * dz_lift pre-split by the BatchNorm backward): the inference engine's LDS-DMA kernels
        (dn_spconv2d_nhwc), 1 / (dz_lift * wmul) in the scale vector."""
        w4 = w.reshape(w.shape[0], w.shape[1], d.ksize, d.ksize)
        if dz_sp is not None and d.stride == 2:
            # all four parity classes of the stride-2 layer's data gradient (include/disconet_train.h ::
            # dn_conv_dgrad_class_weights) as ONE stride-1 launch of the split-f16 engine over dz: the classes are the output
            # channel groups, [n, h_in / 2, w_in / 2, 4 c_in] -- the space-to-depth image of dx, read in place by the next
            # BatchNorm backward (up_a = 2).  27 of its 36 (class, tap) weight blocks are zero: 4 x the MFMAs the masked fp32
            # form runs, on an engine 16 x as fast per MFMA -- and dz is read once instead of four times.
            n_in = (w4.shape[1] - ci_first) if c_in is None else c_in
            dev = dz_sp.data.device
            dd = ops.conv_desc(d.n_images, d.h_in // 2, d.w_in // 2, d.c_out, 4 * n_in, 3, 1, False)
            wmul = self._wmul_of(w)

            def pack_classes():
                wt = torch.empty((4 * n_in, d.c_out, 3, 3), dtype=torch.float32, device=dev)
                for py in (0, 1):
                    for px in (0, 1):
                        k = py * 2 + px
                        T.dgrad_class_weights(w4, py, px, ci_first, n_in, out=wt[k * n_in:(k + 1) * n_in])
                return ops.sp_pack_conv_weights(dd, wt, wmul)[0]
            packed = self._packed_form(w, 2, dd, ci_first, n_in, pack_classes, lift=w)
            out = torch.empty((d.n_images, d.h_in // 2, d.w_in // 2, 4 * n_in), dtype=torch.float32, device=dev)
            ops.sp_conv2d_nhwc(dd, dz_sp, packed, self._const(dev, 4 * n_in, 1.0 / (dz_lift * wmul)),
                               self._const(dev, 4 * n_in, 0.0), out)
            out._dn_s2d = True
            return out
        if dz_sp is not None:
            n_in = (w4.shape[1] - ci_first) if c_in is None else c_in
            dev = dz_sp.data.device
            dd = ops.conv_desc(d.n_images, d.h_in, d.w_in, d.c_out, n_in, 3, 1, False)
            wmul = self._wmul_of(w)
            packed = self._packed_form(w, 1, dd, ci_first, n_in,
                                       lambda: ops.sp_pack_conv_weights(dd, T.dgrad_weights(w4, ci_first, n_in), wmul)[0], lift=w)
            if dx_out is None:
                dx_out = torch.empty((d.n_images, d.h_in, d.w_in, n_in), dtype=torch.float32, device=dev)
            ops.sp_conv2d_nhwc(dd, dz_sp, packed, self._const(dev, n_in, 1.0 / (dz_lift * wmul)), self._const(dev, n_in, 0.0),
                               dx_out)
            return dx_out
        if (d.stride == 2 and d.ksize == 3 and d.h_in % 2 == 0 and d.w_in % 2 == 0
                and os.environ.get("DN_DGRAD_PARITY", "1") != "0"):
            # parity-phase form: four stride-1 convs over dz of 1 / 2 / 2 / 4 taps, each writing one parity class of dx
            # (include/disconet_train.h :: dn_conv_dgrad_class_weights) -- a quarter of the zero-stuffed form's MFMAs
            n_in = (w4.shape[1] - ci_first) if c_in is None else c_in
            dev = dz.device
            if dx_out is None:
                dx_out = torch.empty((d.n_images, d.h_in, d.w_in, n_in), dtype=torch.float32, device=dev)
            ho, wo = d.h_in // 2, d.w_in // 2
            dd = ops.conv_desc(d.n_images, ho, wo, d.c_out, n_in, 3, 1, False, ld0=dz.stride(2), math=0)
            one, zero = self._const(dev, n_in, 1.0), self._const(dev, n_in, 0.0)
            for py in (0, 1):
                for px in (0, 1):
                    v, mask = T.dgrad_class_weights(w4, py, px, ci_first, n_in)
                    ops.conv2d_taps(dd, dz, ops.pack_conv_weights(dd, v), one, zero, dx_out[:, py::2, px::2, :], mask)
            return dx_out
        c_in = (w4.shape[1] - ci_first) if c_in is None else c_in
        dd = ops.conv_desc(d.n_images, d.h_in, d.w_in, d.c_out, c_in, d.ksize, 1, False,
                           up0=2 if d.stride == 2 else 0, ld0=dz.stride(2),
                           ldo=dx_out.stride(2) if dx_out is not None else None, math=0)
        packed = self._packed_form(w, 1, dd, ci_first, c_in, lambda: ops.pack_conv_weights(dd, T.dgrad_weights(w4, ci_first, c_in)),
                                   engine="nhwc")
        dev = dz.device
        if dx_out is None:
            dx_out = torch.empty((d.n_images, d.h_in, d.w_in, c_in), dtype=torch.float32, device=dev)
        ops.conv2d(dd, dz, packed, self._const(dev, c_in, 1.0), self._const(dev, c_in, 0.0), out=dx_out)
        return dx_out

    # ------------------------------------------------------------------
    # fusion lists (host side, from num_agent / only_v2i)
    # ------------------------------------------------------------------
    def _fusion_lists(self, trans, num_agent_cpu, B, dev, poses=True):
        sh = self.shard
        return fusion_lists(self.model.agent_num, self.model.only_v2i, trans, num_agent_cpu, B, dev,
                            ego_first=sh.first if sh is not None else 0, ego_count=sh.count if sh is not None else None, poses=poses)

    # ------------------------------------------------------------------
    # the step's packed weights in one launch
    # ------------------------------------------------------------------
    # A step packs every 3x3 layer's weights for the split-f16 engine twice -- as they are for the forward, flipped and transposed
    # (or as the four parity classes of a stride-2 layer) for the data gradient: ~70 launches of 4-7 us plus the flips' ~35, on a
    # stream with nothing to run beside them and a host that needs ~15 us per launch (profiles/r06_step_start_ab.txt).  A form
    # that was asked for in an earlier forward is a job of ops.SpPackSet: all jobs are packed by ONE launch at the start of each
    # forward (dn_spconv_pack_weights_multi: the same bytes), and _conv / _dgrad pick the images up.  A form not in the set
    # (yet), a tap-merged layer, a temporary weight, DN_TRAIN_PACK_MULTI=0: the single launches, as before.
    def _pack_multi_on(self):
        return os.environ.get("DN_TRAIN_PACK_MULTI", "1") != "0"

    def _pack_multi(self):
        """at the start of a forward (after `generation` moved): pack every known form from the parameters as they are now --
        one launch per conv engine"""
        if not self._pack_multi_on():
            return
        for engine, ps in self.__dict__.get("_packset", {}).items():
            if ps["pending"]:
                jobs = dict(ps["jobs"])
                jobs.update(ps["pending"])
                # forms nobody asked for in the last 8 forwards leave the set (a change of dgrad_math, of the batch shape)
                jobs = {k: j for k, j in jobs.items() if self.generation - ps["used"].get(k, self.generation) <= 8}
                ps["pending"] = {}
                ps["jobs"] = jobs
                keys = list(jobs)
                ps["set"] = ops.PackSet([jobs[k][:6] for k in keys], self.flat_p.device, engine) if keys else None
                ps["image"] = {k: b for k, b in zip(keys, ps["set"].buffers)} if keys else {}
                ps["lifts"] = [jobs[k][6] for k in keys]
            if ps["set"] is not None:
                ps["set"].run([self._wmul_of(w) if w is not None else 1.0 for w in ps["lifts"]])
                ps["generation"] = self.generation

    def _packed_form(self, w, mode, dd, ci_first, n_in, single, engine="sp", lift=None):
        """the packed image, for the conv `dd` on `engine`, of Parameter `w`'s weight form (mode, ci_first, n_in: ops.PackSet;
        lift: the Parameter whose power-of-two lift is multiplied in, None = 1): this forward's one-launch image when the form is
        in the set, else `single()` (the per-layer launches) -- and a place in the next set"""
        ent = self.grad_of.get(id(w))
        off = ent[0] if ent is not None else getattr(w, "_dn_flat_off", None)      # a Parameter, or a labelled view of the flat buffer
        if off is None or not self._pack_multi_on():
            return single()
        ps = self.__dict__.setdefault("_packset", {}).setdefault(
            engine, {"jobs": {}, "pending": {}, "used": {}, "image": {}, "single": set(), "set": None, "generation": -1})
        key = (off, mode, ci_first, n_in, lift is not None) + _desc_key(dd)
        ps["used"][key] = self.generation
        if ps["generation"] == self.generation:
            img = ps["image"].get(key)
            if img is not None:
                return img
        if key not in ps["jobs"] and key not in ps["pending"] and key not in ps["single"]:
            w3 = w.detach().reshape(w.shape[0], w.shape[1], -1)
            if ops.PackSet.supported(dd, engine) and w3.is_contiguous() and w3.dtype == torch.float32:
                ps["pending"][key] = (_desc_copy(dd), w3, mode, w3.shape[1], ci_first, n_in, lift)
            else:
                ps["single"].add(key)       # a tap-merged layer: its own pack kernel
        return single()

    # ------------------------------------------------------------------
    # forward (training mode)
    # ------------------------------------------------------------------
    def check_aliasing(self):
        """The module's Parameters are views into flat_p (what Adam updates).  model.to() / .float() /
        load_state_dict(assign=True) re-point p.data and silently break that: fail loudly instead."""
        base, end = self.flat_p.data_ptr(), self.flat_p.data_ptr() + 4 * self.flat_p.numel()
        for p in self.params:
            off = self.grad_of[id(p)][0]
            if p.data_ptr() != base + 4 * off or not (base <= p.data_ptr() < end):
                raise RuntimeError(
                    "a Parameter of the model no longer aliases the training engine's flat buffer "
                    "(model.to()/.float()/load_state_dict(assign=True) after CoDetModule/TrainEngine was "
                    "built?); rebuild the CoDetModule / TrainEngine after moving or re-assigning parameters")

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size):
        self.check_aliasing()
        self.generation += 1
        self._pack_multi()
        m, L = self.model, self.L
        A, B = m.agent_num, batch_size
        if m.layer != 3 and m.u_encoder.compress_level > 0:
            raise NotImplementedError("compress_level > 0 in training needs layer = 3")
        n = bevs.shape[0] * bevs.shape[1]
        sh = self.shard
        A_loc = A if sh is None else sh.count            # agents whose images this rank holds (agent-parallel training)
        if n != A_loc * B:
            raise ValueError("bevs has %d images, expected %d (this rank's agents) * batch_size %d = %d"
                             % (n, A_loc, B, A_loc * B))
        dev = bevs.device
        x = bevs.reshape(n, bevs.shape[2], bevs.shape[3], bevs.shape[4])
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        self._attach_sp(x)
        trans = trans_matrices.to(device=dev, dtype=torch.float32).contiguous()
        self._num_agent_cpu, self._batch = num_agent_tensor[:, 0].cpu(), B
        F = self._fusion_lists(trans, self._num_agent_cpu, B, dev, poses=False)     # the poses: once the encoder is queued
        self.F = F

        # encoder: groups of layers, the last one of group k yields e[k]; the maps of the fusion layer
        # are written straight into the pair buffer [own maps | warped maps]
        lay_k = m.layer
        C = LAYER_CHANNEL[lay_k]
        hk, wk = x.shape[1] >> lay_k, x.shape[2] >> lay_k
        # pair buffer [every agent's map (A * B, agent-major) | this rank's warps]; this rank's own maps are rows
        # [lo, lo + n) of it -- all of it without a shard -- and the other ranks' arrive by the all-gather below
        NI, NW = A * B, F["n_warps"]
        lo = 0 if sh is None else sh.first * B
        maps = torch.empty((NI + NW, hk, wk, C), dtype=torch.float32, device=dev)
        own = maps[lo:lo + n]
        def group(k, a):
            for name in _ENC_GROUPS[k]:
                into = own if (name == _ENC_GROUPS[k][-1] and k == lay_k and "compress" not in L) else None
                a = self._layer_fwd(L[name], a, y_out=into)
            return a

        e, a = [], x
        for k in range(lay_k + 1):
            a = group(k, a)
            e.append(a)
        fusion_poses(F, trans)
        # the encoder levels above the exchanged one could run beside the fusion block on a second
        # stream (self.overlap_streams, default off and meant to stay off: no kernel-source rule rules out
        # the co-residency defect of DESIGN.md 3.6 (B)), else in stream order
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if self.overlap_streams else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            up = a
            for k in range(lay_k + 1, 5):
                up = group(k, up)
                e.append(up)
        if "compress" in L:
            # the encoder goes on from the uncompressed x3; what is exchanged (and what the decoder's
            # skip sees) goes through the 1x1 compress / decompress pair
            self._layer_fwd(L["decompress"], self._layer_fwd(L["compress"], e[3]), y_out=own)
        if sh is not None:
            sh.gather_rows(maps[:NI], lo, n)             # the V2X exchange: every agent's layer-`layer` map on every rank
        fused = self._fusion_fwd(maps, NI, NW, F)
        main.wait_stream(side)
        for t in e[lay_k + 1:]:
            t.record_stream(main)
        sk = list(e)
        sk[lay_k] = self._attach_sp(fused)                    # the decoder sees the fused map
        a = self._layer_fwd(L["conv5_1"], sk[4], sk[3], up0=1)
        x5 = self._layer_fwd(L["conv5_2"], a)
        a = self._layer_fwd(L["conv6_1"], x5, sk[2], up0=1)
        x6 = self._layer_fwd(L["conv6_2"], a)
        a = self._layer_fwd(L["conv7_1"], x6, sk[1], up0=1)
        x7 = self._layer_fwd(L["conv7_2"], a)
        a = self._layer_fwd(L["conv8_1"], x7, sk[0], up0=1)
        x8 = self._layer_fwd(L["conv8_2"], a)
        x3f = fused

        # heads: both first convs as one 64-channel layer (their parameters are adjacent in the
        # flat buffer), then the two 1x1 prediction convs on the halves
        cls, reg = m.classification, m.regression.box_prediction
        po, _, _ = self.grad_of[id(cls.conv1.weight)]
        w1 = self.flat_p[po:po + 2 * cls.conv1.weight.numel()].view(64, 32, 3, 3)
        w1._dn_flat_off = po                              # (its place in the flat buffer: the key of its packed forms)
        bo, _, _ = self.grad_of[id(cls.conv1.bias)]
        b1 = self.flat_p[bo:bo + 64]
        go, _, _ = self.grad_of[id(cls.bn1.weight)]
        gamma = self.flat_p[go:go + 64]
        beo, _, _ = self.grad_of[id(cls.bn1.bias)]
        beta = self.flat_p[beo:beo + 64]
        self.head1 = _Layer("heads1", w1, b1, None, 3)
        h1 = self._layer_fwd(self.head1, x8, w=w1, b=b1, gamma=gamma, beta=beta, want_sp=False)
        cls_out, dc = self._conv(cls.conv2.weight, cls.conv2.bias, h1[..., :32], ksize=1)
        loc_out, dr = self._conv(reg[3].weight, reg[3].bias, h1[..., 32:], ksize=1)
        self.head_ctx = dict(h1=h1, dc=dc, dr=dr)

        # running statistics (momentum updates in the reference's call order)
        self._tracked = []
        for lay in L.values():
            if lay.name.startswith("mlp"):
                continue
            c = lay.ctx
            self._update_running(lay.bn, c["mean"], c["var"], c["z"].numel() // c["z"].shape[-1], done=c.get("running_done", False))
        hc = self.head1.ctx
        rows = (hc["z"].numel() // 64) * (sh.world if sh is not None else 1)
        T.bn_update_running(hc["mean"][:, :32], hc["var"][:, :32], rows, cls.bn1.running_mean,
                            cls.bn1.running_var, _MOMENTUM)
        T.bn_update_running(hc["mean"][:, 32:], hc["var"][:, 32:], rows, reg[1].running_mean,
                            reg[1].running_var, _MOMENTUM)
        self._tracked += [cls.bn1.num_batches_tracked, reg[1].num_batches_tracked]
        torch._foreach_add_(self._tracked, 1)      # every layer's call counter in one launch (25 one-element adds before)
        self._tracked = []

        nI, h, w = cls_out.shape[0], cls_out.shape[1], cls_out.shape[2]
        self.outs = dict(x5=x5, x6=x6, x7=x7, x8=x8, fused=x3f)
        self.last_result = {"loc": loc_out.view(nI, h, w, m.anchor_num_per_loc, m.out_seq_len, m.box_code_size),
                            "cls": cls_out.view(nI, -1, m.category_num)}
        return self.last_result

    def _fusion_fwd(self, maps, NI, NW, F):
        m, L = self.model, self.L
        f = m.pixel_weighted_fusion
        C = maps.shape[-1]
        P = NI + NW
        hw = maps.shape[1] * maps.shape[2]
        if NW:
            T.warp_list(maps, F["poses"], F["src_image"], out=maps[NI:])
        # W1 = [W_ego | W_nbr]: two 1x1 convs over column cuts of one weight
        E, d_e = self._conv(None, None, maps[:NI], ksize=1, cut=(f.conv1_1.weight, 0))
        z1, d_f = self._conv(None, f.conv1_1.bias, maps, ksize=1, cut=(f.conv1_1.weight, C))
        T.pair_add_ego(z1, E, F["ego_image"])
        mean1, var1 = T.bn_stats(z1, P)
        h1 = T.bn_apply(z1, mean1, var1, f.bn1_1.weight, f.bn1_1.bias, _EPS, relu=True)
        h2 = self._layer_fwd(L["mlp2"], h1, groups=P)
        h3 = self._layer_fwd(L["mlp3"], h2, groups=P)
        z4, d4 = self._conv(f.conv1_4.weight, f.conv1_4.bias, h3, ksize=1)
        n_ego = F["ego_out"].numel()                          # fused maps of this rank's egos (all of them without a shard)
        fused = torch.empty((n_ego,) + tuple(maps.shape[1:]), dtype=torch.float32, device=maps.device)
        weights = T.fuse_combine(z4, maps, F["first"], F["pair_index"], F["map_image"], F["ego_out"], fused)
        self.fctx = dict(maps=maps, NI=NI, NW=NW, z1=z1, mean1=mean1, var1=var1, h1=h1, d_e=d_e, d_f=d_f,
                         h3=h3, z4=z4, d4=d4, weights=weights)
        stats = ((f.bn1_1, mean1, var1), (f.bn1_2, L["mlp2"].ctx["mean"], L["mlp2"].ctx["var"]),
                 (f.bn1_3, L["mlp3"].ctx["mean"], L["mlp3"].ctx["var"]))
        if self.shard is not None and self.shard.world > 1:
            # the reference updates the MLP's running statistics once per CALL, in its loop order (scene, then ego): every
            # rank owns some egos' calls, so the per-call statistics travel (all-gather) and every rank replays all of them
            calls = self.shard.calls_in_reference_order(
                fusion_call_counts(m.agent_num, m.only_v2i, self._num_agent_cpu, self._batch), F["n_calls"])
            if calls["total"]:
                local = F["order"][:F["n_calls"]].long()
                for bn, mean, var in stats:
                    allm = self.shard.gather_padded(mean[local], calls["max_per_rank"])
                    allv = self.shard.gather_padded(var[local], calls["max_per_rank"])
                    T.bn_update_running(allm.reshape(-1, mean.shape[1]), allv.reshape(-1, var.shape[1]), hw, bn.running_mean,
                                        bn.running_var, _MOMENTUM, calls["index"].to(mean.device))
                    bn.num_batches_tracked += calls["total"]
        elif F["n_calls"]:
            for bn, mean, var in stats:
                T.bn_update_running(mean, var, hw, bn.running_mean, bn.running_var, _MOMENTUM,
                                    F["order"][:F["n_calls"]].contiguous())
                bn.num_batches_tracked += F["n_calls"]
        return fused

    # ------------------------------------------------------------------
    # backward: d(loss)/d(cls), d(loss)/d(loc) -> every parameter's gradient (into flat G)
    # ------------------------------------------------------------------
    def backward(self, *args, **kw):
        """The reverse pass (`_backward_pass`) and the split-f16 range check behind it.  A gradient map that outgrew the
        measured power-of-two lift of its split-f16 copy was CLAMPED in that copy: the gradients of the pass are wrong.  The
        reference's CoDetModule.step never throws on a finite loss, so neither does this: the saved activations of the
        forward are all still there -- the lifts are dropped and the SAME backward runs again with every data and weight
        gradient on the exact-fp32 kernels (which is also the calibration pass: it measures the new lifts), counted in
        `f32_fallback_steps`.  Agent-parallel ranks decide on the MAX of their flag words, so that all of them take the second
        pass (it holds collectives) or none does.  Only a pass that is flagged again -- a NaN, or an overflow in the fp32
        pass's own operands -- raises."""
        G = self._pass_with_folds(*args, **kw)
        flags = self._range_flags()
        if flags & 1:
            self._dz_lift.clear()
            self.f32_fallback_steps += 1
            self.last_fallback_step = self.step_count
            G = self._pass_with_folds(*args, **kw)
            flags = self._range_flags()
            if flags & 1:
                raise ops._lib.DnError(
                    "backward: the split-f16 range guard tripped again in the all-fp32 second pass (an activation of the "
                    "forward beyond the f16 range?); the gradients of this step are invalid and were NOT applied")
        if flags & 4:
            raise ops._lib.DnError("backward: a NaN reached a split-f16 epilogue")
        return G

    def _pass_with_folds(self, *args, **kw):
        """one reverse pass; the folds of its bias gradients (sums over dz per channel: leaves, read by the optimizer only) are
        collected and launched together behind it (train_ops.DeferredFolds; DN_TRAIN_DEFER_FOLDS=0: each behind its sum)"""
        if os.environ.get("DN_TRAIN_DEFER_FOLDS", "1") == "0":
            return self._backward_pass(*args, **kw)
        folds = self.__dict__.get("_folds")
        if folds is None or len(folds._ws) > 256:      # (workspaces are kept per output tensor: a caller that hands in a new
            folds = self._folds = T.DeferredFolds()    #  gradient buffer every pass must not grow them without bound)
        self._folds_active = folds
        try:
            G = self._backward_pass(*args, **kw)
            folds.run()
        finally:
            self._folds_active = None
            folds._jobs, folds._keep = [], []
        return G

    def _backward_pass(self, dcls, dloc, G=None, dkd=None):
        """dkd (knowledge distillation): optional dict of dense NHWC gradients w.r.t. the student's
        x5 / x6 / x7 / fused maps; each is a second consumer of that map, added in the BN backward
        that already reads the decoder's gradient.  Idempotent: reads the saved activations and its arguments, writes G
        and fresh buffers only (backward() may run it twice)."""
        m, L = self.model, self.L
        G = self.flat_g if G is None else G
        dkd = dkd or {}
        cls, reg = m.classification, m.regression.box_prediction
        hc = self.head_ctx
        h1 = hc["h1"]
        n, h, w = h1.shape[0], h1.shape[1], h1.shape[2]
        dcls = dcls.reshape(n, h, w, -1)
        dloc = dloc.reshape(n, h, w, -1)
        if not dcls.is_contiguous():
            dcls = dcls.contiguous()
        if not dloc.is_contiguous():
            dloc = dloc.contiguous()
        dh1 = torch.empty_like(h1)
        self._conv_bwd(hc["dc"], cls.conv2.weight, h1[..., :32], None, dcls, self.g(cls.conv2.weight, G),
                       self.g(cls.conv2.bias, G), dx_out=dh1[..., :32])
        self._conv_bwd(hc["dr"], reg[3].weight, h1[..., 32:], None, dloc, self.g(reg[3].weight, G),
                       self.g(reg[3].bias, G), dx_out=dh1[..., 32:])

        def merged(p, count):
            off, _, _ = self.grad_of[id(p)]
            return G[off:off + count]
        dx8 = self._layer_bwd(self.head1, dh1, G, gw=merged(cls.conv1.weight, 64 * 32 * 9).view(64, 32, 3, 3),
                              gb=merged(cls.conv1.bias, 64), ggamma=merged(cls.bn1.weight, 64),
                              gbeta=merged(cls.bn1.bias, 64))

        d = self._layer_bwd(L["conv8_2"], dx8, G)
        dcat8 = self._layer_bwd(L["conv8_1"], d, G)                     # [.., 64 (up x7) | 32 (x0)]
        d = self._layer_bwd(L["conv7_2"], dcat8[..., :64], G, up_a=True, dy_b=dkd.get("x7"))
        dcat7 = self._layer_bwd(L["conv7_1"], d, G)                     # [.., 128 (up x6) | 64 (x1)]
        d = self._layer_bwd(L["conv6_2"], dcat7[..., :128], G, up_a=True, dy_b=dkd.get("x6"))
        dcat6 = self._layer_bwd(L["conv6_1"], d, G)                     # [.., 256 (up x5) | 128 (x2)]
        d = self._layer_bwd(L["conv5_2"], dcat6[..., :256], G, up_a=True, dy_b=dkd.get("x5"))
        dcat5 = self._layer_bwd(L["conv5_1"], d, G)                     # [.., 512 (up x4) | 256 (fused)]

        # decoder-side gradient of each encoder output e[k] (k = 4 arrives at twice its resolution)
        d_dec = [dcat8[..., 64:], dcat7[..., 128:], dcat6[..., 256:], dcat5[..., 512:], dcat5[..., :512]]
        lay_k = m.layer
        # (layer 4: the fused map reaches conv5_1 through the upsample -- undo it before the fusion)
        dfused = T.upsample2_sum(d_dec[4]) if lay_k == 4 else d_dec[lay_k]
        if dkd.get("fused") is not None:
            dfused = T.add_rows(dkd["fused"].clone(), dfused)      # (a copy: the pass may run a second time on the same dkd)
        # encoder, top down: e[k] feeds conv{k+1}_1 (gradient d) and the decoder / the fusion (d_dec[k])
        def group_bwd(k, d):
            names = _ENC_GROUPS[k]
            for name in reversed(names):
                last = name == names[-1]
                if last and d is None:                     # e[4]: one consumer, the decoder's upsample
                    d = self._layer_bwd(L[name], d_dec[4], G, up_a=lay_k != 4)
                elif last:
                    d = self._layer_bwd(L[name], d, G, dy_b=d_dec[k])
                else:
                    # (the group's first layer, for k >= 1 the stride-2 one: its dx goes to the level below's last layer and nowhere else)
                    d = self._layer_bwd(L[name], d, G, need_dx=name != "conv_pre_1", s2d_ok=name == names[0] and k >= 1)
            return d

        # the levels above the exchanged one do not wait for the fusion's backward: second stream
        dev = dcls.device
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if self.overlap_streams else main
        d = None
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for k in range(4, lay_k, -1):
                d = group_bwd(k, d)
        d_fus = self._fusion_bwd(dfused, G)                # gradient w.r.t. the own maps of the fusion layer
        if "compress" in L:
            d_fus = self._layer_bwd(L["compress"], self._layer_bwd(L["decompress"], d_fus, G), G)
        d_dec[lay_k] = d_fus
        main.wait_stream(side)
        if d is not None:
            d.record_stream(main)
        for k in range(lay_k, -1, -1):
            d = group_bwd(k, d)
        return G

    def _range_flags(self):
        """dgrad_math / wgrad_math = "sp": did a dz outgrow its lift (bit 0: |dz| * lift > 65504, or an activation times
        _WGRAD_X_LIFT did in the split-f16 weight gradient), did a NaN reach a split-f16 epilogue (bit 2)?  A read of the
        engine's sticky range flags BEFORE the optimizer step (the step ends in a host read of the losses anyway).  With an
        agent shard the word is the MAX over the ranks: every rank must take the same branch in backward() -- one rank
        re-running the pass (or raising) alone would leave its peers in a collective nobody else posts."""
        sharded_run = self.shard is not None and self.shard.world > 1
        if (self.dgrad_math != "sp" and self.wgrad_math != "sp") or (not self._dz_lift and not sharded_run):
            return 0
        # stream-ordered collect into a word this engine keeps + one host read: the blocking dn_sp_range_flags synchronises the
        # whole device and allocates / frees its scratch word per call (measured ~2 ms per step inside a large process)
        word = self.__dict__.get("_range_word")
        if word is None or word.device != self.flat_p.device:
            word = self._range_word = torch.zeros(1, dtype=torch.int32, device=self.flat_p.device)
        word = ops.sp_range_flags_into(word, zero_first=True, reset=True)
        if self._force_range_flags:                      # tests: pretend the guard tripped (on this rank only)
            word |= int(self._force_range_flags.pop(0))
        if sharded_run:
            self.shard.max_(word)
        return int(word.item()) & 0xffffffff

    def _fusion_bwd(self, dfused, G):
        m, L, F, c = self.model, self.L, self.F, self.fctx
        f = m.pixel_weighted_fusion
        maps, NI, NW = c["maps"], c["NI"], c["NW"]
        C = maps.shape[-1]
        P = NI + NW
        # (agent-parallel: the other ranks' maps are in no list of this rank's egos -- only their warps are -- so their rows
        # of dmaps are written by nothing before the adds below: start from zero)
        dmaps = torch.empty_like(maps) if self.shard is None else torch.zeros_like(maps)
        dz4 = T.fuse_combine_backward(dfused, c["z4"], c["weights"], maps, F["first"], F["pair_index"],
                                      F["map_image"], F["ego_out"], dmaps)
        dh3 = self._conv_bwd(c["d4"], f.conv1_4.weight, c["h3"], None, dz4, self.g(f.conv1_4.weight, G),
                             self.g(f.conv1_4.bias, G))
        dh2 = self._layer_bwd(L["mlp3"], dh3, G)
        dh1 = self._layer_bwd(L["mlp2"], dh2, G)
        dz1 = T.bn_backward(dh1, c["h1"], c["z1"], c["mean1"], c["var1"], f.bn1_1.weight, _EPS,
                            self.g(f.bn1_1.weight, G), self.g(f.bn1_1.bias, G), relu=True)
        T.channel_sum(dz1, self.g(f.conv1_1.bias, G), folds=self.__dict__.get("_folds_active"))
        dE = T.pair_sum_ego(dz1, F["efirst"], F["epairs"], NI)
        gw1 = self.g(f.conv1_1.weight, G).view(128, 2 * C)
        T.conv_wgrad(c["d_e"], maps[:NI], None, dE, gw1[:, :C], dw_cin_total=2 * C)
        T.conv_wgrad(c["d_f"], maps, None, dz1, gw1[:, C:], dw_cin_total=2 * C)
        T.add_rows(dmaps, self._dgrad(c["d_f"], f.conv1_1.weight, dz1, ci_first=C, c_in=C))       # W1 = [W_ego | W_nbr]
        T.add_rows(dmaps[:NI], self._dgrad(c["d_e"], f.conv1_1.weight, dE, ci_first=0, c_in=C))
        if NW:
            T.warp_backward(dmaps[NI:], F["poses"], F["src_image"], dmaps[:NI], rigid=F["rigid"])
        if self.shard is not None:
            # dmaps[:NI] = d(this rank's loss terms) / d(EVERY agent's map): the backward of the all-gather is a
            # reduce-scatter -- each rank receives the sum, over the ranks, of the gradient of its own agents' maps
            n_loc = self.shard.count * (NI // m.agent_num)
            return self.shard.reduce_scatter_rows(dmaps[:NI], self.shard.first * (NI // m.agent_num), n_loc)
        return dmaps[:NI]

    # ------------------------------------------------------------------
    def optimizer_step(self):
        self.step_count += 1
        T.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.step_count, self.lr,
                    self.betas, self.eps, self.weight_decay)
        self.model._plan = None          # eval-mode packed weights are stale now

    # -- what the reference's epoch loop saves / resumes (optimizer_state_dict, scheduler) ------
    def state_dict(self):
        """Adam moments per parameter NAME (independent of the flat layout) + step count + lr"""
        names = {id(p): n for n, p in self.model.named_parameters()}
        view = lambda flat, p: flat[self.grad_of[id(p)][0]:self.grad_of[id(p)][0] + p.numel()].view(p.shape)
        return {"step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "weight_decay": self.weight_decay,
                "exp_avg": {names[id(p)]: view(self.flat_m, p).clone() for p in self.params},
                "exp_avg_sq": {names[id(p)]: view(self.flat_v, p).clone() for p in self.params}}

    def load_state_dict(self, sd):
        self.__dict__.pop("_wmul_cache", None)      # weight lifts belong to the weights that were just replaced
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        self.betas, self.eps, self.weight_decay = tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])
        for n, p in self.model.named_parameters():
            off = self.grad_of[id(p)][0]
            self.flat_m[off:off + p.numel()].view(p.shape).copy_(sd["exp_avg"][n])
            self.flat_v[off:off + p.numel()].view(p.shape).copy_(sd["exp_avg_sq"][n])

    def allreduce_grads(self):
        """DDP's gradient averaging as ONE collective over the flat buffer (RCCL over xGMI).  Agent-parallel training: the
        ranks hold DISJOINT terms of one loss (normalised by the global image count), so their gradients are SUMMED."""
        if self.shard is not None:
            self.shard.sum_(self.flat_g)
            return
        from .sharded import average_gradients_
        average_gradients_(self.flat_g)


class _TrainFn(torch.autograd.Function):
    """The whole forward as one autograd node: lets the reference's own
    `loss.backward(); optimizer.step()` drive the explicit reverse pass.  Outputs: cls, loc and
    (for the KD loss of kd_flag = 1) NCHW-shaped views of x8, x7, x6, x5 and the fused map."""

    @staticmethod
    def forward(ctx, engine, bevs, trans, num_agent, batch_size, *params):
        res = engine.forward(bevs, trans, num_agent, batch_size)
        ctx.engine = engine
        ctx.generation = engine.generation
        ctx.set_materialize_grads(False)      # outputs the loss does not touch come back as None
        o = engine.outs
        nchw = lambda t: t.permute(0, 3, 1, 2)
        return (res["cls"], res["loc"], nchw(o["x8"]), nchw(o["x7"]), nchw(o["x6"]), nchw(o["x5"]),
                nchw(o["fused"]))

    @staticmethod
    def backward(ctx, dcls, dloc, dx8, dx7, dx6, dx5, dfused):
        eng = ctx.engine
        if ctx.generation != eng.generation:
            # the reverse pass reads the activations the engine saved in ITS last forward: a second
            # forward (another micro-batch, a no_grad monitoring pass in train() mode) overwrote them
            raise RuntimeError(
                "DiscoNet (train mode): backward() of forward #%d called after forward #%d ran on the same "
                "model -- the HIP training engine keeps ONE set of saved activations; call backward() "
                "before the next forward (accumulate gradients across steps with "
                "CoDetModule / TrainEngine.flat_g instead)" % (ctx.generation, eng.generation))
        if dx8 is not None:
            raise NotImplementedError("a loss on x8 itself is not part of the reference's KD term")
        nhwc = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous()
        dkd = {"x7": nhwc(dx7), "x6": nhwc(dx6), "x5": nhwc(dx5), "fused": nhwc(dfused)}
        zeros = lambda t, ref: torch.zeros_like(ref) if t is None else t.contiguous()
        G = torch.zeros_like(eng.flat_g)
        eng.backward(zeros(dcls, eng.last_result["cls"]), zeros(dloc, eng.last_result["loc"]), G, dkd)
        return (None, None, None, None, None) + tuple(eng.g(p, G) for p in eng.params)


def train_forward(model, bevs, trans_matrices, num_agent_tensor, batch_size):
    """DiscoNet.forward in train() mode"""
    eng = model.__dict__.get("_train_engine")
    if eng is None:
        eng = TrainEngine(model)
        model.__dict__["_train_engine"] = eng
    if torch.is_grad_enabled():
        cls, loc, x8, x7, x6, x5, fused = _TrainFn.apply(eng, bevs, trans_matrices, num_agent_tensor,
                                                         batch_size, *eng.params)
        result = {"loc": loc, "cls": cls}
    else:
        result = eng.forward(bevs, trans_matrices, num_agent_tensor, batch_size)
        o = eng.outs
        x8, x7, x6, x5, fused = (o[k].permute(0, 3, 1, 2) for k in ("x8", "x7", "x6", "x5", "fused"))
    if model.kd_flag == 1:
        return (result, x8, x7, x6, x5, fused)
    return result


class CoDetModule:
    """upstream:coperception/utils/CoDetModule.py :: CoDetModule, the training surface:
    step(data, batch_size) -> loss values, with the losses, the backward and Adam on the HIP path.
    kd_flag = 1: `teacher` (disconet_amd.TeacherNet, frozen, eval) sees data["bev_seq_teacher"] and
    kd_weight * sum of KLDiv(log_softmax(student), softmax(teacher)) over x5, x6, x7 and the fused
    layer-3 map (vs the teacher's x3) joins the loss."""

    def __init__(self, model, teacher=None, config=None, optimizer=None, kd_flag=0, lr=1e-3,
                 alpha=0.25, gamma=2.0, sigma=3.0, shard=None, dgrad_math=None, wgrad_math=None):
        """shard (sharded.AgentShard): agent-parallel training -- step() then takes THIS rank's agents' images, labels and
        targets (agent-major, [count * B, ...]); trans_matrices / num_agent stay the whole scenes'.
        With kd_flag = 1 (BASELINE configs[2] + [4] together): the frozen teacher is replicated and has no communication --
        one image in, that image's pyramid out -- so every rank runs it on data["bev_seq_teacher"] of ITS agents (the
        holistic views in their frames, [count * B, ...] like bev_seq) and adds its share of the KD term: the KL means are
        normalised by the GLOBAL row count, the ranks' terms (and their gradients, through the one summed all-reduce of the
        flat gradient) add up to the un-sharded step's."""
        if kd_flag and teacher is None:
            raise ValueError("kd_flag = 1 needs the teacher network")
        self.model, self.teacher, self.kd_flag = model, teacher, int(bool(kd_flag))
        if self.kd_flag:
            teacher.eval()
        kw = {}
        if optimizer is not None:      # take the hyper-parameters of the torch optimizer handed in
            grp = optimizer.param_groups[0]
            lr = grp["lr"]
            kw = {"betas": tuple(grp.get("betas", (0.9, 0.999))), "eps": grp.get("eps", 1e-8),
                  "weight_decay": grp.get("weight_decay", 0.0)}
        self.engine = TrainEngine(model, lr=lr, shard=shard, dgrad_math=dgrad_math, wgrad_math=wgrad_math, **kw)
        model.__dict__["_train_engine"] = self.engine
        self.alpha, self.gamma, self.sigma = alpha, gamma, sigma

    def scheduler_step(self, epoch, milestones=(50, 100), gamma=0.5):
        """torch.optim.lr_scheduler.MultiStepLR as the reference's epoch loop steps it: call once
        per finished epoch (1-based count of finished epochs)"""
        if epoch in milestones:
            self.engine.lr *= gamma
        return self.engine.lr

    def step(self, data, batch_size, update=True):
        """update = False: forward, losses and every gradient (engine.flat_g), but no gradient exchange and no Adam step --
        tests, and the calibration pass of dgrad_math = "sp" (the first backward measures the gradient maps' lifts)"""
        bev_seq = data["bev_seq"]
        eng = self.engine
        if not self.model.training:          # (Module.train() walks every submodule: 0.2 ms of host time with the GPU idle)
            self.model.train()
        with torch.no_grad():
            res = eng.forward(bev_seq, data["trans_matrices"], data["num_agent"], batch_size)
            code = res["loc"].shape[-1]
            dev = bev_seq.device
            f32 = lambda t, shape: t.to(device=dev, dtype=torch.float32).reshape(shape).contiguous()
            losses, dcls, dloc = T.det_loss(
                res["cls"].reshape(-1, 2), f32(data["labels"], (-1, 2)), res["loc"].reshape(-1, code),
                f32(data["reg_targets"], (-1, code)), f32(data["reg_loss_mask"], (-1,)),
                norm=bev_seq.shape[0] * (eng.shard.world if eng.shard is not None else 1),      # the reference's N: all A * B images
                alpha=self.alpha, gamma=self.gamma, sigma=self.sigma)
            dkd, kd = None, None
            if self.kd_flag:
                kd_weight = float(data["kd_weight"]) if "kd_weight" in data else 1e5
                t8, t7, t6, t5, t3, t2 = self.teacher.forward_nhwc(data["bev_seq_teacher"])
                kd = torch.zeros(1, dtype=torch.float64, device=dev)
                o = eng.outs
                world = eng.shard.world if eng.shard is not None else 1
                dkd = {k: T.kd_kl_loss(o[k], t, kd_weight, kd,
                                       norm_rows=(o[k].numel() // o[k].shape[-1]) * world if world > 1 else None)
                       for k, t in (("x5", t5), ("x6", t6), ("x7", t7), ("fused", t3))}
                if world > 1:
                    eng.shard.sum_(kd)                # reported KD loss: the whole scenes', as without a shard
            eng.backward(dcls, dloc, dkd=dkd)
            if update:
                eng.allreduce_grads()
                eng.optimizer_step()
        if eng.shard is not None:
            eng.shard.sum_(losses)                    # reported losses: the whole scenes', as without a shard
        l = losses.tolist()
        out = {"loss": l[0] + l[1], "cls_loss": l[0], "loc_loss": l[1]}
        if kd is not None:
            out["kd_loss"] = float(kd)
            out["loss"] += out["kd_loss"]
        return out
