"""Drop-in `mmvit4.MMVit4` for MI355X: same nn.Module surface and state-dict as the reference
(mmvit4.py:391-532), forward/backward executed by the gfx950 kernels of libcorrif_gfx950.so.

    from mmvit4 import MMVit4          # as F2_MAIN.py:122-123 does
    model = MMVit4().to("cuda")
    pred = model(images)               # images [B,3,D,H,W] fp32 -> [B,3,1,224,224] in (0,1), autograd-connected

Internals are channels-last (NDHWC): the reference's `permute(0,2,3,4,1).view(B,-1,512)` tokenisations
(mmvit4.py:458-461, 499-501, 510-513) and the `[B,8,8,8,2048]` re-view (mmvit4.py:526) are free
reinterpretations here, and every 1x1x1 convolution is a row-major GEMM.  Parameters keep the reference
names, shapes and (O,I,kd,kh,kw) layout, so checkpoints are interchangeable in both directions.
There is no CPU path: calling the model on CPU tensors raises.
"""
import math

import torch
import torch.nn as nn

import ops

# The decoder / multimodal-transformer parameters are used by two sample-group lanes on two streams (MMVit4.decoder_split), so the second
# lane's gradient reaches an AccumulateGrad node that lives on the first lane's stream: autograd inserts the stream wait the accumulation
# needs and warns that the streams differ.  The mismatch is intentional here, and the warning is switched off for the duration of THIS
# model's backward pass only (`_QuietBackwardFn` on the prediction: its backward is the first node of the pass and an engine callback
# restores the switch at the end), not for the process (round 2 flipped it at import time).
_WARN_SWITCH = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)


def _restore_warning():
    if _WARN_SWITCH is not None:
        _WARN_SWITCH(True)


class _QuietBackwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred):
        return pred.view_as(pred)

    @staticmethod
    def backward(ctx, g):
        if _WARN_SWITCH is not None:
            _WARN_SWITCH(False)
            torch.autograd.Variable._execution_engine.queue_callback(_restore_warning)
        return g

basic_dims = 8
transformer_basic_dims = 512
mlp_dim = 512
num_heads = 8
depth = 1
num_modals = 3
patch_size = 8
_MODS = ("RGB", "NIR", "SWIR")
CAPTURE_LANES = True  # diagnostics: False leaves the sample-group lanes out of HIP-graph captures (round 2 behaviour, see _run_lanes)
GRAD_TAP = True       # Bottleneck identity blocks: fold the residual gradient into conv1's data-gradient epilogue (ops.grad_tap)


def _rs(t, stream):
    """tensor.record_stream(stream) for tensors that cross streams - except while a HIP graph is being captured: the caching
    allocator cannot honour it for a graph's private pool (its deferred-free events are capture nodes), and inside one captured
    forward every cross-stream tensor is held by the forward's own locals until all forked streams have re-joined."""
    if not torch.cuda.is_current_stream_capturing():
        t.record_stream(stream)


class _Edges:
    """Fork / join edges between the model's HIP streams, built on PERSISTENT events.

    `a.wait_stream(b)` (and a `torch.cuda.Event()` local) creates an event, records it and destroys it again a few lines later.
    In eager mode that is harmless.  While a HIP graph is being captured it is what made the process crash "on the next eager
    forward or capture" in round 1: an event recorded on a capturing stream is entered into that stream's list of captured events,
    `hipStreamEndCapture` walks that list to take every event out of capture mode, and an event that was destroyed in between is a
    dangling pointer there (heap corruption: the fault shows up later, wherever the freed block is reused - which is why it was
    seen after a capture that itself replayed bit-identically, and why the single-stream capture, which records no event, never
    showed it).  So: every edge uses an event from a pool owned by the module, created once, never destroyed while the module is
    alive.  `reset()` rewinds the pool at the start of a forward; re-recording an event is safe because a wait that was already
    enqueued refers to the record that preceded it (hipStreamWaitEvent snapshots the event at the call)."""

    def __init__(self):
        self.events, self.i = [], 0
        self._eager, self._captured = self.events, []

    def reset(self):
        """rewind at the start of a forward.  A forward that is being captured into a HIP graph gets a FRESH set of events, kept alive
        with the module and never recorded again: an event that already sits in one instantiated graph must not be re-recorded into
        another capture (a second GraphedForward on the same model - another batch size - crashed the HIP runtime that way), nor mixed
        with the eager pool."""
        self.i = 0
        if torch.cuda.is_current_stream_capturing():
            self.events = []
            self._captured.append(self.events)
        else:
            self.events = self._eager

    def mark(self, src):
        """a pooled event recorded on `src` now (wait for it later with stream.wait_event)"""
        if self.i == len(self.events):
            self.events.append(torch.cuda.Event())
        ev = self.events[self.i]
        self.i += 1
        ev.record(src)
        return ev

    def edge(self, src, dst):
        """everything enqueued on `src` so far happens before whatever is enqueued on `dst` from here on"""
        dst.wait_event(self.mark(src))


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


class Conv3dP(nn.Module):
    """Parameter holder + launcher for one nn.Conv3d of the reference (weight O,I,kd,kh,kw; optional bias)."""

    def __init__(self, cin, cout, k=1, stride=1, pad=0, bias=True, replicate=False):
        super().__init__()
        self.k, self.stride, self.pad, self.replicate = _triple(k), _triple(stride), _triple(pad), replicate
        self.weight = nn.Parameter(torch.empty(cout, cin, *self.k))
        fan_in = cin * self.k[0] * self.k[1] * self.k[2]
        nn.init.kaiming_normal_(self.weight)                      # mmvit4.py:437-439
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in)))
        else:
            self.register_parameter("bias", None)

    side_wgrad = False      # set on the encoders' convolutions (one gradient per step): ops.SIDE_WGRAD

    def forward(self, x, out=None, stats=None, grad_link=None, bwd_stats=None):
        return ops.conv3d(x, self.weight, self.bias, self.stride, self.pad, self.replicate, out, stats=stats, grad_link=grad_link,
                          side_wgrad=self.side_wgrad, bwd_stats=bwd_stats)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%s, padding=%s%s%s" % (
            self.weight.shape[1], self.weight.shape[0], self.k, self.stride, self.pad,
            ", padding_mode=replicate" if self.replicate else "", "" if self.bias is not None else ", bias=False")


class BatchNorm3dP(nn.Module):
    """nn.BatchNorm3d parameters / buffers (mmvit4.py:121,130-151).  `num_batches_tracked` is bookkeeping only (momentum is fixed at
    0.1, so nothing reads it): training forwards count on the host (`_nbt_pending`) and the int64 buffer is brought up to date
    whenever it is looked at - attribute access, `state_dict()`, `MMVit4.buffers()` - instead of costing one device launch per
    BatchNorm layer and step (159 `add` launches per step in round 1)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_counter())

    def flush_counter(self):
        if self._nbt_pending:
            n, self._nbt_pending = self._nbt_pending, 0
            self._buffers["num_batches_tracked"] += n

    def __getattr__(self, name):
        if name == "num_batches_tracked":
            self.flush_counter()
        return super().__getattr__(name)

    def _load_from_state_dict(self, *a, **kw):
        self._nbt_pending = 0                      # the loaded counter replaces whatever was pending
        return super()._load_from_state_dict(*a, **kw)

    def forward(self, x, residual=None, relu_in=False, relu_out=False, out=None, pre=None, bwd_link=None):
        if self.training:
            self._nbt_pending += 1
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, residual, relu_in, relu_out,
                              self.training, self.momentum, self.eps, out, pre, bwd_link)

    def extra_repr(self):
        return "%d, eps=%g, momentum=%g" % (self.weight.numel(), self.eps, self.momentum)


class InstanceNorm3dP(nn.Module):
    """nn.InstanceNorm3d(affine=False, track_running_stats=False): no parameters, no buffers (mmvit4.py:24)."""

    def forward(self, x, out=None, pre=None):           # applied AFTER the ReLU, fused with it (mmvit4.py:41-45)
        return ops.relu_instnorm(x, 1e-5, out, pre)


class general_conv3d_prenorm(nn.Module):
    """conv(bias) -> ReLU -> InstanceNorm3d (mmvit4.py:29-45)."""

    def __init__(self, in_ch, out_ch, k_size=3, stride=1, padding=1, pad_type="zeros"):
        super().__init__()
        self.conv = Conv3dP(in_ch, out_ch, k_size, stride, padding, True, pad_type == "replicate" and k_size != 1)
        self.norm = InstanceNorm3dP()

    def forward(self, x, out=None):
        st = {"G": x.shape[0], "relu": True}            # the conv's epilogue also produces the InstanceNorm statistics partials
        return self.norm(self.conv(x, stats=st), out, pre=st)


class fusion_prenorm(nn.Module):
    """RFM: 1x1x1, 3x3x3 (zero pad), 1x1x1 (mmvit4.py:47-56)."""

    def __init__(self, in_channel=64, num_cls=1):
        super().__init__()
        self.fusion_layer = nn.Sequential(general_conv3d_prenorm(in_channel * num_cls, in_channel, 1, 1, 0),
                                          general_conv3d_prenorm(in_channel, in_channel, 3, 1, 1),
                                          general_conv3d_prenorm(in_channel, in_channel, 1, 1, 0))

    def forward(self, x):
        return self.fusion_layer(x)


class EarlyFusionBlock(nn.Module):
    """cat(3 modalities) -> 1x1x1 conv -> ReLU -> IN (mmvit4.py:64-81).  The concat buffer is filled in place
    by the three encoders' adapt convolutions, so `forward` receives it already assembled."""

    def __init__(self, in_channels):
        super().__init__()
        c = num_modals * in_channels
        self.conv = Conv3dP(c, c, 1, 1, 0, True)
        self.norm = InstanceNorm3dP()

    def forward(self, cat):
        st = {"G": cat.shape[0], "relu": True}
        return self.norm(self.conv(cat, stats=st), pre=st)


class Bottleneck3D(nn.Module):
    """mmvit4.py:196-212 with the inflated ResNet-50 convolutions of :126-151."""

    def __init__(self, cin, width, stride, down):
        super().__init__()
        cout = 4 * width
        self.conv1 = Conv3dP(cin, width, 1, 1, 0, False)
        self.bn1 = BatchNorm3dP(width)
        self.conv2 = Conv3dP(width, width, (1, 3, 3), (1, stride, stride), (0, 1, 1), False)
        self.bn2 = BatchNorm3dP(width)
        self.conv3 = Conv3dP(width, cout, 1, 1, 0, False)
        self.bn3 = BatchNorm3dP(cout)
        self.downsample = nn.Sequential(Conv3dP(cin, cout, 1, (1, stride, stride), 0, False), BatchNorm3dP(cout)) if down else None

    def forward(self, x, link=None):
        """link: the gradient link of this block's conv1 when the caller has parked further consumers of x on it (Encoder: the adapt
        convolution of the previous layer's output)"""
        return _bottleneck(self, x, link, self.training)


def _bottleneck(L, x, link, train):
    """Bottleneck3D.forward over a layer provider `L` (conv1, bn1, conv2, bn2, conv3, bn3, downsample): the module itself, or a
    _GroupedBlock of the three modality encoders' twin blocks on stacked activations."""
    G = getattr(L, "groups", 1)

    def st():                                   # BatchNorm batch statistics come out of the producing GEMM's epilogue
        return {"G": G, "relu": False} if train else None

    if link is None and GRAD_TAP and torch.is_grad_enabled() and x.requires_grad:
        link = {}          # x feeds conv1 AND the residual add / the downsample conv: conv1's data-gradient epilogue absorbs their gradient
    s1, s2, s3 = st(), st(), st()
    # backward statistics of a BatchNorm come out of the data-gradient epilogue of the convolution that consumes its output
    # (ops._bwd_stats_request): bn1 -> conv2, bn2 -> conv3, and the previous block's bn3 -> this block's conv1 when every other consumer
    # of the block input rides on conv1's gradient link
    tr = train and torch.is_grad_enabled()
    l1, l2, l3 = ({} if tr else None), ({} if tr else None), ({} if tr else None)
    prev = getattr(x, "_corrif_bn_link", None) if link is not None else None
    y = L.bn1(L.conv1(x, stats=s1, grad_link=link, bwd_stats=prev), relu_out=True, pre=s1, bwd_link=l1)
    y = L.bn2(L.conv2(y, stats=s2, bwd_stats=l1), relu_out=True, pre=s2, bwd_link=l2)
    y = L.conv3(y, stats=s3, bwd_stats=l2)
    xt = x if link is None else ops.grad_tap(x, link)      # created after conv1..conv3's nodes: its backward runs before theirs
    if L.downsample is not None:                           # (the downsample path's nodes too: they are created here, not first)
        s0 = st()
        idt = L.downsample[1](L.downsample[0](xt, stats=s0), pre=s0)
    else:
        idt = xt
    out = L.bn3(y, residual=idt, relu_out=True, pre=s3, bwd_link=l3)
    if l3 is not None:
        out._corrif_bn_link = l3         # read by the next block's conv1 (same tensor object)
    return out


def _fire_hooks(mods, x, y, zin, zout):
    """forward hooks of twin modules that ran as one grouped launch: each module sees its own group's input / output slice"""
    G = len(mods)
    for g, m in enumerate(mods):
        if not m._forward_hooks:
            continue

        def cut(t, mode):
            if mode == "stack":
                n = t.shape[0] // G
                return t[g * n:(g + 1) * n]
            c = t.shape[-1] // G
            return t[..., g * c:(g + 1) * c]
        for hook in list(m._forward_hooks.values()):
            hook(m, (cut(x, zin),), cut(y, zout))


class _GConv:
    """the G twin Conv3dP of the modality encoders as one grouped launch (ops.conv3d_grouped); call signature of Conv3dP.forward"""

    def __init__(self, convs, zin="stack", zout="stack"):
        self.convs, self.zin, self.zout = convs, zin, zout

    def __call__(self, x, out=None, stats=None, grad_link=None, bwd_stats=None):
        c0 = self.convs[0]
        y = ops.conv3d_grouped(x, [c.weight for c in self.convs], [c.bias for c in self.convs], c0.stride, c0.pad, self.zin, self.zout,
                               out, stats, grad_link, bwd_stats)
        _fire_hooks(self.convs, x, y, self.zin, self.zout)
        return y


class _GBN:
    """the G twin BatchNorm3dP on activations stacked along the batch axis; call signature of BatchNorm3dP.forward"""

    def __init__(self, bns):
        self.bns = bns

    def __call__(self, x, residual=None, relu_in=False, relu_out=False, out=None, pre=None, bwd_link=None):
        b0 = self.bns[0]
        if b0.training:
            for b in self.bns:
                b._nbt_pending += 1
        y = ops.batch_norm_grouped(x, [b.weight for b in self.bns], [b.bias for b in self.bns], [b.running_mean for b in self.bns],
                                   [b.running_var for b in self.bns], residual, relu_in, relu_out, b0.training, b0.momentum, b0.eps, out,
                                   pre, bwd_link)
        _fire_hooks(self.bns, x, y, "stack", "stack")
        return y


class _GroupedBlock:
    """layer provider for _bottleneck built from the twin Bottleneck3D modules of the modality encoders"""

    def __init__(self, blocks):
        self.groups = len(blocks)
        self.blocks = blocks
        for name in ("conv1", "conv2", "conv3"):
            setattr(self, name, _GConv([getattr(b, name) for b in blocks]))
        for name in ("bn1", "bn2", "bn3"):
            setattr(self, name, _GBN([getattr(b, name) for b in blocks]))
        self.downsample = None
        if blocks[0].downsample is not None:
            self.downsample = (_GConv([b.downsample[0] for b in blocks]), _GBN([b.downsample[1] for b in blocks]))


class _ResLayer(nn.Sequential):
    """nn.Sequential of Bottleneck3D (same state-dict keys `0.`, `1.`, ...; module hooks on the layer keep working) whose first block
    can be handed the gradient link of the layer input (see Encoder.forward)"""

    def forward(self, x, link=None):
        for i, blk in enumerate(self):
            x = blk(x, link) if i == 0 else blk(x)
        return x


def _res_layer(cin, width, n, stride):
    return _ResLayer(Bottleneck3D(cin, width, stride, True), *[Bottleneck3D(4 * width, width, 1, False) for _ in range(n - 1)])


_ADAPT = ((64, basic_dims), (256, basic_dims * 2), (512, basic_dims * 4), (1024, basic_dims * 8), (2048, basic_dims * 8))


class Encoder(nn.Module):
    """mmvit4.py:113-194.  `forward` writes x1..x5 / x6 straight into channel slices of the early-fusion concat buffers."""

    def __init__(self, inflate_time=3):
        super().__init__()
        self.e1_c1 = Conv3dP(1, 64, (inflate_time, 7, 7), (1, 2, 2), (inflate_time // 2, 3, 3), False)
        self.e1_bn = BatchNorm3dP(64)
        self.e2 = _res_layer(64, 64, 3, 1)
        self.e3 = _res_layer(256, 128, 4, 2)
        self.e4 = _res_layer(512, 256, 6, 2)
        self.e5 = _res_layer(1024, 512, 3, 2)
        self.conv6 = Conv3dP(basic_dims * 23, basic_dims * 8, 1)
        for i, (ci, co) in enumerate(_ADAPT):
            setattr(self, "adapt%d" % (i + 1), Conv3dP(ci, co, 1))

    def forward(self, x, m, cats):
        """x: [B, D, H, W] view of modality m; cats[l]: concat buffer of level l (None until the first modality allocates it)."""
        g = self.steps(x, m, cats)
        try:
            while True:
                next(g)
        except StopIteration as done:
            return done.value

    def steps(self, x, m, cats):
        """`forward` as a generator that yields after the stem, after every ResNet layer and before the adapt / conv6 tail, so that
        MMVit4 can enqueue the three modality branches layer by layer in turn (see MMVit4._forward); returns forward's value."""
        f = ops.maxpool133(self.e1_bn(self.e1_c1(x), relu_in=True))          # conv -> ReLU -> BN -> pool (mmvit4.py:172-174)
        yield
        feats, links = [f], []
        tap = GRAD_TAP and torch.is_grad_enabled()
        for layer in (self.e2, self.e3, self.e4, self.e5):
            # a layer input has three consumers: the first block's conv1 and downsample conv and this level's adapt conv; the latter
            # two park their gradients on conv1's link (ops.grad_tap) and ride along in its data-gradient epilogue
            link = {} if tap and f.requires_grad else None
            links.append(link)
            f = layer(f, link)
            feats.append(f)
            yield
        links.append(None)
        outs = []
        B = x.shape[0]
        cube = torch.empty((B, 8, 8, 8, basic_dims * 23), dtype=torch.float32, device=x.device)
        parts, off = [], 0
        for l, f in enumerate(feats):
            c = _ADAPT[l][1]
            if cats[l] is None:
                cats[l] = torch.empty(f.shape[:4] + (num_modals * c,), dtype=torch.float32, device=x.device)
            fa = f if links[l] is None else ops.grad_tap(f, links[l])
            xl = getattr(self, "adapt%d" % (l + 1))(fa, out=cats[l][..., m * c:(m + 1) * c])
            outs.append(xl)
            parts.append(ops.trilinear(xl, (8, 8, 8), out=cube[..., off:off + c]))     # mmvit4.py:187-191
            off += c
        if cats[5] is None:
            cats[5] = torch.empty((B, 8, 8, 8, num_modals * basic_dims * 8), dtype=torch.float32, device=x.device)
        c6 = basic_dims * 8
        x6 = self.conv6(ops.cat_channels(cube, *parts), out=cats[5][..., m * c6:(m + 1) * c6])
        return outs + [x6]


class Decoder_fuse(nn.Module):
    """mmvit4.py:222-292 (seg_* heads are parameter-bearing but unused, as in the reference)."""

    def __init__(self, num_cls=1, reduce5=True):
        """reduce5=False builds the sibling model's decoder (mmmvit2.py:116-219): no RFM5_reduce, d4_c1 is 192 -> 128"""
        super().__init__()
        b, rep = basic_dims, "replicate"
        self.d4_c1 = general_conv3d_prenorm(b * 16 if reduce5 else b * 24, b * 16, pad_type=rep)
        self.d4_c2 = general_conv3d_prenorm(320, b * 8, pad_type=rep)
        self.d4_out = general_conv3d_prenorm(b * 8, b * 8, k_size=1, padding=0, pad_type=rep)
        self.d3_c1 = general_conv3d_prenorm(b * 8, b * 4, pad_type=rep)
        self.d3_c2 = general_conv3d_prenorm(128, b * 4, pad_type=rep)
        self.d3_out = general_conv3d_prenorm(b * 4, b * 4, k_size=1, padding=0, pad_type=rep)
        self.d2_c1 = general_conv3d_prenorm(b * 4, b * 2, pad_type=rep)
        self.d2_c2 = general_conv3d_prenorm(64, b * 2, pad_type=rep)
        self.d2_out = general_conv3d_prenorm(b * 2, b * 2, k_size=1, padding=0, pad_type=rep)
        self.d1_c1 = general_conv3d_prenorm(b * 2, b, pad_type=rep)
        self.d1_c2 = general_conv3d_prenorm(32, b, pad_type=rep)
        self.d1_out = general_conv3d_prenorm(b, b, k_size=1, padding=0, pad_type=rep)
        self.seg_d4 = Conv3dP(b * 8, num_cls, 1)
        self.seg_d3 = Conv3dP(b * 8, num_cls, 1)
        self.seg_d2 = Conv3dP(b * 4, num_cls, 1)
        self.seg_d1 = Conv3dP(b * 2, num_cls, 1)
        self.seg_layer = Conv3dP(b, num_cls, 1)
        self.RFM5 = fusion_prenorm(b * 24)
        if reduce5:
            self.RFM5_reduce = Conv3dP(b * 24, b * 16, 1)
        else:
            self.RFM5_reduce = None
        self.RFM4 = fusion_prenorm(b * 24)
        self.RFM3 = fusion_prenorm(b * 12)
        self.RFM2 = fusion_prenorm(b * 6)
        self.RFM1 = fusion_prenorm(b * 3)
        self.final_conv = Conv3dP(8, 3, 1)
        self.concurrent_skips = True
        self.compact_skips = True       # diagnostics / A-B: False materialises the up-sampled skip tensors and the concat buffers
        self._side = None
        self._edges = {}
        self._fork_from = {}            # lane -> event on the stream the lanes forked from (set by _run_lanes, see forward)

    def forward(self, x1, x2, x3, x4, x5, lane=0):
        B, dev = x5.shape[0], x5.device
        stages = ((self.RFM4, x4, 16, self.d4_c1, self.d4_c2, self.d4_out),
                  (self.RFM3, x3, 32, self.d3_c1, self.d3_c2, self.d3_out),
                  (self.RFM2, x2, 64, self.d2_c1, self.d2_c2, self.d2_out),
                  (self.RFM1, x1, 128, self.d1_c1, self.d1_c2, self.d1_out))
        # Compact skip branch.  F.interpolate(nearest) of the RFM output from Ds to n depth slices (mmvit4.py:271-286) is constant
        # along depth inside each block of slices that share a source (n / Ds slices, or floor / ceil of it when Ds does not divide n:
        # the reference-native 3 bands, 12 bands), so its share of the replicate-padded 3x3x3 convolution d*_c2 takes three
        # distinct values per block (first slice / interior / last slice).  For n / Ds >= 8 that share is evaluated on a grid of 3 * Ds slices
        # - nearest-up-sampling to (3 Ds, n, n), the same convolution with the weight's skip channels, no bias - and broadcast into the
        # convolution of the other (d*_c1) channels: d1_c2 at 4 bands runs 8 + 24 * 12/128 instead of 32 input channels' worth of
        # forward, data-gradient and weight-gradient work, the 128^3 x 24-channel up-sampled tensor and the concat buffer disappear.
        # Same arithmetic, summed in a different order (the sum over input channels is split in two).
        fs = []
        for _, skip, n, _, _, _ in stages:
            Ds = skip.shape[1]
            fs.append(n // Ds if (self.compact_skips and n // Ds >= 8) else 0)       # any ratio: blocks of floor / ceil(n / Ds) slices
        cats = [None if fs[l] else torch.empty((B, n, n, n, skip.shape[-1] + c1.conv.weight.shape[0]), dtype=torch.float32, device=dev)
                for l, (_, skip, n, c1, _, _) in enumerate(stages)]

        def skip_branch(l):
            rfm, skip, n, _, c2, _ = stages[l]
            cs = skip.shape[-1]
            s = rfm(skip)
            if not fs[l]:
                return ops.nearest(s, (n, n, n), out=cats[l][..., :cs])         # F.interpolate nearest (mmvit4.py:271-286)
            w_s, w_y = ops.split_weight(c2.conv.weight, cs)
            sc = ops.nearest(s, (3 * skip.shape[1], n, n))                       # the three depth classes of every block
            return ops.conv3d(sc, w_s, None, c2.conv.stride, c2.conv.pad, c2.conv.replicate), w_y

        # skip branches (RFM -> nearest upsample -> their share of d*_c2 / the concat slice) only depend on the early-fusion outputs:
        # run them on a side stream next to the main chain (RFM5 -> up2 -> d*_c1 ...), which needs each of them only at d*_c2
        parts_s = [None] * 4
        if self._side is None:
            self._side = {}
        side = self._side.get(lane)
        use_side = self.concurrent_skips and x5.is_cuda
        if use_side:
            cur = torch.cuda.current_stream()
            if side is None:
                side = self._side[lane] = torch.cuda.Stream(device=dev)
            edges = self._edges.setdefault(lane, _Edges())
            edges.reset()
            # Inside a sample-group lane the skip stream forks from the event the LANES forked from (recorded by _run_lanes on its
            # caller's stream), never from its own lane's stream: hipStreamEndCapture of ROCm 7.2 segfaults as soon as a stream of a
            # capture waits for an event that a FORKED stream recorded and later joins that stream again (fork -> fork -> join -> join;
            # tools/probe/graph_fork2.py: `raw` / `join1` / `onelane` / `origin_lane` crash with torch streams, events and element-wise
            # kernels alone - the round-2 crash "at batch >= 2" was this, no product code involved - while `origin`, the topology used
            # here, captures and replays).  The branch's inputs are all older than that event, so nothing is lost; the concat buffers
            # it writes are then allocated on ITS stream (their previous use, if any, was its own), see below.
            fork = self._fork_from.pop(lane, None)
            if fork is not None:
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    cats = [None if c is None else torch.empty_like(c) for c in cats]
                for c in cats:
                    if c is not None:
                        _rs(c, cur)
            else:
                edges.edge(cur, side)
                for c in cats:
                    if c is not None:
                        _rs(c, side)          # allocated on the caller's stream, written (and saved) on the side stream
            for t in (x1, x2, x3, x4):
                _rs(t, side)
            events = []
            with torch.cuda.stream(side):
                for l in range(4):
                    parts_s[l] = skip_branch(l)
                    events.append(edges.mark(side))
        y = self.RFM5(x5)
        if self.RFM5_reduce is not None:
            y = self.RFM5_reduce(y)
        for l, (rfm, skip, n, c1, c2, cout) in enumerate(stages):
            cs = skip.shape[-1]
            cat = cats[l]
            up = ops.trilinear(y, (n, n, n))                                   # self.up2, align_corners (mmvit4.py:243)
            part_y = c1(up) if fs[l] else c1(up, out=cat[..., cs:])            # d*_c1: 3x3x3 replicate -> ReLU -> IN
            if use_side:
                torch.cuda.current_stream().wait_event(events[l])
                for t in (parts_s[l] if fs[l] else (parts_s[l],)):
                    _rs(t, torch.cuda.current_stream())
            else:
                parts_s[l] = skip_branch(l)
            if fs[l]:
                ys, w_y = parts_s[l]
                # the skip channels' share is broadcast-added by depth class in the convolution's epilogue, which then also takes the
                # InstanceNorm statistics of the sum (no depth_bcast_add pass, no statistics pass over the 64^3 / 128^3 tensors)
                hb, st = {"ys": ys, "done": False}, {"G": part_y.shape[0], "relu": True, "after_bcast": True}
                yc = ops.conv3d(part_y, w_y, c2.conv.bias, c2.conv.stride, c2.conv.pad, c2.conv.replicate, stats=st, bcast=hb)
                y = cout(c2.norm(ops.depth_bcast_add(yc, ys, fused=hb["done"]), pre=st))
            else:
                y = cout(c2(ops.cat_channels(cat, parts_s[l], part_y)))
        up = ops.trilinear(y, (1, 224, 224))                                   # up_to_224 (mmvit4.py:263): depth slice 0 only
        return ops.head(up, self.final_conv.weight, self.final_conv.bias)      # final_conv + sigmoid (mmvit4.py:290-291)


class _Holder(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class LinearP(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-1.0 / math.sqrt(cin), 1.0 / math.sqrt(cin)))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)

    def extra_repr(self):
        return "in_features=%d, out_features=%d, bias=%s" % (self.weight.shape[1], self.weight.shape[0], self.bias is not None)


class LayerNormP(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class SelfAttention(nn.Module):
    """mmvit4.py:295-315"""

    def __init__(self, dim, heads=8, dropout_rate=0.0):
        super().__init__()
        self.num_heads, self.p = heads, dropout_rate
        self.qkv = LinearP(dim, dim * 3, bias=False)
        self.proj = LinearP(dim, dim)

    def forward(self, x):
        o = ops.attention(self.qkv(x), self.num_heads, self.p, self.training)      # softmax(qk^T/sqrt(hd)) -> attn_drop -> @v
        return ops.dropout(self.proj(o), self.p, self.training)                   # proj -> proj_drop


class FeedForward(nn.Module):
    """mmvit4.py:347-358: net.0 Linear, net.1 GELU, net.2 Dropout, net.3 Linear, net.4 Dropout"""

    def __init__(self, dim, hidden, p):
        super().__init__()
        self.p = p
        self.net = nn.ModuleList([LinearP(dim, hidden), nn.Identity(), nn.Identity(), LinearP(hidden, dim), nn.Identity()])

    def forward(self, x):
        h = ops.dropout(ops.gelu(self.net[0](x)), self.p, self.training)
        return ops.dropout(self.net[3](h), self.p, self.training)


class _PreNorm(nn.Module):
    """PreNorm / PreNormDrop (mmvit4.py:324-339): parameters `norm.*`, `fn.*`"""

    def __init__(self, dim, fn, p=None):
        super().__init__()
        self.norm = LayerNormP(dim)
        self.fn = fn
        self.p = p


class Transformer(nn.Module):
    """mmvit4.py:360-388 (depth blocks of Residual(PreNormDrop(attn)) + Residual(PreNorm(ffn)))."""

    def __init__(self, embedding_dim, depth_, heads, mlp, dropout_rate=0.1, n_levels=1, n_points=4):
        super().__init__()
        self.depth = depth_
        self.cross_attention_list = nn.ModuleList(
            [_Holder(_PreNorm(embedding_dim, SelfAttention(embedding_dim, heads, dropout_rate), dropout_rate)) for _ in range(depth_)])
        self.cross_ffn_list = nn.ModuleList(
            [_Holder(_PreNorm(embedding_dim, FeedForward(embedding_dim, mlp, dropout_rate))) for _ in range(depth_)])

    def forward(self, x, pos):
        for j in range(self.depth):
            att, ffn = self.cross_attention_list[j].fn, self.cross_ffn_list[j].fn
            xs, h = ops.layer_norm(x, att.norm.weight, att.norm.bias, pos=pos)             # x = x + pos ; LN(x)
            a = ops.dropout(att.fn(h), att.p, self.training)                                # PreNormDrop's dropout
            x = ops.add(a, xs)                                                              # Residual
            h = ops.layer_norm(x, ffn.norm.weight, ffn.norm.bias)
            x = ops.add(ffn.fn(h), x)
        return x


class MMVit4(nn.Module):
    # parameters that never receive a gradient: the *_decode_conv and seg_* heads are never called and fusion5's
    # output is never consumed (mmvit4.py:453 vs :532).  torch.optim.Adam skips them (grad is None); the data-parallel layer reads
    # this list so that they are never communicated and its buckets exist before the first backward.
    NOGRAD_PREFIXES = ("RGB_decode_conv.", "NIR_decode_conv.", "SWIR_decode_conv.", "decoder_fuse.seg_d1.", "decoder_fuse.seg_d2.",
                       "decoder_fuse.seg_d3.", "decoder_fuse.seg_d4.", "decoder_fuse.seg_layer.", "fusion5.conv.")

    def __init__(self, num_cls=1):
        super().__init__()
        d8, T = basic_dims * 8, transformer_basic_dims
        self.RGB_encoder, self.NIR_encoder, self.SWIR_encoder = Encoder(), Encoder(), Encoder()
        self.RGB_encode_conv, self.NIR_encode_conv, self.SWIR_encode_conv = Conv3dP(d8, T, 1), Conv3dP(d8, T, 1), Conv3dP(d8, T, 1)
        self.fused6_encode_conv = Conv3dP(d8 * 3, T, 1)
        self.RGB_decode_conv, self.NIR_decode_conv, self.SWIR_decode_conv = Conv3dP(T, d8, 1), Conv3dP(T, d8, 1), Conv3dP(T, d8, 1)
        for m in _MODS + ("fused6",):
            setattr(self, m + "_pos", nn.Parameter(torch.zeros(1, patch_size ** 3, T)))
        self.RGB_transformer = Transformer(T, depth, num_heads, mlp_dim)
        self.NIR_transformer = Transformer(T, depth, num_heads, mlp_dim)
        self.SWIR_transformer = Transformer(T, depth, num_heads, mlp_dim)
        self.qkv_RGB, self.qkv_NIR, self.qkv_SWIR = Conv3dP(T, T * 3, 1), Conv3dP(T, T * 3, 1), Conv3dP(T, T * 3, 1)
        self.multimodal_transformer = Transformer(T, depth, num_heads, mlp_dim, n_levels=3)
        self.multimodal_decode_conv = Conv3dP(T * 4, d8 * 3, 1)
        self.decoder_fuse = Decoder_fuse(num_cls=num_cls)
        for i, c in enumerate((basic_dims, basic_dims * 2, basic_dims * 4, d8, d8, d8)):
            setattr(self, "fusion%d" % (i + 1), EarlyFusionBlock(c))
        # The three modality branches (encoder -> tokens -> intra-modality transformer -> qkv) are independent until the
        # inter-modal correlation: run them on three HIP streams so that their small late-stage launches (e4/e5: 1-2
        # workgroups per CU each) fill the 256 CUs together.  Same kernels, same order per branch: results are unchanged.
        self.concurrent_branches = True
        # The three encoders are the SAME network on three inputs (mmvit4.py:442-447): their twin layers can run as ONE grouped launch
        # each (stacked activations, stacked weights; _encoders_grouped) - 3x fewer, 3x larger launches whose tile grids fill the chip
        # more evenly than three small ones (encoder GEMMs +10-25 % in isolation).  The price is the overlap of one branch's HBM-bound
        # BatchNorm passes with another branch's matrix work, which only the three-stream schedule has.  Measured (round 3, fwd+bwd,
        # 4 bands 224^2, ms per step grouped / per modality): B=32 259 / 247, B=16 135 / 129, B=8 75.0 / 73.4, B=4 45.7 / 62.2,
        # B=2 44.0 / 58.5.  None = choose per call (encoders_grouped_for): grouped when the step is launch-bound (small batches), per
        # modality otherwise; True / False force one.  NOT grouped near the HBM capacity even though that schedule is single-stream anyway:
        # the stacked activations are 3x larger allocations and the caching allocator then fragments (B = 64, 8 bands, 256^2: 3.5 s per
        # step grouped vs 1.0 s per modality at the same 230 GB peak; 12 bands 512^2 at B = 16: 4.5 s vs 1.17 s).
        self.grouped_encoders = None
        self._gcache = None
        self.interleave_branches = True      # enqueue the three branches layer by layer in turn (diagnostics / A-B: False = branch after branch)
        self._streams = None
        # The decoder has no cross-sample coupling (InstanceNorm is per sample, no BatchNorm): run it as two half-batch chains on
        # two streams so that the HBM-bound stages of one half (norm statistics / apply, resampling) overlap the MFMA-bound stages
        # of the other.  Forward values are unchanged; decoder weight gradients become the sum of two half-batch reductions.
        self.decoder_split = 2          # number of sample groups (0 / 1 = off)
        self._dec_streams = None
        self._edges = _Edges()
        for enc in (self.RGB_encoder, self.NIR_encoder, self.SWIR_encoder):      # used once per step: weight gradients may run beside
            for m in enc.modules():                                               # the branch's normalisation passes (ops.SIDE_WGRAD)
                if isinstance(m, Conv3dP):
                    m.side_wgrad = True
        self.auto_streams = True        # single-stream schedule when the step's autograd state nears the HBM capacity
        self._mem_plan = {}             # input shape -> single-stream? (see _wants_single_stream)

    def flush_counters(self):
        """bring every BatchNorm's `num_batches_tracked` buffer up to date (see BatchNorm3dP)"""
        for m in self.modules():
            if isinstance(m, BatchNorm3dP):
                m.flush_counter()

    def named_buffers(self, *a, **kw):
        self.flush_counters()
        return super().named_buffers(*a, **kw)

    @staticmethod
    def _level_shapes(B, D, H, W):
        def o(v, k, s, p):
            return (v + 2 * p - k) // s + 1
        h, w = o(o(H, 7, 2, 3), 3, 2, 1), o(o(W, 7, 2, 3), 3, 2, 1)
        shapes = [(B, D, h, w)]
        shapes.append((B, D, h, w))
        for _ in range(3):
            h, w = o(h, 3, 2, 1), o(w, 3, 2, 1)
            shapes.append((B, D, h, w))
        return shapes

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("MMVit4 (MI355X build) runs on the GPU only: there is no CPU fall-back path")
        if x.dtype != torch.float32 or x.dim() != 5 or x.shape[1] != num_modals:
            raise ValueError("expected fp32 input [B, 3, D, H, W]")
        B = x.shape[0]
        P3, T = patch_size, transformer_basic_dims
        if self.auto_streams and torch.is_grad_enabled() and self._wants_single_stream(x):
            # Near the HBM capacity the multi-stream schedule back-fires: the caching allocator keeps one pool PER STREAM, so blocks
            # cached by one stream cannot serve another and the step degenerates into hipFree / hipMalloc retries (measured at
            # B=64, 8 bands, 256^2 = 250 GB of autograd state: 10.4 s per step on seven streams, 1.16 s on one).
            saved = (self.concurrent_branches, self.decoder_split, self.decoder_fuse.concurrent_skips)
            self.concurrent_branches, self.decoder_split, self.decoder_fuse.concurrent_skips = False, 0, False
            try:
                return self._forward(x)
            finally:
                self.concurrent_branches, self.decoder_split, self.decoder_fuse.concurrent_skips = saved
        return self._forward(x)

    def _wants_single_stream(self, x):
        """Memory plan per input shape.  The first training step at a shape is scheduled from an a-priori estimate of its autograd state
        (`_memory_limited`); the following steps from what the steps before it NEEDED, read off the process-wide high-water mark
        `torch.cuda.max_memory_allocated` - which is only read, never reset (round 2 reset it around its measurements and so clobbered
        the caller's statistics): when the mark rose during this shape's steps it IS their peak, otherwise it is an upper bound of it.
        Multi-stream while that stays under 55 % of the device memory, single-stream above 75 %, no change in between (the multi-stream
        schedule needs more than the single-stream one: per-stream pools).  A change of schedule restarts the three-step window, so a
        flip to multi-stream that then exceeds 75 % is taken back; after three steps without a change the plan of the shape is final."""
        key, dev = tuple(x.shape), x.device
        total = torch.cuda.get_device_properties(dev).total_memory
        peak = torch.cuda.max_memory_allocated(dev)
        plan = self._mem_plan.get(key)
        if plan is None:
            self._mem_plan[key] = plan = {"single": _memory_limited(x), "steps": 0, "mark": peak}
            return plan["single"]
        if plan["steps"] < 3:
            if peak < plan["mark"]:               # the caller reset the statistics in between: this sample says nothing about the step
                plan["mark"] = peak
                return plan["single"]
            rose = peak > plan["mark"]
            if rose and peak > 0.75 * total and not plan["single"]:
                plan["single"], plan["steps"] = True, 0
            elif peak < 0.55 * total and plan["single"]:
                plan["single"], plan["steps"] = False, 0
            else:
                plan["steps"] += 1
            plan["mark"] = peak
        return plan["single"]

    def encoders_grouped_for(self, x):
        """the encoder schedule `forward` uses for input x under the current switches (see __init__)"""
        if self.grouped_encoders is not None:
            return bool(self.grouped_encoders)
        B, _, D, Hh, W = x.shape
        rows_e4 = B * D * ((Hh + 15) // 16) * ((W + 15) // 16)          # GEMM rows of one modality at the e4 level
        return rows_e4 < 5000                  # B <= 6 at 4 bands 224^2: one modality's launches leave most of the 256 CUs idle

    def _encoders_grouped(self, x, cats):
        """The three modality encoders as one stacked pass.  Returns (cat[0..5], tok): the six early-fusion concat buffers as autograd
        tensors (the grouped adapt / conv6 convolutions write every modality's channel slice in one launch) and the encode_conv tokens
        of all modalities stacked along the batch axis [3B, 512, 512]."""
        encs = [getattr(self, m + "_encoder") for m in _MODS]
        G, B = num_modals, x.shape[0]
        if self._gcache is None:
            gc = {"e1_bn": _GBN([e.e1_bn for e in encs]), "layers": [], "adapt": [], }
            for name in ("e2", "e3", "e4", "e5"):
                layers = [getattr(e, name) for e in encs]
                gc["layers"].append((layers, [_GroupedBlock([l[i] for l in layers]) for i in range(len(layers[0]))]))
            for l in range(5):
                gc["adapt"].append(_GConv([getattr(e, "adapt%d" % (l + 1)) for e in encs], "stack", "cat"))
            gc["conv6"] = _GConv([e.conv6 for e in encs], "stack", "cat")
            gc["encode"] = _GConv([getattr(self, m + "_encode_conv") for m in _MODS], "cat", "stack")
            self._gcache = gc
        gc = self._gcache
        train = self.training
        h1, w1 = (x.shape[3] - 1) // 2 + 1, (x.shape[4] - 1) // 2 + 1
        y0 = torch.empty((G * B, x.shape[2], h1, w1, 64), dtype=torch.float32, device=x.device)
        parts = [encs[g].e1_c1(x[:, g], out=y0[g * B:(g + 1) * B]) for g in range(G)]        # the stems: three launches into one buffer
        f = ops.maxpool133(gc["e1_bn"](ops.cat_batch_inplace(y0, *parts), relu_in=True))     # conv -> ReLU -> BN -> pool (mmvit4.py:172-174)
        feats, links = [f], []
        tap = GRAD_TAP and torch.is_grad_enabled()
        for layers, blocks in gc["layers"]:
            link = {} if tap and f.requires_grad else None
            links.append(link)
            fin = f
            for i, blk in enumerate(blocks):
                fprev = f
                f = _bottleneck(blk, f, link if i == 0 else None, train)
                _fire_hooks(blk.blocks, fprev, f, "stack", "stack")
            _fire_hooks(layers, fin, f, "stack", "stack")
            feats.append(f)
        links.append(None)
        cube = torch.empty((G * B, 8, 8, 8, basic_dims * 23), dtype=torch.float32, device=x.device)
        outs, cparts, off = [], [], 0
        for l, f in enumerate(feats):
            c = _ADAPT[l][1]
            fa = f if links[l] is None else ops.grad_tap(f, links[l])
            xl = gc["adapt"][l](fa, out=cats[l])                                         # all three modalities' slices of the concat buffer
            outs.append(xl)
            cparts.append(ops.resample_groups_to_cube(xl, cube, off, c, G))              # mmvit4.py:187-191
            off += c
        x6 = gc["conv6"](ops.cat_channels(cube, *cparts), out=cats[5])
        outs.append(x6)
        tok = gc["encode"](x6).view(G * B, patch_size ** 3, transformer_basic_dims)       # channels-last == token layout (:458-461)
        return outs, tok

    def _forward(self, x):
        B = x.shape[0]
        P3, T = patch_size, transformer_basic_dims
        self._edges.reset()
        # early-fusion concat buffers, allocated up front on the caller's stream (the branches fill their channel slices)
        cats = [torch.empty(sh + (num_modals * c,), dtype=torch.float32, device=x.device)
                for sh, (_, c) in zip(self._level_shapes(B, x.shape[2], x.shape[3], x.shape[4]), _ADAPT)]
        cats.append(torch.empty((B, P3, P3, P3, num_modals * basic_dims * 8), dtype=torch.float32, device=x.device))
        feats, skip, qkv = [None] * 3, [None] * 3, [None] * 3
        catf = None
        if self.encoders_grouped_for(x):
            catf, tok_all = self._encoders_grouped(x, cats)
            toks = ops.split_batch(tok_all, [B * k for k in range(num_modals + 1)])

        def branch(i, m, interleaved):
            """one modality branch as a generator (yields between the encoder's layers and before the transformer)"""
            if catf is None:
                enc = getattr(self, m + "_encoder")
                if interleaved:
                    feats[i] = yield from enc.steps(x[:, i], i, cats)
                else:
                    feats[i] = enc(x[:, i], i, cats)
                tok = getattr(self, m + "_encode_conv")(feats[i][5]).view(B, P3 ** 3, T)       # channels-last == token layout (:458-461)
            else:
                feats[i] = []
                tok = toks[i]
            skip[i] = tok
            yield
            tr = getattr(self, m + "_transformer")(tok, getattr(self, m + "_pos"))
            qkv[i] = getattr(self, "qkv_" + m)(tr.view(B, P3, P3, P3, T)).view(B, P3 ** 3, 3 * T)

        if self.concurrent_branches:
            cur = torch.cuda.current_stream()
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=x.device) for _ in range(num_modals)]
            for i, m in enumerate(_MODS):
                st = self._streams[i]
                self._edges.edge(cur, st)
                # tensors allocated on the caller's stream that this branch reads / writes (also from its saved-for-backward
                # state): tell the caching allocator, or their memory could be re-used while the branch stream still needs it
                if catf is None:
                    _rs(x, st)
                    for c in cats:
                        _rs(c, st)
                else:
                    _rs(tok_all, st)
            # The host enqueues the three branches LAYER BY LAYER in turn.  Branch after branch, the first stream ran alone while the
            # host was still enqueuing it (and, because autograd replays nodes in reverse creation order, the last stream ran alone at
            # the end of the backward while its 1/3 of the nodes were the only ones left): with interleaved creation order all three
            # streams are fed from the first to the last millisecond of both passes.  Per branch the kernels and their order are the
            # same, so results are unchanged.
            gens = [branch(i, m, self.interleave_branches) for i, m in enumerate(_MODS)]
            alive = list(range(num_modals))
            while alive:
                for i in list(alive):
                    with torch.cuda.stream(self._streams[i]):
                        try:
                            next(gens[i])
                        except StopIteration:
                            alive.remove(i)
            for st in self._streams:
                self._edges.edge(st, cur)
            for i in range(num_modals):        # branch outputs are consumed on the caller's stream from here on
                for t in feats[i] + [skip[i], qkv[i]]:
                    _rs(t, cur)
        else:
            for i, m in enumerate(_MODS):
                for _ in branch(i, m, False):
                    pass
        if catf is not None:
            fused = [getattr(self, "fusion%d" % (l + 1))(catf[l]) for l in range(6)]
        else:
            fused = [getattr(self, "fusion%d" % (l + 1))(ops.cat_channels(cats[l], *[feats[i][l] for i in range(num_modals)]))
                     for l in range(6)]        # fused[4] (fusion5) is computed and never consumed, as in the reference (:453)
        corr = ops.inter_corr(qkv[0], qkv[1], qkv[2])                                      # mmvit4.py:481-503
        mm = [ops.add(skip[i], corr[i]) for i in range(num_modals)]
        mm.append(self.fused6_encode_conv(fused[5]).view(B, P3 ** 3, T))
        pos = ops.cat_tokens(self.RGB_pos, self.NIR_pos, self.SWIR_pos, self.fused6_pos)
        tokens = ops.cat_tokens(*mm)                                                       # [B, 2048, 512]

        def tail(tok, f1, f2, f3, f4, lane=0):
            nb = tok.shape[0]
            y = self.multimodal_transformer(tok, pos)
            x6 = self.multimodal_decode_conv(y.view(nb, P3, P3, P3, 4 * T))                # 4 tokens -> one voxel (mmvit4.py:526)
            return self.decoder_fuse(f1, f2, f3, f4, x6, lane=lane)

        pred = _run_lanes(self, tail, pos, tokens, fused[0], fused[1], fused[2], fused[3])
        if pred.requires_grad and int(self.decoder_split or 0) >= 2:
            pred = _QuietBackwardFn.apply(pred)
        return pred


def _memory_limited(x, frac=0.6):
    """True when the autograd state of one training step is estimated to exceed `frac` of the device memory.  Fit to measured peaks
    (B=32: 74.5 GB at 4 x 224^2, 124.5 GB at 8 x 256^2; B=64: 248 GB at 8 x 256^2): the decoder costs 1.36 GB per sample whatever the
    input size, the three encoders 4.82 kB per input voxel."""
    B, _, D, Hh, W = x.shape
    est = B * (1.36e9 + 4.82e3 * D * Hh * W)
    return est > frac * torch.cuda.get_device_properties(x.device).total_memory


def _run_lanes(model, tail, shared, *per_sample):
    """Everything after the inter-modal correlation is per sample: the multimodal transformer and the decoder run as
    `model.decoder_split` sample-group chains ("lanes") on separate streams, so one chain's softmax / LayerNorm / InstanceNorm /
    resampling passes overlap the other's matrix work.  tail(*tensors, lane=k) -> prediction of that sample group; `shared` is a
    tensor every lane reads (the positional embedding)."""
    B = per_sample[0].shape[0]
    lanes = min(int(model.decoder_split), B) if model.decoder_split else 1
    if lanes >= 2 and torch.cuda.is_current_stream_capturing() and not CAPTURE_LANES:
        lanes = 1          # diagnostics switch only (round 2 kept the lanes out of captures; the cause and the fix are in Decoder_fuse.forward)
    if lanes < 2:
        return tail(*per_sample)
    cur = torch.cuda.current_stream()
    if model._dec_streams is None or len(model._dec_streams) < lanes:
        model._dec_streams = [torch.cuda.Stream(device=per_sample[0].device) for _ in range(lanes)]
    bounds = [B * k // lanes for k in range(lanes + 1)]
    ins = [ops.split_batch(t, bounds) for t in per_sample]
    outs = []
    fork = model._edges.mark(cur)            # the lanes AND their decoder skip streams fork from here (one fork level, see Decoder_fuse.forward)
    dec = getattr(model, "decoder_fuse", None)
    for k in range(lanes):
        st = model._dec_streams[k]
        st.wait_event(fork)
        if dec is not None and dec.concurrent_skips:
            dec._fork_from[k] = fork
        _rs(shared, st)
        for t in ins:
            _rs(t[k], st)
        with torch.cuda.stream(st):
            outs.append(tail(*[t[k] for t in ins], lane=k))
    for k in range(lanes):
        model._edges.edge(model._dec_streams[k], cur)
        _rs(outs[k], cur)
    return ops.cat_batch(*outs)


def Jaccard2(y, y_pred, epsilon=1e-8):
    """F5_JACCARD2.py:11-20 on the device; returns a [1] tensor like the reference."""
    return ops.jaccard_all(y, y_pred, epsilon)[0:1]


def Jaccard(y, y_pred, epsilon=1e-8):
    return ops.jaccard_all(y, y_pred, epsilon)[1:2]


def JaccardAndF1(y, y_pred, epsilon=1e-8):
    return ops.jaccard_all(y, y_pred, epsilon)[2:3]
