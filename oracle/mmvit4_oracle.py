"""ORACLE - test infrastructure only.  NOT part of the product path.

CPU restatement (stock PyTorch ops) of the CorrIFNet `MMVit4` hot path, written
from the behavioural spec in SURVEY.md section 8(a).  It exists so that
  * tests/ can check the HIP path against it on identical seeded inputs,
  * __graft_entry__.smoke() can check one small invocation,
  * bench.py's `cpu_baseline` leg can time the CPU path on the GPU box's host.
Nothing under the product package imports this file.

Parity pin: `tests/golden/make_golden.py` imports the reference's own
`mmvit4.py` in the development container, loads the same deterministic
state-dict into it and into this restatement, and commits the reference's
outputs / gradients as fixtures under tests/golden/.  `tests/test_oracle_golden.py`
re-checks this file against those fixtures wherever the tests run.

Reference citations are `file:line` inside the upstream repository.
The parameter tree reproduces the reference's 1140 state-dict keys
(mmvit4.py:391-439) so that reference checkpoints load here unchanged.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# mmvit4.py:10-16
BASE = 8            # basic_dims
TOK = 512           # transformer_basic_dims
MLP = 512           # mlp_dim
HEADS = 8           # num_heads
NMOD = 3            # num_modals
PATCH = 8           # patch_size
MODS = ("RGB", "NIR", "SWIR")
LAYER_BLOCKS = (3, 4, 6, 3)      # torchvision ResNet-50 topology used at mmvit4.py:117,154-157
LAYER_WIDTH = (64, 128, 256, 512)


def _conv(ci, co, k=1, stride=1, pad=0, bias=True, mode="zeros"):
    return nn.Conv3d(ci, co, k, stride, pad, bias=bias, padding_mode=mode)


class ConvReluIN(nn.Module):
    """general_conv3d_prenorm, mmvit4.py:29-45: conv(bias) -> ReLU -> InstanceNorm3d."""

    def __init__(self, ci, co, k=3, pad=1, mode="zeros"):
        super().__init__()
        self.conv = _conv(ci, co, k, 1, pad, True, mode)
        self.norm = nn.InstanceNorm3d(co)

    def forward(self, x):
        return self.norm(F.relu(self.conv(x)))


class RFM(nn.Module):
    """fusion_prenorm, mmvit4.py:47-56: 1x1x1, 3x3x3 (zero pad), 1x1x1; C -> C."""

    def __init__(self, c):
        super().__init__()
        self.fusion_layer = nn.Sequential(ConvReluIN(c, c, 1, 0), ConvReluIN(c, c, 3, 1), ConvReluIN(c, c, 1, 0))

    def forward(self, x):
        return self.fusion_layer(x)


class EarlyFusion(nn.Module):
    """EarlyFusionBlock, mmvit4.py:64-81: cat(3 modalities) -> 1x1x1 -> ReLU -> IN."""

    def __init__(self, c):
        super().__init__()
        self.conv = _conv(NMOD * c, NMOD * c)
        self.norm = nn.InstanceNorm3d(NMOD * c)

    def forward(self, a, b, c):
        return self.norm(F.relu(self.conv(torch.cat([a, b, c], 1))))


class Bottleneck(nn.Module):
    """Bottleneck3D, mmvit4.py:196-212 (convs inflated at :126-151, no conv bias)."""

    def __init__(self, cin, width, stride, down):
        super().__init__()
        cout = 4 * width
        self.conv1 = _conv(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm3d(width)
        self.conv2 = _conv(width, width, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(width)
        self.conv3 = _conv(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm3d(cout)
        self.downsample = (
            nn.Sequential(_conv(cin, cout, 1, (1, stride, stride), bias=False), nn.BatchNorm3d(cout)) if down else None
        )

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt)


def _layer(cin, width, n, stride):
    blocks = [Bottleneck(cin, width, stride, True)]
    blocks += [Bottleneck(4 * width, width, 1, False) for _ in range(n - 1)]
    return nn.Sequential(*blocks)


class Encoder(nn.Module):
    """Encoder, mmvit4.py:113-194."""

    def __init__(self):
        super().__init__()
        self.e1_c1 = _conv(1, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), bias=False)
        self.e1_bn = nn.BatchNorm3d(64)
        cin = 64
        for name, width, n, stride in zip(("e2", "e3", "e4", "e5"), LAYER_WIDTH, LAYER_BLOCKS, (1, 2, 2, 2)):
            setattr(self, name, _layer(cin, width, n, stride))
            cin = 4 * width
        self.conv6 = _conv(BASE * 23, BASE * 8)
        for i, (ci, co) in enumerate(((64, BASE), (256, 2 * BASE), (512, 4 * BASE), (1024, 8 * BASE), (2048, 8 * BASE))):
            setattr(self, "adapt%d" % (i + 1), _conv(ci, co))

    def forward(self, x):
        # mmvit4.py:172-174: conv -> ReLU -> BN (BN after ReLU) -> max-pool
        f1 = F.max_pool3d(self.e1_bn(F.relu(self.e1_c1(x))), (1, 3, 3), (1, 2, 2), (0, 1, 1))
        f2 = self.e2(f1)
        f3 = self.e3(f2)
        f4 = self.e4(f3)
        f5 = self.e5(f4)
        xs = [self.adapt1(f1), self.adapt2(f2), self.adapt3(f3), self.adapt4(f4), self.adapt5(f5)]
        cube = [F.interpolate(t, size=(8, 8, 8), mode="trilinear", align_corners=True) for t in xs]
        return xs + [self.conv6(torch.cat(cube, 1))]


class Decoder(nn.Module):
    """Decoder_fuse, mmvit4.py:222-292 (unused seg_* heads kept for the state-dict)."""

    def __init__(self, num_cls=1):
        super().__init__()
        rep = "replicate"
        b = BASE
        self.d4_c1 = ConvReluIN(16 * b, 16 * b, 3, 1, rep)
        self.d4_c2 = ConvReluIN(320, 8 * b, 3, 1, rep)
        self.d4_out = ConvReluIN(8 * b, 8 * b, 1, 0, rep)
        self.d3_c1 = ConvReluIN(8 * b, 4 * b, 3, 1, rep)
        self.d3_c2 = ConvReluIN(128, 4 * b, 3, 1, rep)
        self.d3_out = ConvReluIN(4 * b, 4 * b, 1, 0, rep)
        self.d2_c1 = ConvReluIN(4 * b, 2 * b, 3, 1, rep)
        self.d2_c2 = ConvReluIN(64, 2 * b, 3, 1, rep)
        self.d2_out = ConvReluIN(2 * b, 2 * b, 1, 0, rep)
        self.d1_c1 = ConvReluIN(2 * b, b, 3, 1, rep)
        self.d1_c2 = ConvReluIN(32, b, 3, 1, rep)
        self.d1_out = ConvReluIN(b, b, 1, 0, rep)
        self.seg_d4 = _conv(8 * b, num_cls)
        self.seg_d3 = _conv(8 * b, num_cls)
        self.seg_d2 = _conv(4 * b, num_cls)
        self.seg_d1 = _conv(2 * b, num_cls)
        self.seg_layer = _conv(b, num_cls)
        self.RFM5 = RFM(24 * b)
        self.RFM5_reduce = _conv(24 * b, 16 * b)
        self.RFM4 = RFM(24 * b)
        self.RFM3 = RFM(12 * b)
        self.RFM2 = RFM(6 * b)
        self.RFM1 = RFM(3 * b)
        self.final_conv = _conv(8, 3)

    @staticmethod
    def _up2(t):
        return F.interpolate(t, scale_factor=2, mode="trilinear", align_corners=True)

    @staticmethod
    def _nearest(t, size):
        """F.interpolate(t, size) (nearest, mmvit4.py:271,276,281,286).  On the CPU that is exactly what runs.  When the tests evaluate
        this module on the GPU box's device (sizes the host would need hours for), ATen's device kernel for the BACKWARD of
        upsample_nearest3d assigns output voxels to sources by ceil(i * out/in) in floating point, which is not the adjoint of its own
        forward floor(o * in/out) when out/in is inexact (128/56: output row 16 is read from source 7 but its gradient goes to source
        6).  The CPU kernel - what the reference runs - scatters with the forward's index.  So on a device the same forward indices
        are applied with index_select, whose backward is the exact adjoint."""
        if not t.is_cuda:
            return F.interpolate(t, size)
        for ax, (n_in, n_out) in enumerate(zip(t.shape[2:], size)):
            scale = torch.tensor(float(n_in), dtype=torch.float32) / torch.tensor(float(n_out), dtype=torch.float32)
            idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * scale).long().clamp_(max=n_in - 1)
            t = t.index_select(2 + ax, idx.to(t.device))
        return t

    def forward(self, x1, x2, x3, x4, x5):
        y = self.d4_c1(self._up2(self.RFM5_reduce(self.RFM5(x5))))
        for rfm, skip, size, c2, out, c1 in (
            (self.RFM4, x4, 16, self.d4_c2, self.d4_out, self.d3_c1),
            (self.RFM3, x3, 32, self.d3_c2, self.d3_out, self.d2_c1),
            (self.RFM2, x2, 64, self.d2_c2, self.d2_out, self.d1_c1),
            (self.RFM1, x1, 128, self.d1_c2, self.d1_out, None),
        ):
            s = self._nearest(rfm(skip), (size, size, size))
            y = out(c2(torch.cat((s, y), 1)))
            if c1 is not None:
                y = c1(self._up2(y))
        y = F.interpolate(y, size=(1, 224, 224), mode="trilinear", align_corners=True)   # mmvit4.py:263,289
        return torch.sigmoid(self.final_conv(y))


class Attention(nn.Module):
    """SelfAttention, mmvit4.py:295-315."""

    def __init__(self, dim, heads, p):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.attn_drop = nn.Dropout(p)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(p)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        a = self.attn_drop(torch.softmax((q @ k.transpose(-2, -1)) * self.scale, -1))
        return self.proj_drop(self.proj((a @ v).transpose(1, 2).reshape(B, N, C)))


class _Fn(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class NormFn(nn.Module):
    """PreNorm / PreNormDrop, mmvit4.py:324-339 (dropout present only in the attention branch)."""

    def __init__(self, dim, fn, p=None):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        if p is not None:
            self.dropout = nn.Dropout(p)
        self.fn = fn
        self._p = p

    def forward(self, x):
        y = self.fn(self.norm(x))
        return self.dropout(y) if self._p is not None else y


class FFN(nn.Module):
    """FeedForward, mmvit4.py:347-358; net.{0,3} are the two Linears."""

    def __init__(self, dim, hidden, p):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(p), nn.Linear(hidden, dim), nn.Dropout(p))

    def forward(self, x):
        return self.net(x)


class Res(_Fn):
    def forward(self, x):
        return self.fn(x) + x


class TransformerBlock(nn.Module):
    """Transformer(depth=1), mmvit4.py:360-388."""

    def __init__(self, dim=TOK, heads=HEADS, mlp=MLP, p=0.1):
        super().__init__()
        self.cross_attention_list = nn.ModuleList([Res(NormFn(dim, Attention(dim, heads, p), p))])
        self.cross_ffn_list = nn.ModuleList([Res(NormFn(dim, FFN(dim, mlp, p)))])

    def forward(self, x, pos):
        for att, ffn in zip(self.cross_attention_list, self.cross_ffn_list):
            x = ffn(att(x + pos))
        return x


def inter_corr(q, ks, vs):
    """inter_attn closure, mmvit4.py:481-487.

    scores_i = q * k_i (element-wise); the three flattened score rows are
    soft-maxed along the key axis after division by sqrt(3); the [3, B*C*S]
    buffer is then RE-VIEWED as [B, 3*C, S...] before weighting the values,
    which mixes samples when B > 1.  Reproduced literally.
    """
    B, C = q.shape[:2]
    rows = torch.stack([(q * k).reshape(-1) for k in ks], 0)
    attn = torch.softmax(rows / math.sqrt(len(ks)), 0).reshape(B, len(ks) * C, *q.shape[2:])
    out = attn[:, 0:C] * vs[0]
    for i in range(1, len(vs)):
        out = out + attn[:, i * C:(i + 1) * C] * vs[i]
    return out


class MMVit4(nn.Module):
    """MMVit4, mmvit4.py:391-532."""

    def __init__(self, num_cls=1):
        super().__init__()
        for m in MODS:
            setattr(self, m + "_encoder", Encoder())
        for m in MODS:
            setattr(self, m + "_encode_conv", _conv(8 * BASE, TOK))
        self.fused6_encode_conv = _conv(24 * BASE, TOK)
        for m in MODS:                                  # never called (mmvit4.py:404-406), state-dict only
            setattr(self, m + "_decode_conv", _conv(TOK, 8 * BASE))
        for m in MODS + ("fused6",):
            setattr(self, m + "_pos", nn.Parameter(torch.zeros(1, PATCH ** 3, TOK)))
        for m in MODS:
            setattr(self, m + "_transformer", TransformerBlock())
        for m in MODS:
            setattr(self, "qkv_" + m, _conv(TOK, 3 * TOK))
        self.multimodal_transformer = TransformerBlock()
        self.multimodal_decode_conv = _conv(4 * TOK, 24 * BASE)
        self.decoder_fuse = Decoder(num_cls)
        for i, c in enumerate((BASE, 2 * BASE, 4 * BASE, 8 * BASE, 8 * BASE, 8 * BASE)):
            setattr(self, "fusion%d" % (i + 1), EarlyFusion(c))
        for m in self.modules():                        # mmvit4.py:437-439
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        B = x.shape[0]
        feats = [getattr(self, m + "_encoder")(x[:, i:i + 1]) for i, m in enumerate(MODS)]
        fused = [getattr(self, "fusion%d" % (l + 1))(feats[0][l], feats[1][l], feats[2][l]) for l in range(6)]
        # (fused[4] = fusion5 output is never consumed: mmvit4.py:453 vs :532)

        def tokens(t):          # NCDHW -> [B, D*H*W, C]   (mmvit4.py:458-461)
            return t.permute(0, 2, 3, 4, 1).reshape(B, -1, TOK)

        skip, trans = [], []
        for i, m in enumerate(MODS):
            tok = tokens(getattr(self, m + "_encode_conv")(feats[i][5]))
            skip.append(tok)
            trans.append(getattr(self, m + "_transformer")(tok, getattr(self, m + "_pos")))

        qs, ks, vs = [], [], []
        for i, m in enumerate(MODS):
            vol = trans[i].reshape(B, PATCH, PATCH, PATCH, TOK).permute(0, 4, 1, 2, 3)     # mmvit4.py:473-475
            q, k, v = getattr(self, "qkv_" + m)(vol).chunk(3, 1)
            qs.append(q), ks.append(k), vs.append(v)
        corr = [tokens(inter_corr(q, ks, vs)) for q in qs]            # mmvit4.py:489-503
        mm = [s + c for s, c in zip(skip, corr)]
        mm.append(tokens(self.fused6_encode_conv(fused[5])))
        pos = torch.cat([getattr(self, m + "_pos") for m in MODS + ("fused6",)], 1)
        y = self.multimodal_transformer(torch.cat(mm, 1), pos)          # [B, 2048, 512]
        # mmvit4.py:526: 4 consecutive tokens become one voxel's 2048 channels
        vol = y.reshape(B, PATCH, PATCH, PATCH, 4 * TOK).permute(0, 4, 1, 2, 3).contiguous()   # :526-528 (dense NCDHW copy)
        x6 = self.multimodal_decode_conv(vol)
        return self.decoder_fuse(fused[0], fused[1], fused[2], fused[3], x6)


def set_dropout(model, on):
    """'train-nodrop' mode helper: BN in batch-stat mode, Dropout modules off."""
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.train(on)
    return model


def train_step_loss(pred, mask):
    """F4_TRAIN.py:58-60: BCEWithLogitsLoss applied to the already-sigmoided output (reference quirk, kept)."""
    return F.binary_cross_entropy_with_logits(pred, mask)
