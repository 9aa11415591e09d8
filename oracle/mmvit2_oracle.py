"""ORACLE - test infrastructure only.  NOT part of the product path.

CPU restatement (stock PyTorch ops) of the sibling model `MMVit2` (SURVEY.md section 8(f), row N4;
upstream `mmmvit2.py:345-478`).  It shares the decoder blocks, the transformers and the inter-modal
correlation construct with `MMVit4` (restated in `mmvit4_oracle.py`); what differs is
  * the encoder: a plain full-resolution 3-D conv pyramid (mmmvit2.py:57-104), replicate padding,
    stride-2 3x3x3 down-sampling in ALL three axes, `x + c3(c2(x))` residuals, nearest (not trilinear)
    resampling of the five levels to 8^3;
  * no early-fusion convs: the decoder's skips are plain channel concatenations of the three
    modalities (mmmvit2.py:420-434);
  * the multimodal transformer sees the three correlation outputs only (1536 tokens, no residual with
    the intra-modality tokens, no fused stream) and 3 consecutive tokens are re-viewed as one voxel's
    1536 channels (mmmvit2.py:463-472);
  * the decoder has no `RFM5_reduce`: `d4_c1` is 192 -> 128 (mmmvit2.py:120,163-166).

Parity pin: `tests/golden/make_golden_mmvit2.py` imports the reference's own `mmmvit2.py` in the
development container (it needs nothing but torch/numpy), loads the same deterministic state-dict into it
and into this restatement and commits the reference's outputs / gradients as fixtures under tests/golden/.
The parameter tree reproduces the reference's state-dict keys so that reference checkpoints load unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mmvit4_oracle import BASE, MODS, NMOD, PATCH, TOK, ConvReluIN, RFM, TransformerBlock, _conv, inter_corr


class ConvReluINs(ConvReluIN):
    """general_conv3d_prenorm with a stride (mmmvit2.py:27-43)."""

    def __init__(self, ci, co, k=3, pad=1, mode="zeros", stride=1):
        super().__init__(ci, co, k, pad, mode)
        self.conv = nn.Conv3d(ci, co, k, stride, pad, bias=True, padding_mode=mode)


class Encoder2(nn.Module):
    """Encoder, mmmvit2.py:57-104."""

    def __init__(self):
        super().__init__()
        rep, b = "replicate", BASE
        self.e1_c1 = nn.Conv3d(1, b, 3, 1, 1, bias=True, padding_mode=rep)
        self.e1_c2 = ConvReluINs(b, b, mode=rep)
        self.e1_c3 = ConvReluINs(b, b, mode=rep)
        for lvl, (ci, co) in zip((2, 3, 4, 5), ((b, 2 * b), (2 * b, 4 * b), (4 * b, 8 * b), (8 * b, 8 * b))):
            setattr(self, "e%d_c1" % lvl, ConvReluINs(ci, co, mode=rep, stride=2))
            setattr(self, "e%d_c2" % lvl, ConvReluINs(co, co, mode=rep))
            setattr(self, "e%d_c3" % lvl, ConvReluINs(co, co, mode=rep))
        self.conv = nn.Conv3d(23 * b, 8 * b, 1, 1, 0, bias=True, padding_mode=rep)

    def forward(self, x):
        x1 = self.e1_c1(x)
        x1 = x1 + self.e1_c3(self.e1_c2(x1))
        xs = [x1]
        for lvl in (2, 3, 4, 5):
            t = getattr(self, "e%d_c1" % lvl)(xs[-1])
            xs.append(t + getattr(self, "e%d_c3" % lvl)(getattr(self, "e%d_c2" % lvl)(t)))
        x6 = self.conv(torch.cat([F.interpolate(t, (8, 8, 8)) for t in xs], 1))       # nearest, mmmvit2.py:97-103
        return xs + [x6]


class Decoder2(nn.Module):
    """Decoder_fuse, mmmvit2.py:116-219 (unused seg_* heads kept for the state-dict)."""

    def __init__(self, num_cls=1):
        super().__init__()
        rep, b = "replicate", BASE
        self.d4_c1 = ConvReluIN(24 * b, 16 * b, 3, 1, rep)
        self.d4_c2 = ConvReluIN(40 * b, 8 * b, 3, 1, rep)
        self.d4_out = ConvReluIN(8 * b, 8 * b, 1, 0, rep)
        self.d3_c1 = ConvReluIN(8 * b, 4 * b, 3, 1, rep)
        self.d3_c2 = ConvReluIN(16 * b, 4 * b, 3, 1, rep)
        self.d3_out = ConvReluIN(4 * b, 4 * b, 1, 0, rep)
        self.d2_c1 = ConvReluIN(4 * b, 2 * b, 3, 1, rep)
        self.d2_c2 = ConvReluIN(8 * b, 2 * b, 3, 1, rep)
        self.d2_out = ConvReluIN(2 * b, 2 * b, 1, 0, rep)
        self.d1_c1 = ConvReluIN(2 * b, b, 3, 1, rep)
        self.d1_c2 = ConvReluIN(4 * b, b, 3, 1, rep)
        self.d1_out = ConvReluIN(b, b, 1, 0, rep)
        self.seg_d4 = _conv(8 * b, num_cls)
        self.seg_d3 = _conv(8 * b, num_cls)
        self.seg_d2 = _conv(4 * b, num_cls)
        self.seg_d1 = _conv(2 * b, num_cls)
        self.seg_layer = _conv(b, num_cls)
        self.RFM5 = RFM(24 * b)
        self.RFM4 = RFM(24 * b)
        self.RFM3 = RFM(12 * b)
        self.RFM2 = RFM(6 * b)
        self.RFM1 = RFM(3 * b)
        self.final_conv = _conv(8, 3)

    @staticmethod
    def _up2(t):
        return F.interpolate(t, scale_factor=2, mode="trilinear", align_corners=True)

    def forward(self, x1, x2, x3, x4, x5):
        y = self.d4_c1(self._up2(self.RFM5(x5)))
        for rfm, skip, size, c2, out, c1 in (
            (self.RFM4, x4, 16, self.d4_c2, self.d4_out, self.d3_c1),
            (self.RFM3, x3, 32, self.d3_c2, self.d3_out, self.d2_c1),
            (self.RFM2, x2, 64, self.d2_c2, self.d2_out, self.d1_c1),
            (self.RFM1, x1, 128, self.d1_c2, self.d1_out, None),
        ):
            s = F.interpolate(rfm(skip), (size, size, size))           # nearest, mmmvit2.py:171,183,193,203
            y = out(c2(torch.cat((s, y), 1)))
            if c1 is not None:
                y = c1(self._up2(y))
        y = F.interpolate(y, size=(1, 224, 224), mode="trilinear", align_corners=True)   # mmmvit2.py:154,210
        return torch.sigmoid(self.final_conv(y))


class MMVit2(nn.Module):
    """MMVit2, mmmvit2.py:345-478."""

    def __init__(self, num_cls=1):
        super().__init__()
        for m in MODS:
            setattr(self, m + "_encoder", Encoder2())
        for m in MODS:
            setattr(self, m + "_encode_conv", _conv(8 * BASE, TOK))
        for m in MODS:                                  # never called, state-dict only (mmmvit2.py:358-360)
            setattr(self, m + "_decode_conv", _conv(TOK, 8 * BASE))
        for m in MODS:
            setattr(self, m + "_pos", nn.Parameter(torch.zeros(1, PATCH ** 3, TOK)))
        for m in MODS:
            setattr(self, m + "_transformer", TransformerBlock())
        for m in MODS:
            setattr(self, "qkv_" + m, _conv(TOK, 3 * TOK))
        self.multimodal_transformer = TransformerBlock()
        self.multimodal_decode_conv = _conv(NMOD * TOK, NMOD * 8 * BASE)
        self.decoder_fuse = Decoder2(num_cls)
        for m in self.modules():                        # mmmvit2.py:390-392
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        B = x.shape[0]
        feats = [getattr(self, m + "_encoder")(x[:, i:i + 1]) for i, m in enumerate(MODS)]

        def tokens(t):          # NCDHW -> [B, D*H*W, C]   (mmmvit2.py:403-405)
            return t.permute(0, 2, 3, 4, 1).reshape(B, -1, TOK)

        qs, ks, vs = [], [], []
        for i, m in enumerate(MODS):
            tok = tokens(getattr(self, m + "_encode_conv")(feats[i][5]))
            tr = getattr(self, m + "_transformer")(tok, getattr(self, m + "_pos"))
            vol = tr.reshape(B, PATCH, PATCH, PATCH, TOK).permute(0, 4, 1, 2, 3).contiguous()   # mmmvit2.py:411-413 (dense NCDHW copy)
            q, k, v = getattr(self, "qkv_" + m)(vol).chunk(3, 1)
            qs.append(q), ks.append(k), vs.append(v)
        skips = [torch.cat([feats[i][l] for i in range(NMOD)], 1) for l in range(4)]        # stack(...).view == channel concat (:420-431)
        corr = [inter_corr(q, ks, vs) for q in qs]                                          # mmmvit2.py:440-453
        mm = torch.cat([tokens(c) for c in corr], 1)                                        # [B, 1536, 512]  (:461-465)
        pos = torch.cat([getattr(self, m + "_pos") for m in MODS], 1)
        y = self.multimodal_transformer(mm, pos)
        # mmmvit2.py:470: 3 consecutive tokens become one voxel's 1536 channels
        vol = y.reshape(B, PATCH, PATCH, PATCH, NMOD * TOK).permute(0, 4, 1, 2, 3).contiguous()
        x6 = self.multimodal_decode_conv(vol)
        return self.decoder_fuse(skips[0], skips[1], skips[2], skips[3], x6)
