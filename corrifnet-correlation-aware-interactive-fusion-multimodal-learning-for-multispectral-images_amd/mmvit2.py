"""Drop-in `MMVit2` (the reference's sibling model, `mmmvit2.py:345-478`) on the same gfx950 kernels.

    from mmvit2 import MMVit2
    model = MMVit2().to("cuda")
    pred = model(images)               # images [B,3,D,H,W] fp32 -> [B,3,1,224,224] in (0,1), autograd-connected

Same nn.Module surface and state-dict (229 keys, 16,012,600 parameters) as the reference.  It re-uses the decoder, the
transformers and the inter-modal correlation kernel of `mmvit4`; the encoder is the reference's plain 3-D conv pyramid
(replicate padding, stride-2 3x3x3 down-sampling in all three axes, `x + c3(c2(x))` residuals, nearest resampling of the five
levels to 8^3).  Channels-last internally, as in `mmvit4`.  There is no CPU path.
"""
import torch
import torch.nn as nn

import ops
from mmvit4 import (_Edges, _QuietBackwardFn, _rs, _run_lanes, Conv3dP, Decoder_fuse, Transformer, _MODS, basic_dims, depth, general_conv3d_prenorm, mlp_dim, num_heads,
                    num_modals, patch_size, transformer_basic_dims)


class Encoder(nn.Module):
    """mmmvit2.py:57-104"""

    def __init__(self):
        super().__init__()
        b, rep = basic_dims, "replicate"
        self.e1_c1 = Conv3dP(1, b, 3, 1, 1, True, True)                       # Ci = 1: the scalar-gather implicit GEMM, replicate clamp
        self.e1_c2 = general_conv3d_prenorm(b, b, pad_type=rep)
        self.e1_c3 = general_conv3d_prenorm(b, b, pad_type=rep)
        for lvl, (ci, co) in zip((2, 3, 4, 5), ((b, 2 * b), (2 * b, 4 * b), (4 * b, 8 * b), (8 * b, 8 * b))):
            setattr(self, "e%d_c1" % lvl, general_conv3d_prenorm(ci, co, stride=2, pad_type=rep))
            setattr(self, "e%d_c2" % lvl, general_conv3d_prenorm(co, co, pad_type=rep))
            setattr(self, "e%d_c3" % lvl, general_conv3d_prenorm(co, co, pad_type=rep))
        self.conv = Conv3dP(b * 23, b * 8, 1)

    def forward(self, x):
        """x: [B, D, H, W] view of one modality -> (x1..x5 channels-last, x6 [B,8,8,8,64])"""
        t = self.e1_c1(x)
        xs = [ops.add(t, self.e1_c3(self.e1_c2(t)))]
        for lvl in (2, 3, 4, 5):
            t = getattr(self, "e%d_c1" % lvl)(xs[-1])
            xs.append(ops.add(t, getattr(self, "e%d_c3" % lvl)(getattr(self, "e%d_c2" % lvl)(t))))
        B = x.shape[0]
        cube = torch.empty((B, 8, 8, 8, basic_dims * 23), dtype=torch.float32, device=x.device)
        parts, off = [], 0
        for t in xs:                                                          # F.interpolate(x, (8,8,8)): nearest (mmmvit2.py:97-101)
            c = t.shape[-1]
            parts.append(ops.nearest(t, (8, 8, 8), out=cube[..., off:off + c]))
            off += c
        return xs + [self.conv(ops.cat_channels(cube, *parts))]


class MMVit2(nn.Module):
    NOGRAD_PREFIXES = ("RGB_decode_conv.", "NIR_decode_conv.", "SWIR_decode_conv.", "decoder_fuse.seg_d1.", "decoder_fuse.seg_d2.",
                       "decoder_fuse.seg_d3.", "decoder_fuse.seg_d4.", "decoder_fuse.seg_layer.")      # never called (mmmvit2.py)

    def __init__(self, num_cls=1):
        super().__init__()
        d8, T = basic_dims * 8, transformer_basic_dims
        self.RGB_encoder, self.NIR_encoder, self.SWIR_encoder = Encoder(), Encoder(), Encoder()
        self.RGB_encode_conv, self.NIR_encode_conv, self.SWIR_encode_conv = Conv3dP(d8, T, 1), Conv3dP(d8, T, 1), Conv3dP(d8, T, 1)
        self.RGB_decode_conv, self.NIR_decode_conv, self.SWIR_decode_conv = Conv3dP(T, d8, 1), Conv3dP(T, d8, 1), Conv3dP(T, d8, 1)
        for m in _MODS:
            setattr(self, m + "_pos", nn.Parameter(torch.zeros(1, patch_size ** 3, T)))
        self.RGB_transformer = Transformer(T, depth, num_heads, mlp_dim)
        self.NIR_transformer = Transformer(T, depth, num_heads, mlp_dim)
        self.SWIR_transformer = Transformer(T, depth, num_heads, mlp_dim)
        self.qkv_RGB, self.qkv_NIR, self.qkv_SWIR = Conv3dP(T, T * 3, 1), Conv3dP(T, T * 3, 1), Conv3dP(T, T * 3, 1)
        self.multimodal_transformer = Transformer(T, depth, num_heads, mlp_dim, n_levels=num_modals)
        self.multimodal_decode_conv = Conv3dP(T * num_modals, d8 * num_modals, 1)
        self.decoder_fuse = Decoder_fuse(num_cls=num_cls, reduce5=False)
        self.concurrent_branches = True
        self.decoder_split = 2          # sample-group lanes from the multimodal transformer on (see mmvit4._run_lanes)
        self._streams = None
        self._dec_streams = None
        self._edges = _Edges()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("mmvit2.MMVit2 runs on the MI355X kernels only: move the model and the input to a HIP device")
        B, T, P3 = x.shape[0], transformer_basic_dims, patch_size
        x = x.contiguous()
        self._edges.reset()
        feats, qkv = [None] * 3, [None] * 3

        def branch(i, m):
            feats[i] = getattr(self, m + "_encoder")(x[:, i])
            tok = getattr(self, m + "_encode_conv")(feats[i][5]).view(B, P3 ** 3, T)       # channels-last == token layout (:403-405)
            tr = getattr(self, m + "_transformer")(tok, getattr(self, m + "_pos"))
            qkv[i] = getattr(self, "qkv_" + m)(tr.view(B, P3, P3, P3, T)).view(B, P3 ** 3, 3 * T)

        if self.concurrent_branches:           # the three modality branches are independent up to the correlation: three streams
            cur = torch.cuda.current_stream()
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=x.device) for _ in range(num_modals)]
            for i, m in enumerate(_MODS):
                st = self._streams[i]
                self._edges.edge(cur, st)
                _rs(x, st)
                with torch.cuda.stream(st):
                    branch(i, m)
            for st in self._streams:
                self._edges.edge(st, cur)
            for i in range(num_modals):
                for t in feats[i] + [qkv[i]]:
                    _rs(t, cur)
        else:
            for i, m in enumerate(_MODS):
                branch(i, m)
        # stack((RGB, NIR, SWIR), 1).view(B, -1, ...) == channel concatenation (mmmvit2.py:420-431)
        skips = [ops.cat_channels_copy(*[feats[i][l] for i in range(num_modals)]) for l in range(4)]
        corr = ops.inter_corr(qkv[0], qkv[1], qkv[2])                                      # mmmvit2.py:440-453
        pos = ops.cat_tokens(self.RGB_pos, self.NIR_pos, self.SWIR_pos)

        def tail(tok, s1, s2, s3, s4, lane=0):
            nb = tok.shape[0]
            y = self.multimodal_transformer(tok, pos)                                      # [nb, 1536, 512]
            x6 = self.multimodal_decode_conv(y.view(nb, P3, P3, P3, num_modals * T))       # 3 tokens -> one voxel (mmmvit2.py:470)
            return self.decoder_fuse(s1, s2, s3, s4, x6, lane=lane)

        pred = _run_lanes(self, tail, pos, ops.cat_tokens(*corr), skips[0], skips[1], skips[2], skips[3])
        if pred.requires_grad and int(self.decoder_split or 0) >= 2:
            pred = _QuietBackwardFn.apply(pred)        # scoped suppression of the intended stream-mismatch warning (mmvit4._QuietBackwardFn)
        return pred
