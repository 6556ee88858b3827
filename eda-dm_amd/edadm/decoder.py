"""First-stage decoder on the HIP kernels: the graph of ldm/modules/diffusionmodules/model.py:465-572
(+ the 1x1 `post_quant_conv` of VQModelInterface.decode, ldm/models/autoencoder.py:274-282) executed in NHWC fp32:

    3x3 / 1x1 convolutions .... implicit GEMM; bias, residual and the nearest-2x upsample of `Upsample` folded into
                                 the same launch.  Large layers: edadm_split_f16 + edadm_qgemm_f16 (fp32 operands as
                                 two-term f16 expansions, three products on the f16 MFMA, fp32-grade result); the
                                 rest: edadm_conv2d_f32_nhwc (exact-fp32 MFMA)
    GroupNorm (+ swish) ....... edadm_gn_split_f16: normalise + swish + the f16 expansion the next convolution reads, in two
                                 passes over the input (the normalised tensor is never stored); edadm_groupnorm_stats /
                                 _apply (K5) in front of the exact-fp32 layers
    attention block ........... edadm_gemm_f32_nt (q k^T, p v) + edadm_softmax_f32
    layout .................... NCHW <-> NHWC once at the boundary

The vector-quantisation lookup of VQModelInterface.decode (taming's VectorQuantizer2, not vendored by the
reference: parity unpinned) is `nearest_code`, a plain argmin over the codebook."""
import torch

from . import ops
from .contract import F16X3


def _w(conv):
    """Conv2d weight [N][C][KH][KW] -> [N][KH][KW][C4] (input channels zero-padded to a multiple of 4)."""
    w = conv.weight.detach().float().permute(0, 2, 3, 1)
    pad = (-w.shape[-1]) % 4
    if pad:
        w = torch.nn.functional.pad(w, (0, pad))
    return w.contiguous()


class DecoderEngine:
    def __init__(self, decoder, post_quant_conv=None, codebook=None, chunk_pixels=25 << 16):
        self.dec, self.pq = decoder, post_quant_conv
        for mod in (decoder, post_quant_conv):
            prm = next(mod.parameters(), None) if isinstance(mod, torch.nn.Module) else None
            if prm is not None:
                ops.init_device(prm.device)
                break
        self.codebook = None if codebook is None else codebook.detach().float().contiguous()
        # output pixels per batch chunk: keeps every activation under the gather's 2 GiB (32-bit byte offsets) and bounds the attention
        # scores.  25 images of 256 x 256 (the largest tensor, 256^2 x 256 channels fp32, is 1.68 GB): a 50-image batch decodes as
        # 25 + 25 (2.148 -> 2.113 ms per image against 13 + 13 + 12 + 12 under the 16-image cap of round 3, tools/decoder_time.py)
        self.chunk_pixels = chunk_pixels
        self._wc = {}
        self.fuse_gn_split = True                   # norm -> swish -> conv: the conv's f16 operand straight from the GroupNorm pass

    def w(self, conv):
        if id(conv) not in self._wc:
            self._wc[id(conv)] = (_w(conv), None if conv.bias is None else conv.bias.detach().float().contiguous())
        return self._wc[id(conv)]

    def conv(self, conv, x, residual=None, ups=False):
        w, b = self.w(conv)
        if x.shape[-1] != w.shape[-1]:                          # latent channels (3) -> 4
            x = torch.nn.functional.pad(x, (0, w.shape[-1] - x.shape[-1])).contiguous()
        if self._f16x3(conv, x, ups):
            return ops.conv2d_f16x3_nhwc(x, w, b, residual=residual, stride=1, pad=conv.padding[0], ups=ups,
                                         presplit=self._presplit(conv))
        if (w.shape[0] <= 4 and w.shape[1] == 3 and w.shape[2] == 3 and conv.padding[0] == 1 and x.shape[-1] <= 320 and residual is None
                and not ups and x.shape[0] * x.shape[1] < (1 << 31)):
            # conv_out: 3 output channels -- a GEMM tile would compute 64 columns for them (edadm_conv3x3_f32_smalln: fp32 FMAs, one
            # wave per image row, the same kernel as the UNet's last layer)
            return ops.conv3x3_f32_smalln(x, w, b)
        return ops.conv2d_f32_nhwc(x, w, b, residual=residual, stride=1, pad=conv.padding[0], ups=ups)

    def _presplit(self, conv):
        """(two-term f16 expansion of the filter, its per-row inverse scales, the same packed for the direct 3x3 kernel or None), once"""
        key = ("h", id(conv))
        if key not in self._wc:
            w, _ = self.w(conv)
            wb, inv_b = ops.split_f16(w, w.shape[0], w.shape[1] * w.shape[2], w.shape[3], 2, True)[:2]
            wdc = None
            if w.shape[1] == 3 and w.shape[2] == 3 and w.shape[3] % 16 == 0 and ops.lib.load().edadm_conv3_packed_rows(int(w.shape[0])):
                wdc = ops.conv3_f16x3_pack_w(wb, w.shape[0], w.shape[3])
            self._wc[key] = (wb, inv_b, wdc)
        return self._wc[key]

    def _f16x3(self, conv, x, ups=False):
        w, _ = self.w(conv)
        return (F16X3 and x.shape[-1] == w.shape[-1] and ops.f16x3_conv_ok(x, w)
                and x.shape[0] * x.shape[1] * x.shape[2] * (4 if ups else 1) >= 16384)

    def gn_conv(self, norm, x, silu, conv, residual=None):
        """conv(swish(norm(x))) (+ residual).  When the convolution runs on the three-product path its operand is the f16
        expansion of the normalised tensor: edadm_gn_split_f16 writes that expansion from x directly (two passes over x instead
        of four over the normalised tensor, which is never stored)."""
        if not (self.fuse_gn_split and self._f16x3(conv, x) and ops.gn_split_ok(x, norm.num_groups)):
            return self.conv(conv, self.gn(norm, x, silu), residual=residual)
        w, b = self.w(conv)
        wb, inv_b, wdc = self._presplit(conv)
        xa, comb = ops.gn_split_f16(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu, inv_b, w.shape[0])
        return ops.conv2d_f16x3_pre(xa, comb, x.shape, w.shape, wb, b, residual=residual, stride=1, pad=conv.padding[0], wdc=wdc)

    def gn(self, norm, x, silu):
        st = ops.groupnorm_stats(x, norm.num_groups, norm.eps)
        y, _ = ops.groupnorm_apply(x, st, norm.weight, norm.bias, norm.num_groups, silu, want_f32=True)
        return y

    def res(self, blk, x):
        h = self.gn_conv(blk.norm1, x, True, blk.conv1)
        if blk.in_channels != blk.out_channels:
            x = self.conv(blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut, x)
        return self.gn_conv(blk.norm2, h, True, blk.conv2, residual=x)

    def attn(self, blk, x):
        B, H, W, C = x.shape
        N = H * W
        h = self.gn(blk.norm, x, False)
        q, k, v = (self.conv(m, h).reshape(B, N, C) for m in (blk.q, blk.k, blk.v))
        vt = ops.nhwc_to_nchw(v.reshape(B, N, 1, C)).reshape(B, C, N)                  # [B][C][N]: B operand of p . v
        if F16X3 and C % 16 == 0 and N % 16 == 0 and N >= 1024:
            # the two products of a 4096-token, 512-channel attention are 17 GFLOP each per image: on the three-product f16 path
            # (fp32-grade, section 4 of DESIGN.md) like the convolutions, per image through edadm_qgemm_f16x3 on pre-expanded
            # operands; the probabilities lie in [0, 1], so their expansion needs no maximum scan
            ka, inv_k, _ = ops.split_f16(k.reshape(B * N, C), B * N, 1, C, 2, False)
            qa, _, comb = ops.split_f16(q.reshape(B * N, C), B * N, 1, C, 2, False, other=inv_k, N=N)
            comb = comb * (int(C) ** (-0.5))
            s = torch.empty(B, N, N, dtype=torch.float32, device=x.device)
            for i in range(B):
                ops.qgemm_f16x3_pre(qa[i * N:(i + 1) * N], 2 * C, ka[i * N:(i + 1) * N], 2 * C, N, N, 2 * C, comb, s[i])
            p = ops.softmax_f32(s.reshape(B * N, N))
            if ("one", x.device) not in self._wc:
                self._wc[("one", x.device)] = torch.ones(1024, dtype=torch.float32, device=x.device)
            vta, inv_v, _ = ops.split_f16(vt.reshape(B * C, N), B * C, 1, N, 2, False)
            pa, _, comb2 = ops.split_f16(p, B * N, 1, N, 2, False, other=inv_v, N=C, amax=self._wc[("one", x.device)])
            # one batched launch over the images (16 x 32 tiles of 256 x 256 instead of 32 per launch); the common power-of-two factor
            # of the two expansions comes off afterwards
            o = ops.gemm_f16x3_nt(pa, 2 * N, N * 2 * N, vta, 2 * N, C * 2 * N, B, N, C, 2 * N).mul_(comb2)
        else:
            s = ops.gemm_f32_nt(q, k, N, N, C, alpha=int(C) ** (-0.5), batch=B, strideA=N * C, strideB=N * C, strideC=N * N)
            p = ops.softmax_f32(s.reshape(B * N, N)).reshape(B, N, N)
            o = ops.gemm_f32_nt(p, vt, N, C, N, batch=B, strideA=N * N, strideB=C * N, strideC=N * C)
        return self.conv(blk.proj_out, o.reshape(B, H, W, C), residual=x)

    def nearest_code(self, z_nhwc):
        """z -> codebook[argmin ||z - e||^2] (the VectorQuantizer2 lookup); identity when no codebook is attached."""
        if self.codebook is None:
            return z_nhwc
        flat = z_nhwc.reshape(-1, z_nhwc.shape[-1]).contiguous()
        if flat.shape[1] <= 8:
            # edadm_vq_nearest: the codebook staged through LDS, no [pixels][codes] distance matrix (2 GB for 16 VQ-f4 latents)
            return ops.vq_nearest(flat, self.codebook).reshape(z_nhwc.shape)
        d = (flat * flat).sum(1, keepdim=True) - 2 * flat @ self.codebook.t() + (self.codebook * self.codebook).sum(1)[None]
        return self.codebook[d.argmin(1)].reshape(z_nhwc.shape)

    @torch.no_grad()
    def decode_nhwc(self, z):
        d = self.dec
        h = self.nearest_code(z)
        if self.pq is not None:
            h = self.conv(self.pq, h)
        h = self.conv(d.conv_in, h)
        h = self.res(d.mid.block_2, self.attn(d.mid.attn_1, self.res(d.mid.block_1, h)))
        for i_level in reversed(range(d.num_resolutions)):
            up = d.up[i_level]
            for i_block in range(d.num_res_blocks + 1):
                h = self.res(up.block[i_block], h)
                if len(up.attn) > 0:
                    h = self.attn(up.attn[i_block], h)
            if i_level != 0:
                h = self.conv(up.upsample.conv, h, ups=True) if up.upsample.with_conv else ops.upsample2_nhwc(h)
        if d.give_pre_end:
            return h
        h = self.gn_conv(d.norm_out, h, True, d.conv_out)
        return torch.tanh(h) if d.tanh_out else h

    @torch.no_grad()
    def __call__(self, z_nchw):
        """z [B][zc][h][w] -> image [B][out_ch][H][W], decoded in batch chunks."""
        B = z_nchw.shape[0]
        up = 2 ** (self.dec.num_resolutions - 1)
        per = z_nchw.shape[2] * z_nchw.shape[3] * up * up
        nb = max(1, min(B, self.chunk_pixels // per))
        # balanced chunks: 50 images under a 16-image cap decode as 13 + 13 + 12 + 12, not 16 + 16 + 16 + 2 (a 2-image chunk leaves the
        # 64x64 level with 32 tiles for 256 CUs)
        n_chunks = (B + nb - 1) // nb
        base, extra = divmod(B, n_chunks)
        outs, b0 = [], 0
        for i in range(n_chunks):
            n = base + (1 if i < extra else 0)
            zc = ops.nchw_to_nhwc(z_nchw[b0:b0 + n].contiguous().float())
            outs.append(ops.nhwc_to_nchw(self.decode_nhwc(zc)))
            b0 += n
        return torch.cat(outs) if len(outs) > 1 else outs[0]
