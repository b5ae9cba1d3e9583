"""MI355X-native VQ-VAE tokenizer ("SoundStream") with the reference's `models/vq_brain.py` surface: CausalConv1d,
CausalConvTranspose1d, ResidualUnit, EncoderBlock, DecoderBlock, Encoder, Decoder, SoundStream — same constructor arguments,
`SoundStream(x, targets=None, date_info=None) -> (total_loss, reconstruction)`, same state-dict keys for the convolution stack
(`encoder.layers.0.weight` [Cout, Cin, K], `decoder.layers.2.layers.0.weight` [Cin, Cout, 2s], ...).

Reference map: CausalConv1d models/vq_brain.py:22-28, CausalConvTranspose1d :31-45, ResidualUnit :48-64, EncoderBlock :67-91,
DecoderBlock :94-118, Encoder :121-139, Decoder :142-160, SoundStream :163-243 (custom_l1_loss :222-229, calculate_perp :239-243).

Layout: activations stay channels-last [B, T, C] (the reference rearranges to [B, C, T] around its stack; the public
Encoder / Decoder / SoundStream signatures are channels-last in both).  Every convolution is im2col + the MFMA GEMM with bias and
the residual add fused into its epilogue (engine.CausalConv1dFn); the sub-modules therefore take channels-last tensors.

Vector quantisation: the reference delegates to `vector_quantize_pytorch.VectorQuantize`, a third-party package that is neither
vendored nor version-pinned in the reference (**parity unpinned**).  `VectorQuantize` below states its own semantics
(cosine-similarity lookup, straight-through estimator, commitment loss, EMA codebook) in its docstring; the convolution stack,
the loss and the perplexity are pinned against the reference (tests/golden/vq_conv_small.npz)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import engine as E
from .. import kernels as K
from .brainformer import _prep


class ELU(nn.Module):
    def forward(self, x):
        return E.EluFn.apply(_prep(x))


class CausalConv1d(nn.Module):
    """nn.Conv1d parameters ([Cout, Cin, K] weight, bias) with left padding dilation * (K - 1); input / output [B, T, C]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = (kernel_size,), (stride,), (dilation,)
        self.causal_padding = dilation * (kernel_size - 1)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))            # nn.Conv1d.reset_parameters
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, residual=None):
        return E.CausalConv1dFn.apply(_prep(x), self.weight, self.bias, self.stride[0], self.dilation[0], "conv", residual)


class Conv1d(CausalConv1d):
    """kernel-size-1 nn.Conv1d of the residual units (no padding needed)."""


class CausalConvTranspose1d(nn.Module):
    """nn.ConvTranspose1d parameters ([Cin, Cout, K] weight, bias), K = 2 * stride, output trimmed to stride * T (causal)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        assert kernel_size == 2 * stride, "built for kernel_size = 2 * stride (models/vq_brain.py:99-101)"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.output_padding = (kernel_size,), (stride,), (1,), (0,)
        self.causal_padding = (kernel_size - 1) + 1 - stride
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(out_channels * kernel_size)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, output_size=None):
        return E.CausalConv1dFn.apply(_prep(x), self.weight, self.bias, self.stride[0], 1, "convT", None)


class ResidualUnit(nn.Module):
    def __init__(self, in_channels, out_channels, dilation):
        super().__init__()
        self.dilation = dilation
        self.layers = nn.Sequential(
            CausalConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, dilation=dilation),
            ELU(),
            Conv1d(in_channels=out_channels, out_channels=in_channels, kernel_size=1),
        )

    def forward(self, x):
        x = _prep(x)
        h = self.layers[1](self.layers[0](x))
        return self.layers[2](h, residual=x)          # x + conv1x1(...): the add is the GEMM epilogue's residual


class EncoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.layers = nn.Sequential(
            ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=1), ELU(),
            ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=1), ELU(),
            ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=1), ELU(),
            CausalConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=2 * stride, stride=stride),
        )

    def forward(self, x):
        return self.layers(x)


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.layers = nn.Sequential(
            CausalConvTranspose1d(in_channels=in_channels, out_channels=out_channels, kernel_size=2 * stride, stride=stride), ELU(),
            ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=1), ELU(),
            ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=1), ELU(),
            ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=1),
        )

    def forward(self, x):
        return self.layers(x)


class Encoder(nn.Module):
    """[B, T, n_electrodes] -> [B, T / 4, D]"""

    def __init__(self, C, D, n_electrodes):
        super().__init__()
        self.layers = nn.Sequential(
            CausalConv1d(in_channels=n_electrodes, out_channels=C, kernel_size=5), ELU(),
            EncoderBlock(in_channels=C, out_channels=C, stride=2), ELU(),
            EncoderBlock(in_channels=C, out_channels=C, stride=2), ELU(),
            CausalConv1d(in_channels=C, out_channels=D, kernel_size=3),
        )

    def forward(self, x):
        return self.layers(_prep(x))


class Decoder(nn.Module):
    """[B, T / 4, D] -> [B, T, n_channels_out]"""

    def __init__(self, C, D, n_channels_out):
        super().__init__()
        self.layers = nn.Sequential(
            CausalConv1d(in_channels=D, out_channels=C, kernel_size=3), ELU(),
            DecoderBlock(in_channels=C, out_channels=C, stride=2), ELU(),
            DecoderBlock(in_channels=C, out_channels=C, stride=2), ELU(),
            CausalConv1d(in_channels=C, out_channels=n_channels_out, kernel_size=5),
        )

    def forward(self, x):
        return self.layers(_prep(x))


class _Codebook(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        embed = torch.nn.functional.normalize(torch.randn(1, codebook_size, dim), dim=-1)
        self.register_buffer("initted", torch.tensor([True]))
        self.register_buffer("cluster_size", torch.ones(1, codebook_size))
        self.register_buffer("embed_avg", embed.clone())
        self.register_buffer("embed", embed)


class VectorQuantize(nn.Module):
    """Build-defined semantics (the reference's third-party VQ is unpinned), chosen to follow the published cosine-similarity
    VQ-VAE recipe the reference configures (commitment_weight 0.25, cosine similarity, EMA codebook, dead-code threshold 2):
      * lookup: idx = argmax_j <x / |x|, e_j>  with unit-norm codes e_j (GEMM + fk_argmax_rows);  quantized = e_idx
      * output: straight-through, i.e. forward value `quantized`, gradient passed to x unchanged
      * commit_loss = commitment_weight * mean((x - quantized)^2), gradient to x only
      * training: EMA (decay 0.8) of per-code counts and of the summed unit-norm inputs, codes re-normalised to unit length;
        codes whose EMA count falls below threshold_ema_dead_code are re-seeded from random inputs of the batch.
    Buffers use the key names of the usual implementation (`_codebook.embed` [1, K, D], `_codebook.cluster_size`, ...)."""

    def __init__(self, dim, codebook_size, commitment_weight=0.25, decay=0.8, eps=1e-5, threshold_ema_dead_code=2,
                 use_cosine_sim=True, channel_last=True, kmeans_init=False):
        super().__init__()
        assert use_cosine_sim and channel_last
        self.dim, self.codebook_size = dim, codebook_size
        self.commitment_weight, self.decay, self.eps, self.dead = commitment_weight, decay, eps, threshold_ema_dead_code
        self._codebook = _Codebook(dim, codebook_size)

    def forward(self, x):
        B, T, D = x.shape
        x = _prep(x)
        x2 = x.reshape(B * T, D)
        cb = self._codebook
        ones = torch.full((D,), 1.0 / math.sqrt(D), dtype=torch.float32, device=x.device)
        xn, _, _ = K.norm_fwd(x2.detach(), ones, None, 1e-12, K.NORM_RMS)            # x / |x|
        codes = cb.embed[0].to(x.dtype) if cb.embed.dtype != x.dtype else cb.embed[0]
        sims = K.gemm_nt(xn, codes.contiguous(), out_dtype=torch.float32)            # [B*T, K]
        idx = K.argmax_rows(sims)
        q = K.gather_rows(codes.contiguous().unsqueeze(0), idx.view(1, -1)).view(B, T, D)
        if self.training:
            self._ema_update(xn.float(), idx)
        commit = E.mse_loss(x, q.detach()) * self.commitment_weight
        return E.StraightThrough.apply(x, q), idx.view(B, T), commit

    @torch.no_grad()
    def _ema_update(self, xn, idx):
        cb = self._codebook
        Kc = self.codebook_size
        counts = torch.zeros(Kc, dtype=torch.float32, device=xn.device)
        counts.index_add_(0, idx, torch.ones_like(idx, dtype=torch.float32))
        sums = torch.zeros((Kc, self.dim), dtype=torch.float32, device=xn.device)
        K.scatter_add_rows_(sums, idx, xn.contiguous())
        cb.cluster_size[0].mul_(self.decay).add_(counts, alpha=1 - self.decay)
        cb.embed_avg[0].mul_(self.decay).add_(sums, alpha=1 - self.decay)
        n = cb.cluster_size[0].sum()
        smoothed = (cb.cluster_size[0] + self.eps) / (n + Kc * self.eps) * n
        cb.embed[0].copy_(torch.nn.functional.normalize(cb.embed_avg[0] / smoothed[:, None], dim=-1))
        # dead-code re-seeding without a host round trip (no data-dependent shapes: capturable in a hipGraph)
        dead = cb.cluster_size[0] < self.dead
        seed = xn[torch.randint(0, xn.shape[0], (Kc,), device=xn.device)]
        cb.embed[0].copy_(torch.where(dead[:, None], seed, cb.embed[0]))
        cb.embed_avg[0].copy_(torch.where(dead[:, None], seed * self.dead, cb.embed_avg[0]))
        cb.cluster_size[0].masked_fill_(dead, float(self.dead))


class SoundStream(nn.Module):
    def __init__(self, C, D, codebook_size, n_electrodes, use_cosine_sim=True):
        super().__init__()
        self.codebook_size = codebook_size
        self.encoder = Encoder(C=C, D=D, n_electrodes=n_electrodes)
        self.quantizer = VectorQuantize(dim=D, codebook_size=codebook_size, commitment_weight=0.25, channel_last=True,
                                        kmeans_init=True, threshold_ema_dead_code=2, use_cosine_sim=use_cosine_sim)
        self.decoder = Decoder(C=C, D=D, n_channels_out=n_electrodes)

    def forward(self, x, targets=None, date_info=None):
        e = self.encoder(x)
        quantized, indices, commit_loss = self.quantizer(e)
        o = self.decoder(quantized)
        self.last_perplexity = self.calculate_perp(indices)
        rec_loss = self.custom_l1_loss(o, x)
        return rec_loss + commit_loss, o

    def custom_l1_loss(self, pred, gt):
        """mean |pred - gt| over the frames whose input row is not all zeros (padding), models/vq_brain.py:222-229."""
        real = (~torch.all(gt == 0, dim=2)).reshape(-1).to(torch.float32).contiguous()
        return E.L1Loss.apply(_prep(pred), gt, False, real)

    def get_quantize_vectors(self, x):
        e = self.encoder(x)
        quantized, indices, _ = self.quantizer(e)
        return indices, quantized

    def calculate_perp(self, indices):
        flat = indices.reshape(-1)
        counts = torch.zeros(self.codebook_size, dtype=torch.float32, device=flat.device)
        counts.index_add_(0, flat, torch.ones_like(flat, dtype=torch.float32))     # bincount() would sync for its size
        p = counts / counts.sum()
        return (-(p * torch.log(p + 1e-10)).sum()).exp()
