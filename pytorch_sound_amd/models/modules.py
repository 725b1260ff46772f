"""Transformer blocks on the (N, C, T) layout of pytorch_sound/models/modules.py.

Semantics pinned by tests/golden/modules.npz (generated from the imported reference):
  * the fused 1x1 projection yields chunks in the order K, V, Q (modules.py:34)
  * heads are folded into the batch dimension head-major: (H*N, C/H, T) (modules.py:38,48)
  * scores[b, t_key, t_query] = k^T q / sqrt(d); softmax runs over KEYS (dim 1) (modules.py:66-74)
  * ``mask`` is (N, T) bool, True = padding: padded key rows get -inf before the softmax, padded
    query columns are zeroed after it (modules.py:69-76); the attention tensor is returned
  * the "layer norm" is nn.GroupNorm(1, C): statistics over (C x T) jointly per sample, per-channel
    affine; it is applied to (x + input) (modules.py:30,58,98,114)
"""
import math
from typing import Optional, Tuple

import torch
from pytorch_sound_amd import _switches as _sw
import torch.nn as nn
import torch.nn.functional as F


def _hip_ok(x: torch.Tensor) -> bool:
    """A tensor that lives on the GPU ALWAYS takes the hand-written kernels (psnd_linear1x1_*, psnd_mha_*, psnd_groupnorm1_*):
    the kernels compute in fp32 from fp32 memory, so another floating dtype is cast on the way in and the result cast back
    (`_to_kernel_dtype`) - there is no library path for a HIP tensor and no switch in this module that opens one (a parity test that
    wants the torch formulation evaluated on the GPU as its fp32 yardstick patches this function).  CPU tensors use the torch
    formulation that the golden tests pin (tests/test_modules_golden.py)."""
    return x.is_cuda


def _to_kernel_dtype(x: torch.Tensor) -> torch.Tensor:
    if _hip_ok(x) and x.dtype != torch.float32:
        if not x.is_floating_point():
            raise TypeError('expected a floating-point activation tensor, got %s' % x.dtype)
        return x.float()
    return x


def _add_norm(norm: nn.GroupNorm, x: torch.Tensor, residual: torch.Tensor, relu: bool = False, link=None) -> torch.Tensor:
    """GroupNorm(1, C)(x + residual) [-> ReLU]; `link`: the kernels.ResidualLink the branch's first projection was given - the residual's
    gradient then reaches the input through that projection's input-gradient GEMM instead of a separate accumulation pass"""
    if _hip_ok(x):
        from pytorch_sound_amd import kernels as K
        if link is not None and residual.dtype != torch.float32:
            link = None                                      # (the gradient would be of the converted copy, not of `residual`)
        return K.GroupNorm1.apply(x.float(), residual.float(), norm.weight.float(), norm.bias.float(), norm.eps, relu, link)
    y = norm(x + residual)
    return F.relu(y) if relu else y


def _conv1x1(conv: nn.Conv1d, x: torch.Tensor, relu: bool = False, link=None, relu_link=None, hidden_out: bool = False, t_len=None,
             pad_rows: bool = False) -> torch.Tensor:
    """the 1x1 Conv1d projections of modules.py:21-22, 93-95 as what they are - one GEMM over (C_in, N*T), on the exact-fp32
    matrix-core kernel (psnd_linear1x1_*, bias and the following ReLU fused).  CPU tensors (and the tests' fp32 yardstick) keep
    the torch formulation."""
    if not _hip_ok(x):
        y = conv(x)
        return F.relu(y) if relu else y
    from pytorch_sound_amd import kernels as K
    # under torch.autocast(bfloat16) the products take bf16 operands (fp32 accumulation, fp32 activations in memory): 16x the
    # matrix rate; without autocast they are exact fp32
    bf16 = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
    if link is not None and x.dtype != torch.float32:
        link = None
    # hidden_out (the first projection of Conv1d -> ReLU -> Conv1d, modules.py:93-95) under autocast: the tensor between the two is STORED as
    # bf16 - what torch.autocast's own conv output is; the second projection takes it as it comes
    out_h = bool(hidden_out and bf16 and K.HIDDEN_BF16 and (not relu or (relu_link is not None and K.RELU_LINKS) or not torch.is_grad_enabled()))
    xin = x if (bf16 and x.dtype == torch.bfloat16) else x.float()
    # pad_rows: that bf16 tensor with rows of the next multiple of 64 frames (each starts on a 128-byte line); the call that takes it names the
    # frames in use (t_len)
    return K.Linear1x1.apply(xin, conv.weight.float(), None if conv.bias is None else conv.bias.float(), relu, bf16, link, relu_link, out_h, t_len,
                             bool(pad_rows and K.PAD_HIDDEN_ROWS))


class MultiHeadAttention(nn.Module):

    def __init__(self, hidden_dim: int, heads: int, dropout_rate: float):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.heads = heads
        self.linear_kvq = nn.Conv1d(hidden_dim, hidden_dim * 3, 1, bias=False)
        self.linear = nn.Conv1d(hidden_dim, hidden_dim, 1, bias=False)
        self.drop_out = nn.Dropout(dropout_rate) if 0 < dropout_rate < 1 else None
        self.layernorm = nn.GroupNorm(1, hidden_dim)

    def _fold_heads(self, x: torch.Tensor) -> torch.Tensor:
        n, c, t = x.shape
        # (N, H, d, T) -> (H, N, d, T) -> (H*N, d, T): head-major on the batch axis
        return x.view(n, self.heads, c // self.heads, t).transpose(0, 1).reshape(self.heads * n, c // self.heads, t)

    def _unfold_heads(self, x: torch.Tensor) -> torch.Tensor:
        hn, d, t = x.shape
        n = hn // self.heads
        return x.view(self.heads, n, d, t).transpose(0, 1).reshape(n, self.heads * d, t)

    # (not in the reference) False: forward returns (x, None) and the (H*N, T, T) attention tensor is never written to memory -
    # 855 MB per layer at 32 clips x 1292 frames; the reference always materialises and returns it
    return_att = True
    # (not in the reference) under torch.autocast(bfloat16) with return_att = False: kvq and its gradient as bf16 tensors between the
    # projection and the attention kernels; False keeps them fp32
    kvq_bf16 = _sw.lab('PSND_MHA_KVQ_BF16', '1') == '1'

    def forward(self, input: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        in_dtype = input.dtype
        input = _to_kernel_dtype(input)
        link = None
        if _hip_ok(input) and input.dtype == torch.float32 and input.requires_grad and torch.is_grad_enabled():
            from pytorch_sound_amd import kernels as K
            link = K.ResidualLink()                     # input feeds the projection AND the residual: one gradient pass for both
        # under autocast, and when the attention tensor is not returned, the projection STORES kvq as bf16 (the attention kernels round their
        # operands to bf16 when they load them: same products, half the bytes, no conversion in their loops); its gradient comes back as bf16
        kvq = _conv1x1(self.linear_kvq, input, link=link, hidden_out=_hip_ok(input) and not self.return_att and self.kvq_bf16)
        if _hip_ok(input):
            # gfx950: projection -> attention over all heads in one kernel pair (psnd_mha_*), scores stay on the chip
            if self.hidden_dim % self.heads != 0 or self.hidden_dim // self.heads > 128:
                from pytorch_sound_amd._lib import PsndError
                raise PsndError('MultiHeadAttention(hidden_dim=%d, heads=%d): psnd_mha_* covers head dimensions up to 128 that divide '
                                'hidden_dim; there is no library path for a HIP tensor' % (self.hidden_dim, self.heads))
            from pytorch_sound_amd import kernels as K
            # (a bool mask already is one byte of 0 / 1 per frame: a view, no conversion launch)
            mask_u8 = None if mask is None else (mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous())
            # under torch.autocast(bfloat16) the score / accumulate products take bf16 operands, as the projections do
            bf16 = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            x, att = K.AttentionKVQ.apply(kvq, mask_u8, self.heads, self.return_att, bf16)
            att = att if self.return_att else None
        else:
            k, v, q = (self._fold_heads(p) for p in kvq.chunk(3, 1))
            if mask is not None:
                mask = mask.repeat(self.heads, 1)
            x, att = self.scale_dot_att(k, v, q, att_mask=mask)
            x = self._unfold_heads(x)
        x = _conv1x1(self.linear, x)
        if self.drop_out is not None:
            x = self.drop_out(x)
        x = _add_norm(self.layernorm, x, input, link=link)
        if in_dtype != x.dtype:
            x, att = x.to(in_dtype), (None if att is None else att.to(in_dtype))
        return x, att

    @staticmethod
    def scale_dot_att(k: torch.Tensor, v: torch.Tensor, q: torch.Tensor,
                      att_mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """k, v, q: (B, d, T); att_mask: (B, T) bool or None -> (B, d, T), (B, T_key, T_query)."""
        if _hip_ok(q):
            # a public static method in the reference (modules.py:61-79): called directly on HIP tensors it runs on psnd_mha_* as well -
            # one "head" per batch entry (the heads are folded into the batch here already), scores on the chip, no library bmm
            from pytorch_sound_amd import kernels as K
            from pytorch_sound_amd._lib import PsndError
            if not (k.shape == v.shape == q.shape) or k.dim() != 3 or k.size(1) > 128:
                raise PsndError('scale_dot_att on HIP tensors: (B, d, T) operands of one shape with d <= 128 (psnd_mha_*), got %s / %s / %s; '
                                'there is no library path for a HIP tensor' % (tuple(k.shape), tuple(v.shape), tuple(q.shape)))
            dt = q.dtype
            kvq = torch.cat([k.float(), v.float(), q.float()], dim=1)
            mask_u8 = None if att_mask is None else att_mask.to(torch.uint8).contiguous()
            bf16 = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            x, att = K.AttentionKVQ.apply(kvq, mask_u8, 1, True, bf16)
            return (x, att) if dt == torch.float32 else (x.to(dt), att.to(dt))
        scores = torch.bmm(k.transpose(1, 2), q) / math.sqrt(k.size(1))
        if att_mask is not None:
            scores = scores.masked_fill(att_mask.unsqueeze(2), -float('inf'))     # padded keys
        att = F.softmax(scores, 1)
        if att_mask is not None:
            att = att.masked_fill(att_mask.unsqueeze(1), 0.)                      # padded queries
        return torch.bmm(v, att), att


class PointwiseFeedForward(nn.Module):
    """Conv1d(C,4C,1) -> ReLU -> Conv1d(4C,C,1) -> [dropout] -> GroupNorm(1,C)(x + input) -> ReLU
    (modules.py:82-116)."""

    def __init__(self, hidden_dim: int, dropout_rate: float):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.ff = nn.Sequential(
            nn.Conv1d(hidden_dim, hidden_dim * 4, 1),
            nn.ReLU(),
            nn.Conv1d(hidden_dim * 4, hidden_dim, 1),
        )
        self.layernorm = nn.GroupNorm(1, hidden_dim)
        self.act = nn.ReLU()
        self.drop_out = nn.Dropout(dropout_rate) if 0 < dropout_rate < 1 else None

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        in_dtype = input.dtype
        input = _to_kernel_dtype(input)
        link = relu_link = None
        if _hip_ok(input) and input.dtype == torch.float32 and torch.is_grad_enabled():
            from pytorch_sound_amd import kernels as K
            relu_link = K.ReluLink()          # the ReLU's backward in the epilogue of the GEMM that produces its gradient
            if input.requires_grad:
                link = K.ResidualLink()
        x = _conv1x1(self.ff[2], _conv1x1(self.ff[0], input, relu=True, link=link, relu_link=relu_link, hidden_out=True, pad_rows=True),
                     relu_link=relu_link, t_len=input.size(-1))
        if self.drop_out is not None:
            x = self.drop_out(x)
        x = _add_norm(self.layernorm, x, input, relu=True, link=link)
        return x if x.dtype == in_dtype else x.to(in_dtype)


class PositionalEncoding(nn.Module):
    """x * sqrt(C) + pe[..., :T], pe (1, C, max_len): angle = pos / 10000^(2*(c//2)/C), sin on even
    channels, cos on odd ones (modules.py:119-145)."""

    def __init__(self, dim: int, max_seq_len: int):
        super().__init__()
        self.dim = dim
        self.register_buffer('pe', self.get_embedding(max_seq_len, dim).T.unsqueeze(0))

    @staticmethod
    def get_embedding(num_embeddings: int, embedding_dim: int) -> torch.Tensor:
        chan = torch.arange(embedding_dim, dtype=torch.float32)
        denom = (10000 ** (2 * (chan // 2) / embedding_dim)).unsqueeze(0)                 # (1, C)
        pos = torch.arange(num_embeddings, dtype=torch.float32).unsqueeze(1).repeat(1, embedding_dim)
        table = pos / denom
        table[:, 0::2] = torch.sin(table[:, 0::2])
        table[:, 1::2] = torch.cos(table[:, 1::2])
        return table

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _hip_ok(x) and x.dim() == 3 and x.dtype == torch.float32 and x.size(-1) <= self.pe.size(-1) and self.pe.device == x.device:
            from pytorch_sound_amd import kernels as K
            # the buffer is the reference's transposed view (1, C, max_len) of a (max_len, C) table: the kernel reads (C, max_len) rows
            key = (self.pe.data_ptr(), self.pe._version, self.pe.device)
            if getattr(self, '_pe_rows_key', None) != key:
                self._pe_rows, self._pe_rows_key = self.pe[0].contiguous(), key
            return K.PosEnc.apply(x, self._pe_rows, self.dim ** 0.5)    # one pass instead of a scalar multiply + a broadcast add
        return x * (self.dim ** 0.5) + self.pe[..., :x.size(-1)]
