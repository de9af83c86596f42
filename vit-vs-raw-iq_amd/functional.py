"""Stand-alone native ops behind the reference's per-layer modules.

The whole-model plan (csrc/model.hip) is what training runs; these wrappers let the individual layers of the
reference's module tree run on their own (forward hooks, per-layer feature extraction, unit tests), each through
the per-op C ABI of include/iqvit.h and with an autograd backward through the matching native gradient kernel:

  layer_norm     LayerNorm.forward                     V/models/layers/layers_norm.py:11-19        iq_ln_fwd / iq_ln_bwd
  linear         nn.Linear (w_q/w_k/w_v/w_concat,      V/models/layers/multi_head_attention.py:18,28
                 linear1/linear2)                      V/models/layers/position_wise_feed_forward.py:13-16
                                                                              iq_gemm_bf16_nt / iq_gemm_bf16_wgrad
  attention      ScaleDotProductAttention.forward      V/models/layers/scale_dot_product_attention.py:18-39
                 (+ split / concat, multi_head_attention.py:34-47)            iq_attn_fwd / iq_attn_bwd
  patch_embed    PatchEmbedding / SequenceEmbedding    V/.../patch_embedding.py:11-15, R/.../patch_embedding.py:47-60
                                                                              iq_patchify + iq_gemm_bf16_nt

Mixed precision is the build's stated policy: fp32 tensors in and out (like the reference), bf16 operands and fp32
accumulation inside.  GPU tensors only -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _native as N


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise N.IqError("this framework runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                            "(the CPU restatement used for parity checks lives in oracle/ and is test infrastructure)")


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.bfloat16).contiguous()


def _f32(t):
    return None if t is None else t.detach().float().contiguous()


# ------------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_cuda(x, gamma, beta)
        L = N.lib()
        D = x.shape[-1]
        if not L.iq_ln_supported(D):
            raise N.IqError(f"LayerNorm width {D} is not supported by the native kernel (needs D % 8 == 0, D <= 2048)")
        z = _bf(x).view(-1, D)
        M = z.shape[0]
        g, b = _f32(gamma), _f32(beta)
        out = torch.empty_like(z)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        N.check(L.iq_ln_fwd(z.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                            M, D, float(eps), N.stream_handle()), "iq_ln_fwd")
        ctx.save_for_backward(z, mean, rstd, g)
        ctx.shape = x.shape
        return out.view(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        z, mean, rstd, g = ctx.saved_tensors
        L = N.lib()
        M, D = z.shape
        dx = _bf(gout).view(M, D)
        dz = torch.empty_like(z)
        dg = torch.empty(D, dtype=torch.float32, device=z.device)
        db = torch.empty_like(dg)
        ws = torch.empty(L.iq_ln_bwd_ws_bytes(D), dtype=torch.uint8, device=z.device)
        N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dz.data_ptr(),
                            None, None, dg.data_ptr(), db.data_ptr(), ws.data_ptr(), 0, M, D, N.stream_handle()),
                "iq_ln_bwd")
        return dz.view(ctx.shape).to(gout.dtype), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-12):
    return _LayerNormFn.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    """y = [relu](x W^T + b); bf16 operands, fp32 accumulate.  x: (..., K) fp32, W: (N, K), b: (N) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        _need_cuda(x, weight, bias)
        L = N.lib()
        Nn, K = weight.shape
        if x.shape[-1] != K:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} x {K}x{Nn})")
        if K % 8 or Nn % 8:
            raise N.IqError(f"native GEMM needs in/out features that are multiples of 8, got {K} -> {Nn}")
        a = _bf(x).view(-1, K)
        M = a.shape[0]
        w = _bf(weight)
        b = _f32(bias)
        if b is not None and b.data_ptr() % 16:
            b = b.clone()
        y = torch.empty(M, Nn, dtype=torch.bfloat16, device=x.device)
        e = N.Epilogue()
        e.bias = N.ptr(b)
        e.relu = 1 if relu else 0
        N.check(L.iq_gemm_bf16_nt(a.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), Nn, M, Nn, K, C.byref(e),
                                  N.stream_handle()), "iq_gemm_bf16_nt")
        ctx.save_for_backward(a, weight, y if relu else None)
        ctx.relu, ctx.has_bias, ctx.xshape = bool(relu), bias is not None, x.shape
        return y.view(*x.shape[:-1], Nn).to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        a, weight, y = ctx.saved_tensors
        L = N.lib()
        M, K = a.shape
        Nn = weight.shape[0]
        st = N.stream_handle()
        dy = _bf(gout).view(M, Nn)
        if ctx.relu:                     # gate by the saved activation (y > 0), exactly what the plan's backward does
            dy = torch.where(y > 0, dy, torch.zeros_like(dy))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _bf(weight.t())         # [K, N]: the transposed shadow the plan keeps per layer
            dxb = torch.empty(M, K, dtype=torch.bfloat16, device=a.device)
            N.check(L.iq_gemm_bf16_nt(dy.data_ptr(), Nn, wt.data_ptr(), Nn, dxb.data_ptr(), K, M, K, Nn, None, st),
                    "iq_gemm_bf16_nt (dgrad)")
            dx = dxb.view(ctx.xshape).to(gout.dtype)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty(Nn, K, dtype=torch.float32, device=a.device)
            db = torch.empty(Nn, dtype=torch.float32, device=a.device) if ctx.has_bias else None
            nbytes = L.iq_wgrad_ws_bytes(M, Nn, K)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=a.device)
            N.check(L.iq_gemm_bf16_wgrad(dy.data_ptr(), Nn, a.data_ptr(), K, dw.data_ptr(), N.ptr(db), M, Nn, K,
                                         ws.data_ptr(), ws.numel(), 0, st), "iq_gemm_bf16_wgrad")
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    return _LinearFn.apply(x, weight, bias, relu)


# ------------------------------------------------------------------------------------------------
class _AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v per (batch, head) on head-split tensors (B, H, S, dh)."""

    @staticmethod
    def forward(ctx, q, k, v, mask):
        _need_cuda(q, k, v, mask)
        L = N.lib()
        B, H, S, dh = q.shape
        if k.shape != q.shape or v.shape != q.shape:
            raise N.IqError("native attention is self-attention shaped: q, k, v must have equal (B, H, S, dh) shapes")
        if dh not in (16, 32, 64) or not L.iq_attn_supported(S, dh):
            raise N.IqError(f"head dim {dh} / sequence {S} not supported by the native attention kernels "
                            "(head dim must be 16, 32 or 64)")
        D = H * dh
        # the kernels read the packed projection layout [B*S, 3D] (q | k | v, head h at columns h*dh)
        qkv = torch.empty(B * S, 3 * D, dtype=torch.bfloat16, device=q.device)
        for i, t in enumerate((q, k, v)):
            qkv[:, i * D:(i + 1) * D] = t.detach().permute(0, 2, 1, 3).reshape(B * S, D)
        out = torch.empty(B * S, D, dtype=torch.bfloat16, device=q.device)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=q.device)
        mk = _mask_bytes(mask, B, H, S, q.device)
        N.check(L.iq_attn_fwd_masked(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), N.ptr(mk), _mask_hstride(mk, H, S),
                                     B, S, H, dh, N.stream_handle()), "iq_attn_fwd")
        ctx.save_for_backward(qkv, out, lse, mk)
        ctx.dims = (B, H, S, dh)
        return out.view(B, S, H, dh).permute(0, 2, 1, 3).to(q.dtype)

    @staticmethod
    def backward(ctx, gout):
        qkv, out, lse, mk = ctx.saved_tensors
        B, H, S, dh = ctx.dims
        D = H * dh
        L = N.lib()
        do = _bf(gout.permute(0, 2, 1, 3)).view(B * S, D)
        dqkv = torch.empty_like(qkv)
        N.check(L.iq_attn_bwd_masked(qkv.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                     N.ptr(mk), _mask_hstride(mk, H, S), B, S, H, dh, N.stream_handle()), "iq_attn_bwd")
        g = [dqkv[:, i * D:(i + 1) * D].view(B, S, H, dh).permute(0, 2, 1, 3).to(gout.dtype) for i in range(3)]
        return g[0], g[1], g[2], None


def _mask_bytes(mask, B, H, S, device):
    """scale_dot_product_attention.py:30-31: positions with mask == 0 get a score of -10000.  Any mask broadcastable to
    (B, H, S, S) is accepted and materialised as uint8 (B, 1 or H, S, S)."""
    if mask is None:
        return None
    m = (mask != 0)
    while m.dim() < 4:
        m = m.unsqueeze(0)
    hh = H if m.shape[1] != 1 else 1
    return m.expand(B, hh, S, S).to(device=device, dtype=torch.uint8).contiguous()


def _mask_hstride(mk, H, S):
    if mk is None:
        return 0
    return S * S if mk.shape[1] == H and H > 1 else 0


def attention(q, k, v, mask=None):
    return _AttentionFn.apply(q, k, v, mask)


def attention_probabilities(q, k, mask=None):
    """The `score` tensor ScaleDotProductAttention.forward also returns (:34, discarded by MultiHeadAttention :24).
    The fused kernel never materialises it; callers that ask for it get it rebuilt from q, k and the kernel's
    log-sum-exp is not needed: a plain fp32 softmax over (B, H, S, S) on the device."""
    dh = q.shape[-1]
    s = (q.float() @ k.float().transpose(2, 3)) / math.sqrt(dh)
    if mask is not None:
        s = s.masked_fill(mask == 0, -10000)
    return torch.softmax(s, dim=-1)


# ------------------------------------------------------------------------------------------------
def patch_embed(x, weight, bias, kind, patch):
    """Non-overlapping conv embedding as patchify + GEMM.  kind 0: x (B,C,H,W), weight (D,C,p,p) -> (B, N, D);
    kind 1: x (B,C,L), weight (D,C,k) -> (B, L/k, D).  Differentiable w.r.t. weight and bias through `linear`."""
    _need_cuda(x, weight, bias)
    L = N.lib()
    D = weight.shape[0]
    Bn = x.shape[0]
    P = weight[0].numel()
    Kpad = (P + 31) // 32 * 32
    xs = x.detach().float().contiguous()
    if kind == 0:
        _, Cc, Hh, Ww = xs.shape
        tok = (Hh // patch) * (Ww // patch)
        dims = (Cc, Hh, Ww)
    else:
        _, Cc, Ll = xs.shape
        tok = Ll // patch
        dims = (Cc, Ll, 0)
    patches = torch.empty(Bn * tok, Kpad, dtype=torch.bfloat16, device=x.device)
    N.check(L.iq_patchify(xs.data_ptr(), patches.data_ptr(), kind, Bn, dims[0], dims[1], dims[2], patch, Kpad,
                          N.stream_handle()), "iq_patchify")
    w2 = weight.reshape(D, P)
    if Kpad != P:
        w2 = torch.nn.functional.pad(w2, (0, Kpad - P))
    return linear(patches.float(), w2, bias).view(Bn, tok, D)
