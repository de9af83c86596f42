"""Host-side mirror of the reference's nn.Module surface over the MI355X-native path.

Class names, constructor kwargs, attribute names (hence state_dict keys), parameter creation
order (hence seeded init) and error behaviour follow the reference:
  ViT tree      /root/reference/Transformer_Thesis/ViT/models/{amc_transformer,encoder}.py,
                .../embedding/{patch_embedding,positional_encoding}.py
  rawIQ tree    /root/reference/Transformer_Thesis/transformer_rawIQ/models/{transformer_rawIQ,encoder}.py,
                .../embedding/{patch_embedding,positional_encoding}.py
  shared        .../layers/{layers_norm,multi_head_attention,scale_dot_product_attention,
                position_wise_feed_forward}.py, .../blocks/encoder_layer.py
The modules own ordinary fp32 nn.Parameters (so .parameters(), optimizers, state_dict,
load_state_dict and .to() behave as in the reference).  The arithmetic does not live here:
AMCTransformer.forward / Encoder.forward hand the whole forward (and, through one
autograd.Function, the whole backward) to the native runtime in csrc/model.hip.  On first use on a
GPU the parameters are re-homed as views of one flat fp32 HBM buffer whose layout the native side
defines; nothing else about them changes.  A model and its encoder share ONE plan (one flat buffer):
`model.encoder(x)` runs the model's plan up to the encoder output.

The leaf layers (LayerNorm, MultiHeadAttention, PositionwiseFeedForward, EncoderLayer, the embeddings)
also run on their own -- forward hooks, per-layer feature extraction -- through the per-op C ABI
(functional.py: iq_ln_*, iq_gemm_bf16_*, iq_attn_*), with autograd; they are the reference's layer
bodies (V/models/blocks/encoder_layer.py:18-35 etc.) over native ops.  Training runs the plan, not these.

There is no CPU path: a CPU tensor raises.  The CPU restatement lives in oracle/ (tests only).
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from typing import Dict, Optional

import torch
from torch import nn

from . import _native as N
from . import functional as NF


# ------------------------------------------------------------------------------------------------
# leaf layers: same attribute names as the reference (hence state_dict keys); each forward is the
# reference's layer body over the native per-op entry points (functional.py)
# ------------------------------------------------------------------------------------------------
class LayerNorm(nn.Module):
    """layers_norm.py:4-19 -- params are `gamma` / `beta`, eps 1e-12."""

    def __init__(self, d_model, eps=1e-12):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(d_model))
        self.beta = nn.Parameter(torch.zeros(d_model))
        self.eps = eps

    def forward(self, x):
        return NF.layer_norm(x, self.gamma, self.beta, self.eps)


class ScaleDotProductAttention(nn.Module):
    """scale_dot_product_attention.py:5-39 (no parameters): softmax(q k^T / sqrt(dh)) v on (B, H, S, dh) tensors,
    optional mask (mask == 0 -> score -10000, :30-31).  Returns (v, score) like the reference; the fused kernel never
    forms `score`, so it is rebuilt with a plain softmax only when asked for (`need_score`, default as the reference;
    MultiHeadAttention discards it, multi_head_attention.py:24, and passes need_score=False)."""

    def forward(self, q, k, v, mask=None, e=1e-12, need_score=True):
        out = NF.attention(q, k, v, mask)
        score = NF.attention_probabilities(q, k, mask) if need_score else None
        return out, score


class MultiHeadAttention(nn.Module):
    """multi_head_attention.py:6-47 -- four separate Linear(D,D) with bias."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.n_head = n_head
        self.attention = ScaleDotProductAttention()
        self.w_q = nn.Linear(d_model, d_model)
        self.w_k = nn.Linear(d_model, d_model)
        self.w_v = nn.Linear(d_model, d_model)
        self.w_concat = nn.Linear(d_model, d_model)

    def forward(self, q, k, v, mask=None):
        q = NF.linear(q, self.w_q.weight, self.w_q.bias)
        k = NF.linear(k, self.w_k.weight, self.w_k.bias)
        v = NF.linear(v, self.w_v.weight, self.w_v.bias)
        q, k, v = self.split(q), self.split(k), self.split(v)
        out, _ = self.attention(q, k, v, mask=mask, need_score=False)
        out = self.concat(out)
        return NF.linear(out, self.w_concat.weight, self.w_concat.bias)

    def split(self, tensor):
        """(B, L, D) -> (B, H, L, dh)  (multi_head_attention.py:34-40)."""
        batch_size, length, d_model = tensor.size()
        d_tensor = d_model // self.n_head
        return tensor.view(batch_size, length, self.n_head, d_tensor).transpose(1, 2)

    def concat(self, tensor):
        """inverse of split (multi_head_attention.py:41-47)."""
        batch_size, head, length, d_tensor = tensor.size()
        return tensor.transpose(1, 2).contiguous().view(batch_size, length, head * d_tensor)


class PositionwiseFeedForward(nn.Module):
    """position_wise_feed_forward.py:3-17 -- Linear, ReLU, Dropout, Linear."""

    def __init__(self, d_model, hidden, drop_prob=0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, hidden)
        self.linear2 = nn.Linear(hidden, d_model)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=drop_prob)

    def forward(self, x):
        x = NF.linear(x, self.linear1.weight, self.linear1.bias, relu=True)     # linear1 + relu in one epilogue
        x = self.dropout(x)
        return NF.linear(x, self.linear2.weight, self.linear2.bias)


class EncoderLayer(nn.Module):
    """encoder_layer.py:7-35 -- post-norm block."""

    def __init__(self, d_model, ffn_hidden, n_head, drop_prob):
        super().__init__()
        self.attention = MultiHeadAttention(d_model=d_model, n_head=n_head)
        self.norm1 = LayerNorm(d_model=d_model)
        self.dropout1 = nn.Dropout(p=drop_prob)
        self.ffn = PositionwiseFeedForward(d_model=d_model, hidden=ffn_hidden, drop_prob=drop_prob)
        self.norm2 = LayerNorm(d_model=d_model)
        self.dropout2 = nn.Dropout(p=drop_prob)

    def forward(self, x, src_mask=None):
        _x = x
        x = self.attention(q=x, k=x, v=x, mask=src_mask)
        x = self.dropout1(x)
        x = self.norm1(x + _x)
        _x = x
        x = self.ffn(x)
        x = self.dropout2(x)
        return self.norm2(x + _x)


class PatchEmbedding(nn.Module):
    """ViT patch_embedding.py:3-15 -- Conv2d(C, D, k=p, s=p), flatten, transpose: (B,C,H,W) -> (B,N,D)."""

    def __init__(self, in_channels, patch_size, embedding_dim):
        super().__init__()
        self.projection = nn.Conv2d(in_channels, embedding_dim, kernel_size=patch_size, stride=patch_size)
        self._patch = patch_size

    def forward(self, x):
        return NF.patch_embed(x, self.projection.weight, self.projection.bias, 0, self._patch)


class SequenceEmbedding(nn.Module):
    """rawIQ patch_embedding.py:5-60 -- Conv1d k=1 ('conv1d') or k=s=segment ('segment')."""

    def __init__(self, in_channels=2, embedding_dim=256, method="conv1d", segment_size=None):
        super().__init__()
        self.in_channels = in_channels
        self.embedding_dim = embedding_dim
        self.method = method
        self.segment_size = segment_size
        if method == "conv1d":
            self.projection = nn.Conv1d(in_channels, embedding_dim, kernel_size=1)
        elif method == "segment":
            if segment_size is None:
                raise ValueError("segment_size is required for 'segment' method")
            self.projection = nn.Conv1d(in_channels, embedding_dim, kernel_size=segment_size, stride=segment_size)
        else:
            raise ValueError(f"Unknown method: {method}. Use 'conv1d' or 'segment'")

    def forward(self, x):
        """(B, C, L) -> (B, L/k, D)  (R/.../patch_embedding.py:47-60)."""
        k = 1 if self.method == "conv1d" else self.segment_size
        return NF.patch_embed(x, self.projection.weight, self.projection.bias, 1, k)


class PositionalEncodingViT(nn.Module):
    """ViT positional_encoding.py:4-29: table via pow(10000, 2i/D) then divide; buffer `encoding`."""

    def __init__(self, d_model, max_len=5000, device="cpu"):
        super().__init__()
        enc = torch.zeros(max_len, d_model)
        pos = torch.arange(0, max_len).float().unsqueeze(1)
        two_i = torch.arange(0, d_model, step=2).float()
        den = torch.pow(10000.0, two_i / d_model)
        enc[:, 0::2] = torch.sin(pos / den)
        enc[:, 1::2] = torch.cos(pos / den)
        self.register_buffer("encoding", enc)

    def forward(self, x):
        """x + encoding[:S]  (V/.../positional_encoding.py:21-29)."""
        return x + self.encoding[: x.size(1), :].unsqueeze(0)


class PositionalEncodingRawIQ(nn.Module):
    """rawIQ positional_encoding.py:6-82: table via exp(-ln(1e4) 2i/D) then multiply."""

    def __init__(self, d_model, max_len=5000, device="cpu", dropout=0.0):
        super().__init__()
        enc = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        enc[:, 0::2] = torch.sin(position * div_term)
        enc[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("encoding", enc)
        self.dropout = nn.Dropout(p=dropout) if dropout > 0 else None

    def forward(self, x):
        """R/.../positional_encoding.py:55-82."""
        seq_len = x.size(1)
        if seq_len > self.encoding.size(0):
            raise ValueError(f"Sequence length {seq_len} exceeds maximum length {self.encoding.size(0)}. "
                             "Increase max_len parameter.")
        x = x + self.encoding[:seq_len, :].unsqueeze(0)
        return self.dropout(x) if self.dropout is not None else x


# ------------------------------------------------------------------------------------------------
# native runner: flat parameter storage + workspace + calls into libiqvit.so
# ------------------------------------------------------------------------------------------------
class NativePlan:
    """One iq_model_t plus the HBM buffers it is bound to.  Owned by an AMCTransformer (shared with its
    encoder) or by a stand-alone Encoder."""
    _count = 0

    def __init__(self, owner: nn.Module, cfg: N.ModelCfg, prefix_strip: str = ""):
        self._owner = weakref.ref(owner)
        self.cfg = cfg
        self.strip = prefix_strip
        self.L = N.lib()
        h = C.c_void_p()
        rc = self.L.iq_model_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise N.IqError(
                "configuration not supported by the MI355X kernels "
                f"(d_model={cfg.d_model}, n_head={cfg.n_head}, ffn_hidden={cfg.ffn_hidden}): requires "
                "d_model % n_head == 0, d_model/n_head in {16,32,64}, d_model % 8 == 0, ffn_hidden % 8 == 0 and a "
                "sequence that fits the attention kernel's LDS")
        self.h = h
        self.S = self.L.iq_model_tokens(h)
        self.nparam = self.L.iq_model_param_floats(h)
        self.entries = []
        name = C.create_string_buffer(256)
        off, nd, dims = C.c_size_t(), C.c_int(), (C.c_int * 4)()
        for i in range(self.L.iq_model_param_entries(h)):
            N.check(self.L.iq_model_param_entry(h, i, name, 256, C.byref(off), C.byref(nd), dims), "param_entry")
            self.entries.append((name.value.decode(), off.value, tuple(dims[k] for k in range(nd.value))))
        self.flat: Optional[torch.Tensor] = None
        self.gflat: Optional[torch.Tensor] = None
        self.shadow: Optional[torch.Tensor] = None
        self.ws: Optional[torch.Tensor] = None
        self.ws_batch = 0
        self.shadow_version = -1
        self.generation = 0          # bumped by every forward that writes the workspace
        # dropout seed: derived from torch's seed WITHOUT drawing from the global generator (a draw here, at the first
        # forward, would shift a seeded caller's random stream relative to the reference's)
        NativePlan._count += 1
        self.seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + NativePlan._count * 0xD1B54A32D192ED03) & ((1 << 62) - 1)
        self.step = 0
        self.step_ctr: Optional[torch.Tensor] = None    # persistent device u32: the dropout step the kernels read
        self.ctr_value = -1                             # host mirror of step_ctr (-1: unknown)
        self._probe = None

    def __deepcopy__(self, memo):      # copies of a module build their own plan lazily
        return None

    def __reduce__(self):              # pickling a module drops the native handle
        return (type(None), ())

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.iq_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------------------------
    def _named(self) -> Dict[str, nn.Parameter]:
        owner = self._owner()
        return dict(owner.named_parameters())

    def _pe(self) -> torch.Tensor:
        owner = self._owner()
        enc = owner.encoder if hasattr(owner, "encoder") else owner
        return enc.positional_encoding.encoding

    def bind(self, device: torch.device):
        """(Re)home the owner's parameters as views of one flat fp32 buffer on `device`."""
        named = self._named()
        flat = torch.zeros(self.nparam, dtype=torch.float32, device=device)
        first = None
        bound = []
        with torch.no_grad():
            for name, off, shape in self.entries:
                key = name[len(self.strip):] if self.strip and name.startswith(self.strip) else name
                p = named.get(key)
                if p is None:
                    continue            # e.g. head parameters when a bare Encoder owns the plan
                if tuple(p.shape) != shape:
                    raise N.IqError(f"parameter {key} has shape {tuple(p.shape)}, native layout expects {shape}")
                n = p.numel()
                view = flat[off:off + n].view(shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                bound.append(p)
                if first is None:
                    first = p
        self.flat = flat
        self._probe = first
        self._bound = bound
        pe = self._pe()
        if pe.device != device or pe.dtype != torch.float32 or not pe.is_contiguous():
            raise N.IqError("positional encoding buffer must be a contiguous fp32 tensor on the model's device")
        self.shadow = torch.empty(self.L.iq_model_shadow_bytes(self.h), dtype=torch.uint8, device=device)
        self.gflat = None
        self.shadow_version = -1
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=device)
        self.ctr_value = 0
        N.check(self.L.iq_model_bind_step_counter(self.h, N.ptr(self.step_ctr)), "iq_model_bind_step_counter", self.h)
        self._bind_native(None)

    def _bind_native(self, gflat):
        N.check(self.L.iq_model_bind(self.h, N.ptr(self.flat), N.ptr(gflat), N.ptr(self._pe()), N.ptr(self.shadow)),
                "iq_model_bind", self.h)
        self.gflat = gflat

    def is_bound(self, device=None) -> bool:
        """True while the owner's parameters are still views of this plan's flat buffer."""
        p = self._probe
        return not (self.flat is None or (device is not None and self.flat.device != device) or p is None
                    or p.data.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr())

    def ensure(self, device: torch.device):
        if not self.is_bound(device):
            self.bind(device)
        # `p.data = view` does not share version counters with the flat buffer, so sum the parameters'
        # own counters: every in-place update (optimizer step, load_state_dict, p.add_()) bumps one.
        # Writes through `p.data` are invisible to autograd versions: call mark_dirty() after those.
        v = 0
        for q in self._bound:
            v += q._version
        if v != self.shadow_version:
            N.check(self.L.iq_model_refresh_shadow(self.h, N.stream_handle()), "refresh_shadow", self.h)
            self.shadow_version = v

    def mark_dirty(self):
        """Force a bf16 shadow refresh on the next forward (after writing parameters via `.data`)."""
        self.shadow_version = -1

    def workspace(self, batch: int, device) -> torch.Tensor:
        need = self.L.iq_model_workspace_bytes(self.h, batch, 1)
        if self.ws is None or self.ws.numel() < need or self.ws.device != device:
            self.ws = None
            self.ws = torch.empty(need, dtype=torch.uint8, device=device)
        self.ws_batch = batch
        return self.ws

    # -- forward / backward -------------------------------------------------------------------
    def forward(self, src: torch.Tensor, training: bool, want_logits: bool, want_enc: bool):
        dev = src.device
        self.ensure(dev)
        B = src.shape[0]
        ws = self.workspace(B, dev)
        logits = torch.empty(B, self.cfg.num_classes, dtype=torch.float32, device=dev) if want_logits else None
        enc = torch.empty(B, self.S, self.cfg.d_model, dtype=torch.float32, device=dev) if want_enc else None
        if training:
            self.step += 1
            self.ctr_value = self.step & 0x7FFFFFFF     # the forward writes this value into step_ctr
        self.generation += 1
        N.check(self.L.iq_model_forward(self.h, N.ptr(src), B, N.ptr(ws), ws.numel(), 1 if training else 0,
                                        self.seed, self.step & 0x7FFFFFFF, N.ptr(enc), N.ptr(logits),
                                        N.stream_handle()), "iq_model_forward", self.h)
        return logits, enc

    def backward(self, batch: int, dlogits, denc, gflat: torch.Tensor, accumulate=False, stage_hi=None, stage_lo=0):
        if gflat is not self.gflat:
            self._bind_native(gflat)
        hi = self.cfg.n_layers + 1 if stage_hi is None else stage_hi
        N.check(self.L.iq_model_backward(self.h, N.ptr(dlogits), N.ptr(denc), batch, N.ptr(self.ws), self.ws.numel(),
                                         1 if accumulate else 0, hi, stage_lo, N.stream_handle()),
                "iq_model_backward", self.h)

    def grad_range(self, stage_hi: int, stage_lo: int):
        off, ln = C.c_size_t(), C.c_size_t()
        N.check(self.L.iq_model_grad_range(self.h, stage_hi, stage_lo, C.byref(off), C.byref(ln)), "grad_range")
        return off.value, ln.value

    def grad_views(self, gflat: torch.Tensor, params_in_order):
        """Views of the flat gradient matching `params_in_order` (list of (name, param))."""
        index = {}
        for name, off, shape in self.entries:
            key = name[len(self.strip):] if self.strip and name.startswith(self.strip) else name
            index[key] = (off, shape)
        out = []
        for name, p in params_in_order:
            off, shape = index[name]
            out.append(gflat[off:off + p.numel()].view(shape))
        return out


class _PlanFn(torch.autograd.Function):
    """Whole-model autograd node: forward and backward are single native calls."""

    @staticmethod
    def forward(ctx, plan: NativePlan, src, training, want, names, *params):
        logits, enc = plan.forward(src, training, want == "logits", want == "enc")
        ctx.plan = plan
        ctx.want = want
        ctx.names = names
        ctx.batch = src.shape[0]
        ctx.generation = plan.generation
        ctx.nparams = len(params)
        ctx.set_materialize_grads(False)
        return logits if want == "logits" else enc

    @staticmethod
    def backward(ctx, gout):
        plan: NativePlan = ctx.plan
        none = (None,) * 5
        if gout is None:
            return none + (None,) * ctx.nparams
        if ctx.generation != plan.generation:
            raise RuntimeError(
                "backward() after a later forward() on the same model: the native workspace holding the saved "
                "activations has been overwritten. Call backward before running the model again.")
        gout = gout.contiguous().float()
        gflat = torch.empty_like(plan.flat)
        if ctx.want == "logits":
            plan.backward(ctx.batch, gout, None, gflat)
        else:
            plan.backward(ctx.batch, None, gout, gflat)
        named = plan._named()
        grads = plan.grad_views(gflat, [(n, named[n]) for n in ctx.names])
        return none + tuple(grads)


def _check_src(src, ndim, what):
    if not isinstance(src, torch.Tensor) or src.dim() != ndim:
        raise ValueError(f"expected a {ndim}-D tensor {what}, got {tuple(src.shape) if hasattr(src, 'shape') else src}")
    if not src.is_cuda:
        raise N.IqError(
            "this framework runs on MI355X only: the input is a CPU tensor and there is no CPU fallback "
            "(the CPU restatement used for parity checks lives in oracle/ and is test infrastructure)")
    return src.contiguous().float()


class _WeakLink:
    """Weak back-reference from an encoder to the AMCTransformer that owns it.  Copies and pickles carry an empty
    link; the owner re-links in its constructor / __setstate__ (copy.deepcopy and torch.save of whole modules)."""

    def __init__(self, target=None):
        self._ref = weakref.ref(target) if target is not None else None

    def __call__(self):
        return self._ref() if self._ref is not None else None

    def __deepcopy__(self, memo):
        return _WeakLink()

    def __reduce__(self):
        return (_WeakLink, ())


def _link(parent):
    object.__setattr__(parent.encoder, "_parent_ref", _WeakLink(parent))


def _parent_of(encoder):
    """The AMCTransformer that owns `encoder`, if it still does."""
    ref = getattr(encoder, "_parent_ref", None)
    parent = ref() if ref is not None else None
    return parent if parent is not None and getattr(parent, "encoder", None) is encoder else None


def _run(plan: NativePlan, owner: nn.Module, src, want, training=None):
    """`training` = the flag of the module that was CALLED (an encoder reached through its parent's plan obeys its own
    train()/eval() state, as Encoder.forward does in the reference); default: the owner's."""
    training = owner.training if training is None else bool(training)
    names, params = [], []
    for n, p in owner.named_parameters():
        names.append(n)
        params.append(p)
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if needs_grad:
        plan.ensure(src.device)      # re-home BEFORE autograd records the parameter tensors
        return _PlanFn.apply(plan, src, training, want, tuple(names), *params)
    logits, enc = plan.forward(src, training, want == "logits", want == "enc")
    return logits if want == "logits" else enc


# ------------------------------------------------------------------------------------------------
# ViT tree
# ------------------------------------------------------------------------------------------------
class EncoderViT(nn.Module):
    """ViT/models/encoder.py:7-53."""

    def __init__(self, in_channels, img_size_h, img_size_w, patch_size, d_model, ffn_hidden, n_head, n_layers,
                 drop_prob, device):
        super().__init__()
        self.device = device
        self.patch_embedding = PatchEmbedding(in_channels=in_channels, patch_size=patch_size, embedding_dim=d_model)
        num_patches = (img_size_h // patch_size) * (img_size_w // patch_size)
        self.positional_encoding = PositionalEncodingViT(d_model=d_model, max_len=num_patches + 1, device=device)
        self.cls_token = nn.Parameter(torch.randn(1, 1, d_model))
        self.layers = nn.ModuleList([EncoderLayer(d_model=d_model, ffn_hidden=ffn_hidden, n_head=n_head,
                                                  drop_prob=drop_prob) for _ in range(n_layers)])
        self.dropout = nn.Dropout(p=drop_prob)
        self._geom = dict(kind=0, in_channels=in_channels, img_h=img_size_h, img_w=img_size_w, patch=patch_size,
                          seq_length=0, conv_k=0, use_cls=1, d_model=d_model, n_head=n_head, n_layers=n_layers,
                          ffn_hidden=ffn_hidden, drop_prob=float(drop_prob))
        self._plan: Optional[NativePlan] = None

    def _cfg(self, num_classes=1):
        return N.ModelCfg(num_classes=num_classes, **self._geom)

    def _forward_layers(self, src, src_mask):
        """V/models/encoder.py:34-53 layer by layer (the only path that takes a src_mask; no reference caller passes
        one, so the fused plan has no mask input)."""
        x = self.patch_embedding(src)
        cls = self.cls_token.repeat(x.shape[0], 1, 1)
        x = torch.cat([cls, x], dim=1)
        x = self.dropout(self.positional_encoding(x))
        for layer in self.layers:
            x = layer(x, src_mask)
        return x

    def _expect(self, src):
        g = self._geom
        src = _check_src(src, 4, "(batch, in_channels, img_size_h, img_size_w)")
        if tuple(src.shape[1:]) != (g["in_channels"], g["img_h"], g["img_w"]):
            raise RuntimeError(f"input shape {tuple(src.shape)} does not match the model geometry "
                               f"(B, {g['in_channels']}, {g['img_h']}, {g['img_w']})")
        return src

    def forward(self, src, src_mask=None):
        if src_mask is not None:
            return self._forward_layers(src, src_mask)
        src = self._expect(src)
        parent = _parent_of(self)
        if parent is not None:           # one plan / one flat parameter buffer per model: run the model's plan
            return _run(parent.native_plan(), parent, src, "enc", training=self.training)
        if self._plan is None:
            self._plan = NativePlan(self, self._cfg(), prefix_strip="encoder.")
        return _run(self._plan, self, src, "enc")


class AMCTransformerViT(nn.Module):
    """ViT/models/amc_transformer.py:5-31."""

    def __init__(self, in_channels, img_size_h, img_size_w, patch_size, num_classes, d_model, n_head, n_layers,
                 ffn_hidden, drop_prob, device):
        super().__init__()
        self.encoder = EncoderViT(in_channels=in_channels, img_size_h=img_size_h, img_size_w=img_size_w,
                                  patch_size=patch_size, d_model=d_model, n_head=n_head, ffn_hidden=ffn_hidden,
                                  drop_prob=drop_prob, n_layers=n_layers, device=device)
        self.mlp_head = nn.Linear(d_model, num_classes)
        self._num_classes = num_classes
        self._plan: Optional[NativePlan] = None
        _link(self)

    def __setstate__(self, state):
        super().__setstate__(state)
        _link(self)

    def native_plan(self) -> NativePlan:
        if self._plan is None:
            _link(self)
            self._plan = NativePlan(self, self.encoder._cfg(self._num_classes))
        return self._plan

    def forward(self, src):
        src = self.encoder._expect(src)
        return _run(self.native_plan(), self, src, "logits")


# ------------------------------------------------------------------------------------------------
# raw-IQ tree
# ------------------------------------------------------------------------------------------------
class EncoderRawIQ(nn.Module):
    """transformer_rawIQ/models/encoder.py:8-153."""

    def __init__(self, in_channels, seq_length, d_model, ffn_hidden, n_head, n_layers, drop_prob, device,
                 use_cls_token=True, embedding_type="conv1d", segment_size=64):
        super().__init__()
        self.device = device
        self.use_cls_token = use_cls_token
        self.embedding_type = embedding_type
        if embedding_type == "conv1d":
            self.sequence_embedding = SequenceEmbedding(in_channels=in_channels, embedding_dim=d_model, method="conv1d")
            num_tokens = seq_length
            conv_k = 1
        elif embedding_type == "segment":
            if seq_length % segment_size != 0:
                raise ValueError(f"seq_length ({seq_length}) must be divisible by segment_size ({segment_size})")
            self.sequence_embedding = SequenceEmbedding(in_channels=in_channels, embedding_dim=d_model,
                                                        segment_size=segment_size, method="segment")
            num_tokens = seq_length // segment_size
            conv_k = segment_size
        else:
            raise ValueError(f"Unknown embedding_type: {embedding_type}")
        max_len = num_tokens + (1 if use_cls_token else 0)
        self.positional_encoding = PositionalEncodingRawIQ(d_model=d_model, max_len=max_len, device=device, dropout=0.0)
        if use_cls_token:
            self.cls_token = nn.Parameter(torch.randn(1, 1, d_model))
        self.layers = nn.ModuleList([EncoderLayer(d_model=d_model, ffn_hidden=ffn_hidden, n_head=n_head,
                                                  drop_prob=drop_prob) for _ in range(n_layers)])
        self.dropout = nn.Dropout(p=drop_prob)
        self._geom = dict(kind=1, in_channels=in_channels, img_h=0, img_w=0, patch=0, seq_length=seq_length,
                          conv_k=conv_k, use_cls=1 if use_cls_token else 0, d_model=d_model, n_head=n_head,
                          n_layers=n_layers, ffn_hidden=ffn_hidden, drop_prob=float(drop_prob))
        self._plan: Optional[NativePlan] = None

    def _cfg(self, num_classes=1):
        return N.ModelCfg(num_classes=num_classes, **self._geom)

    def _forward_layers(self, src, src_mask):
        """R/models/encoder.py:86-117 layer by layer (the only path that takes a src_mask)."""
        x = self.sequence_embedding(src)
        if self.use_cls_token:
            x = torch.cat([self.cls_token.repeat(x.shape[0], 1, 1), x], dim=1)
        x = self.dropout(self.positional_encoding(x))
        for layer in self.layers:
            x = layer(x, src_mask)
        return x

    def _expect(self, src):
        g = self._geom
        src = _check_src(src, 3, "(batch, in_channels, seq_length)")
        if src.shape[1] != g["in_channels"]:
            raise RuntimeError(f"expected {g['in_channels']} input channels, got {src.shape[1]}")
        n_tok = src.shape[2] // g["conv_k"] + g["use_cls"]
        max_len = self.positional_encoding.encoding.size(0)
        if n_tok > max_len:
            raise ValueError(f"Sequence length {n_tok} exceeds maximum length {max_len}. Increase max_len parameter.")
        if src.shape[2] != g["seq_length"]:
            raise RuntimeError(f"input length {src.shape[2]} does not match seq_length {g['seq_length']} the native "
                               "plan was built for")
        return src

    def forward(self, src, src_mask=None):
        if src_mask is not None:
            return self._forward_layers(src, src_mask)
        src = self._expect(src)
        parent = _parent_of(self)
        if parent is not None:           # one plan / one flat parameter buffer per model: run the model's plan
            return _run(parent.native_plan(), parent, src, "enc", training=self.training)
        if self._plan is None:
            self._plan = NativePlan(self, self._cfg(), prefix_strip="encoder.")
        return _run(self._plan, self, src, "enc")

    def get_cls_token_output(self, src, src_mask=None):
        if not self.use_cls_token:
            raise ValueError("CLS token is not enabled. Set use_cls_token=True")
        return self.forward(src, src_mask)[:, 0, :]

    def get_sequence_output(self, src, src_mask=None):
        x = self.forward(src, src_mask)
        return x[:, 1:, :] if self.use_cls_token else x


class AMCTransformerRawIQ(nn.Module):
    """transformer_rawIQ/models/transformer_rawIQ.py:7-97."""

    def __init__(self, in_channels, seq_length, num_classes, d_model, n_head, n_layers, ffn_hidden, drop_prob, device,
                 use_cls_token=True, embedding_type="segment", segment_size=64):
        super().__init__()
        self.use_cls_token = use_cls_token
        self.d_model = d_model
        self.encoder = EncoderRawIQ(in_channels=in_channels, seq_length=seq_length, d_model=d_model, n_head=n_head,
                                    ffn_hidden=ffn_hidden, drop_prob=drop_prob, n_layers=n_layers, device=device,
                                    use_cls_token=use_cls_token, embedding_type=embedding_type,
                                    segment_size=segment_size)
        self.mlp_head = nn.Sequential(nn.LayerNorm(d_model), nn.Linear(d_model, num_classes))
        self._num_classes = num_classes
        self._plan: Optional[NativePlan] = None
        _link(self)

    def __setstate__(self, state):
        super().__setstate__(state)
        _link(self)

    def native_plan(self) -> NativePlan:
        if self._plan is None:
            _link(self)
            self._plan = NativePlan(self, self.encoder._cfg(self._num_classes))
        return self._plan

    def forward(self, src):
        src = self.encoder._expect(src)
        return _run(self.native_plan(), self, src, "logits")
