"""CPU oracle for the ViT / raw-IQ training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *functional* restatement (plain PyTorch, fp32, CPU)
of the algorithm the reference implements with nn.Modules.  It exists so the
HIP path can be checked; it is never imported by the product package
(`vit-vs-raw-iq_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.

Parity pin: the reference ships no golden vectors.  The
oracle is pinned against the reference itself, imported in the build
container by `tests/golden/make_golden.py`, which asserts equality of logits,
loss, every gradient and the post-step parameters and writes the fixtures in
`tests/golden/*.npz`.  `tests/test_oracle_golden.py` re-checks the oracle
against those fixtures on every CPU run.

Citations (all under /root/reference/Transformer_Thesis/):
  V/ = ViT/, R/ = transformer_rawIQ/
  init order .............. V/models/encoder.py:16-30, V/models/amc_transformer.py:13-24,
                            R/models/encoder.py:34-84, R/models/transformer_rawIQ.py:52-70
  patch embedding ......... V/models/embedding/patch_embedding.py:9-15
  sequence embedding ...... R/models/embedding/patch_embedding.py:27-60
  positional encoding ..... V/models/embedding/positional_encoding.py:9-29 (pow + divide)
                            R/models/embedding/positional_encoding.py:28-43 (exp + multiply)
  layer norm .............. V/models/layers/layers_norm.py:11-19 (eps 1e-12, biased var)
  attention ............... V/models/layers/scale_dot_product_attention.py:23-39
  multi-head .............. V/models/layers/multi_head_attention.py:16-47
  feed forward ............ V/models/layers/position_wise_feed_forward.py:12-17 (ReLU)
  encoder layer ........... V/models/blocks/encoder_layer.py:18-35 (post-norm)
  heads ................... V/models/amc_transformer.py:26-31, R/models/transformer_rawIQ.py:88-96
  training step ........... V/training/train.py:185-207,405-412; R/training/train.py:252-277,504-511
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, Optional

import torch

State = Dict[str, torch.Tensor]


@dataclass
class OracleConfig:
    kind: str = "vit"              # "vit" | "rawiq"
    in_channels: int = 1
    img_size_h: int = 32           # vit only
    img_size_w: int = 32           # vit only
    patch_size: int = 16           # vit only
    seq_length: int = 1024         # rawiq only
    embedding_type: str = "segment"  # rawiq only: "segment" | "conv1d"
    segment_size: int = 64         # rawiq only
    use_cls_token: bool = True     # rawiq only (vit always has one)
    num_classes: int = 11
    d_model: int = 128
    n_head: int = 8
    n_layers: int = 2
    ffn_hidden: int = 512
    drop_prob: float = 0.1

    def tokens(self) -> int:
        """Embedded tokens per frame, cls excluded."""
        if self.kind == "vit":
            return (self.img_size_h // self.patch_size) * (self.img_size_w // self.patch_size)
        if self.embedding_type == "conv1d":
            return self.seq_length
        if self.embedding_type == "segment":
            if self.seq_length % self.segment_size != 0:
                raise ValueError(
                    f"seq_length ({self.seq_length}) must be divisible by segment_size ({self.segment_size})")
            return self.seq_length // self.segment_size
        raise ValueError(f"Unknown embedding_type: {self.embedding_type}")

    def has_cls(self) -> bool:
        return True if self.kind == "vit" else bool(self.use_cls_token)

    def seq(self) -> int:
        return self.tokens() + (1 if self.has_cls() else 0)

    def to_dict(self):
        return asdict(self)


# ----------------------------------------------------------------------------
# parameter construction, in the reference's RNG consumption order
# ----------------------------------------------------------------------------

def _uniform_fan_in(shape, fan_in):
    """What nn.Linear / nn.ConvNd.reset_parameters draws: kaiming_uniform_(a=sqrt(5))
    is U(-1/sqrt(fan_in), 1/sqrt(fan_in)); the bias uses the same bound."""
    bound = 1.0 / math.sqrt(fan_in)
    return torch.empty(shape, dtype=torch.float32).uniform_(-bound, bound)


def _linear(sd: State, name: str, out_f: int, in_f: int):
    sd[name + ".weight"] = _uniform_fan_in((out_f, in_f), in_f)
    sd[name + ".bias"] = _uniform_fan_in((out_f,), in_f)


def positional_table(cfg: OracleConfig) -> torch.Tensor:
    S, D = cfg.seq(), cfg.d_model
    enc = torch.zeros(S, D)
    pos = torch.arange(0, S, dtype=torch.float32).unsqueeze(1)
    idx = torch.arange(0, D, 2, dtype=torch.float32)
    if cfg.kind == "vit":           # V/.../positional_encoding.py:11-16
        den = torch.pow(torch.tensor(10000.0), idx / D)
        enc[:, 0::2] = torch.sin(pos / den)
        enc[:, 1::2] = torch.cos(pos / den)
    else:                            # R/.../positional_encoding.py:31-43
        mul = torch.exp(idx * -(math.log(10000.0) / D))
        enc[:, 0::2] = torch.sin(pos * mul)
        enc[:, 1::2] = torch.cos(pos * mul)
    return enc


def init_state(cfg: OracleConfig, seed: Optional[int] = None) -> State:
    """Fresh parameters + buffer keyed exactly like the reference's state_dict.
    With the same torch seed this is bit-identical to constructing the reference
    model (asserted by tests/golden/make_golden.py)."""
    if seed is not None:
        torch.manual_seed(seed)
    D, F = cfg.d_model, cfg.ffn_hidden
    sd: State = {}
    if cfg.kind == "vit":
        p = cfg.patch_size
        fan = cfg.in_channels * p * p
        sd["encoder.patch_embedding.projection.weight"] = _uniform_fan_in((D, cfg.in_channels, p, p), fan)
        sd["encoder.patch_embedding.projection.bias"] = _uniform_fan_in((D,), fan)
    else:
        k = 1 if cfg.embedding_type == "conv1d" else cfg.segment_size
        cfg.tokens()  # raises the reference's ValueErrors
        fan = cfg.in_channels * k
        sd["encoder.sequence_embedding.projection.weight"] = _uniform_fan_in((D, cfg.in_channels, k), fan)
        sd["encoder.sequence_embedding.projection.bias"] = _uniform_fan_in((D,), fan)
    sd["encoder.positional_encoding.encoding"] = positional_table(cfg)
    if cfg.has_cls():
        sd["encoder.cls_token"] = torch.randn(1, 1, D)
    for i in range(cfg.n_layers):
        pre = f"encoder.layers.{i}."
        for w in ("w_q", "w_k", "w_v", "w_concat"):
            _linear(sd, pre + "attention." + w, D, D)
        sd[pre + "norm1.gamma"] = torch.ones(D)
        sd[pre + "norm1.beta"] = torch.zeros(D)
        _linear(sd, pre + "ffn.linear1", F, D)
        _linear(sd, pre + "ffn.linear2", D, F)
        sd[pre + "norm2.gamma"] = torch.ones(D)
        sd[pre + "norm2.beta"] = torch.zeros(D)
    if cfg.kind == "vit":
        _linear(sd, "mlp_head", cfg.num_classes, D)
    else:
        sd["mlp_head.0.weight"] = torch.ones(D)
        sd["mlp_head.0.bias"] = torch.zeros(D)
        _linear(sd, "mlp_head.1", cfg.num_classes, D)
    return sd


BUFFER_KEYS = ("encoder.positional_encoding.encoding",)


def param_keys(sd: State):
    return [k for k in sd if k not in BUFFER_KEYS]


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------

def _dropout(x, p, train):
    if not train or p <= 0.0:
        return x
    keep = (torch.rand_like(x) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


def custom_layer_norm(x, gamma, beta, eps=1e-12):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)      # biased
    return gamma * ((x - mean) / torch.sqrt(var + eps)) + beta


def attention_core(q, k, v, mask=None):
    """q,k,v: (B,H,S,dh).  softmax(q k^T / sqrt(dh)) v; no dropout.  Optional mask: positions with mask == 0 get the
    score -10000 after scaling (scale_dot_product_attention.py:26-31; no reference caller passes one)."""
    dh = q.shape[-1]
    score = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(dh)
    if mask is not None:
        score = score.masked_fill(mask == 0, -10000)
    score = score - score.max(dim=-1, keepdim=True).values
    e = torch.exp(score)
    prob = e / e.sum(dim=-1, keepdim=True)
    return torch.matmul(prob, v)


def embed(cfg: OracleConfig, sd: State, src):
    """(B,C,H,W) or (B,C,L) -> (B,N,D): the non-overlapping conv written as a GEMM."""
    D = cfg.d_model
    B = src.shape[0]
    if cfg.kind == "vit":
        p = cfg.patch_size
        C, H, W = src.shape[1:]
        gh, gw = H // p, W // p
        x = src[:, :, :gh * p, :gw * p].reshape(B, C, gh, p, gw, p)
        x = x.permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
        w = sd["encoder.patch_embedding.projection.weight"].reshape(D, -1)
        return x @ w.t() + sd["encoder.patch_embedding.projection.bias"]
    w3 = sd["encoder.sequence_embedding.projection.weight"]
    k = w3.shape[2]
    C, L = src.shape[1:]
    n = L // k
    x = src[:, :, :n * k].reshape(B, C, n, k).permute(0, 2, 1, 3).reshape(B, n, C * k)
    return x @ w3.reshape(D, -1).t() + sd["encoder.sequence_embedding.projection.bias"]


def encoder_forward(cfg: OracleConfig, sd: State, src, train=False):
    p = cfg.drop_prob
    x = embed(cfg, sd, src)
    B = x.shape[0]
    if cfg.has_cls():
        x = torch.cat([sd["encoder.cls_token"].expand(B, 1, cfg.d_model), x], dim=1)
    S = x.shape[1]
    pe = sd["encoder.positional_encoding.encoding"]
    if S > pe.shape[0]:
        raise ValueError(
            f"Sequence length {S} exceeds maximum length {pe.shape[0]}. Increase max_len parameter.")
    x = _dropout(x + pe[:S].unsqueeze(0), p, train)
    for i in range(cfg.n_layers):
        x = encoder_layer(sd, f"encoder.layers.{i}.", x, cfg.n_head, p, train)
    return x


def multi_head_attention(sd: State, pre: str, x, n_head: int, mask=None):
    """MultiHeadAttention.forward(q=x, k=x, v=x) (multi_head_attention.py:16-32); `pre` = state_dict prefix of the module."""
    B, S, D = x.shape
    dh = D // n_head

    def lin(t, name):
        return t @ sd[pre + name + ".weight"].t() + sd[pre + name + ".bias"]

    q = lin(x, "w_q").view(B, S, n_head, dh).transpose(1, 2)
    k = lin(x, "w_k").view(B, S, n_head, dh).transpose(1, 2)
    v = lin(x, "w_v").view(B, S, n_head, dh).transpose(1, 2)
    a = attention_core(q, k, v, mask).transpose(1, 2).reshape(B, S, D)
    return lin(a, "w_concat")


def feed_forward(sd: State, pre: str, x, p=0.0, train=False):
    """PositionwiseFeedForward.forward (position_wise_feed_forward.py:12-17)."""
    h = torch.relu(x @ sd[pre + "linear1.weight"].t() + sd[pre + "linear1.bias"])
    h = _dropout(h, p, train)
    return h @ sd[pre + "linear2.weight"].t() + sd[pre + "linear2.bias"]


def encoder_layer(sd: State, pre: str, x, n_head: int, p=0.0, train=False, mask=None):
    """EncoderLayer.forward, post-norm (encoder_layer.py:18-35)."""
    a = multi_head_attention(sd, pre + "attention.", x, n_head, mask)
    x = custom_layer_norm(_dropout(a, p, train) + x, sd[pre + "norm1.gamma"], sd[pre + "norm1.beta"])
    h = feed_forward(sd, pre + "ffn.", x, p, train)
    return custom_layer_norm(_dropout(h, p, train) + x, sd[pre + "norm2.gamma"], sd[pre + "norm2.beta"])


def model_forward(cfg: OracleConfig, sd: State, src, train=False):
    enc = encoder_forward(cfg, sd, src, train)
    feat = enc[:, 0] if cfg.has_cls() else enc.mean(dim=1)
    if cfg.kind == "vit":
        return feat @ sd["mlp_head.weight"].t() + sd["mlp_head.bias"]
    mean = feat.mean(-1, keepdim=True)
    var = ((feat - mean) ** 2).mean(-1, keepdim=True)
    feat = (feat - mean) / torch.sqrt(var + 1e-5) * sd["mlp_head.0.weight"] + sd["mlp_head.0.bias"]
    return feat @ sd["mlp_head.1.weight"].t() + sd["mlp_head.1.bias"]


# ----------------------------------------------------------------------------
# loss / clip / AdamW  (restating torch.nn.CrossEntropyLoss(label_smoothing),
# clip_grad_norm_ and torch.optim.AdamW as the reference configures them)
# ----------------------------------------------------------------------------

def smoothed_cross_entropy(logits, labels, smoothing=0.1):
    logp = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    nll = -logp.gather(1, labels.view(-1, 1)).squeeze(1)
    uni = -logp.mean(dim=1)
    return ((1.0 - smoothing) * nll + smoothing * uni).mean()


def clip_coefficient(grads, max_norm=1.0):
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, coef


@dataclass
class AdamWState:
    step: int
    exp_avg: State
    exp_avg_sq: State


def adamw_init(sd: State) -> AdamWState:
    keys = param_keys(sd)
    return AdamWState(0, {k: torch.zeros_like(sd[k]) for k in keys},
                      {k: torch.zeros_like(sd[k]) for k in keys})


def adamw_update(sd: State, grads: State, st: AdamWState, lr=1e-4, betas=(0.9, 0.99),
                 eps=1e-8, weight_decay=1e-3):
    """Decoupled weight decay on EVERY parameter (reference passes model.parameters()
    without groups: V/training/train.py:407-412)."""
    st.step += 1
    b1, b2 = betas
    bc1 = 1.0 - b1 ** st.step
    bc2 = 1.0 - b2 ** st.step
    for k in param_keys(sd):
        g = grads[k]
        p = sd[k]
        p.mul_(1.0 - lr * weight_decay)
        st.exp_avg[k].mul_(b1).add_(g, alpha=1.0 - b1)
        st.exp_avg_sq[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (st.exp_avg_sq[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st.exp_avg[k], denom, value=-(lr / bc1))


def loss_and_grads(cfg: OracleConfig, sd: State, src, labels, smoothing=0.1, train=False):
    """Returns (logits, loss, grads dict).  Backward through torch autograd of the
    functional forward above."""
    leaf = {k: (v.detach().clone().requires_grad_(True) if k not in BUFFER_KEYS else v)
            for k, v in sd.items()}
    logits = model_forward(cfg, leaf, src, train)
    loss = smoothed_cross_entropy(logits, labels, smoothing)
    keys = param_keys(sd)
    gs = torch.autograd.grad(loss, [leaf[k] for k in keys])
    return logits.detach(), loss.detach(), dict(zip(keys, gs))


def train_step(cfg: OracleConfig, sd: State, st: AdamWState, src, labels, lr=1e-4,
               weight_decay=1e-3, smoothing=0.1, max_norm=1.0, train=True):
    """One reference training step (V/training/train.py:191-201) on `sd`, in place.
    Returns (loss, n_correct, grad_norm)."""
    logits, loss, grads = loss_and_grads(cfg, sd, src, labels, smoothing, train)
    total, coef = clip_coefficient(list(grads.values()), max_norm)
    grads = {k: g * coef for k, g in grads.items()}
    with torch.no_grad():
        adamw_update(sd, grads, st, lr=lr, weight_decay=weight_decay)
    correct = int((logits.argmax(1) == labels).sum())
    return float(loss), correct, float(total)


def count_parameters(sd: State) -> int:
    return sum(sd[k].numel() for k in param_keys(sd))
