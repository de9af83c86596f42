#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE in this build container.

The reference (aliftffd/ViT-vs-Raw-IQ, mounted read-only at /root/reference) is
imported here -- and only here -- to
  (1) assert that oracle/iq_oracle.py equals it (init under the same seed,
      logits, loss, every gradient, parameters after one clip+AdamW step), and
  (2) write small input/output vectors as fixtures.
The fixtures are data only (inputs, weights drawn by the reference's own
constructors, expected outputs); no reference source travels.

The two reference trees both call their top package `models`, so each tree runs
in its own subprocess:   python tests/golden/make_golden.py            (driver)
                         python tests/golden/make_golden.py --tree vit (worker)

`typing.LiteralString` does not exist on Python 3.10; the reference imports it
(unused) at ViT/models/layers/multi_head_attention.py:1, so the worker defines
it before importing.  Reference files are not modified.
"""
import argparse
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Transformer_Thesis"

# name -> (tree, constructor kwargs, batch, store_full_weights)
CASES = {
    # BASELINE.json configs[0] / V/test_model.py intent
    "vit_A": ("vit", dict(in_channels=1, img_size_h=32, img_size_w=32, patch_size=16, num_classes=11,
                          d_model=128, n_head=8, n_layers=2, ffn_hidden=512), 4, True),
    # reference training default geometry (V/training/train.py:83-88,378-390), 2 layers
    "vit_ref_L2": ("vit", dict(in_channels=1, img_size_h=32, img_size_w=64, patch_size=4, num_classes=19,
                               d_model=128, n_head=8, n_layers=2, ffn_hidden=512), 2, False),
    # configs[1] ViT-Tiny/16 224x224 truncated to 2 layers
    "vit_tiny224_L2": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19,
                                   d_model=192, n_head=3, n_layers=2, ffn_hidden=768), 2, False),
    # multi-channel, non-square, dh=32
    "vit_c2_dh32": ("vit", dict(in_channels=2, img_size_h=16, img_size_w=48, patch_size=8, num_classes=7,
                                d_model=64, n_head=2, n_layers=1, ffn_hidden=96), 3, False),
    # R/test_model.py config
    "rawiq_R": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=11, d_model=128, n_head=8, n_layers=2,
                              ffn_hidden=512, use_cls_token=True, embedding_type="segment", segment_size=64), 4, True),
    # configs[2] rawIQ train.py defaults truncated to 2 layers
    "rawiq_C_L2": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=2,
                                 ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 2, False),
    # published best geometry (d256 h8 -> dh 32), 1 layer
    "rawiq_Cp_L1": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=256, n_head=8, n_layers=1,
                                  ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 2, False),
    # conv1d embedding: S = 1025
    "rawiq_conv1d": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=1,
                                   ffn_hidden=256, use_cls_token=True, embedding_type="conv1d", segment_size=64), 2, False),
    # mean pooling head (use_cls_token=False)
    "rawiq_nocls": ("rawiq", dict(in_channels=2, seq_length=512, num_classes=5, d_model=64, n_head=4, n_layers=1,
                                  ffn_hidden=128, use_cls_token=False, embedding_type="segment", segment_size=32), 3, False),
    # ---- full-depth cases of the benchmarked configurations (weights regenerated from the seed by the build's own
    #      init, whose bit-equality to the reference's is asserted below) ----------------------------
    # configs[1] ViT-Tiny/16 224x224 at its full depth of 12
    "vit_tiny224_L12": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19,
                                    d_model=192, n_head=3, n_layers=12, ffn_hidden=768), 2, False),
    # configs[3] ViT-Base/16 geometry (D768 / H12 / F3072), 2-layer truncation (341 MB of fp32 weights at L12)
    "vit_base_L2": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19,
                                d_model=768, n_head=12, n_layers=2, ffn_hidden=3072), 2, False),
    # configs[2] rawIQ train.py defaults at full depth 6
    "rawiq_C_L6": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=6,
                                 ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 2, False),
    # published best rawIQ geometry at full depth 9
    "rawiq_Cp_L9": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=256, n_head=8, n_layers=9,
                                  ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 2, False),
}

# parameter-count known answers (V/main.ipynb:694,758-773)
COUNTS = {
    "vit": [
        (dict(in_channels=1, img_size_h=32, img_size_w=64, patch_size=4, num_classes=19, d_model=256, n_head=16,
              n_layers=6, ffn_hidden=1024), 4748051),
        (dict(in_channels=1, img_size_h=32, img_size_w=32, patch_size=16, num_classes=11, d_model=128, n_head=8,
              n_layers=2, ffn_hidden=512), 430987),
        (dict(in_channels=1, img_size_h=32, img_size_w=64, patch_size=4, num_classes=19, d_model=128, n_head=8,
              n_layers=6, ffn_hidden=512), 1194387),
        (dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192, n_head=3,
              n_layers=12, ffn_hidden=768), 5391571),
        (dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=768, n_head=12,
              n_layers=12, ffn_hidden=3072), 85267219),
    ],
    "rawiq": [
        (dict(in_channels=2, seq_length=1024, num_classes=11, d_model=128, n_head=8, n_layers=2, ffn_hidden=512,
              use_cls_token=True, embedding_type="segment", segment_size=64), 414859),
        (dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=6, ffn_hidden=1024,
              use_cls_token=True, embedding_type="segment", segment_size=16), 1986195),
        (dict(in_channels=2, seq_length=1024, num_classes=19, d_model=256, n_head=8, n_layers=9, ffn_hidden=1024,
              use_cls_token=True, embedding_type="segment", segment_size=16), 7121939),
        (dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=6, ffn_hidden=512,
              use_cls_token=True, embedding_type="conv1d", segment_size=64), 1192851),
    ],
}

SEED = 1234
LR, WD, SMOOTH, CLIP = 1e-4, 1e-3, 0.1, 1.0


def digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def worker(tree, only=None):
    import typing
    sys.dont_write_bytecode = True
    if not hasattr(typing, "LiteralString"):
        typing.LiteralString = str
    sys.path.insert(0, os.path.join(REF, "ViT" if tree == "vit" else "transformer_rawIQ"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import iq_oracle as O
    if tree == "vit":
        from models.amc_transformer import AMCTransformer
    else:
        from models.transformer_rawIQ import AMCTransformer
    torch.set_num_threads(4)

    # ---- parameter-count known answers -------------------------------------------------
    counts = []
    for kw, expect in COUNTS[tree]:
        cfg = O.OracleConfig(kind=tree, drop_prob=0.0, **kw)
        ref = AMCTransformer(drop_prob=0.0, device="cpu", **kw)
        n_ref = sum(p.numel() for p in ref.parameters())
        n_or = O.count_parameters(O.init_state(cfg, 0))
        assert n_ref == expect == n_or, (kw, n_ref, expect, n_or)
        counts.append((kw, expect))
        del ref
    print(f"[{tree}] parameter counts OK: {[c[1] for c in counts]}")

    for name, (t, kw, B, full) in CASES.items():
        if t != tree or (only and name not in only):
            continue
        cfg = O.OracleConfig(kind=tree, drop_prob=0.0, **kw)
        # ---- init parity: same seed -> bit-identical parameters -------------------------
        torch.manual_seed(SEED)
        ref = AMCTransformer(drop_prob=0.0, device="cpu", **kw)
        sd_ref = {k: v.detach().clone() for k, v in ref.state_dict().items()}
        sd = O.init_state(cfg, SEED)
        assert set(sd) == set(sd_ref), (sorted(set(sd) ^ set(sd_ref)))
        for k in sd:
            assert sd[k].shape == sd_ref[k].shape, k
            if k.endswith("positional_encoding.encoding"):
                assert torch.allclose(sd[k], sd_ref[k], atol=1e-6, rtol=0), k
                pe_bit_equal = bool(torch.equal(sd[k], sd_ref[k]))
                sd[k] = sd_ref[k].clone()
            else:
                assert torch.equal(sd[k], sd_ref[k]), f"init differs: {k}"
        # ---- data ----------------------------------------------------------------------
        g = torch.Generator().manual_seed(SEED + 1)
        if tree == "vit":
            x = torch.randn(B, kw["in_channels"], kw["img_size_h"], kw["img_size_w"], generator=g)
        else:
            x = torch.randn(B, kw["in_channels"], kw["seq_length"], generator=g)
        y = torch.randint(0, kw["num_classes"], (B,), generator=g)
        # ---- reference: eval logits, then one training step (train mode, p=0) ------------
        ref.eval()
        with torch.no_grad():
            logits_ref = ref(x)
        ref.train()
        crit = torch.nn.CrossEntropyLoss(label_smoothing=SMOOTH)
        opt = torch.optim.AdamW(ref.parameters(), lr=LR, weight_decay=WD, betas=(0.9, 0.99))
        opt.zero_grad()
        out = ref(x)
        loss_ref = crit(out, y)
        loss_ref.backward()
        grads_ref = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
        gn_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=CLIP)
        opt.step()
        post_ref = {k: p.detach().clone() for k, p in ref.named_parameters()}
        # ---- oracle --------------------------------------------------------------------
        logits_or = O.model_forward(cfg, sd, x)
        _, loss_or, grads_or = O.loss_and_grads(cfg, sd, x, y, SMOOTH)
        # optimizer restatement is checked on the REFERENCE's gradients: the first AdamW step is
        # lr*g/(|g|+eps), so elements with |g| ~ eps (e.g. the K bias, whose true gradient is 0)
        # amplify 1e-10 gradient noise to ~lr and must not be compared through two backward passes.
        sd2 = {k: v.clone() for k, v in sd.items()}
        st = O.adamw_init(sd2)
        gn_or, coef = O.clip_coefficient([grads_ref[k] for k in O.param_keys(sd2)], CLIP)
        with torch.no_grad():
            O.adamw_update(sd2, {k: grads_ref[k] * coef for k in grads_ref}, st, lr=LR, weight_decay=WD)
        gn_or = float(gn_or)
        assert torch.allclose(logits_or, logits_ref, atol=1e-5, rtol=1e-5), (name, (logits_or - logits_ref).abs().max())
        assert abs(float(loss_or) - float(loss_ref.detach())) < 1e-6, name
        # the oracle sums squares in fp64; torch's clip_grad_norm_ reduces in fp32 (3.4e-5 off at ViT-Base's 14 M elements)
        assert abs(gn_or - float(gn_ref)) < 1e-4 * max(1.0, float(gn_ref)), (gn_or, float(gn_ref))
        worst = 0.0
        for k in grads_ref:
            d = (grads_or[k] - grads_ref[k]).abs().max().item()
            s = grads_ref[k].abs().max().item() + 1e-6
            worst = max(worst, d / s)
            assert torch.allclose(grads_or[k], grads_ref[k], atol=1e-6, rtol=1e-4), (name, k, d, s)
            assert torch.allclose(sd2[k], post_ref[k], atol=1e-7, rtol=1e-6), (name, k)
        print(f"[{tree}] {name}: oracle == reference (logits {float((logits_or-logits_ref).abs().max()):.2e}, "
              f"worst rel grad {worst:.2e}, |g| {float(gn_ref):.4f}, PE bit-equal {pe_bit_equal})")
        # ---- fixture -------------------------------------------------------------------
        keys = list(grads_ref)
        fx = {
            "cfg_kind": np.array(tree), "cfg_json": np.array(repr(kw)), "seed": np.array(SEED),
            "x": x.numpy(), "y": y.numpy(), "logits": logits_ref.numpy(), "loss": np.array(float(loss_ref.detach())),
            "grad_norm": np.array(float(gn_ref)), "keys": np.array(keys),
            "grad_l2": np.array([grads_ref[k].double().norm().item() for k in keys]),
            "grad_sum": np.array([grads_ref[k].double().sum().item() for k in keys]),
            "post_l2": np.array([post_ref[k].double().norm().item() for k in keys]),
            "post_sum": np.array([post_ref[k].double().sum().item() for k in keys]),
            "init_digest": np.array([digest(sd_ref[k]) for k in keys]),
            "hyper": np.array([LR, WD, SMOOTH, CLIP]),
            "n_params": np.array(sum(p.numel() for p in ref.parameters())),
            "pe": sd_ref["encoder.positional_encoding.encoding"].numpy(),
        }
        if full:
            for k, v in sd_ref.items():
                fx["w:" + k] = v.numpy()
            for k in keys:
                fx["g:" + k] = grads_ref[k].numpy()
        else:
            # a few gradients in full: embedding, first-layer q, last norm, head
            for k in keys:
                if kw["d_model"] > 256 and k.endswith("weight") and "mlp_head" not in k:
                    continue        # ViT-Base: keep the fixture small (norms of every gradient are stored above)
                if ("projection" in k or k.endswith("layers.0.attention.w_q.weight") or "mlp_head" in k
                        or k.endswith("cls_token") or k.endswith(f"layers.{kw['n_layers']-1}.norm2.gamma")):
                    fx["g:" + k] = grads_ref[k].numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    if tree == "vit" and (not only or "sublayers" in only):
        sublayer_pins(torch, O)
    if not only:
        np.savez_compressed(os.path.join(HERE, f"param_counts_{tree}.npz"),
                            kwargs=np.array([repr(c[0]) for c in counts]), counts=np.array([c[1] for c in counts]))


def sublayer_pins(torch, O):
    """The reference's layer modules run on their own (the files are byte-identical in both trees): EncoderLayer with and
    without a src_mask (V/models/blocks/encoder_layer.py:18-35 -> scale_dot_product_attention.py:30-31), MultiHeadAttention,
    PositionwiseFeedForward, LayerNorm.  Asserts the oracle's layer functions equal them and writes sublayers.npz."""
    from models.blocks.encoder_layer import EncoderLayer
    D, F, H, B, S = 64, 128, 2, 2, 10
    torch.manual_seed(SEED)
    layer = EncoderLayer(d_model=D, ffn_hidden=F, n_head=H, drop_prob=0.0)
    with torch.no_grad():                      # non-trivial affine parameters
        for n in ("norm1", "norm2"):
            getattr(layer, n).gamma.uniform_(0.5, 1.5)
            getattr(layer, n).beta.uniform_(-0.5, 0.5)
    layer.eval()
    g = torch.Generator().manual_seed(SEED + 7)
    x = torch.randn(B, S, D, generator=g)
    mask = (torch.rand(B, 1, S, S, generator=g) > 0.3).to(torch.int64)
    mask[:, :, 2, :] = 0                       # one fully masked query row
    sd = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    with torch.no_grad():
        out = layer(x, None)
        out_m = layer(x, mask)
        mha = layer.attention(q=x, k=x, v=x, mask=None)
        mha_m = layer.attention(q=x, k=x, v=x, mask=mask)
        ffn = layer.ffn(x)
        ln = layer.norm1(x)
        q = layer.attention.split(layer.attention.w_q(x))
        k = layer.attention.split(layer.attention.w_k(x))
        v = layer.attention.split(layer.attention.w_v(x))
        core, score = layer.attention.attention(q, k, v, mask=mask)
    checks = [(O.encoder_layer(sd, "", x, H), out), (O.encoder_layer(sd, "", x, H, mask=mask), out_m),
              (O.multi_head_attention(sd, "attention.", x, H), mha), (O.multi_head_attention(sd, "attention.", x, H, mask), mha_m),
              (O.feed_forward(sd, "ffn.", x), ffn), (O.custom_layer_norm(x, sd["norm1.gamma"], sd["norm1.beta"]), ln),
              (O.attention_core(q, k, v, mask), core)]
    for i, (a, b) in enumerate(checks):
        assert torch.allclose(a, b, atol=2e-6, rtol=1e-5), (i, (a - b).abs().max())
    fx = {"dims": np.array([D, F, H, B, S]), "x": x.numpy(), "mask": mask.numpy(), "out": out.numpy(), "out_masked": out_m.numpy(),
          "mha": mha.numpy(), "mha_masked": mha_m.numpy(), "ffn": ffn.numpy(), "ln": ln.numpy(), "core_masked": core.numpy(),
          "score_masked": score.numpy()}
    for k_, v_ in sd.items():
        fx["w:" + k_] = v_.numpy()
    np.savez_compressed(os.path.join(HERE, "sublayers.npz"), **fx)
    print("[vit] sublayers: oracle layer functions == reference EncoderLayer / MultiHeadAttention / FFN / LayerNorm (mask branch included)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", choices=["vit", "rawiq"])
    ap.add_argument("--only", default="", help="comma-separated case names (default: all; existing fixtures are kept)")
    a = ap.parse_args()
    only = [n for n in a.only.split(",") if n]
    if a.tree:
        worker(a.tree, only)
        return
    for tree in ("vit", "rawiq"):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--tree", tree] + (["--only", a.only] if only else []))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
