"""CPU: the oracle (oracle/iq_oracle.py) against the fixtures the REFERENCE produced
(tests/golden/make_golden.py).  This is the pin that lets the GPU tests trust the oracle."""
import ast
import hashlib
import os

import numpy as np
import pytest
import torch

import iq_oracle as O
from conftest import GOLDEN, golden_names, load_golden


def _digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()[:16]


def _state_for(name):
    kind, kw, z = load_golden(name)
    cfg = O.OracleConfig(kind=kind, drop_prob=0.0, **kw)
    sd = O.init_state(cfg, int(z["seed"]))
    return cfg, sd, z


@pytest.mark.parametrize("name", golden_names())
def test_init_is_the_references_init(name):
    cfg, sd, z = _state_for(name)
    keys = [str(k) for k in z["keys"]]
    assert sorted(keys) == sorted(O.param_keys(sd))
    for k, d in zip(keys, z["init_digest"]):
        assert _digest(sd[k]) == str(d), k
    assert O.count_parameters(sd) == int(z["n_params"])
    assert np.array_equal(sd["encoder.positional_encoding.encoding"].numpy(), z["pe"])
    for k in z.files:
        if k.startswith("w:"):
            assert np.array_equal(sd[k[2:]].numpy(), z[k]), k


@pytest.mark.parametrize("name", golden_names())
def test_forward_loss_grads_match_reference(name):
    cfg, sd, z = _state_for(name)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    lr, wd, smooth, clip = [float(v) for v in z["hyper"]]
    logits, loss, grads = O.loss_and_grads(cfg, sd, x, y, smooth)
    np.testing.assert_allclose(logits.numpy(), z["logits"], atol=1e-5, rtol=1e-5)
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    keys = [str(k) for k in z["keys"]]
    l2 = np.array([grads[k].double().norm().item() for k in keys])
    np.testing.assert_allclose(l2, z["grad_l2"], rtol=2e-4, atol=1e-7)
    total, _ = O.clip_coefficient([grads[k] for k in keys], clip)
    assert abs(float(total) - float(z["grad_norm"])) < 1e-4 * float(z["grad_norm"])
    for k in z.files:
        if k.startswith("g:"):
            np.testing.assert_allclose(grads[k[2:]].numpy(), z[k], atol=1e-6, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("name", ["vit_A", "rawiq_R"])
def test_clip_adamw_step_matches_reference(name):
    """Optimizer restatement on the reference's own gradients (full-gradient fixtures)."""
    cfg, sd, z = _state_for(name)
    lr, wd, smooth, clip = [float(v) for v in z["hyper"]]
    keys = [str(k) for k in z["keys"]]
    grads = {k: torch.from_numpy(z["g:" + k]) for k in keys}
    total, coef = O.clip_coefficient([grads[k] for k in keys], clip)
    st = O.adamw_init(sd)
    with torch.no_grad():
        O.adamw_update(sd, {k: g * coef for k, g in grads.items()}, st, lr=lr, weight_decay=wd)
    l2 = np.array([sd[k].double().norm().item() for k in keys])
    sm = np.array([sd[k].double().sum().item() for k in keys])
    np.testing.assert_allclose(l2, z["post_l2"], rtol=1e-6)
    np.testing.assert_allclose(sm, z["post_sum"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tree", ["vit", "rawiq"])
def test_parameter_count_known_answers(tree):
    z = np.load(os.path.join(GOLDEN, f"param_counts_{tree}.npz"), allow_pickle=False)
    for kw, n in zip(z["kwargs"], z["counts"]):
        kw = ast.literal_eval(str(kw))
        if int(n) > 20_000_000:
            # ViT-Base: closed form instead of allocating 341 MB
            D, F, L, K = kw["d_model"], kw["ffn_hidden"], kw["n_layers"], kw["num_classes"]
            P = kw["in_channels"] * kw["patch_size"] ** 2
            got = P * D + D + D + L * (4 * (D * D + D) + 2 * (D * F) + F + D + 4 * D) + D * K + K
        else:
            got = O.count_parameters(O.init_state(O.OracleConfig(kind=tree, drop_prob=0.0, **kw), 0))
        assert got == int(n), kw


def test_reference_error_conventions():
    with pytest.raises(ValueError, match="must be divisible by segment_size"):
        O.init_state(O.OracleConfig(kind="rawiq", seq_length=1000, segment_size=64, in_channels=2), 0)
    with pytest.raises(ValueError, match="Unknown embedding_type"):
        O.init_state(O.OracleConfig(kind="rawiq", embedding_type="patch", in_channels=2), 0)


def test_smoothed_ce_equals_torch():
    g = torch.Generator().manual_seed(0)
    lg = torch.randn(16, 19, generator=g)
    y = torch.randint(0, 19, (16,), generator=g)
    a = O.smoothed_cross_entropy(lg, y, 0.1)
    b = torch.nn.functional.cross_entropy(lg, y, label_smoothing=0.1)
    assert abs(float(a) - float(b)) < 1e-6


def test_dropout_train_mode_statistics():
    cfg = O.OracleConfig(kind="vit", drop_prob=0.5)
    x = torch.ones(200, 200)
    y = O._dropout(x, 0.5, True)
    assert abs(float((y == 0).float().mean()) - 0.5) < 0.02
    assert abs(float(y.mean()) - 1.0) < 0.05


def test_oracle_layer_functions_match_the_reference_layer_outputs():
    """tests/golden/sublayers.npz holds the reference's own EncoderLayer / MultiHeadAttention / FFN / LayerNorm outputs
    (mask branch included, scale_dot_product_attention.py:30-31); the oracle's layer-level functions must reproduce them."""
    import os
    import numpy as np
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "sublayers.npz"), allow_pickle=False)
    D, F, H, B, S = [int(v) for v in z["dims"]]
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    x = torch.from_numpy(z["x"])
    mask = torch.from_numpy(z["mask"])
    pairs = {"out": O.encoder_layer(sd, "", x, H), "out_masked": O.encoder_layer(sd, "", x, H, mask=mask),
             "mha": O.multi_head_attention(sd, "attention.", x, H), "mha_masked": O.multi_head_attention(sd, "attention.", x, H, mask),
             "ffn": O.feed_forward(sd, "ffn.", x), "ln": O.custom_layer_norm(x, sd["norm1.gamma"], sd["norm1.beta"])}
    for name, got in pairs.items():
        assert torch.allclose(got, torch.from_numpy(z[name]), atol=2e-6, rtol=1e-5), name
