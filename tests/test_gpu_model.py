"""GPU (MI355X): the nn.Module surface over the native plan against the golden fixtures the reference
produced (tests/golden/*.npz) and against the CPU oracle (oracle/iq_oracle.py, pinned to the reference by
tests/test_oracle_golden.py).

Stated floating-point tolerance of the bf16 path (BASELINE.json north_star "within a stated fp tolerance"):
  logits ........ |err| <= 3e-2 absolute (logit range here ~ +-1.5) and identical argmax
  loss .......... |err| <= 1e-2
  gradients ..... per parameter, ||g - g_ref|| <= 5% of ||g_ref|| (12% for ffn.linear1, see below) (+ 2e-3 of the
                  global gradient norm for parameters whose true gradient is ~0, e.g. the K bias), global norm
                  within 3%.  The limits are ~2x the measured errors, so a regression in any one kernel shows.
                  Measured: 1.5-3% everywhere except ffn.linear1 (4-8%): the ReLU mask is taken from the bf16
                  hidden activation, and the ~0.3% of pre-activations with |pre| below the bf16 forward error
                  flip sign relative to the fp32 reference; a flipped unit changes its gradient entry by 100%,
                  i.e. a relative L2 error of sqrt(flip rate).  Inherent to any bf16 forward, not to the kernels
                  (tests/test_gpu_kernels.py checks each kernel to bf16-ulp level on identical inputs).
The reference itself is fp32; bf16 activations with fp32 accumulation give ~1e-2 relative error
end to end (SURVEY.md section 7 measured 7e-3 for autocast-bf16 on the same model).
"""
import math

import numpy as np
import pytest
import torch

import iq_oracle as O
from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 3e-2
LOSS_ATOL = 1e-2
GRAD_REL = 5e-2            # every parameter class but ffn.linear1
GRAD_REL_FFN1 = 12e-2      # ffn.linear1.{weight,bias}: ReLU-mask sign flips of a bf16 forward (docstring)
GRAD_ABS_OF_TOTAL = 2e-3
# deep stacks: per-layer bf16 rounding accumulates through 12 / 9 post-norm layers (logit scale ~1.5)
DEEP = {"vit_tiny224_L12": 5e-2, "rawiq_Cp_L9": 5e-2, "rawiq_C_L6": 4e-2, "vit_base_L2": 4e-2}


def grad_rel(key):
    return GRAD_REL_FFN1 if "ffn.linear1" in key else GRAD_REL


def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def build(kind, kw, drop_prob=0.0):
    import vit_vs_raw_iq_amd as P
    cls = P.AMCTransformerViT if kind == "vit" else P.AMCTransformerRawIQ
    return cls(drop_prob=drop_prob, device="cuda", **kw)


def oracle_state(kind, kw, z):
    cfg = O.OracleConfig(kind=kind, drop_prob=0.0, **kw)
    return cfg, O.init_state(cfg, int(z["seed"]))


@pytest.mark.parametrize("name", golden_names())
def test_state_dict_layout_and_seeded_init(name):
    """Same keys/shapes as the reference and, under the same torch seed, bit-identical initial values."""
    kind, kw, z = load_golden(name)
    torch.manual_seed(int(z["seed"]))
    m = build(kind, kw)
    cfg, sd = oracle_state(kind, kw, z)
    msd = m.state_dict()
    assert set(msd) == set(sd)
    for k in sd:
        assert tuple(msd[k].shape) == tuple(sd[k].shape), k
        assert torch.equal(msd[k], sd[k]), k
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])


@pytest.mark.parametrize("name", golden_names())
def test_logits_loss_and_grads_match_reference(name):
    d = dev()
    kind, kw, z = load_golden(name)
    cfg, sd = oracle_state(kind, kw, z)
    m = build(kind, kw)
    m.load_state_dict(sd)
    m.to(d)
    x = torch.from_numpy(z["x"]).to(d)
    y = torch.from_numpy(z["y"]).to(d)
    # ---- eval forward vs the reference's logits -------------------------------------------------
    m.eval()
    with torch.no_grad():
        logits = m(x)
    ref = torch.from_numpy(z["logits"])
    err = (logits.cpu() - ref).abs().max().item()
    assert err <= DEEP.get(name, LOGIT_ATOL), f"{name}: logits max err {err:.4g}"
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * DEEP.get(name, LOGIT_ATOL)      # argmax must agree wherever it is not a near tie
    assert torch.equal(logits.cpu().argmax(1)[clear], ref.argmax(1)[clear])
    # state survived the re-homing into the flat buffer
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    # ---- training-mode step (drop_prob 0), torch loss + autograd through the native backward ----
    m.train()
    out = m(x)
    assert (out - logits).abs().max().item() == 0.0      # p = 0: train == eval, deterministic
    loss = torch.nn.functional.cross_entropy(out, y, label_smoothing=float(z["hyper"][2]))
    assert abs(loss.item() - float(z["loss"])) <= LOSS_ATOL
    loss.backward()
    _, _, gref = O.loss_and_grads(cfg, sd, torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), float(z["hyper"][2]))
    total_ref = float(z["grad_norm"])
    total = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()))
    assert abs(total - total_ref) <= 0.03 * total_ref, (total, total_ref)
    worst = (0.0, None)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        g, r = p.grad.cpu().double(), gref[k].double()
        e = (g - r).norm().item()
        lim = grad_rel(k) * r.norm().item() + GRAD_ABS_OF_TOTAL * total_ref
        if e / lim > worst[0]:
            worst = (e / lim, k)
        assert e <= lim, f"{name}: grad {k}: ||err|| {e:.4g} > {lim:.4g} (||ref|| {r.norm().item():.4g})"
    # golden full-precision gradient vectors the fixture carries (reference's own numbers)
    for key in z.files:
        if key.startswith("g:"):
            r = torch.from_numpy(z[key]).double()
            g = dict(m.named_parameters())[key[2:]].grad.cpu().double()
            assert (g - r).norm().item() <= grad_rel(key) * r.norm().item() + GRAD_ABS_OF_TOTAL * total_ref, key
    # per-parameter gradient norms the REFERENCE produced (every fixture carries them, full depth included)
    keys = [str(k) for k in z["keys"]]
    named = dict(m.named_parameters())
    for k, l2 in zip(keys, z["grad_l2"]):
        got = named[k].grad.double().norm().item()
        assert abs(got - float(l2)) <= grad_rel(k) * float(l2) + GRAD_ABS_OF_TOTAL * total_ref, (k, got, float(l2))


def test_reference_smoke_shapes():
    """V/test_model.py:55 and R/test_model.py:91-92,110-114: (4,11) and batch sizes 1, 8, 16."""
    d = dev()
    import vit_vs_raw_iq_amd as P
    v = P.AMCTransformerViT(in_channels=1, img_size_h=32, img_size_w=32, patch_size=16, num_classes=11, d_model=128,
                            n_head=8, n_layers=2, ffn_hidden=512, drop_prob=0.1, device="cuda").to(d)
    assert v(torch.randn(4, 1, 32, 32, device=d)).shape == (4, 11)
    r = P.AMCTransformerRawIQ(in_channels=2, seq_length=1024, num_classes=11, d_model=128, n_head=8, n_layers=2,
                              ffn_hidden=512, drop_prob=0.1, device="cuda", use_cls_token=True,
                              embedding_type="segment", segment_size=64).to(d)
    r.eval()
    with torch.no_grad():
        assert r(torch.randn(4, 2, 1024, device=d)).shape == (4, 11)
        for b in (1, 8, 16):
            assert r(torch.randn(b, 2, 1024, device=d)).shape == (b, 11)


def test_encoder_surface_and_errors():
    d = dev()
    import vit_vs_raw_iq_amd as P
    kind, kw, z = load_golden("rawiq_R")
    cfg, sd = oracle_state(kind, kw, z)
    m = build(kind, kw)
    m.load_state_dict(sd)
    m.to(d).eval()
    x = torch.from_numpy(z["x"]).to(d)
    with torch.no_grad():
        enc = m.encoder(x)
        ref = O.encoder_forward(cfg, sd, torch.from_numpy(z["x"]))
        assert enc.shape == ref.shape
        assert (enc.cpu() - ref).abs().max().item() <= 6e-2          # LayerNorm outputs, |x| up to ~4
        assert torch.equal(m.encoder.get_cls_token_output(x), m.encoder(x)[:, 0, :])
        assert m.encoder.get_sequence_output(x).shape == (4, 16, 128)
        # the full model still works after the encoder borrowed the parameters
        assert (m(x).cpu() - torch.from_numpy(z["logits"])).abs().max().item() <= LOGIT_ATOL
    with pytest.raises(ValueError, match="must be divisible by segment_size"):
        P.AMCTransformerRawIQ(in_channels=2, seq_length=1000, num_classes=3, d_model=64, n_head=4, n_layers=1,
                              ffn_hidden=64, drop_prob=0.0, device="cuda", segment_size=64)
    with pytest.raises(ValueError, match="Unknown embedding_type"):
        P.AMCTransformerRawIQ(in_channels=2, seq_length=1024, num_classes=3, d_model=64, n_head=4, n_layers=1,
                              ffn_hidden=64, drop_prob=0.0, device="cuda", embedding_type="patch")
    nc = P.EncoderRawIQ(in_channels=2, seq_length=256, d_model=64, ffn_hidden=64, n_head=4, n_layers=1, drop_prob=0.0,
                        device="cuda", use_cls_token=False, embedding_type="segment", segment_size=32)
    with pytest.raises(ValueError, match="CLS token is not enabled"):
        nc.get_cls_token_output(torch.zeros(1, 2, 256, device=d))
    with pytest.raises(P.IqError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 1024))
    # sub-layers run on their own (per-op C ABI): tests/test_gpu_sublayers.py
    assert m.encoder.layers[0].norm1(torch.zeros(1, 1, 128, device=d)).shape == (1, 1, 128)
    # model and encoder share ONE plan / ONE flat parameter buffer: the encoder call above did not re-home anything
    assert m.native_plan().is_bound(d) and m.encoder._plan is None


def test_encoder_backward_from_sequence_output():
    """Gradient entering at the encoder output (not the logits) reaches every encoder parameter."""
    d = dev()
    kind, kw, z = load_golden("rawiq_nocls")
    cfg, sd = oracle_state(kind, kw, z)
    m = build(kind, kw)
    m.load_state_dict(sd)
    m.to(d).train()
    x = torch.from_numpy(z["x"]).to(d)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(3, 16, 64, generator=g)
    enc = m.encoder(x)
    (enc * w.to(d)).sum().backward()
    leaf = {k: (v.clone().requires_grad_(True) if k not in O.BUFFER_KEYS else v) for k, v in sd.items()}
    (O.encoder_forward(cfg, leaf, torch.from_numpy(z["x"])) * w).sum().backward()
    for k, p in m.encoder.named_parameters():
        r = leaf["encoder." + k].grad.double()
        e = (p.grad.cpu().double() - r).norm().item()
        assert e <= grad_rel(k) * r.norm().item() + 0.05, (k, e, r.norm().item())


def test_dropout_training_mode():
    """drop_prob > 0: same (seed, step) -> same output; different steps differ; eval is deterministic;
    mean loss stays close to the no-dropout loss (inverted dropout keeps expectations)."""
    d = dev()
    kind, kw, z = load_golden("vit_A")
    cfg, sd = oracle_state(kind, kw, z)
    m = build(kind, kw, drop_prob=0.3)
    m.load_state_dict(sd)
    m.to(d)
    x = torch.from_numpy(z["x"]).to(d)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x), m(x)
    assert torch.equal(e1, e2)
    assert (e1.cpu() - torch.from_numpy(z["logits"])).abs().max().item() <= LOGIT_ATOL
    m.train()
    with torch.no_grad():
        t1 = m(x)
        t2 = m(x)
        plan = m.native_plan()
        plan.step -= 1
        t2b = m(x)
    assert not torch.equal(t1, t2)
    assert torch.equal(t2, t2b)
    # backward with dropout on: masks regenerated consistently -> finite, non-zero grads everywhere
    out = m(x)
    torch.nn.functional.cross_entropy(out, torch.from_numpy(z["y"]).to(d)).backward()
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k
    assert m.encoder.layers[0].ffn.linear1.weight.grad.abs().sum().item() > 0


@pytest.mark.parametrize("pdrop", [0.0, 0.2])
def test_gradient_is_exact_for_the_sampled_mask(pdrop):
    """With dropout ON, loss.backward() must be the gradient of the SAME masked network: central difference
    of the loss along g at fixed (seed, step) must equal 2*eps*|g|^2 to first order.  p=0 is the control."""
    d = dev()
    kind, kw, z = load_golden("rawiq_C_L2")
    cfg, sd = oracle_state(kind, kw, z)
    m = build(kind, kw, drop_prob=pdrop)
    m.load_state_dict(sd)
    m.to(d).train()
    x = torch.from_numpy(z["x"]).to(d)
    y = torch.from_numpy(z["y"]).to(d)
    plan = m.native_plan()
    loss0 = torch.nn.functional.cross_entropy(m(x), y)
    loss0.backward()
    step_used = plan.step
    grads = [p.grad.clone() for p in m.parameters()]
    gnorm2 = sum(float(g.double().pow(2).sum()) for g in grads)
    eps = 4e-3 / math.sqrt(gnorm2)

    def loss_at(alpha):
        with torch.no_grad():
            for p, g in zip(m.parameters(), grads):
                p.add_(g, alpha=alpha)
            plan.step = step_used - 1            # replay the same dropout masks
            val = torch.nn.functional.cross_entropy(m(x), y).item()
            for p, g in zip(m.parameters(), grads):
                p.add_(g, alpha=-alpha)
        return val

    got = loss_at(eps) - loss_at(-eps)
    pred = 2 * eps * gnorm2
    assert abs(got - pred) <= 0.12 * abs(pred) + 2e-3, (got, pred)


def test_torch_optimizer_loop_reduces_loss():
    """The reference's own loop shape (V/training/train.py:191-201): zero_grad, forward, CE(label_smoothing),
    backward, clip_grad_norm_, AdamW.step -- unchanged torch components around the native model."""
    d = dev()
    import vit_vs_raw_iq_amd as P
    torch.manual_seed(0)
    m = P.AMCTransformerRawIQ(in_channels=2, seq_length=256, num_classes=4, d_model=64, n_head=4, n_layers=2,
                              ffn_hidden=128, drop_prob=0.1, device="cuda", segment_size=16).to(d)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3, weight_decay=1e-3, betas=(0.9, 0.99))
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    g = torch.Generator().manual_seed(1)
    y = torch.randint(0, 4, (64,), generator=g)
    x = torch.randn(64, 2, 256, generator=g) * 0.3
    x[:, 0, :] += (y.float().view(-1, 1) - 1.5)          # class-dependent offset: learnable quickly
    x, y = x.to(d), y.to(d)
    m.train()
    first = last = None
    for it in range(60):
        opt.zero_grad()
        loss = crit(m(x), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
        opt.step()
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.7 * first, (first, last)
    m.eval()
    with torch.no_grad():
        acc = (m(x).argmax(1) == y).float().mean().item()
    assert acc > 0.9, acc


def test_full_size_properties_vit_tiny():
    """BASELINE configs[1] geometry (ViT-Tiny/16, 224x224, S=197) at a batch the oracle cannot do in seconds:
    size-independent properties -- frames are independent (batch permutation equivariance, bit exact),
    and a 2-layer slice of the same batch agrees with the oracle on a few frames."""
    d = dev()
    import vit_vs_raw_iq_amd as P
    kw = dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192, n_head=3,
              n_layers=12, ffn_hidden=768)
    torch.manual_seed(3)
    m = P.AMCTransformerViT(drop_prob=0.0, device="cuda", **kw).to(d).eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(96, 1, 224, 224, generator=g).to(d)
    perm = torch.randperm(96, generator=g).to(d)
    with torch.no_grad():
        a = m(x)
        b = m(x[perm])
    assert torch.isfinite(a).all()
    assert torch.equal(a[perm], b)
    cfg = O.OracleConfig(kind="vit", drop_prob=0.0, **kw)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref = O.model_forward(cfg, sd, x[:3].cpu())
    assert (a[:3].cpu() - ref).abs().max().item() <= 5e-2          # 12 layers deep
    assert torch.equal(a[:3].cpu().argmax(1), ref.argmax(1))


FULL = {
    "B": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192,
                      n_head=3, n_layers=12, ffn_hidden=768), 256),
    "C": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=6,
                        ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 256),
    "Cp": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=256, n_head=8, n_layers=9,
                         ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 128),
    "D_L2": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=768,
                         n_head=12, n_layers=2, ffn_hidden=3072), 256),   # 256 x 197 rows = 197 whole 256-row blocks: the full
                                                                          # batch runs gemm_big / wgrad_big, its halves the tiled kernels
    # BASELINE configs[3]'s per-GPU share at FULL depth and batch: ViT-Base L12 x 512 frames (global 4096 / 8 GPUs) -- the
    # 12-layer grouped wgrad_big launches, gemm_big on every Linear, the 23 GB workspace plan, LayerNorm at D = 768
    "D": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=768,
                      n_head=12, n_layers=12, ffn_hidden=3072), 512),
}


@pytest.mark.parametrize("cid", sorted(FULL))
def test_full_batch_training_step_properties(cid):
    """The benchmarked configurations at their benchmarked batch (BASELINE.json configs[1..3]; ViT-Base's per-GPU share
    at 2 layers x 256 frames and at its full 12 layers x 512 frames) -- size-independent properties of one whole training
    step (the gradient itself is compared with the oracle at these batches in
    test_benchmarked_batch_gradient_matches_oracle):
      * every gradient is finite and the gradient norm is positive;
      * the flat gradient does not depend on the order of the frames in the batch (frames are independent; the
        weight-gradient sums run in a different order, so equality is to fp32 summation error, not bit exact);
      * the gradient of a batch is the mean of the gradients of its halves (linearity of the mean loss);
      * hipGraph replay of the step lands on the same parameters as eager launches, bit for bit (dropout ON)."""
    from vit_vs_raw_iq_amd.trainer import FusedTrainer
    d = dev()
    kind, kw, B = FULL[cid]
    torch.manual_seed(11)
    m = build(kind, kw).to(d).train()
    g = torch.Generator().manual_seed(12)
    shape = (B, kw["in_channels"], kw["img_size_h"], kw["img_size_w"]) if kind == "vit" else (B, kw["in_channels"], kw["seq_length"])
    x = torch.randn(*shape, generator=g).to(d)
    y = torch.randint(0, kw["num_classes"], (B,), generator=g).to(d)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)

    def flat_grad(xb, yb):
        for p in m.parameters():
            p.grad = None
        crit(m(xb), yb).backward()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double()

    g_all = flat_grad(x, y)
    assert torch.isfinite(g_all).all() and g_all.norm().item() > 0
    perm = torch.randperm(B, generator=g).to(d)
    g_perm = flat_grad(x[perm], y[perm])
    rel = ((g_all - g_perm).norm() / g_all.norm()).item()
    assert rel < 2e-3, f"{cid}: gradient depends on the frame order ({rel:.3g})"
    # cfg B: the batch and its halves must take the same kernels for this to be a statement about linearity -- at 256 frames
    # (50,432 rows) the feed-forward sub-layer runs as ONE launch (ffn_chain.hip, M > 32,768 rows), at 128 frames as two, and
    # the two agree only up to rounding ties (the chain feeds the MFMAs their k-values in another order), which 12 post-norm
    # layers amplify to ~1 % of the gradient (measured 1.4e-2; one layer: 2.8e-5).  So: 512 frames against its two halves of
    # 256 -- the benchmarked batch itself.
    if cid == "B":
        x2 = torch.cat([x, torch.randn(*shape, generator=g).to(d)])
        y2 = torch.cat([y, torch.randint(0, kw["num_classes"], (B,), generator=g).to(d)])
        g_full = flat_grad(x2, y2)
        g_half = 0.5 * (g_all + flat_grad(x2[B:], y2[B:]))
    else:
        h = B // 2
        g_full = g_all
        g_half = 0.5 * (flat_grad(x[:h], y[:h]) + flat_grad(x[h:], y[h:]))
    rel = ((g_full - g_half).norm() / g_full.norm()).item()
    assert rel < 2e-3, f"{cid}: batch gradient != mean of half-batch gradients ({rel:.3g})"
    # graph == eager over three fused steps, dropout ON (device-side step counter, regenerated masks)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    outs = []
    for use_graph in (False, True):
        mm = build(kind, kw, drop_prob=0.1)
        mm.load_state_dict(sd0)
        mm.to(d).train()
        tr = FusedTrainer(mm, lr=1e-3, weight_decay=1e-3, use_graph=use_graph, dropout_seed=21)
        for _ in range(3):
            tr.step(x, y)
        loss, _, frames = tr.read_stats()
        assert frames == 3 * B and math.isfinite(loss)
        outs.append({k: v.detach().clone() for k, v in mm.state_dict().items()})
        del tr, mm
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{cid}: graph replay diverged from eager on {k}"


@pytest.mark.parametrize("cid", ["B", "C"])
def test_benchmarked_batch_gradient_matches_oracle(cid):
    """BASELINE configs[1] / configs[2] at the batch bench.py times (256 frames per GPU), dropout off: logits, loss and
    EVERY per-parameter gradient against the CPU oracle's O.loss_and_grads on the same seeded weights and frames, at the
    limits of the small-batch fixtures.  The batch-2 fixtures never reach ragged row tiles (256 x 197 = 394 x 128 rows),
    the band variant of the GEMM + LayerNorm kernel (one frame per CU), the M-splits of the weight-gradient tiles or tiles
    that straddle frames; this does.  (The oracle needs ~2 s for cfg C and ~8 s for cfg B on the box's host cores.)"""
    d = dev()
    kind, kw, B = FULL[cid]
    cfg = O.OracleConfig(kind=kind, drop_prob=0.0, **kw)
    sd = O.init_state(cfg, 5)
    m = build(kind, kw)
    m.load_state_dict(sd)
    m.to(d).train()
    g = torch.Generator().manual_seed(6)
    shape = (B, kw["in_channels"], kw["img_size_h"], kw["img_size_w"]) if kind == "vit" else (B, kw["in_channels"], kw["seq_length"])
    x = torch.randn(*shape, generator=g)
    y = torch.randint(0, kw["num_classes"], (B,), generator=g)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref_logits, ref_loss, gref = O.loss_and_grads(cfg, sd, x, y, 0.1)
    out = m(x.to(d))
    loss = torch.nn.functional.cross_entropy(out, y.to(d), label_smoothing=0.1)
    loss.backward()
    lim_logit = {"B": 5e-2, "C": 4e-2}[cid]                       # the DEEP limits of the same geometries' fixtures
    err = (out.detach().cpu() - ref_logits).abs().max().item()
    assert err <= lim_logit, f"{cid}@{B}: logits max err {err:.4g}"
    assert abs(loss.item() - float(ref_loss)) <= LOSS_ATOL
    total_ref = math.sqrt(sum(float(v.double().pow(2).sum()) for v in gref.values()))
    total = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()))
    assert abs(total - total_ref) <= 0.03 * total_ref, (total, total_ref)
    worst = (0.0, None)
    for k, p in m.named_parameters():
        r = gref[k].double()
        e = (p.grad.cpu().double() - r).norm().item()
        lim = grad_rel(k) * r.norm().item() + GRAD_ABS_OF_TOTAL * total_ref
        worst = max(worst, (e / lim, k))
        assert e <= lim, f"{cid}@{B}: grad {k}: ||err|| {e:.4g} > {lim:.4g} (||ref|| {r.norm().item():.4g})"
    print(f"{cid}@{B}: logits err {err:.3g}, worst gradient at {worst[0]:.2f} of its limit ({worst[1]})")


def test_backward_beyond_65536_rows():
    """D = 192 with B*S > 65536 rows (cfg B geometry, 384 frames x 197 tokens = 75,648): the fused data-gradient GEMM +
    LayerNorm backward writes one gamma/beta partial row per 128-row block, more than the stand-alone kernel's 512-block
    cap -- the plan must size the partial-row scratch for it.  The batch gradient equals the mean of its halves (halves,
    not thirds: a power-of-two loss scale keeps every bf16 rounding of the activation gradients identical)."""
    d = dev()
    kind, kw, _ = FULL["B"]
    kw = dict(kw, n_layers=2)
    torch.manual_seed(31)
    m = build(kind, kw).to(d).train()
    g = torch.Generator().manual_seed(32)
    B = 384
    x = torch.randn(B, 1, 224, 224, generator=g).to(d)
    y = torch.randint(0, kw["num_classes"], (B,), generator=g).to(d)

    def flat_grad(xb, yb):
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(m(xb), yb, label_smoothing=0.1).backward()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double()

    g_all = flat_grad(x, y)
    assert torch.isfinite(g_all).all() and g_all.norm().item() > 0
    t = B // 2
    g_halves = (flat_grad(x[:t], y[:t]) + flat_grad(x[t:], y[t:])) / 2
    rel = ((g_all - g_halves).norm() / g_all.norm()).item()
    assert rel < 2e-3, rel
    gam = m.encoder.layers[0].norm1.gamma.grad
    assert torch.isfinite(gam).all() and gam.abs().sum().item() > 0
