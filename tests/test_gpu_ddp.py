"""GPU (MI355X, one card): two ranks (gloo, both on cuda:0) run the data-parallel fused step on half batches;
the result must equal one rank stepping on the whole batch (gradient mean over the global batch, identical clip
scale on every rank).  RCCL refuses two ranks on one device, so gloo carries the collective here; the bucket /
stage logic under test is backend independent (bench.py uses nccl = RCCL on a multi-GPU node)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
    import torch, torch.distributed as dist
    import iq_oracle as O
    import vit_vs_raw_iq_amd as P
    from vit_vs_raw_iq_amd.trainer import FusedTrainer
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    d = torch.device("cuda:0")
    kw = dict(in_channels=2, seq_length=256, num_classes=5, d_model=64, n_head=4, n_layers=3, ffn_hidden=128,
              use_cls_token=True, embedding_type="segment", segment_size=16)
    sd = O.init_state(O.OracleConfig(kind="rawiq", drop_prob=0.0, **kw), 3)
    m = P.AMCTransformerRawIQ(drop_prob=0.0, device="cuda", **kw)
    m.load_state_dict(sd); m.to(d).train()
    tr = FusedTrainer(m, lr=1e-3, weight_decay=1e-2, n_buckets=3)
    assert tr.world == world and (world == 1 or len(tr.buckets) == 3)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(32, 2, 256, generator=g); y = torch.randint(0, 5, (32,), generator=g)
    per = 32 // world
    for _ in range(4):
        tr.step(x[rank * per:(rank + 1) * per].to(d), y[rank * per:(rank + 1) * per].to(d))
    loss, acc, frames = tr.read_stats()
    assert frames == 4 * 32, frames
    torch.save({"loss": loss, "acc": acc, "sd": {k: v.cpu() for k, v in m.state_dict().items()}}, sys.argv[1] + f".w{world}.r{rank}")
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    print("ok", world, rank)
""") % (ROOT, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(script, out, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(out)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, f"world {world} rank {r} failed:\n{o}"


def test_two_ranks_equal_one_rank_on_the_global_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    out = tmp_path / "res"
    _run(script, out, 1)
    _run(script, out, 2)
    one = torch.load(str(out) + ".w1.r0", weights_only=True)
    r0 = torch.load(str(out) + ".w2.r0", weights_only=True)
    r1 = torch.load(str(out) + ".w2.r1", weights_only=True)
    assert abs(one["loss"] - r0["loss"]) < 2e-3 and abs(r0["loss"] - r1["loss"]) < 1e-9
    for k in one["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"ranks diverged on {k}"
        # bf16 activations: half-batch row tiles are rounded identically, only the fp32 summation order of the
        # weight-gradient slabs differs -> parameters agree to a small fraction of the 4e-3 total movement
        # AdamW moves an element by ~lr per step whatever its gradient's size, so elements whose true gradient
        # is ~0 (the K bias: softmax is shift invariant; weights of dead ReLU units) turn 1e-9 summation-order
        # noise into +-lr steps of random sign.  Robust statement: such elements are rare and bounded by
        # steps*lr; everything else agrees to a small fraction of the movement.
        diff = (one["sd"][k] - r0["sd"][k]).abs()
        assert diff.max().item() <= 4.5e-3, k
        if not k.endswith("w_k.bias"):
            assert (diff > 1e-4).float().mean().item() < 0.02, (k, (diff > 1e-4).float().mean().item())
