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
    use_graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
    if rank == 1:                      # a rank that starts from different weights: the start-up broadcast must repair it
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.01)
    tr = FusedTrainer(m, lr=1e-3, weight_decay=1e-2, n_buckets=3, use_graph=use_graph)
    assert tr.world == world and (world == 1 or len(tr.buckets) == 3)
    if use_graph:
        assert tr.use_graph            # hipGraph replay stays on under data parallelism (one graph per bucket segment)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(32, 2, 256, generator=g); y = torch.randint(0, 5, (32,), generator=g)
    per = 32 // world
    for _ in range(4):
        tr.step(x[rank * per:(rank + 1) * per].to(d), y[rank * per:(rank + 1) * per].to(d))
    loss, acc, frames = tr.read_stats()
    assert frames == 4 * 32, frames
    if use_graph:
        assert tr._graphs is not None and len(tr._graphs) == len(tr.buckets) + 1
    torch.save({"loss": loss, "acc": acc, "sd": {k: v.cpu() for k, v in m.state_dict().items()}},
               sys.argv[1] + f".w{world}.r{rank}" + (".graph" if use_graph else ""))
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


def _run(script, out, world, *extra):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(out), *extra], env=env, stdout=subprocess.PIPE,
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


def test_two_ranks_with_graph_replay_equal_two_ranks_eager(tmp_path):
    """hipGraph replay under data parallelism: the step is captured as one graph per bucket segment and the all-reduces
    run between the replays.  Same trajectory as eager launches, bit for bit; and the rank that started from perturbed
    weights was repaired by the start-up broadcast (otherwise the ranks would differ)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    out = tmp_path / "res"
    _run(script, out, 2)
    _run(script, out, 2, "graph")
    e0 = torch.load(str(out) + ".w2.r0", weights_only=True)
    g0 = torch.load(str(out) + ".w2.r0.graph", weights_only=True)
    g1 = torch.load(str(out) + ".w2.r1.graph", weights_only=True)
    assert abs(e0["loss"] - g0["loss"]) < 1e-7
    for k in e0["sd"]:
        assert torch.equal(g0["sd"][k], g1["sd"][k]), f"ranks diverged on {k}"
        assert torch.equal(e0["sd"][k], g0["sd"][k]), f"graph replay diverged from eager on {k}"


SWEEP_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from vit_vs_raw_iq_amd import sweep as SW, data as D
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    d = torch.device("cuda:0")
    classes = ["BPSK", "QPSK", "16QAM", "OOK"]
    X, Y, _ = D.make_dataset(320, seed=3, classes=classes, snrs_db=(8.0,), n_symbols=1024)
    mean, std = D.zscore_stats(X)
    raw = torch.from_numpy(D.to_rawiq(X, mean, std)); img = torch.from_numpy(D.to_vit_images(X, mean, std)); Yt = torch.from_numpy(Y)
    data_raw = ((raw[:256], Yt[:256]), (raw[256:], Yt[256:])); data_vit = ((img[:256], Yt[:256]), (img[256:], Yt[256:]))
    vit_cfg = dict(in_channels=1, img_h=32, img_w=64, num_classes=4, device="cuda")
    raw_cfg = dict(in_channels=2, seq_length=1024, num_classes=4, device="cuda")
    rng = np.random.default_rng(0)
    lo, hi = SW.MIN_BOUNDS.copy(), SW.MAX_BOUNDS.copy(); hi[1], hi[3], hi[4] = 128, 2, 256
    Xp = rng.uniform(lo, hi, size=(5, 9))
    Xp[:, 5] = 0.0                      # no dropout: the score of a particle does not depend on which rank ran it
    orig = SW.build_models
    def seeded(params, rc, vc):         # the same initial weights for a particle whichever rank builds it
        torch.manual_seed(int(abs(float(params[1])) * 1000) %% 100000)
        return orig(params, rc, vc)
    SW.build_models = seeded
    s = SW.fitness_function(Xp, data_vit, data_raw, raw_cfg, vit_cfg, d)
    assert s.shape == (5,)
    np.save(sys.argv[1] + f".sweep.w{world}.r{rank}.npy", s)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    print("ok")
""") % ROOT


def test_sweep_task_parallel_two_ranks_equal_one_rank(tmp_path):
    """hyperparameter_tuning.py's particles are independent models: rank r scores particles r, r+W, ... and the ranks
    combine the scalars (sweep.fitness_function, world > 1 branch).  Two ranks must return, on every rank, exactly the
    scores one rank computes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    script = tmp_path / "s.py"
    script.write_text(SWEEP_WORKER)
    out = tmp_path / "res"
    _run(script, out, 1)
    _run(script, out, 2)
    one = np.load(str(out) + ".sweep.w1.r0.npy")
    a, b = np.load(str(out) + ".sweep.w2.r0.npy"), np.load(str(out) + ".sweep.w2.r1.npy")
    assert np.array_equal(a, b)
    assert np.allclose(one, a, atol=1e-12), (one, a)
    assert np.all(one <= 0) and np.all(one >= -1)
