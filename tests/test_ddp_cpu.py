"""CPU, world_size 2, gloo: the data-parallel host logic of vit-vs-raw-iq_amd/trainer.py
(bucketed asynchronous gradient all-reduce, shard sampler)."""
import os
import socket
import subprocess
import sys
import textwrap

import torch

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from vit_vs_raw_iq_amd.trainer import BucketReducer, shard_indices, make_buckets
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    # flat "gradient": rank r holds (r+1) * pattern ; mean over ranks = 1.5 * pattern
    n = 10_000
    pattern = torch.arange(n, dtype=torch.float32) %% 97 - 48
    flat = pattern * (rank + 1)
    red = BucketReducer()
    assert red.world == 2
    ranges = [(0, 1000), (1000, 4000), (5000, 5000)]
    order = [2, 1, 0]                      # backward order: last bucket first
    for i in order:
        off, ln = ranges[i]
        red.launch(flat, off, ln)          # asynchronous; the next "stage" would run here
    red.wait()
    flat *= 1.0 / world                    # the 1/world factor lives in the gradnorm / AdamW kernels on GPU
    assert torch.allclose(flat, pattern * 1.5), (flat - pattern * 1.5).abs().max()
    # every rank computes the identical clip coefficient from the averaged gradient
    norm = flat.double().norm()
    t = torch.tensor([float(norm)], dtype=torch.float64)
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    assert lst[0].item() == lst[1].item()
    # sampler: disjoint, equal-sized shards, same permutation on every rank, reshuffled per epoch
    mine = shard_indices(1001, rank, world, epoch=3, seed=7)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.numel()]))
    assert sizes[0].item() == sizes[1].item() == 501
    both = [torch.zeros(501, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(both, mine)
    allidx = torch.cat(both)
    assert set(allidx.tolist()) == set(range(1001))           # full coverage (one wrapped duplicate)
    assert len(set(both[0].tolist()) & set(both[1].tolist())) <= 1
    assert not torch.equal(mine, shard_indices(1001, rank, world, epoch=4, seed=7))
    # epoch metric reduce (c2 in SURVEY 2.1): 3 scalars
    stats = torch.tensor([2.0 * (rank + 1), 10.0 * (rank + 1), 100.0], dtype=torch.float64)
    dist.all_reduce(stats)
    assert stats.tolist() == [6.0, 30.0, 200.0]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bucketed_allreduce_and_sampler_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\\n{out}"
        assert f"rank {r} ok" in out
