#!/usr/bin/env python3
"""Headline benchmark: IQ frames/s of one full training step (forward + label-smoothed CE + backward +
gradient all-reduce + clip + AdamW) on synthetic frames already resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--config B|C|Cp|D|A|ref] [--batch per-GPU]

N > 1 without a launcher: bench.py starts its own N ranks (one process per GPU, RCCL) BEFORE touching the GPU and
exits non-zero unless all N joined.  Under `python -m torch.distributed.run ... bench.py --gpus N` it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment instead.

Default workload = BASELINE.json configs[1]: ViT-Tiny/16 on 224x224 single-channel frames, 19 classes,
256 frames per GPU, dropout 0.1 ON, bf16 activations / fp32 master weights.  Weak scaling: per-GPU batch fixed.
Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- dominant kernel family, timed live with HIP event pairs on the launch stream
                  (iq_prof_* in include/iqvit.h) over profiled steps of the same workload
  cpu_baseline -- the CPU oracle (oracle/iq_oracle.py, "port") timed on this host's cores on a bounded sample
  step         -- whole-step fractions of the SURVEY 8(d) roofline: frames/s x algorithmic bytes (flops) per frame
  secondary    -- the raw-IQ transformer (BASELINE.json configs[2]) measured the same way in the same run
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

CONFIGS = {
    # name: (kind, ctor kwargs, per-GPU batch, drop_prob, weight_decay, description)
    "B": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192,
                      n_head=3, n_layers=12, ffn_hidden=768), 256, 0.1, 1e-3,
          "ViT-Tiny/16 224x224 C=1 19cls (BASELINE configs[1])"),
    "C": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=128, n_head=8, n_layers=6,
                        ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 256, 0.2, 1e-4,
          "transformer_rawIQ seg16 d128 h8 L6 F1024 (BASELINE configs[2])"),
    "Cp": ("rawiq", dict(in_channels=2, seq_length=1024, num_classes=19, d_model=256, n_head=8, n_layers=9,
                         ffn_hidden=1024, use_cls_token=True, embedding_type="segment", segment_size=16), 128, 0.1, 1e-3,
           "transformer_rawIQ published best d256 L9"),
    "D": ("vit", dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=768,
                      n_head=12, n_layers=12, ffn_hidden=3072), 512, 0.1, 1e-3,
          "ViT-Base/16 224x224 (BASELINE configs[3], 512/GPU)"),
    "A": ("vit", dict(in_channels=1, img_size_h=32, img_size_w=32, patch_size=16, num_classes=11, d_model=128,
                      n_head=8, n_layers=2, ffn_hidden=512), 256, 0.1, 1e-3, "ViT 32x32 p16 (BASELINE configs[0])"),
    "ref": ("vit", dict(in_channels=1, img_size_h=32, img_size_w=64, patch_size=4, num_classes=19, d_model=128,
                        n_head=8, n_layers=6, ffn_hidden=512), 256, 0.1, 1e-3, "reference train.py default ViT 32x64 p4"),
}

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16
RIDGE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
FAMILIES = ["gemm_nt", "wgrad", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "misc", "optimizer"]
PROFILE_ROUND = "r03"        # profiles/<round>_pmc_family_cfg<id>.json holds the PMC passes of this round's kernels


def geometry(kind, kw):
    D, F, L, H = kw["d_model"], kw["ffn_hidden"], kw["n_layers"], kw["n_head"]
    if kind == "vit":
        tok = (kw["img_size_h"] // kw["patch_size"]) * (kw["img_size_w"] // kw["patch_size"])
        P = kw["in_channels"] * kw["patch_size"] ** 2
        S = tok + 1
        in_elems = kw["in_channels"] * kw["img_size_h"] * kw["img_size_w"]
    else:
        k = 1 if kw["embedding_type"] == "conv1d" else kw["segment_size"]
        tok = kw["seq_length"] // k
        P = kw["in_channels"] * k
        S = tok + (1 if kw["use_cls_token"] else 0)
        in_elems = kw["in_channels"] * kw["seq_length"]
    return dict(D=D, F=F, L=L, H=H, dh=D // H, tok=tok, P=P, Ppad=(P + 31) // 32 * 32, S=S, K=kw["num_classes"],
                in_elems=in_elems)


def param_count(g):
    D, F, L, K, P = g["D"], g["F"], g["L"], g["K"], g["P"]
    per_layer = 4 * (D * D + D) + 2 * D * F + F + D + 4 * D
    return D * P + D + D + L * per_layer + D * K + K


def family_work(g, B, training_dropout, fused_ln, ln_bwd_launches):
    """Algorithmic bytes / flops per training step and launches per step for each kernel family
    (DESIGN.md section 'Kernels and rooflines' derives the same numbers).  fused_ln: the LayerNorm forward rides in the
    out-projection / FFN2 GEMM (it also writes the normalised rows); ln_bwd_launches: stand-alone LayerNorm backward
    launches left per step -- the others ride in the FFN1 / QKV data-gradient GEMM, which then reads Z and writes dZ (+ dY
    under dropout) instead of writing dX."""
    D, F, L, H, dh, S, tok, Ppad = g["D"], g["F"], g["L"], g["H"], g["dh"], g["S"], g["tok"], g["Ppad"]
    M, MT = B * S, B * tok
    nt = []   # (M, N, K, extra_MN_reads, extra_MN_writes)
    nt.append((MT, D, Ppad, 0, 0))                 # embedding
    ln_out = 1 if fused_ln else 0                  # out-proj / ffn2 also write the LayerNorm output when fused
    fused_bwd = 2 * L - ln_bwd_launches            # LayerNorm backwards inside a data-gradient GEMM
    dy = 1 if training_dropout else 0
    for l in range(L):
        f1 = 1 if fused_bwd >= 2 * L - 1 else 0                                  # norm1 backward in the FFN1 data gradient
        f2 = 1 if fused_bwd >= 2 * L - 1 and l > 0 else 0                        # norm2 backward of layer l-1 in layer l's QKV data gradient
        nt += [(M, 3 * D, D, 0, 0), (M, D, D, 1, ln_out), (M, F, D, 0, 0), (M, D, F, 1, ln_out)]   # fwd: qkv, out(+res), ffn1, ffn2(+res)
        nt += [(M, F, D, 1, 0), (M, D, F, 1 + f1, f1 * dy), (M, D, D, 0, 0), (M, D, 3 * D, 1 + f2, f2 * dy)]   # dgrad: ffn2(+gate), ffn1(+res), out, qkv(+res)
    b_nt = sum(2 * (m * k + n * k + m * n) + 2 * m * n * (ex + wx) for m, n, k, ex, wx in nt)
    f_nt = sum(2 * m * n * k for m, n, k, ex, wx in nt)
    wg = [(MT, D, Ppad)]
    for _ in range(L):
        wg += [(M, D, F), (M, F, D), (M, D, D), (M, 3 * D, D)]
    b_wg = sum(2 * (m * n + m * k) + 4 * n * k for m, n, k in wg)
    f_wg = sum(2 * m * n * k for m, n, k in wg)
    b_af = L * (2 * M * 3 * D + 2 * M * D + 4 * B * H * S)
    f_af = L * 4 * B * H * S * S * dh
    b_ab = L * (2 * M * 3 * D * 2 + 2 * M * D * 2 + 4 * B * H * S)
    f_ab = L * 14 * B * H * S * S * dh          # 7 MFMA products (S and dP are computed in both phases)
    b_lf = 0 if fused_ln else 2 * L * (2 * M * D * 2 + 8 * M)
    b_lb = ln_bwd_launches * (2 * M * D * (3 + (1 if training_dropout else 0)) + 8 * M)
    return {
        "gemm_nt": dict(bytes=b_nt, flops=f_nt, launches=len(nt)),
        "wgrad": dict(bytes=b_wg, flops=f_wg, launches=len(wg)),
        "attn_fwd": dict(bytes=b_af, flops=f_af, launches=L),
        "attn_bwd": dict(bytes=b_ab, flops=f_ab, launches=L),
        "ln_fwd": dict(bytes=b_lf, flops=0, launches=0 if fused_ln else 2 * L),
        "ln_bwd": dict(bytes=b_lb, flops=0, launches=ln_bwd_launches),
    }


def train_flops_per_frame(g):
    D, F, L, S, tok, P, K = g["D"], g["F"], g["L"], g["S"], g["tok"], g["P"], g["K"]
    fwd = L * (8 * S * D * D + 4 * S * S * D + 4 * S * D * F) + 2 * tok * P * D + 2 * D * K
    return 3 * fwd


def algorithmic_bytes_per_frame(g, B):
    """input read + 2 x saved activations (written forward, read backward; the flash-style minimum
    L*(8*S*D + S*F) bf16 elements + L*H*S fp32 log-sum-exp) + 3 x parameter bytes / per-GPU batch.
    cfg B: 22.1 MB, cfg C: 3.3 MB, cfg D: 90 MB per frame."""
    D, F, L, H, S = g["D"], g["F"], g["L"], g["H"], g["S"]
    saved = L * (8 * S * D + S * F) * 2 + L * H * S * 4
    return g["in_elems"] * 2 + 2 * saved + 3 * 4 * param_count(g) / B


def host_cores():
    """Cores this process may use: the scheduler affinity, cut to the cgroup CPU quota when there is one (a 1-GPU box is a
    16-core share of a larger host; more threads than that oversubscribe)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    visible = n
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2
            q, period = fh.read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:      # cgroup v1
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = int(fh.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    elif n > 16:
        n = 16          # no quota readable: a 1-GPU box is documented as a 16-core share of its host
    return n, visible


def cpu_baseline(kind, kw, drop, wd, gpu_batch, budget_s=40.0):
    """The CPU oracle's full training step (dropout ON, clip, AdamW) on this host's cores, SURVEY 8(d) "CPU baseline beside
    it": the GPU run's batch -- or, when one step of it would not fit the budget, the largest batch whose step does (stated
    in `sample`) --, 1 warm-up step + 3 timed steps, torch threads = the cores this process owns (printed)."""
    import torch
    import iq_oracle as O
    ncpu, visible = host_cores()
    if os.environ.get("IQ_CPU_THREADS"):
        ncpu = int(os.environ["IQ_CPU_THREADS"])
    torch.set_num_threads(max(1, ncpu))
    cfg = O.OracleConfig(kind=kind, drop_prob=drop, **kw)
    sd = O.init_state(cfg, 0)
    st = O.adamw_init(sd)
    g = torch.Generator().manual_seed(0)

    def batch_of(b):
        shape = (b, kw["in_channels"], kw["img_size_h"], kw["img_size_w"]) if kind == "vit" else (b, kw["in_channels"], kw["seq_length"])
        return torch.randn(*shape, generator=g), torch.randint(0, kw["num_classes"], (b,), generator=g)

    # probe: a small step to size the batch.  Its time per frame UNDER-estimates the large batch's on a wide model (cfg B: 36
    # frames/s at batch 8, 15 at batch 256 -- activations fall out of the host caches): the estimate is doubled.
    pb = min(gpu_batch, 16 if kw["d_model"] <= 256 else 2)
    x, y = batch_of(pb)
    O.train_step(cfg, sd, st, x, y, weight_decay=wd)
    t0 = time.perf_counter()
    O.train_step(cfg, sd, st, x, y, weight_decay=wd)
    per_frame = 2.0 * (time.perf_counter() - t0) / pb
    step_cap = min(60.0, budget_s / 4.0)           # a step under 60 s; here also 1 + 3 steps inside the budget
    b = int(max(1, min(gpu_batch, step_cap / per_frame)))
    x, y = batch_of(b)
    O.train_step(cfg, sd, st, x, y, weight_decay=wd)        # warm-up at the timed batch
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.train_step(cfg, sd, st, x, y, weight_decay=wd)
        times.append(time.perf_counter() - t0)
    el = sum(times)
    note = "the GPU run's batch" if b == gpu_batch else f"largest batch with a step under {step_cap:.0f} s (GPU run: {gpu_batch})"
    print(f"[bench] cpu_baseline: {torch.get_num_threads()} torch threads on {ncpu} usable cores ({visible} visible), batch {b}, "
          f"steps {', '.join(f'{t:.2f}' for t in times)} s", file=sys.stderr)
    return {"value": round(3 * b / el, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "cores_visible": visible,
            "kind": "port", "batch": b, "steps": 3, "warmup": 1,
            "sample": f"3 full training steps of batch {b} ({note}; same model/config, fp32, dropout on, clip + AdamW) after 1 "
                      f"warm-up, {el:.1f} s"}


def accuracy_reproduction(dev):
    """SURVEY 8(d) metric (2): top-1 accuracy on the shared synthetic IQ set (data.accuracy_task: cfg A's ViT and the
    R/test_model.py raw-IQ geometry, N = 1000, 800 / 200, seed 42, fixed step budget), MI355X path vs the CPU oracle from the
    same initial state.  The oracle half is part of the cpu_baseline leg (rank 0, N = 1 only)."""
    import torch
    import accuracy_oracle as AO
    from vit_vs_raw_iq_amd import accuracy as A
    from vit_vs_raw_iq_amd import data as D
    out = {}
    for name in ("vit_A", "rawiq_R"):
        task = D.accuracy_task(name)
        _, sd0 = AO.initial_state(task)
        t0 = time.perf_counter()
        gpu = A.train_and_score(task, sd0, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cpu = AO.train_and_score(task)
        t2 = time.perf_counter()
        h = task["hyper"]
        out[name] = {"gpu": {k: round(v, 4) for k, v in gpu.items()}, "cpu_oracle": {k: round(v, 4) for k, v in cpu.items()},
                     "chance": round(task["chance"], 4), "delta_heldout": round(abs(gpu["heldout"] - cpu["heldout"]), 4),
                     "delta_fresh": round(abs(gpu["fresh"] - cpu["fresh"]), 4),
                     "gpu_s": round(t1 - t0, 2), "cpu_s": round(t2 - t1, 2),
                     "task": f"{h['n_frames']} frames ({h['n_train']} train / {h['n_frames'] - h['n_train']} held out, seed "
                             f"{h['data_seed']}), {task['kw']['num_classes']} classes, {h['steps']} steps of {h['batch']}, lr {h['lr']}, "
                             f"dropout {h['drop_prob']}; `fresh` = {h['fresh_frames']} more held-out frames (seed {h['fresh_seed']})"}
    return out


def self_launch(a, argv):
    """`--gpus N` with no launcher: start N ranks (one per GPU) before this process has touched the GPU, relay
    rank 0's JSON line, exit non-zero unless every rank finished."""
    import socket
    import torch
    ndev = torch.cuda.device_count()          # does not initialise the GPU
    backend = os.environ.get("IQ_DIST_BACKEND", "nccl")
    if backend == "nccl" and ndev < a.gpus:
        print(f"[bench] --gpus {a.gpus} needs {a.gpus} GPUs for {a.gpus} RCCL ranks, this node shows {ndev}", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), IQ_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    if any(codes):
        print(f"[bench] rank exit codes {codes}: fewer than {a.gpus} ranks completed", file=sys.stderr)
        return 1
    return 0


def measure(a, config_id, dev, rank, world, steps, warmup, prof_steps, want_cpu):
    """Timed steps + profiled leg (+ CPU baseline) for one named configuration -> dict of results (rank 0) or None."""
    import ctypes
    import torch
    import torch.distributed as dist
    import vit_vs_raw_iq_amd as P
    from vit_vs_raw_iq_amd.trainer import FusedTrainer
    import vit_vs_raw_iq_amd._native as N

    kind, kw, batch, drop, wd, desc = CONFIGS[config_id]
    B = a.batch or batch
    if a.drop >= 0:
        drop = a.drop
    torch.manual_seed(0)                       # identical init on every rank; FusedTrainer also broadcasts rank 0's
    cls = P.AMCTransformerViT if kind == "vit" else P.AMCTransformerRawIQ
    model = cls(drop_prob=drop, device="cuda", **kw).to(dev).train()
    use_graph = True if a.graph < 0 else bool(a.graph)
    tr = FusedTrainer(model, lr=1e-4, weight_decay=wd, betas=(0.9, 0.99), label_smoothing=0.1, max_norm=1.0,
                      n_buckets=a.buckets, use_graph=use_graph, dropout_seed=1234)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    shape = (B, kw["in_channels"], kw["img_size_h"], kw["img_size_w"]) if kind == "vit" else (B, kw["in_channels"], kw["seq_length"])
    x = torch.randn(*shape, device=dev, generator=g)
    y = torch.randint(0, kw["num_classes"], (B,), device=dev, generator=g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 1)):
        tr.step(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss, acc, _ = tr.read_stats()
    frames = B * world * steps
    value = frames / el

    geo = geometry(kind, kw)
    roof = None
    if world > 1 and not a.no_roofline:
        # every rank must take the same number of steps (each step all-reduces): the profiled leg runs everywhere,
        # only rank 0 times its kernels
        if rank != 0:
            tr.use_graph = False
            for _ in range(1 + prof_steps):
                tr.step(x, y)
            torch.cuda.synchronize()
    if rank == 0 and not a.no_roofline:
        # profiled leg: same workload, eager launches, HIP event pair around every kernel-family call
        L = N.lib()
        tr.use_graph = False
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_longlong * 8)()
        tr.step(x, y)
        torch.cuda.synchronize()
        L.iq_prof_enable(1)
        L.iq_prof_collect(ms, cnt)
        for _ in range(prof_steps):
            tr.step(x, y)
        L.iq_prof_collect(ms, cnt)
        L.iq_prof_enable(0)
        need = L.iq_prof_kernels(None, 0, 0)
        kbuf = ctypes.create_string_buffer(need + 1)
        L.iq_prof_kernels(kbuf, need + 1, 1)
        kernels = []
        for line in kbuf.value.decode().splitlines():
            name, fam, n, kms, kbytes, kflops = line.split("\t")
            n, kms, kbytes, kflops = int(n), float(kms), float(kbytes), float(kflops)
            if n == 0 or kms <= 0:
                continue
            us = kms * 1e3 / n
            gbs, tfs = kbytes / (kms * 1e-3) / 1e9, kflops / (kms * 1e-3) / 1e12
            bound = "mfma" if kflops / max(kbytes, 1.0) > RIDGE else "hbm"
            kernels.append({"kernel": name, "family": FAMILIES[int(fam)], "launches_per_step": round(n / prof_steps, 2),
                            "avg_us": round(us, 2), "ms_per_step": round(kms / prof_steps, 4),
                            "algorithmic_bytes_per_launch": int(kbytes / n), "flops_per_launch": int(kflops / n),
                            "bound": bound, "achieved_gbs": round(gbs, 1), "achieved_tflops": round(tfs, 2),
                            "frac": round(tfs / MFMA_PEAK_TFLOPS if bound == "mfma" else gbs / HBM_PEAK_GBS, 4)})
        kernels.sort(key=lambda k: -k["ms_per_step"])
        per_step = {f: ms[i] / prof_steps for i, f in enumerate(FAMILIES)}
        fused_ln = int(cnt[FAMILIES.index("ln_fwd")]) == 0
        work = family_work(geo, B, drop > 0, fused_ln, int(cnt[FAMILIES.index("ln_bwd")]) // prof_steps)
        # Algorithmic bytes / flops per family as the launch sites themselves recorded them (operands read once, results written
        # once, per launch: iq_prof_kernels) -- they follow whatever kernels the plan chose (e.g. the one-launch feed-forward,
        # which has no hidden-activation re-read to count); the analytic model above stays as the fallback and cross-check.
        rec = {}
        for kk in kernels:
            r = rec.setdefault(kk["family"], dict(bytes=0.0, flops=0.0, launches=0.0))
            r["bytes"] += kk["algorithmic_bytes_per_launch"] * kk["launches_per_step"]
            r["flops"] += kk["flops_per_launch"] * kk["launches_per_step"]
            r["launches"] += kk["launches_per_step"]
        for f in work:
            if f in rec and rec[f]["bytes"] > 0:
                work[f] = dict(bytes=rec[f]["bytes"], flops=rec[f]["flops"], launches=int(round(rec[f]["launches"])), model_bytes=work[f]["bytes"])
        dom = max(work, key=lambda f: per_step[f])
        w = work[dom]
        dur_ms = per_step[dom]
        ai = w["flops"] / max(w["bytes"], 1)
        launches = max(int(cnt[FAMILIES.index(dom)]) // prof_steps, 1)
        if ai > RIDGE:
            ach = w["flops"] / (dur_ms * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None}
        else:
            ach = w["bytes"] / (dur_ms * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
        # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 --pmc passes,
        # scripts/pmc_family.sh); report the committed measurement of the same workload when there is one.
        pmc_path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_family_cfg{config_id}.json")
        if os.path.exists(pmc_path) and a.batch in (0, batch) and a.drop < 0 and world == 1:
            try:
                with open(pmc_path) as fh:
                    pmc = json.load(fh)
                if dom in pmc:
                    roof["traffic"] = int(pmc[dom]["hbm_bytes_per_launch"])
                    roof["traffic_source"] = os.path.relpath(pmc_path, ROOT)
                if "mfma_util" in pmc:
                    roof["mfma_util_step"] = pmc["mfma_util"]       # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs)
            except (OSError, ValueError, KeyError):
                pass
        roof["avg_launch_us"] = round(dur_ms * 1e3 / launches, 2)
        roof["launches_per_step"] = launches
        roof["algorithmic_bytes_per_launch"] = int(w["bytes"] / launches)
        roof["family_ms_per_step"] = {f: round(v, 4) for f, v in per_step.items()}
        # the five kernels (not families) that take the most time per step, under the names rocprofv3 --kernel-trace --stats
        # prints (profiles/*_kernel_stats_*.csv): each with its own algorithmic bytes / flops per launch and roofline fraction
        roof["kernels"] = kernels[:5]
        tot = sum(per_step.values())
        print(f"[bench:{config_id}] kernel time per step by family (HIP events, eager): "
              + ", ".join(f"{f} {v:.3f} ms" for f, v in per_step.items()) + f"; sum {tot:.3f} ms", file=sys.stderr)
        for f, wk in work.items():
            d = per_step[f] * 1e-3
            if d > 0 and wk["launches"]:
                print(f"[bench:{config_id}]   {f:9s} {wk['bytes'] / d / 1e9:8.1f} GB/s algorithmic  {wk['flops'] / d / 1e12:7.2f} TFLOP/s"
                      f"  ({wk['launches']} launches/step)", file=sys.stderr)

    cpu = None
    if rank == 0 and want_cpu:
        cpu = cpu_baseline(kind, kw, drop, wd, B, budget_s=a.cpu_budget)
    del tr, model
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    fl = train_flops_per_frame(geo)
    by = algorithmic_bytes_per_frame(geo, B)
    return {
        "value": round(value, 1), "ms_per_step": round(el / steps * 1e3, 3),
        "config": {"workload": desc, "config_id": config_id, "per_gpu_batch": B, "global_batch": B * world,
                   "tokens_per_frame": geo["S"], "parallelism": f"dp{world}", "dropout": drop,
                   "hipgraph": bool(use_graph), "train_gflop_per_frame": round(fl / 1e9, 4),
                   "algorithmic_mb_per_frame": round(by / 1e6, 3)},
        "step": {"model_tflops": round(value * fl / 1e12, 2),
                 "step_mfma_frac": round(value * fl / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
                 "step_hbm_gbs": round(value * by / 1e9, 1),
                 "step_hbm_frac": round(value * by / 1e9 / (HBM_PEAK_GBS * world), 4)},
        "final_loss": round(loss, 4), "train_acc": round(acc, 4), "roofline": roof, "cpu_baseline": cpu,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--graph", type=int, default=-1, help="hipGraph replay of the step (default: on)")
    ap.add_argument("--buckets", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the raw-IQ (cfg C) measurement beside cfg B")
    ap.add_argument("--prof-steps", type=int, default=5)
    ap.add_argument("--cpu-budget", type=float, default=80.0, help="seconds of CPU-oracle work per configuration (1 warm-up + 3 timed steps)")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the accuracy reproduction (cfg A ViT + raw-IQ test geometry, GPU vs CPU oracle)")
    ap.add_argument("--drop", type=float, default=-1.0, help="override the config's dropout probability (diagnostics)")
    a = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(self_launch(a, sys.argv[1:]))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a "
              f"{world}-GPU number as a {a.gpus}-GPU one", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IQ_DIST_BACKEND", "nccl")     # "gloo" only to rehearse the N>1 path on a 1-GPU box
        if backend == "nccl":
            if ndev < world:
                raise SystemExit(f"{world} RCCL ranks need {world} GPUs, {ndev} visible")
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world

    main_res = measure(a, a.config, dev, rank, world, a.steps, a.warmup, a.prof_steps,
                       want_cpu=(world == 1 and not a.no_cpu_baseline))
    second = None
    if a.config == "B" and not a.no_secondary and a.batch == 0:
        second = measure(a, "C", dev, rank, world, a.steps, a.warmup, a.prof_steps,
                         want_cpu=(world == 1 and not a.no_cpu_baseline))

    acc = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.no_accuracy and a.config == "B" and a.batch == 0:
        acc = accuracy_reproduction(dev)

    if rank == 0:
        out = {
            "metric": "IQ frames/sec training (fwd+loss+bwd+clip+AdamW)", "value": main_res["value"], "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": main_res["config"], "step": main_res["step"], "final_loss": main_res["final_loss"],
            "train_acc": main_res["train_acc"],
            "dist": {"backend": backend, "rccl_ranks": world if backend == "nccl" else 0, "ranks": world},
            "roofline": main_res["roofline"], "cpu_baseline": main_res["cpu_baseline"],
        }
        if acc is not None:
            out["accuracy"] = acc
        if second is not None:
            out["secondary"] = {"metric": out["metric"], "unit": "frames/s", **second}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
