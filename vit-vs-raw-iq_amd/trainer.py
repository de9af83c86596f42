"""Fused native training step and its data-parallel driver.

Counterpart of the body of train_epoch (reference: Transformer_Thesis/ViT/training/train.py:185-207,
transformer_rawIQ/training/train.py:252-277) with the reference's criterion / optimizer settings
(ViT/training/train.py:405-412): forward, CrossEntropyLoss(label_smoothing), backward,
clip_grad_norm_(max_norm), AdamW(betas (0.9, 0.99), decoupled weight decay on every parameter).
Everything between "batch is in HBM" and "parameters updated" is native kernels on one stream:
no host synchronisation per step (the reference's two .item() calls per step become device-side
accumulators read once per epoch), optional hipGraph replay of the whole step.

Data parallel (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm): the native
backward runs in stages (head, layers L-1..0, embedding); after each bucket of stages the
corresponding contiguous range of the flat gradient is all-reduced asynchronously while the next
stages run, and the clip + AdamW kernels wait for all buckets (the clip needs the global-batch
gradient norm).  Gradient mean = SUM all-reduce, 1/world folded into the norm and AdamW kernels.
hipGraph replay also works data parallel: the step is captured as one graph per segment between bucket
boundaries, and the all-reduces are issued between the replays.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _native as N
from .modules import NativePlan


def make_buckets(n_layers: int, n_buckets: int) -> List[Tuple[int, int]]:
    """Split backward stages n_layers+1 (head) .. 0 (embedding) into contiguous (hi, lo) ranges,
    in execution order.  The head rides with the last layers, the embedding with the first."""
    stages = list(range(n_layers + 1, -1, -1))
    n_buckets = max(1, min(n_buckets, len(stages)))
    per = math.ceil(len(stages) / n_buckets)
    out = []
    for i in range(0, len(stages), per):
        chunk = stages[i:i + per]
        out.append((chunk[0], chunk[-1]))
    return out


class BucketReducer:
    """Asynchronous SUM all-reduce of contiguous ranges of one flat gradient tensor.
    Device-agnostic (works on CPU tensors with gloo: tests/test_ddp_cpu.py)."""

    def __init__(self, group=None):
        self.group = group
        self.pending = []

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def launch(self, flat: torch.Tensor, off: int, length: int):
        if self.world == 1 or length == 0:
            return
        self.pending.append(dist.all_reduce(flat[off:off + length], op=dist.ReduceOp.SUM, group=self.group,
                                            async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


def shard_indices(n: int, rank: int, world: int, epoch: int, seed: int = 0) -> torch.Tensor:
    """DistributedSampler-equivalent: one seeded permutation shared by all ranks, rank r takes
    positions r, r+world, ... (padded by wrapping so every rank gets the same count)."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    perm = torch.randperm(n, generator=g)
    per = math.ceil(n / world)
    total = per * world
    if total > n:
        perm = torch.cat([perm, perm[: total - n]])
    return perm[rank:total:world]


class FusedTrainer:
    """Native fwd + loss + bwd + (all-reduce) + clip + AdamW on a model built from modules.py."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.99), eps=1e-8, label_smoothing=0.1,
                 max_norm=1.0, device=None, group=None, n_buckets: int = 4, use_graph: bool = False,
                 dropout_seed: Optional[int] = None, sync_params: bool = True):
        self.model = model
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        if self.device.type != "cuda":
            raise N.IqError("FusedTrainer needs the model on an MI355X (cuda) device; there is no CPU fallback")
        self.plan: NativePlan = model.native_plan()
        self.L = N.lib()
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.smoothing, self.max_norm = label_smoothing, max_norm
        self.reducer = BucketReducer(group)
        self.world = self.reducer.world
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if dropout_seed is not None:
            self.plan.seed = int(dropout_seed)
        self.plan.seed = (self.plan.seed + 0x9E3779B97F4A7C15 * self.rank) & ((1 << 63) - 1)   # per-rank stream
        with torch.cuda.device(self.device):
            self.plan.ensure(self.device)
            if self.world > 1 and sync_params:
                self._sync_parameters()
        n = self.plan.nparam
        d = self.device
        self.gflat = torch.zeros(n, dtype=torch.float32, device=d)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=d)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=d)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=d)
        self.n_correct = torch.zeros(1, dtype=torch.int32, device=d)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=d)
        self.gn_ws = torch.empty(self.L.iq_gradnorm_ws_bytes(n), dtype=torch.uint8, device=d)
        self.dyn = torch.tensor([lr, 0.0], dtype=torch.float32, device=d)    # device-side {lr, step}
        self.frames_seen = 0
        self.steps = 0
        self.buckets = make_buckets(self.plan.cfg.n_layers, n_buckets if self.world > 1 else 1)
        self.ranges = [self.plan.grad_range(hi, lo) for hi, lo in self.buckets]
        self.use_graph = bool(use_graph)
        # per batch size: logits / dlogits buffers, static inputs and the captured graphs (one hipGraph per segment, see
        # _segments) with the device pointers baked into them.  An epoch's last partial batch and the return to the full
        # batch each find their own slot again instead of discarding every capture twice per epoch.
        self._slots = {}
        self._slot = None
        self._logits = None
        self._dlogits = None
        self._batch = 0

    # ------------------------------------------------------------------------------------------
    def _sync_parameters(self):
        """Data-parallel start-up: every rank starts from rank 0's parameters, then all ranks check
        that they hold the same bytes (a rank that constructed an unseeded or differently loaded model would
        otherwise all-reduce gradients onto different weights and diverge silently).  Also the first collective of
        the job: a broken RCCL set-up fails here, not in the middle of a step."""
        plan, g = self.plan, self.reducer.group
        dist.broadcast(plan.flat, src=dist.get_global_rank(g, 0) if g is not None else 0, group=g)
        plan.mark_dirty()
        plan.ensure(self.device)
        flat = plan.flat.double()
        sig = torch.stack([flat.sum(), flat.abs().sum(), (flat * flat).sum()])
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=g)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=g)
        if not torch.equal(lo, hi):
            raise N.IqError(f"rank {self.rank}: parameters differ across ranks after the start-up broadcast "
                            f"(checksums {sig.tolist()} vs min {lo.tolist()} / max {hi.tolist()})")

    def set_lr(self, lr: float):
        self.lr = float(lr)
        self.dyn[0] = self.lr

    MAX_SLOTS = 4        # batch sizes kept captured at once (full batch, epoch tail, a validation-sized one, ...)

    @property
    def _graphs(self):
        return self._slot["graphs"] if self._slot is not None else None

    @_graphs.setter
    def _graphs(self, value):
        """`trainer._graphs = None` (checkpoint load, hyper-parameter change) drops EVERY capture."""
        if value is None:
            for sl in self._slots.values():
                sl["graphs"] = None
        elif self._slot is not None:
            self._slot["graphs"] = value

    def _buffers(self, B: int):
        if self._batch != B:
            sl = self._slots.pop(B, None)
            if sl is None:
                K = self.plan.cfg.num_classes
                sl = {"logits": torch.empty(B, K, dtype=torch.float32, device=self.device),
                      "dlogits": torch.empty(B, K, dtype=torch.float32, device=self.device),
                      "static": None, "graphs": None, "captured": None}
                while len(self._slots) >= self.MAX_SLOTS:
                    self._slots.pop(next(iter(self._slots)))          # oldest
            self._slots[B] = sl                                       # most recent last
            self._slot = sl
            self._logits, self._dlogits = sl["logits"], sl["dlogits"]
            self._batch = B
        # every call: an inference forward with a larger batch (evaluate(), model(x)) may have replaced the workspace
        self.plan.workspace(B, self.device)

    def _pointers(self):
        p = self.plan
        return (p.ws.data_ptr(), p.ws.numel(), p.flat.data_ptr(), p.shadow.data_ptr(), p.step_ctr.data_ptr(),
                self._logits.data_ptr(), self._dlogits.data_ptr(), self.gflat.data_ptr(), self._batch)

    def _segments(self, x: torch.Tensor, y: torch.Tensor, auto_step: bool):
        """The training step as a list of launch segments, cut where a gradient bucket becomes complete:
        [forward + loss + backward of bucket 0], [backward of bucket 1], ..., [clip + AdamW + shadow refresh].
        Each segment issues only native kernels on the current stream (capture-safe: no alloc, no sync); the caller
        starts the bucket's all-reduce between segments."""
        plan, L = self.plan, self.L
        B = x.shape[0]
        ws = plan.ws
        step_arg = 0xFFFFFFFF if auto_step else (self.steps + 1) & 0x7FFFFFFF

        def fwd_and_first_bucket():
            st = N.stream_handle()
            N.check(L.iq_model_forward(plan.h, x.data_ptr(), B, ws.data_ptr(), ws.numel(), 1, plan.seed, step_arg, None,
                                       self._logits.data_ptr(), st), "iq_model_forward", plan.h)
            N.check(L.iq_ce_fwd_bwd(self._logits.data_ptr(), y.data_ptr(), B, plan.cfg.num_classes, self.smoothing,
                                    float(B), self.loss_sum.data_ptr(), self.n_correct.data_ptr(),
                                    self._dlogits.data_ptr(), st), "iq_ce_fwd_bwd")
            bwd(0)()

        def bwd(i):
            hi, lo = self.buckets[i]

            def run():
                N.check(L.iq_model_backward(plan.h, self._dlogits.data_ptr(), None, B, ws.data_ptr(), ws.numel(), 0, hi,
                                            lo, N.stream_handle()), "iq_model_backward", plan.h)
            return run

        def optimizer():
            st = N.stream_handle()
            gscale = 1.0 / self.world
            N.check(L.iq_gradnorm_sq(self.gflat.data_ptr(), plan.nparam, gscale, self.gn_ws.data_ptr(),
                                     self.gnorm_sq.data_ptr(), st), "iq_gradnorm_sq")
            N.check(L.iq_counter_add(None, 0, self.dyn.data_ptr() + 4, 1.0, st), "iq_counter_add")
            N.check(L.iq_adamw_step(plan.flat.data_ptr(), self.gflat.data_ptr(), self.exp_avg.data_ptr(),
                                    self.exp_avg_sq.data_ptr(), plan.shadow.data_ptr(), plan.nparam, self.lr,
                                    self.betas[0], self.betas[1], self.eps, self.wd, 0, self.gnorm_sq.data_ptr(),
                                    self.max_norm, gscale, self.dyn.data_ptr(), st), "iq_adamw_step")
            # transposed / padded shadows for the next step's dgrads
            N.check(L.iq_model_refresh_transposed(plan.h, st), "iq_model_refresh_transposed", plan.h)

        return [fwd_and_first_bucket] + [bwd(i) for i in range(1, len(self.buckets))] + [optimizer]

    def _run_segments(self, runners):
        """runners[i]() issues segment i; bucket i's all-reduce starts as soon as segment i is on the stream and
        overlaps the following segments; the optimizer segment waits for all of them."""
        nb = len(self.buckets)
        for i, run in enumerate(runners):
            if i == nb:
                self.reducer.wait()
            run()
            if i < nb:
                off, ln = self.ranges[i]
                self.reducer.launch(self.gflat, off, ln)

    def step(self, x: torch.Tensor, y: torch.Tensor):
        """One optimizer step on a batch already resident in HBM.  Asynchronous."""
        if not x.is_cuda or not y.is_cuda:
            raise N.IqError("FusedTrainer.step expects device tensors (batch resident in HBM)")
        x = x.contiguous().float()
        y = y.contiguous().long()
        B = x.shape[0]
        plan = self.plan
        if not plan.is_bound(self.device):
            # the parameters were re-homed behind the trainer's back (model.to(), a second plan, p.data = ...): adopt
            # their CURRENT values instead of updating a buffer nobody reads any more
            plan.ensure(self.device)
        self._buffers(B)
        if self.gflat is not plan.gflat:
            plan._bind_native(self.gflat)
        plan.generation += 1
        if self.use_graph:
            sl = self._slot
            if sl["graphs"] is not None and sl["captured"] != self._pointers():
                sl["graphs"] = None          # a captured buffer moved (workspace regrown by an eval forward, rebinding)
            if plan.ctr_value != self.steps:
                plan.step_ctr.fill_(self.steps)      # something else ran a training forward: re-sync the device step
                plan.ctr_value = self.steps
            if sl["graphs"] is None:
                # first call at this batch size: one eager step on static buffers (this call's step; also warms lazy
                # kernel attributes and the allocator), then capture the identical launch sequence for later replays
                if sl["static"] is None or sl["static"][0].shape != x.shape:
                    sl["static"] = (torch.empty_like(x), torch.empty_like(y))
                static = sl["static"]
                static[0].copy_(x)
                static[1].copy_(y)
                self._run_segments(self._segments(static[0], static[1], auto_step=True))
                torch.cuda.synchronize()
                graphs = []
                for seg in self._segments(static[0], static[1], auto_step=True):
                    g = torch.cuda.CUDAGraph()
                    # thread_local: with a process group alive, its watchdog thread queries events while we capture;
                    # in the default (global) mode that invalidates the capture.  Only this thread's stream is captured.
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        seg()
                    graphs.append(g)
                sl["graphs"] = graphs
                sl["captured"] = self._pointers()
            else:
                sl["static"][0].copy_(x, non_blocking=True)
                sl["static"][1].copy_(y, non_blocking=True)
                self._run_segments([g.replay for g in sl["graphs"]])
        else:
            self._run_segments(self._segments(x, y, auto_step=False))
        self.steps += 1
        self.frames_seen += B
        plan.step = self.steps
        plan.ctr_value = self.steps

    # ------------------------------------------------------------------------------------------
    def read_stats(self, reset=True):
        """(mean loss per frame, accuracy, frames) since the last reset -- ONE host sync, all-reduced
        over ranks (the reference syncs twice per step: V/training/train.py:204-207)."""
        t = torch.tensor([float(self.loss_sum.item()), float(self.n_correct.item()), float(self.frames_seen)],
                         dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t, group=self.reducer.group)
        loss, correct, frames = t.tolist()
        if reset:
            self.loss_sum.zero_()
            self.n_correct.zero_()
            self.frames_seen = 0
        return loss / max(frames, 1.0), correct / max(frames, 1.0), int(frames)

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, y: torch.Tensor, batch: int = 256):
        """Loss / accuracy with the model in eval mode (validate_epoch, V/training/train.py:223-260)."""
        was = self.model.training
        self.model.eval()
        K = self.plan.cfg.num_classes
        ls = torch.zeros(1, device=self.device)
        nc = torch.zeros(1, dtype=torch.int32, device=self.device)
        for i in range(0, x.shape[0], batch):
            xb, yb = x[i:i + batch].to(self.device), y[i:i + batch].to(self.device).long()
            logits = self.model(xb)
            N.check(self.L.iq_ce_fwd_bwd(logits.data_ptr(), yb.data_ptr(), xb.shape[0], K, self.smoothing,
                                         float(xb.shape[0]), ls.data_ptr(), nc.data_ptr(), None, N.stream_handle()),
                    "iq_ce_fwd_bwd")
        self.model.train(was)
        t = torch.tensor([ls.item(), float(nc.item()), float(x.shape[0])], dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t, group=self.reducer.group)
        loss, correct, frames = t.tolist()
        return loss / frames, correct / frames
