"""Data-parallel training across the GPUs of one node: one process per GPU, plots sharded by rank,
one exchange step per iteration = bucketed gradient all-reduce (RCCL over xGMI; ``backend="nccl"`` on ROCm),
launched from autograd hooks so it overlaps the rest of the backward pass.

The reference only has single-process ``nn.DataParallel`` (torch_points3d/trainer.py:149-150); there is no
NCCL call pattern to translate.  Sizing: xGMI is point-to-point (7 links x ~153 GB/s per GPU); SENet14's
gradient is 57.8 MB, SENet50's 195 MB, so a handful of ~16 MB buckets keeps every all-reduce in the
bandwidth regime while the first bucket still starts early in the backward pass.  BatchNorm statistics stay
per rank (the reference has no SyncBN).
"""
from typing import List

import torch
import torch.distributed as dist


class GradAllReduce:
    """Flat gradient buckets + async all-reduce per bucket as soon as all of its gradients are produced.

    Gradients are left to autograd as fresh tensors (``zero_grad(set_to_none=True)``: assignment, no accumulate
    kernel per parameter); when the last gradient of a bucket has arrived, ONE multi-tensor copy packs them into the
    bucket's flat buffer, the all-reduce is launched on it and every ``param.grad`` is re-pointed at its slice of the
    buffer, so the optimiser reads the averaged values once ``__call__`` has waited for the exchange.

    Usage:  sync = GradAllReduce(model.parameters()); model.grad_sync = sync   (called after backward)
    """

    def __init__(self, params, bucket_bytes: int = 16 << 20, process_group=None, average: bool = True,
                 force_collective: bool = False):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        # force_collective: issue the all-reduce per bucket also in a group of ONE rank (a no-op arithmetically: the
        # average over one rank) — how a single-GPU box exercises RCCL's stream hand-off against the HIP kernels' stream
        # (tests/test_dist_gpu.py); needs an initialised process group
        self.force = bool(force_collective) and dist.is_initialized()
        params = [p for p in params if p.requires_grad]
        # reverse registration order ~ order in which backward produces gradients
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(self._make_bucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._pending = []
        self._handles = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_bucket(self, params):
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        views, off = [], 0
        for p in params:
            n = p.numel()
            views.append(flat[off:off + n].view_as(p))
            off += n
        return dict(params=params, flat=flat, views=views, ready=0, launched=False)

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            b["ready"] += 1
            if b["ready"] == len(b["params"]) and not b["launched"]:
                self._launch(b)
        return hook

    def _launch(self, b):
        b["launched"] = True
        with torch.no_grad():
            src, dst = [], []
            for p, v in zip(b["params"], b["views"]):
                if p.grad is None:
                    v.zero_()                      # parameter unused in this step
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad)
                    dst.append(v)
            if src:
                torch._foreach_copy_(dst, src)      # one multi-tensor launch per bucket
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        if self.world == 1 and not self.force:
            return
        backend = dist.get_backend(self.pg)
        if self.average and backend == "nccl":
            work = dist.all_reduce(b["flat"], op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            self._pending.append((work, None))
        else:
            work = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending.append((work, b["flat"] if self.average else None))

    def __call__(self):
        """Finish the exchange: launch buckets whose hooks did not all fire (unused parameters), wait, average."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        for work, flat in self._pending:
            work.wait()
            if flat is not None:
                flat.div_(self.world)
        self._pending.clear()
        for b in self.buckets:
            b["ready"] = 0
            b["launched"] = False

    def remove(self):
        for h in self._handles:
            h.remove()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """Rank ``src``'s parameters and buffers to every rank (once, at start)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
    # (a write through .data does not move the parameter's version counter: the cached bf16 operand forms of the
    # convolution weights — sparse_ops.weight_twins — are stale now)
    from .sparse_ops import bump_weight_epoch
    bump_weight_epoch()


def shard_seeds(global_batch: int, rank: int, world: int, step: int, base: int = 0) -> List[int]:
    """Disjoint synthetic-plot seeds: plots of step ``step`` are split evenly by rank."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    start = base + step * global_batch + rank * per
    return list(range(start, start + per))
