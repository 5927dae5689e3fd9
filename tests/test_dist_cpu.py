"""Data-parallel gradient exchange (dpcr-agb_amd/dist.py) on CPU with the gloo backend, world_size 2:
bucketed all-reduce launched from autograd hooks averages gradients, parameters stay identical across ranks, and
plots are sharded disjointly.  (The same code runs over RCCL/xGMI with backend='nccl' on the GPU box.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpcr_agb_amd.dist import GradAllReduce, broadcast_parameters, shard_seeds
    from dpcr_agb_amd.optim import AdaBelief
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.GELU(),
                                torch.nn.Linear(32, 2))
    broadcast_parameters(model)
    sync = GradAllReduce(model.parameters(), bucket_bytes=2048)  # several buckets
    opt = AdaBelief(model.parameters(), lr=0.005, weight_decay=1e-2)
    seeds = shard_seeds(8, rank, world, step=0)
    g = torch.Generator().manual_seed(seeds[0])
    x, y = torch.randn(4, 6, generator=g), torch.randn(4, 2, generator=g)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.smooth_l1_loss(model(x), y)
        loss.backward()
        sync()
        torch.nn.utils.clip_grad_value_(model.parameters(), 100)
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    q.put((rank, flat.numpy().copy(), grads.numpy().copy(), seeds, len(sync.buckets)))  # plain arrays: no fd passing
    dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, s0, nb), (_, p1, g1, s1, _) = res
    assert nb > 1
    assert (p0 == p1).all()              # identical parameters after 3 synchronous steps
    assert (g0 == g1).all()              # averaged gradients are the same on both ranks
    assert set(s0).isdisjoint(s1) and len(s0) == len(s1) == 4


def test_single_process_matches_manual_average():
    """world_size 1: the hook/bucket machinery must leave gradients untouched."""
    from dpcr_agb_amd.dist import GradAllReduce
    torch.manual_seed(0)
    m = torch.nn.Linear(5, 3)
    ref = torch.nn.Linear(5, 3)
    ref.load_state_dict(m.state_dict())
    sync = GradAllReduce(m.parameters())
    x = torch.randn(7, 5)
    m(x).sum().backward()
    sync()
    ref(x).sum().backward()
    for a, b in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad)
    sync.remove()


def test_unused_parameter_and_repacking():
    """A parameter that receives no gradient is exchanged as zeros; gradients are fresh tensors every step and end up as
    slices of the bucket's flat buffer (what the optimiser reads after the exchange)."""
    from dpcr_agb_amd.dist import GradAllReduce
    torch.manual_seed(1)
    used, unused = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    params = list(used.parameters()) + list(unused.parameters())
    sync = GradAllReduce(params, bucket_bytes=1 << 20)
    x = torch.randn(5, 4)
    for _ in range(2):
        for p in params:
            p.grad = None
        used(x).sum().backward()
        sync()
        flat = sync.buckets[0]["flat"]
        for p in params:
            assert p.grad is not None and p.grad.data_ptr() >= flat.data_ptr()
            assert p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
        assert float(unused.weight.grad.abs().sum()) == 0.0
        assert torch.allclose(used.weight.grad, x.sum(0).expand(3, 4))
    sync.remove()
