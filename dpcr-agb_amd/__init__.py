"""dpcr-agb_amd — MI355X-native (gfx950) hot path of StefOe/DPCR-AGB: point-cloud encoder forward/backward for
above-ground-biomass regression behind the reference's torch_points3d model/backbone API.

Layout:
  csrc/            hand-written HIP kernels + the flat C ABI (include/agb_hip.h) -> libagbhip.so
  _lib.py          ctypes binding (fails loudly when the library is missing; no CPU fallback)
  coords.py        device coordinate manager (hash, strided levels, kernel maps)
  sparse_ops.py    autograd bindings of the sparse-voxel kernels
  me_compat.py     the MinkowskiEngine API subset the reference's backbones use
  backbones/       SENet/ResNet (MSENet14/50), MinkowskiPointNet, KPCNN
  instance/        MinkowskiBaselineModel / KPConv model wrappers (set_input / optimize_parameters contract)
  optim.py         AdaBelief;  dist.py  data-parallel gradient all-reduce (RCCL)
  synthetic.py     seeded synthetic LiDAR plots + batch container

The directory name carries a hyphen (it mirrors the reference's repository name); import it as
``dpcr_agb_amd`` through the shim module at the repository root.
"""
__version__ = "0.1.0"


def limit_host_threads(n=None):
    """Cap torch's intra-op CPU thread pool for a training / serving process whose device work is all on the GPU.

    The host side of a step only issues launches and touches a few small CPU tensors.  torch sizes its OpenMP pool from the
    VISIBLE cores (256 on an MI355X host) even when a cgroup quota allows far fewer (16 on the GPU boxes of this project):
    the spinning workers exhaust the quota and the kernel throttles the whole process for the rest of the scheduling
    period — measured as 50-150 ms host stalls in the KPConv loop (the step rate doubled once the pool was capped).
    n: thread count (default: the AGB_HOST_THREADS environment variable, else 4 — fewer when the cgroup's CPU quota
    divided among the ranks of this node (LOCAL_WORLD_SIZE) is smaller: eight ranks of four spinning threads would
    exhaust a 16-core quota just the same).  Returns the previous setting."""
    import os
    import torch
    old = torch.get_num_threads()
    if n is None:
        n = os.environ.get("AGB_HOST_THREADS")
    if n is None:
        n = 4
        quota = cpu_quota()
        if quota is not None:
            ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            n = max(1, min(4, int(quota // ranks)))
    torch.set_num_threads(max(1, int(n)))
    return old


def cpu_quota():
    """CPU cores the cgroup of this process may use (cgroup v2 cpu.max or v1 cfs quota), or None when unlimited/unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None
