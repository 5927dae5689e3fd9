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
    n: thread count (default: the AGB_HOST_THREADS environment variable, else 4).  Returns the previous setting."""
    import os
    import torch
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, int(n if n is not None else os.environ.get("AGB_HOST_THREADS", "4"))))
    return old
