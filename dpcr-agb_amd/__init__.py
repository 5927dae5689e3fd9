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
