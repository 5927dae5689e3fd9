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
    pin_to_gpu_numa_node()
    return old


def gpu_local_cpus(index=0):
    """CPUs of the NUMA node the index-th visible GPU hangs off, from the KFD topology and PCI sysfs — without touching the
    HIP runtime (so that it can be used before anything creates a thread).  None when it cannot be determined."""
    import glob
    import os
    nodes = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
        try:
            with open(os.path.join(d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except (OSError, ValueError):
            continue
        if int(props.get("simd_count", "0")) > 0:
            nodes.append(props)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):       # (integer lists re-index the devices; UUID forms: give up)
        v = os.environ.get(var)
        if v:
            try:
                index = [int(t) for t in v.split(",")][index]
            except (ValueError, IndexError):
                return None
            if len(nodes) <= 1:       # (the container already shows only the permitted device)
                index = 0
    if not 0 <= index < len(nodes):
        return None
    try:
        loc, dom = int(nodes[index]["location_id"]), int(nodes[index].get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus or None
    except (OSError, ValueError, KeyError):
        return None


def pin_to_gpu_numa_node(index=None):
    """Restrict this process (the calling thread and every thread / worker process it starts afterwards) to the CPUs of its
    GPU's NUMA node: the enqueuing thread rings the GPU's doorbells and reads pinned memory the runtime places near it —
    on the two-socket MI355X hosts the host floor of a step was 5-8 % higher from the far socket (and the scheduler moves an
    unpinned process between the sockets).  index: the visible device (default LOCAL_RANK, else 0).  AGB_NUMA_PIN=0 disables.
    Returns the CPU set applied, or None."""
    import os
    if os.environ.get("AGB_NUMA_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:        # (a convenience, never a reason to stop a rank: any surprise in the topology files means "do not pin")
        if index is None:
            index = int(os.environ.get("LOCAL_RANK", "0"))
        cpus = gpu_local_cpus(index)
    except Exception:
        return None
    if not cpus:
        return None
    cpus &= os.sched_getaffinity(0)
    if not cpus:
        return None
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus


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
