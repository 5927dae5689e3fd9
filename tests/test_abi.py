"""The C-ABI library loads on a machine without a GPU and exports every symbol include/agb_hip.h declares;
the Python binding declares the same set (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "agb_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(agb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "agb_spconv_fwd" in syms and "agb_last_error" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol():
    from dpcr_agb_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} is not built: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in agb_hip.h but not exported: {missing}"


def test_binding_covers_header():
    from dpcr_agb_amd import _lib
    import dpcr_agb_amd.kp_index, dpcr_agb_amd.kpconv_ops, dpcr_agb_amd.norm_ops, dpcr_agb_amd.voxelize  # noqa: F401,E401
    import dpcr_agb_amd.sparse_ops, dpcr_agb_amd.transforms, dpcr_agb_amd.se_ops, dpcr_agb_amd.train_transforms  # noqa: F401,E401
    import dpcr_agb_amd.fused_blocks, dpcr_agb_amd.head_ops  # noqa: F401,E401
    _lib.load()
    bound = set(_lib._SIGNATURES) | {"agb_last_error", "agb_last_kernel"}
    assert set(declared_symbols()) <= bound, sorted(set(declared_symbols()) - bound)


def test_host_helpers_and_error_path():
    from dpcr_agb_amd import _lib
    assert _lib.hash_capacity(1000) == 2048
    assert _lib.hash_capacity(0) == 1024
    lib = _lib.load()
    # argument validation happens before any launch: safe without a GPU
    rc = lib.agb_hash_clear(None, None, 1000, None)
    assert rc == -1 and b"power of two" in lib.agb_last_error()
    with pytest.raises(_lib.AgbError):
        _lib.call("agb_spconv_fwd", None, 3, None, None, 0, 0, None, None, 4, 10, 27, 3, 4, None)


def test_product_path_refuses_cpu_tensors():
    import torch
    from dpcr_agb_amd import _lib
    import dpcr_agb_amd.me_compat as ME
    with pytest.raises(_lib.AgbError):
        ME.SparseTensor(torch.zeros(2, 3), coordinates=torch.tensor([[0, 0, 0, 0], [0, 1, 0, 0]]), device="cpu")
    with pytest.raises(_lib.AgbError):
        _lib.ptr(torch.zeros(3))


def test_host_helpers_without_gpu():
    """Sizing helpers and the process-level knobs are pure host code."""
    import torch
    import dpcr_agb_amd
    from dpcr_agb_amd import _lib
    import dpcr_agb_amd.norm_ops, dpcr_agb_amd.kp_index, dpcr_agb_amd.voxelize, dpcr_agb_amd.transforms  # noqa: F401,E401
    assert _lib.size_call("agb_plot_workspace_bytes", 1000, 4) > 1000 * 5 * 4
    small, big = (_lib.size_call("agb_grid_subsample_workspace_bytes", 1000, 2, c) for c in (100, 10000))
    assert big > small > 0
    assert _lib.size_call("agb_pointnet_mlp_workspace_bytes", 1000, 2, 64, 128, 1024) >= 1000 * (2 * 64 + 2 * 128 + 1024) * 4
    assert _lib.load().agb_pointnet_pool_splits(64 * 13000, 64) >= 8
    old = dpcr_agb_amd.limit_host_threads(3)
    try:
        assert torch.get_num_threads() == 3
    finally:
        torch.set_num_threads(old)
    # NUMA pinning: the GPU's local CPUs from KFD / PCI sysfs without the HIP runtime; harmless where there is no GPU
    import os
    before = os.sched_getaffinity(0)
    try:
        cpus = dpcr_agb_amd.gpu_local_cpus(0)
        assert cpus is None or (len(cpus) > 0 and all(isinstance(c, int) for c in cpus))
        got = dpcr_agb_amd.pin_to_gpu_numa_node()
        assert got is None or (got <= before and os.sched_getaffinity(0) == got)
        os.environ["AGB_NUMA_PIN"] = "0"
        assert dpcr_agb_amd.pin_to_gpu_numa_node() is None
    finally:
        os.environ.pop("AGB_NUMA_PIN", None)
        os.sched_setaffinity(0, before)
    # option validation of the per-call kernel knobs happens before any launch
    rc = _lib.load().agb_spconv_fwd_opt(None, 4, None, None, 0, 0, None, None, 4, 10, 27, 4, 4, None, None, None, 0, 1, None,
                                        7, -1, None)
    assert rc == -1 and b"cmp_mode" in _lib.load().agb_last_error()


def test_split_hints_take_zero_rows():
    """Host helpers on an empty level: no division by the (zero) tile count."""
    from dpcr_agb_amd import _lib
    L = _lib.load()
    assert L.agb_spconv_split_hint_opt(0, 27, 64, 64, 1) == 1
    assert L.agb_spconv_split_hint(0, 27, 512, 512) == 1
    assert L.agb_dense_split_hint(0, 3840, 256) == 1
    assert L.agb_dense_bn_chunks(0, 64, 64) == 0


def test_round6_entry_points_validate_before_any_launch():
    """The block-level, head and alias entry points: field table from the library itself, arena sizing on the host, argument
    errors before any launch (safe without a GPU)."""
    import ctypes
    from dpcr_agb_amd import _lib
    import dpcr_agb_amd.fused_blocks as FB
    import dpcr_agb_amd.head_ops, dpcr_agb_amd.kpconv_ops, dpcr_agb_amd.sparse_ops  # noqa: F401,E401
    L = _lib.load()
    idx = FB._fields()
    assert len(idx) == L.agb_net_field_count() >= 100
    for name in ("x", "n_out", "c1_w", "c2_nbr", "cd_tile_cls", "se_w1", "keep", "gzero", "pool_nbrT", "c1_wt"):
        assert name in idx, name
    tab = FB._Table()
    tab.set(n_in=211000, n_out=211000, B=32, has_down=0, cmp_mode=1, cmp_il=-1, se_H=4)
    tab.setp("c1_", K3=27, cin=64, cout=64)
    tab.setp("c2_", K3=27, cin=64, cout=64)
    saved, fwd, bwd = (L.agb_net_block_bytes(tab.buf, w) for w in (0, 1, 2))
    assert saved >= 3 * 211000 * 64 * 4 and fwd > 0 and bwd >= 4 * 211000 * 64 * 4      # z1, a1, z2 / dz2, dr, da1, dz1
    tab.set(has_down=1, n_in=211000, n_out=61000)
    tab.setp("c1_", cout=128)
    tab.setp("c2_", cin=128, cout=128)
    tab.setp("cd_", K3=1, cin=64, cout=128)
    assert L.agb_net_block_bytes(tab.buf, 0) >= 5 * 61000 * 128 * 4                      # + zd, r
    assert L.agb_net_block_fwd(None, None, 0, None, 0, None) == -1 and b"arenas are required" in L.agb_last_error()
    assert L.agb_net_stem_bwd(tab.buf, None, 0, None, 0, None) == -1
    # head + loss: T out of range / null arguments
    assert L.agb_reg_head_fwd(None, 512, 32, 512, 0, None, None, None, None, None, None, 1, None, None, None, None, None) == -1
    assert b"T 0" in L.agb_last_error()
    assert L.agb_reg_head_fwd(None, 512, 32, 512, 2, None, None, None, None, None, None, 9, None, None, None, None, None) == -1
    assert b"loss mask" in L.agb_last_error()
    # data gradient with an addend: the small-Cin kernels do not take one
    # (dY rows 4 floats wide = the "Cin" of the forward kernels: the stem-like small-Cin path; a non-NULL map pointer)
    rc = L.agb_spconv_bwd_data(None, 4, None, ctypes.c_void_p(64), 10, 0, None, 64, 10, 27, 64, 4, None, None, None, 0, 1, None,
                               ctypes.c_void_p(16), 64, None)
    assert rc == -1 and b"addend" in L.agb_last_error()
    assert _lib.size_call("agb_kpconv_bwd_workspace_bytes", 1000, 15, 16, 32) >= (1000 * 15 * 16 + 15 * 16 * 32) * 4
    assert L.agb_spconv_weight_transpose_batched(None, 0, 0, None) == -1
    assert L.agb_weight_twins_batched(None, 3, 10, None) == -1
    # the fused KPConv layer: coverage, workspace sizing, argument errors
    assert L.agb_kpconv_fused_supported(15, 16, 16) == 1 and L.agb_kpconv_fused_supported(15, 32, 32) == 1
    assert L.agb_kpconv_fused_supported(15, 64, 64) == 0 and L.agb_kpconv_fused_supported(15, 16, 32) == 0
    assert L.agb_kpconv_fused_supported(17, 16, 16) == 0
    ws = _lib.size_call("agb_kpconv_fused_bwd_workspace_bytes", 100000, 15, 32, 32)
    assert ws >= 15 * 32 * 32 * 4 and ws % (15 * 32 * 32 * 4) == 0          # one partial weight gradient per workgroup
    assert _lib.size_call("agb_kpconv_fused_bwd_workspace_bytes", 100000, 15, 64, 64) == 0
    f = ctypes.c_float
    assert L.agb_kpconv_fused_fwd(None, None, None, 40, 100, None, 64, None, 15, f(0.1), None, None, 64, 64, 64, None) == -1
    assert b"not covered" in L.agb_last_error()
    assert L.agb_kpconv_fused_fwd(None, None, None, 40, 100, None, 16, None, 15, f(0.1), None, None, 16, 16, 16, None) == -1
    assert b"null pointer" in L.agb_last_error()
    assert L.agb_kpconv_fused_fwd(None, None, None, 40, 1 << 24, None, 16, None, 15, f(0.1), None, None, 16, 16, 16, None) == -1
    assert L.agb_kpconv_fused_fwd(None, None, None, 40, 0, None, 16, None, 15, f(0.1), None, None, 16, 16, 16, None) == 0
    p1 = ctypes.c_void_p(256)
    assert L.agb_kpconv_fused_bwd(p1, p1, p1, 40, 100, p1, 16, p1, 15, f(0.1), p1, None, 16, None, 16, p1, 0, None, 0, 16, 16,
                                  None) == -1
    assert b"workspace" in L.agb_last_error()
