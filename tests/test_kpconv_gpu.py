"""GPU parity of the KPConv path: fused neighbourhood gather + GEMM, pooled shortcut, KPCNN wiring and the on-device
input pyramid, against golden vectors produced by the reference's own Python (tests/golden/kpconv_layer_golden.npz)
and against the pinned oracle.  fp32 results within 1e-4 relative; indices bit-exact."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import kpconv_index as K
from oracle import kpconv_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kpconv_layer_golden.npz")
RTOL = 1e-4


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def D(a, device, grad=False):
    t = torch.from_numpy(np.asarray(a)).to(device)
    return t.requires_grad_(True) if grad else t


def test_kpconv_layer_matches_reference(device, g):
    from dpcr_agb_amd.backbones.kpconv import KPConv
    conv = KPConv(15, 3, 8, 12, float(g["L_ext"]), 0.08).to(device)
    with torch.no_grad():
        conv.weights.copy_(D(g["L_w"], device))
        conv.kernel_points.copy_(D(g["L_kp"], device))
    for tag, q, idx in (("L", g["points0"], g["neighbors0"]), ("S", g["points1"], g["pools0"])):
        conv.zero_grad()
        x = D(g[f"{tag}_x"], device, True)
        y = conv(D(q, device), D(g["points0"], device), D(idx, device), x)
        y.backward(D(g[f"{tag}_g"], device))
        assert rel(y, g[f"{tag}_y"]) < RTOL
        assert rel(x.grad, g[f"{tag}_dx"]) < RTOL
        assert rel(conv.weights.grad, g[f"{tag}_dw"]) < RTOL
    # the reference hands int64 index matrices: accepted as well
    y64 = conv(D(g["points0"], device), D(g["points0"], device), D(g["neighbors0"], device).long(), D(g["L_x"], device))
    assert rel(y64, g["L_y"]) < RTOL


def _is_symmetric(nb, ns):
    pairs = {(i, int(j)) for i, row in enumerate(nb) for j in row if j < ns}
    return all((j, i) in pairs for i, j in pairs)


@pytest.mark.parametrize("cin,cout", [(16, 12), (16, 16), (32, 32)])
def test_kpconv_symmetric_backward(device, g, cin, cout):
    """A layer on ONE point set with a symmetric neighbour relation runs its backward without the scatter (the forward
    gather on dy with mirrored kernel points): same output and gradients as an fp64 evaluation of the reference formula,
    whichever form runs."""
    import dpcr_agb_amd.backbones.kpconv as KB
    from dpcr_agb_amd import _lib
    from dpcr_agb_amd.kpconv_ops import KPConvSymmetricFunction
    assert _is_symmetric(g["neighbors0"], len(g["points0"]))      # the reference's own radius search, uncropped
    rng = np.random.default_rng(5)
    conv = KB.KPConv(15, 3, cin, cout, float(g["L_ext"]), 0.08).to(device)
    with torch.no_grad():
        conv.kernel_points.copy_(D(g["L_kp"], device))
    assert KPConvSymmetricFunction.supported(15, cin, cout)
    pts = D(g["points0"], device)
    xs = rng.standard_normal((len(g["points0"]), cin)).astype(np.float32)
    gy = rng.standard_normal((len(g["points0"]), cout)).astype(np.float32)
    res = {}
    for form in ("scatter", "symmetric"):
        idx = D(g["neighbors0"], device)
        idx.agb_symmetric = form != "scatter"
        calls = []
        orig = _lib.call
        _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            conv.zero_grad()
            x = D(xs, device, True)
            y = conv(pts, pts, idx, x)
            y.backward(D(gy, device))
        finally:
            _lib.call = orig
        assert ("agb_kpconv_gather_bwd" in calls) == (form == "scatter"), calls
        res[form] = (y.detach(), x.grad.clone(), conv.weights.grad.clone())
    # fp64 evaluation of the reference formula
    xr = torch.from_numpy(xs).double().requires_grad_(True)
    wr = conv.weights.detach().cpu().double().requires_grad_(True)
    yr = R.kpconv(torch.from_numpy(g["points0"]).double(), torch.from_numpy(g["points0"]).double(),
                  torch.from_numpy(g["neighbors0"]).long(), xr, torch.from_numpy(g["L_kp"]).double(), wr, float(g["L_ext"]))
    yr.backward(torch.from_numpy(gy).double())
    for form, (y, dx, dw) in res.items():
        assert rel(y, yr) < RTOL, form
        assert rel(dx, xr.grad) < RTOL, form
        assert rel(dw, wr.grad) < RTOL, form



def _ragged(nb, ns, device):
    """kp_index.Neighbors (ragged rows) of a padded neighbour matrix (entries >= ns are padding)."""
    from dpcr_agb_amd import kp_index
    nb = np.asarray(nb)
    counts = (nb < ns).sum(1).astype(np.int32)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    indices = np.concatenate([row[:c] for row, c in zip(nb, counts)]).astype(np.int32)
    mc = int(counts.max())
    out = kp_index.Neighbors(D(row_ptr, device), D(indices, device), len(nb), ns, mc,
                             torch.tensor([mc], dtype=torch.int32, device=device))
    out.agb_symmetric = True
    return out


@pytest.mark.parametrize("cin", [16, 32])
def test_kpconv_fused_layer_matches_reference(device, g, cin):
    """csrc/kpfused.hip (gather + influence + kernel-weight contraction in ONE kernel per direction, wf never in HBM) on the
    reference's own uncropped radius search: output, dx and dW against an fp64 evaluation of the reference formula
    (blocks.py:304-400) at 1e-4 and against the two-kernel form of this library at 1e-5; bitwise repeatable."""
    import dpcr_agb_amd.backbones.kpconv as KB
    from dpcr_agb_amd import _lib
    from dpcr_agb_amd.sparse_ops import KernelOptions
    ns = len(g["points0"])
    assert _is_symmetric(g["neighbors0"], ns)
    rng = np.random.default_rng(7)
    conv = KB.KPConv(15, 3, cin, cin, float(g["L_ext"]), 0.08).to(device)
    with torch.no_grad():
        conv.kernel_points.copy_(D(g["L_kp"], device))
    pts = D(g["points0"], device)
    xs = rng.standard_normal((ns, cin)).astype(np.float32)
    gy = rng.standard_normal((ns, cin)).astype(np.float32)
    res = {}
    for form in ("two-kernel", "fused", "fused again"):
        idx = _ragged(g["neighbors0"], ns, device)
        calls = []
        orig = _lib.call
        _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            with KernelOptions(fused_kpconv=form != "two-kernel"):
                conv.zero_grad()
                x = D(xs, device, True)
                y = conv(pts, pts, idx, x)
                y.backward(D(gy, device))
        finally:
            _lib.call = orig
        fused = [c for c in calls if c.startswith("agb_kpconv_fused")]
        assert fused == (["agb_kpconv_fused_fwd", "agb_kpconv_fused_bwd"] if form != "two-kernel" else []), calls
        assert ("agb_kpconv_gather_fwd_csr" in calls) == (form == "two-kernel"), calls
        res[form] = (y.detach().clone(), x.grad.clone(), conv.weights.grad.clone())
    xr = torch.from_numpy(xs).double().requires_grad_(True)
    wr = conv.weights.detach().cpu().double().requires_grad_(True)
    p64 = torch.from_numpy(g["points0"]).double()
    yr = R.kpconv(p64, p64, torch.from_numpy(g["neighbors0"]).long(), xr, torch.from_numpy(g["L_kp"]).double(), wr, float(g["L_ext"]))
    yr.backward(torch.from_numpy(gy).double())
    for form, (y, dx, dw) in res.items():
        assert rel(y, yr) < RTOL and rel(dx, xr.grad) < RTOL and rel(dw, wr.grad) < RTOL, form
    for a, b in zip(res["fused"], res["two-kernel"]):
        assert rel(a, b) < 1e-5
    assert all(torch.equal(a, b) for a, b in zip(res["fused"], res["fused again"]))


@pytest.mark.parametrize("n,radius", [(1, 0.3), (37, 0.5), (700, 0.45)])
def test_kpconv_fused_edge_shapes(device, n, radius):
    """Rows with more than 64 neighbours (a second pass over the row), tiles with fewer than 16 rows, a single point; only the
    input gradient / only the weight gradient wanted."""
    from dpcr_agb_amd import kp_index
    from dpcr_agb_amd.kpconv_ops import KPConvFusedFunction, KPConvSymmetricFunction
    from dpcr_agb_amd.sparse_ops import current
    torch.manual_seed(n)
    pts = torch.rand(n, 3, device=device)
    lens = np.array([n], dtype=np.int64)
    nb = kp_index.batch_neighbors_ragged(pts, pts, lens, lens, radius)
    nb.agb_symmetric = True
    if n == 700:
        assert nb.max_count > 64
    kp = (torch.rand(15, 3, device=device) - 0.5) * radius
    for C in (16, 32):
        w = torch.randn(15, C, C, device=device) * 0.1
        x0, gy = torch.randn(n, C, device=device), torch.randn(n, C, device=device)
        assert KPConvFusedFunction.supported(15, C, C, nb, current())
        out = {}
        for name, fn in (("ref", KPConvSymmetricFunction), ("fused", KPConvFusedFunction)):
            x, wt = x0.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = fn.apply(x, pts, nb, kp, 0.6 * radius, wt)
            y.backward(gy)
            out[name] = (y.detach(), x.grad, wt.grad)
        for a, b in zip(out["fused"], out["ref"]):
            assert rel(a, b) < 1e-5, (n, C)
        # one gradient only
        x, wt = x0.clone().requires_grad_(True), w.clone()
        KPConvFusedFunction.apply(x, pts, nb, kp, 0.6 * radius, wt).backward(gy)
        # (the launch without a weight gradient may split the reduction over another number of waves: rounding, not bits)
        assert rel(x.grad, out["fused"][1]) < 2e-6
        x, wt = x0.clone(), w.clone().requires_grad_(True)
        KPConvFusedFunction.apply(x, pts, nb, kp, 0.6 * radius, wt).backward(gy)
        assert torch.equal(wt.grad, out["fused"][2])

@pytest.mark.parametrize("n,cin,cout", [(20000, 64, 16), (20000, 128, 32), (5000, 256, 64), (300, 1024, 256), (1, 16, 16)])
def test_linear_join_adds_the_branch_gradient_in_the_data_gradient_kernel(device, n, cin, cout):
    """DenseLinearFunction's join form (the input of a bottleneck block feeds unary1 and the shortcut, blocks.py:640-668):
    dx = d(branch) + dy @ W from one kernel — against the fp64 value, and bit for bit against a separate addition where
    both forms run the same product kernel (every shape here but the 16 -> 64 data gradient of many rows, which the
    streaming kernel takes when no addend comes with it)."""
    from dpcr_agb_amd import sparse_ops as so
    gen = torch.Generator().manual_seed(n + cin)
    x0 = torch.randn(n, cin, generator=gen)
    w = (torch.randn(cout, cin, generator=gen) / cin ** 0.5).to(device).requires_grad_(True)
    m1, m2 = torch.randn(n, cout, generator=gen).to(device), torch.randn(n, cin, generator=gen).to(device)

    def run(join):
        x = x0.to(device).requires_grad_(True)
        w.grad = None
        with so.KernelOptions(join_dgrad=join):
            y, branch = so.dense_linear_join(x, w)
            assert (branch is not x) == join       # (the join form hands out an alias that carries its autograd node)
            ((y * m1).sum() + (torch.tanh(branch) * m2).sum()).backward()
        return y.detach(), x.grad, w.grad.clone()

    ya, dxa, dwa = run(True)
    yb, dxb, dwb = run(False)
    assert torch.equal(ya, yb) and rel(dwa, dwb) < 1e-5      # (the weight gradient of many rows sums with atomics)
    xd, wd = x0.double().to(device), w.detach().double()
    want = (1 - torch.tanh(xd) ** 2) * m2.double() + m1.double() @ wd
    assert rel(dxa, want) < 2e-6 and rel(dxb, want) < 2e-6
    if not (n >= 16384 and cout == 16):
        assert torch.equal(dxa, dxb)


def test_pool_helpers_match_reference(device, g):
    from dpcr_agb_amd.backbones.kpconv import GlobalSumBlock
    from dpcr_agb_amd.kpconv_ops import KPMaxPoolFunction
    x = D(g["P_x"], device, True)
    y = KPMaxPoolFunction.apply(x, D(g["pools0"], device))
    assert np.array_equal(y.detach().cpu().numpy(), g["P_maxpool"])
    y.sum().backward()
    xr = torch.from_numpy(g["P_x"]).double().requires_grad_(True)
    R.max_pool(xr, torch.from_numpy(g["pools0"]).long()).sum().backward()
    assert rel(x.grad, xr.grad) < 1e-6
    batch = types.SimpleNamespace(lengths=[torch.from_numpy(g["lens0"].astype(np.int64))])
    gs = GlobalSumBlock()(D(g["P_x"], device), batch)
    assert rel(gs, g["P_globalsum"]) < 1e-5


def _mini_cfg():
    from dpcr_agb_amd.config import kpconv_config
    cfg = kpconv_config(in_features_dim=3, first_subsampling_dl=0.032)
    cfg["first_features_dim"] = 16
    cfg["architecture"] = ["simple", "resnetb", "resnetb_strided", "resnetb", "global_sum"]
    return cfg


def test_kpcnn_end_to_end_matches_reference(device, g):
    from dpcr_agb_amd.backbones.kpconv import KPCNN
    net = KPCNN(_mini_cfg())
    sd = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("N_sd/")}
    net.load_state_dict(sd, strict=True)     # the reference's state_dict loads unchanged
    net.to(device).train()
    batch = types.SimpleNamespace(
        features=D(g["N_feats"], device), points=[D(g["points0"], device), D(g["points1"], device)],
        neighbors=[D(g["neighbors0"], device), D(g["neighbors1"], device)], pools=[D(g["pools0"], device)],
        lengths=[torch.from_numpy(g["lens0"].astype(np.int64)), torch.from_numpy(g["lens1"].astype(np.int64))])
    y = net(batch)
    y.backward(D(g["N_g"], device))
    assert rel(y, g["N_y"]) < RTOL
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("N_grad/"))
    worst, worst_name = 0.0, None
    for k in g.files:
        if k.startswith("N_grad/"):
            p = dict(net.named_parameters())[k[7:]]
            e = float(np.abs(p.grad.cpu().double().numpy() - g[k]).max()) / max(float(np.abs(g[k]).max()), 1e-3 * gmax)
            if e > worst:
                worst, worst_name = e, k
        if k.startswith("N_after/"):
            assert rel(net.state_dict()[k[8:]], g[k]) < 1e-4, k
    # measured floor: the reference's fp32 gradients (the golden vectors) sit 1.3e-6 from an fp64 evaluation of the same
    # network (oracle/kpconv_ref.py in double), the fp32 oracle 2.4e-6 — the 1e-4 bar applies to every tensor
    print(f"KPCNN: worst gradient rel err {worst:.2e} ({worst_name})")
    assert worst < RTOL, (worst, worst_name)


def test_input_pyramid_and_network_vs_oracle(device):
    """prepare_inputs on the GPU (5 levels, reference configuration) vs the pinned index oracle with the same
    grid orientations; then the full KPConv model output vs the layer oracle on that pyramid."""
    from dpcr_agb_amd import kp_index, synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import KPConvModel
    torch.manual_seed(0)
    np.random.seed(3)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    opt = Opt(MODEL_OPTIONS["KPConv"])
    model = KPConvModel(opt, "kpconv", ds).to(device).train()
    b = synthetic.make_point_batch([40, 41], n_points=3000)
    lens = np.bincount(b.batch.numpy()).astype(np.int64)
    rots = [kp_index.random_grid_rotations(2) for _ in range(4)]
    inp = model.prepare_inputs(b.pos, b.x, lens, device, rotations=rots)
    # oracle pyramid
    cfg = opt.config
    pts, ln, r = b.pos.numpy(), lens, cfg.first_subsampling_dl * cfg.conv_radius
    for lvl in range(5):
        assert np.array_equal(inp["points"][lvl].cpu().numpy(), pts)
        assert np.array_equal(inp["neighbors"][lvl].cpu().numpy(), K.batch_neighbors(pts, pts, ln, ln, r))
        if lvl == 4:
            break
        rot = pts.copy()
        i0 = 0
        for bi, n in enumerate(ln):
            rot[i0:i0 + n] = np.sum(np.expand_dims(pts[i0:i0 + n], 2) * rots[lvl][bi], axis=1)
            i0 += n
        sp, sb = K.batch_grid_subsampling(rot, ln, sampleDl=2 * r / cfg.conv_radius, order="canonical")
        i0 = 0
        for bi, n in enumerate(sb):
            sp[i0:i0 + n] = np.sum(np.expand_dims(sp[i0:i0 + n], 2) * rots[lvl][bi].T, axis=1)
            i0 += n
        assert K.same_up_to_ties(inp["pools"][lvl].cpu().numpy(), K.batch_neighbors(sp, pts, sb, ln, r), sp, pts)
        pts, ln, r = sp, sb.astype(np.int64), r * 2
    # network on that pyramid vs the oracle (fp64)
    from dpcr_agb_amd.config import Opt as O
    out = model.model(O(inp))
    sd = {k: v.detach().cpu().double() for k, v in model.model.state_dict().items()}
    ob = dict(features=b.x.double(), points=[p.cpu().double() for p in inp["points"]],
              neighbors=[n.cpu().long() for n in inp["neighbors"]], pools=[p.cpu().long() for p in inp["pools"]],
              lengths=[l.numpy() for l in inp["lengths"]])
    ocfg = dict(first_subsampling_dl=cfg.first_subsampling_dl, conv_radius=cfg.conv_radius, KP_extent=cfg.KP_extent,
                in_features_dim=3, first_features_dim=cfg.first_features_dim, architecture=list(cfg.architecture),
                batch_norm_momentum=cfg.batch_norm_momentum)
    ref = R.kpcnn_forward(sd, ocfg, ob, training=True)
    assert rel(out, ref) < RTOL
    # every self-search of the pyramid is marked symmetric (and is); the scatter-free backward they select gives the
    # same parameter gradients as the scatter form
    for lvl, nb in enumerate(inp["neighbors"]):
        assert nb.agb_symmetric
        if lvl >= 3:
            assert _is_symmetric(nb.cpu().numpy(), len(inp["points"][lvl]))
    from dpcr_agb_amd.sparse_ops import KernelOptions
    grads = {}
    gy = torch.randn_like(out)
    for form in ("scatter", "symmetric", "fused"):
        for nb in inp["neighbors"]:
            nb.agb_symmetric = form != "scatter"
        model.model.zero_grad()
        with KernelOptions(fused_kpconv=form == "fused"):
            o = model.model(O(inp))
            assert rel(o, ref) < RTOL, form
            o.backward(gy)
        grads[form] = {k: p.grad.clone() for k, p in model.model.named_parameters() if p.grad is not None}
    gmax = max(float(v.abs().max()) for v in grads["scatter"].values())

    def worst(a, b):
        return max(float((b[k] - v).abs().max()) / max(float(v.abs().max()), 1e-3 * gmax) for k, v in a.items())
    # same forward kernels, different backward: the two forms agree to rounding
    assert worst(grads["scatter"], grads["symmetric"]) < RTOL
    # The one-kernel layers (csrc/kpfused.hip) sum the forward in another order.  Through 15 blocks of batch-normalised fp32
    # layers (a handful of rows on the last level) the parameter gradients of EVERY form sit 2e-2 .. 6e-2 from the fp64
    # gradients of the oracle in the max-norm above (and move by that much between two runs of the atomics of the strided
    # layers); the fused form must be on that same floor — measured as the mean relative L2 error over the parameter tensors.
    # (The layer itself is held to 1e-4 / 1e-5 in test_kpconv_fused_layer_matches_reference.)
    sd64 = {k: v.detach().cpu().double() for k, v in model.model.state_dict().items()}
    names = dict(model.model.named_parameters())
    for k in sd64:
        if k in names:
            sd64[k].requires_grad_(True)
    ref64 = R.kpcnn_forward(sd64, ocfg, ob, training=True)
    ref64.backward(gy.detach().cpu().double())
    og = {k: v.grad.float().to(device) for k, v in sd64.items() if v.grad is not None and k in grads["scatter"]}
    assert len(og) > 100

    def mean_l2(a, b):
        return float(np.mean([float((b[k] - v).norm() / (v.norm() + 1e-3 * gmax)) for k, v in a.items()]))
    floor = mean_l2(og, grads["scatter"])
    assert mean_l2(og, grads["fused"]) < 3.0 * floor + 1e-3, (mean_l2(og, grads["fused"]), floor)


def test_kpconv_training_step(device):
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import KPConvModel
    torch.manual_seed(0)
    np.random.seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds).to(device).train()
    model.init_train_objects(TRAINING_NFI)
    batch = synthetic.make_point_batch([1, 2, 3], n_points=2048)
    losses = []
    for _ in range(3):
        model.set_input(batch, device)
        model.optimize_parameters(epoch=0, batch_size=3, num_batches=10)
        losses.append(float(model.loss.detach()))
    assert all(np.isfinite(losses))
    assert model.get_reg_output().shape == (3, 2)


def test_pyramid_of_plots_far_apart(device):
    """Plots of one batch at different WORLD positions (not centred): the randomly oriented grid subsampling sizes its cells
    from the largest single-plot diagonal, not from the batch's extent (which made B * cells exceed the grid budget and
    raised 'sampleDl too small'); the pyramid equals that of the same plots moved together, index for index."""
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import KPConvModel
    b = synthetic.make_point_batch([21, 22, 23], n_points=1500)
    lens = np.bincount(b.batch.numpy()).astype(np.int32)
    ds = synthetic.SyntheticDataset(stat_seeds=range(0, 8))
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds).to(device).train()
    np.random.seed(3)
    from dpcr_agb_amd import kp_index
    rots = [kp_index.random_grid_rotations(3) for _ in range(4)]
    pos = torch.round(b.pos * 16384.0) / 16384.0     # 14 fractional bits: the shifts below are exact in fp32
    near = model.prepare_inputs(pos, b.x, lens, device, rotations=rots)
    shift = torch.zeros_like(pos)
    shift[b.batch == 1, 0] = 64.0
    shift[b.batch == 2, 1] = -128.0
    far = model.prepare_inputs(pos + shift, b.x, lens, device, rotations=rots)      # (raised before the fix)
    # level 0: identical differences, identical neighbour matrix; deeper levels: the grid is laid out from the rotated cloud's
    # own corner, and a cloud rotated about the origin from another position meets it in another phase — a few cells differ
    assert torch.equal(near["neighbors"][0].padded(), far["neighbors"][0].padded())
    for lvl in range(len(near["points"])):
        a, c = near["lengths"][lvl].double(), far["lengths"][lvl].double()
        assert float((a - c).abs().max()) <= 0.1 * float(a.max()) + 2, (lvl, a, c)    # (other grid phase: a few cells differ)


def test_kpconv_layer_entry_points_of_the_survey_abi(device, g):
    """agb_kpconv_fwd / agb_kpconv_bwd (SURVEY.md 8(b) names: the whole rigid KPConv layer as one C call each, csrc/aliases.hip)
    against the golden vectors of the reference's blocks.py:264-400 on a layer whose widths the dense kernels take (Cin 8 -> K*Cin
    = 120, Cout 12); and agb_hash_build == agb_coords_insert."""
    from dpcr_agb_amd import _lib
    import dpcr_agb_amd.kpconv_ops  # noqa: F401  (declares the entry points)
    P = _lib.ptr
    q, s = D(g["points0"], device).contiguous(), D(g["points0"], device).contiguous()
    idx = D(g["neighbors0"], device).to(torch.int32).contiguous()
    x = D(g["L_x"], device).contiguous()
    W = D(g["L_w"], device).contiguous()            # [K, Cin, Cout]
    kp = D(g["L_kp"], device).contiguous()
    K, cin, cout = W.shape
    N, H = idx.shape
    Ns = x.shape[0]
    ext = float(g["L_ext"])
    wf = torch.empty(N, K * cin, device=device)
    y = torch.empty(N, cout, device=device)
    _lib.call("agb_kpconv_fwd", P(q), P(s), P(idx), H, Ns, P(x), x.stride(0), P(kp), K, ext, P(W), P(wf), P(y), cout, N, cin,
              cout, _lib.stream())
    assert rel(y, g["L_y"]) < RTOL
    dy = D(g["L_g"], device).contiguous()
    dx = torch.zeros(Ns, cin, device=device)
    dW = torch.zeros(K * cin, cout, device=device)
    nbytes = _lib.size_call("agb_kpconv_bwd_workspace_bytes", N, K, cin, cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _lib.call("agb_kpconv_bwd", P(q), P(s), P(idx), H, Ns, P(wf), P(dy), cout, P(kp), K, ext, P(W), P(dx), cin, P(dW), N, cin,
              cout, P(ws), nbytes, _lib.stream())
    assert rel(dx, g["L_dx"]) < RTOL
    assert rel(dW.view(K, cin, cout), g["L_dw"]) < RTOL
    # agb_hash_build: the same table as agb_coords_insert
    coords = torch.tensor([[0, 1, 2, 3], [0, 4, 5, 6], [1, 1, 2, 3]], dtype=torch.int32, device=device)
    cap = _lib.hash_capacity(3)
    out = []
    for name in ("agb_coords_insert", "agb_hash_build"):
        keys = torch.empty(cap, dtype=torch.int64, device=device)
        vals = torch.empty(cap, dtype=torch.int32, device=device)
        slot = torch.empty(3, dtype=torch.int32, device=device)
        status = torch.empty(4, dtype=torch.int32, device=device)
        _lib.call(name, P(coords), 3, None, P(keys), P(vals), cap, P(slot), P(status), _lib.stream())
        out.append((keys.clone(), vals.clone(), slot.clone(), status.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*out))
