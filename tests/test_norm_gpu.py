"""Fused BatchNorm(+activation) and residual-tail HIP kernels vs plain PyTorch (fp64 reference of the same op;
reference semantics: nn.BatchNorm1d on .F, common.py:215-226 / resnet_block.py:62-73)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


ACTS = {"none": lambda z: z, "relu": F.relu, "gelu": F.gelu}


@pytest.mark.parametrize("n,c", [(1, 64), (37, 4), (5000, 64), (70001, 128), (333, 1024), (300017, 64),
                                 # narrower than one 64-channel slab: the threads regroup (csrc/norm.hip: bn_lanes)
                                 (50021, 16), (40003, 32), (9001, 48), (777, 12), (200, 8)])
@pytest.mark.parametrize("act", ["none", "relu", "gelu"])
def test_bn_act_training(device, n, c, act):
    from dpcr_agb_amd.norm_ops import batch_norm_act
    torch.manual_seed(n + c)
    # a large common offset makes single-pass E[x^2]-E[x]^2 statistics fail; Chan's combine must not
    # ReLU's derivative jumps at z = 0: keep the offset small there so fp32-vs-fp64 sign flips of z cannot happen
    x = (torch.randn(n, c) * 0.5 + (0.5 if act == "relu" else 30.0) * torch.randn(1, c)).float()
    bn = torch.nn.BatchNorm1d(c, momentum=0.1)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = torch.nn.BatchNorm1d(c, momentum=0.1).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bn = bn.to(device).train()
    xg = x.to(device).requires_grad_(True)
    g = torch.randn(n, c)
    if n == 1:
        ref.eval(), bn.eval()  # batch statistics of one row are degenerate: exercise the eval path instead
    y = batch_norm_act(xg, bn, act)
    y.backward(g.to(device))
    xr = x.double().requires_grad_(True)
    yr = ACTS[act](ref(xr))
    yr.backward(g.double())
    assert rel(y, yr) < 5e-5  # fp32 input with |mean| ~ 60 sigma: x - mean alone costs ~6 bits
    if act == "relu":  # elements within rounding distance of the kink are excluded from the gradient check
        xd = x.double()
        z = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + ref.eps) * ref.weight.detach() + ref.bias.detach()
        keep = (z.abs() > 1e-4)
        assert rel(xg.grad.cpu() * keep, xr.grad * keep) < 2e-4
    else:
        assert rel(xg.grad, xr.grad) < 2e-4
    if act == "relu":
        # a kink element whose sign differs between fp32 and fp64 moves dbeta by its |g| and dgamma by |g * xhat|:
        # allow that per channel for the elements within rounding distance of z = 0 (about two expected in 9 M)
        near = (z.abs() <= 1e-4).sum(0).double()
        slack = near * float(g.abs().max())
        db, dw = bn.bias.grad.detach().double().cpu(), bn.weight.grad.detach().double().cpu()
        assert ((db - ref.bias.grad).abs() <= 2e-4 * ref.bias.grad.abs().max() + slack).all()
        assert ((dw - ref.weight.grad).abs() <= 2e-4 * ref.weight.grad.abs().max() + 6 * slack).all()
    else:
        assert rel(bn.weight.grad, ref.weight.grad) < 2e-4
        assert rel(bn.bias.grad, ref.bias.grad) < 2e-4
    if n > 1:
        assert rel(bn.running_mean, ref.running_mean) < 1e-5
        assert rel(bn.running_var, ref.running_var) < 1e-4
        assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("training", [True, False])
def test_bn_backward_column_sums(device, training):
    """The column sums of dx handed to the preceding convolution as its bias gradient (closed form in the fold kernel)
    equal the actual sums of dx: gamma * rstd * dbeta with running statistics, zero (up to the rounding of the dx
    values themselves) with batch statistics."""
    from dpcr_agb_amd.norm_ops import batch_norm_act
    torch.manual_seed(11)
    n, c = 20011, 64
    bn = torch.nn.BatchNorm1d(c).to(device)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    x = (torch.randn(n, c, device=device) * 2 + 1).requires_grad_(True)
    captured = {}
    def keep(g):
        hint = getattr(g, "agb_colsum", None)     # (column sums, tensor version the hint is valid for)
        captured["colsum"] = None if hint is None else hint[0]
        captured["version_ok"] = hint is not None and hint[1] == g._version

    x.register_hook(keep)
    y = batch_norm_act(x, bn, "gelu")
    y.backward(torch.randn(n, c, device=device))
    cs = captured["colsum"]
    assert cs is not None and cs.shape == (c,) and captured["version_ok"]
    true = x.grad.double().sum(0)
    scale = float(x.grad.double().abs().sum(0).max())
    assert float((cs.double() - true).abs().max()) < 1e-5 * scale
    if training:
        assert float(cs.abs().max()) == 0.0


@pytest.mark.parametrize("momentum", [0.1, None])
def test_bn_running_stat_bookkeeping(device, momentum):
    """num_batches_tracked is bumped on the device by the fold kernel (momentum given) or on the host (cumulative
    average, momentum None); running statistics follow nn.BatchNorm1d over several steps either way."""
    from dpcr_agb_amd.norm_ops import batch_norm_act
    torch.manual_seed(3)
    bn = torch.nn.BatchNorm1d(16, momentum=momentum).to(device)
    ref = torch.nn.BatchNorm1d(16, momentum=momentum).double()
    for i in range(3):
        x = torch.randn(500 + 7 * i, 16, device=device) * (1 + i) + i
        batch_norm_act(x, bn, None)
        ref(x.double().cpu())
        assert int(bn.num_batches_tracked) == i + 1
    assert rel(bn.running_mean, ref.running_mean) < 1e-5
    assert rel(bn.running_var, ref.running_var) < 1e-4
    bn.eval()
    batch_norm_act(x, bn, None)
    assert int(bn.num_batches_tracked) == 3


@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_add_act(device, act):
    from dpcr_agb_amd.norm_ops import ACT_IDS, AddActFunction
    torch.manual_seed(3)
    n, c, B = 4097, 64, 5
    a, r = torch.randn(n, c), torch.randn(n, c)
    batch = torch.sort(torch.randint(0, B, (n,))).values
    coords = torch.zeros(n, 4, dtype=torch.int32)
    coords[:, 0] = batch.int()
    scale = torch.tensor([1 / 0.9, 0.0, 1 / 0.9, 1 / 0.9, 0.0])
    g = torch.randn(n, c)
    for use_scale in (False, True):
        ag, rg = a.to(device).requires_grad_(True), r.to(device).requires_grad_(True)
        y = AddActFunction.apply(ag, rg, scale.to(device) if use_scale else None, coords.to(device), ACT_IDS[act])
        y.backward(g.to(device))
        ar, rr = a.double().requires_grad_(True), r.double().requires_grad_(True)
        s = scale.double()[batch].unsqueeze(1) if use_scale else 1.0
        yr = ACTS[act](ar * s + rr)
        yr.backward(g.double())
        assert rel(y, yr) < 1e-6
        assert rel(ag.grad, ar.grad) < 1e-5
        assert rel(rg.grad, rr.grad) < 1e-5


def test_prefetched_input_gives_identical_step(device):
    """Building the coordinate plan on a side stream ahead of time must not change a single bit of the step."""
    import random
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    batch = synthetic.make_sparse_batch([11, 12, 13], n_points=2000).to(device)
    outs = []
    for prefetch in (False, True):
        torch.manual_seed(0)
        random.seed(5)
        model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds).to(device).train()
        if prefetch:
            model.prefetch_input(batch, device)
        model.set_input(batch, device)
        model.forward()
        model.loss.backward()
        torch.cuda.synchronize()
        outs.append((model.output.detach().clone(), model.model.blocks[1][0].conv1.kernel.grad.detach().clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    # weight gradients use float atomics across row chunks: equal up to summation order
    assert rel(outs[0][1], outs[1][1]) < 1e-5


@pytest.mark.parametrize("act", ["relu", "gelu"])
@pytest.mark.parametrize("B,C", [(32, 64), (5, 512), (1, 2048), (32, 2048), (32, 100)])
def test_se_excitation_mlp(device, act, B, C):
    """Fused squeeze-excite MLP (csrc/se.hip) vs nn.Linear -> act -> nn.Linear -> Sigmoid in fp64
    (senet_block.py:35-42), forward and every gradient."""
    from dpcr_agb_amd.se_ops import se_excite
    torch.manual_seed(B * 7 + C)
    H = max(C // 16, 1)
    l1, l2 = torch.nn.Linear(C, H), torch.nn.Linear(H, C)
    r1, r2 = torch.nn.Linear(C, H).double(), torch.nn.Linear(H, C).double()
    r1.load_state_dict({k: v.double() for k, v in l1.state_dict().items()})
    r2.load_state_dict({k: v.double() for k, v in l2.state_dict().items()})
    l1, l2 = l1.to(device), l2.to(device)
    p = torch.randn(B, C)
    g = torch.randn(B, C)
    pg = p.to(device).requires_grad_(True)
    s = se_excite(pg, l1, act, l2)
    s.backward(g.to(device))
    pr = p.double().requires_grad_(True)
    sr = torch.sigmoid(r2(ACTS[act](r1(pr))))
    sr.backward(g.double())
    assert rel(s, sr) < 1e-5
    assert rel(pg.grad, pr.grad) < 1e-4
    for a, b in ((l1.weight, r1.weight), (l1.bias, r1.bias), (l2.weight, r2.weight), (l2.bias, r2.bias)):
        assert rel(a.grad, b.grad) < 1e-4


@pytest.mark.parametrize("act,C", [("relu", 64), ("gelu", 128)])
def test_se_layer_one_node(device, act, C):
    """The whole SELayer (average pooling -> fc -> broadcast multiplication, senet_block.py:33-50) as one autograd node
    vs the same composition in fp64: output, input gradient (both branches summed in the fused kernel) and the four
    parameter gradients.  Ragged batch: plots of different sizes, one of them a single row."""
    from dpcr_agb_amd.se_ops import se_layer
    torch.manual_seed(C)
    sizes = [700, 1, 1300, 257]
    B, n = len(sizes), sum(sizes)
    batch = torch.cat([torch.full((s,), i, dtype=torch.int32) for i, s in enumerate(sizes)])
    coords = torch.zeros(n, 4, dtype=torch.int32)
    coords[:, 0] = batch
    ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32)
    H = C // 16
    l1, l2 = torch.nn.Linear(C, H), torch.nn.Linear(H, C)
    r1, r2 = torch.nn.Linear(C, H).double(), torch.nn.Linear(H, C).double()
    r1.load_state_dict({k: v.double() for k, v in l1.state_dict().items()})
    r2.load_state_dict({k: v.double() for k, v in l2.state_dict().items()})
    l1, l2 = l1.to(device), l2.to(device)
    x, g = torch.randn(n, C), torch.randn(n, C)
    xg = x.to(device).requires_grad_(True)
    out = se_layer(xg, coords.to(device), ptr.to(device), B, l1, act, l2)
    out.backward(g.to(device))
    xr = x.double().requires_grad_(True)
    pooled = torch.stack([xr[ptr[i]:ptr[i + 1]].mean(0) for i in range(B)])
    sr = torch.sigmoid(r2(ACTS[act](r1(pooled))))
    outr = xr * sr[batch.long()]
    outr.backward(g.double())
    assert rel(out, outr) < 1e-5
    assert rel(xg.grad, xr.grad) < 1e-4
    for a, b in ((l1.weight, r1.weight), (l1.bias, r1.bias), (l2.weight, r2.weight), (l2.bias, r2.bias)):
        assert rel(a.grad, b.grad) < 1e-4


@pytest.mark.parametrize("n,cin,cout", [(70001, 128, 1024), (5000, 64, 256), (3001, 256, 64), (130, 32, 128)])
def test_bn_statistics_from_the_product_epilogue(device, n, cin, cout):
    """Linear -> BatchNorm in training: the dense product leaves (count, mean, M2) per row tile and column in its epilogue
    (agb_dense_fwd_bn) and the BatchNorm folds those instead of reading the layer output again — same output, same
    running statistics, same gradients as the separate statistics pass and as torch in fp64."""
    from dpcr_agb_amd import _lib, sparse_ops
    from dpcr_agb_amd.norm_ops import batch_norm_act
    from dpcr_agb_amd.sparse_ops import dense_linear
    torch.manual_seed(n + cout)
    x = torch.randn(n, cin) + 3.0
    lin = torch.nn.Linear(cin, cout)
    bn0 = torch.nn.BatchNorm1d(cout, momentum=0.1)
    g = torch.randn(n, cout)
    res = {}
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(cout, momentum=0.1)
        bn.load_state_dict(bn0.state_dict())
        bn = bn.to(device).train()
        w, b = lin.weight.detach().clone().to(device).requires_grad_(True), lin.bias.detach().clone().to(device)
        calls = []
        orig = _lib.call
        _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            xg = x.to(device).requires_grad_(True)
            with sparse_ops.KernelOptions(bn_stats_in_epilogue=fused):
                z = dense_linear(xg, w, b)
                y = batch_norm_act(z, bn, "gelu")   # (smooth: a ReLU kink element flipping between the two roundings moves its gradient)
            y.backward(g.to(device))
        finally:
            _lib.call = orig
        assert ("agb_bn_stats_fold" in calls) == fused and ("agb_bn_stats_tracked" in calls) == (not fused), calls
        res[fused] = (y.detach(), bn.running_mean.clone(), bn.running_var.clone(), xg.grad.clone(), w.grad.clone())
    for a, b_ in zip(res[True], res[False]):
        assert rel(a, b_) < 1e-5      # (two fp32 roundings of the same statistics)
    ref = torch.nn.BatchNorm1d(cout, momentum=0.1).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn0.state_dict().items()})
    yr = F.gelu(ref(x.double() @ lin.weight.detach().double().t() + lin.bias.detach().double()))
    assert rel(res[True][0], yr) < 1e-4
    assert rel(res[True][1], ref.running_mean) < 1e-5 and rel(res[True][2], ref.running_var) < 1e-4


@pytest.mark.parametrize("n,c", [(50021, 64), (4099, 16), (333, 256)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_add_act_fused(device, n, c, training):
    """act(BatchNorm(z) + r) as one node (the tail of KPConv's bottleneck blocks; csrc/norm.hip k_tail_* without an
    excitation) against batch_norm_act + the residual kernel and against torch in fp64."""
    from dpcr_agb_amd.norm_ops import ACT_IDS, AddActFunction, batch_norm_act, batch_norm_add_act
    torch.manual_seed(n + c)
    z0, r0, g = torch.randn(n, c) * 0.7 + 2.0, torch.randn(n, c), torch.randn(n, c)
    bn0 = torch.nn.BatchNorm1d(c, momentum=0.1)
    with torch.no_grad():
        bn0.weight.uniform_(0.5, 1.5), bn0.bias.uniform_(-0.5, 0.5)
        bn0.running_mean.uniform_(1.5, 2.5), bn0.running_var.uniform_(0.3, 0.8)
    res = {}
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(c, momentum=0.1)
        bn.load_state_dict(bn0.state_dict())
        bn = bn.to(device).train(training)
        z, r = z0.to(device).requires_grad_(True), r0.to(device).requires_grad_(True)
        if fused:
            y = batch_norm_add_act(z, r, bn, "gelu")
        else:
            y = AddActFunction.apply(batch_norm_act(z, bn, None), r, None, None, ACT_IDS["gelu"])
        y.backward(g.to(device))
        res[fused] = (y.detach(), z.grad.clone(), r.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                      bn.running_mean.clone(), bn.running_var.clone())
    for a, b in zip(res[True], res[False]):
        assert rel(a, b) < 2e-5
    ref = torch.nn.BatchNorm1d(c, momentum=0.1).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn0.state_dict().items()})
    ref.train(training)
    zr, rr = z0.double().requires_grad_(True), r0.double().requires_grad_(True)
    yr = F.gelu(ref(zr) + rr)
    yr.backward(g.double())
    assert rel(res[True][0], yr) < 5e-5
    assert rel(res[True][1], zr.grad) < 2e-4 and rel(res[True][2], rr.grad) < 2e-4
    assert rel(res[True][3], ref.weight.grad) < 2e-4 and rel(res[True][4], ref.bias.grad) < 2e-4
