"""How much of the 3^3 convolution kernels' time is the gather's memory locality?  Same map with every present neighbour index
folded into the first 4096 rows (all gathers hit L2) vs the real map."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpcr_agb_amd import sparse_ops, synthetic
from dpcr_agb_amd.coords import CoordinateManager
from dpcr_agb_amd.sparse_ops import spconv_forward_raw
dev = torch.device("cuda", 0)
b = synthetic.make_sparse_batch(list(range(32)))
coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
cm = CoordinateManager(coords, device=dev, batch_size=32, bounds=b.coord_bounds)
ts = 1
for ts_in, c in ((2, 64), (4, 128), (8, 256)):
    while ts < ts_in:
        cm.stride(ts, 2); ts *= 2
    n = cm.level(ts_in).n
    nbr = cm.kernel_map(ts_in, 3, 1)
    folded = torch.where(nbr >= 0, nbr % 4096, nbr)
    x = torch.randn(n, c, device=dev)
    w = torch.randn(27 * c, c, device=dev) * 0.05
    wk = w.view(27, c, c).transpose(1, 2).contiguous()
    for prec in ("fp32", "bf16"):
        sparse_ops.set_conv_precision(prec)
        for name, m in (("real", nbr), ("folded", folded)):
            f = lambda: spconv_forward_raw(x, w, m, 0, None, n, 27, c, c, w_kmajor=wk if prec != "fp32" else None)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            print(f"ts{ts_in} {c}->{c} rows {n} {prec:5s} {name:7s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
sparse_ops.set_conv_precision("fp32")
