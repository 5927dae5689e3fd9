import os, sys, subprocess
sys.path.insert(0, "/root/repo")
# each mode in its own process (the switch is read once per process)
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from dpcr_agb_amd import sparse_ops, synthetic
from dpcr_agb_amd.coords import CoordinateManager
from dpcr_agb_amd.sparse_ops import spconv_forward_raw
dev = torch.device("cuda", 0)
torch.manual_seed(0)
b = synthetic.make_sparse_batch(list(range(32)))
coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
cm = CoordinateManager(coords, device=dev, batch_size=32, bounds=b.coord_bounds)
ts = 1
sparse_ops.DEFAULTS.cmp_mode = 128
outs = {}
for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
    while ts < ts_in:
        cm.stride(ts, 2); ts *= 2
    n = cm.level(ts_in).n
    nbr = cm.kernel_map(ts_in, 3, 1)
    x = torch.randn(n, c, device=dev); w = torch.randn(27 * c, c, device=dev) * 0.05
    y = spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
    e1.record(); torch.cuda.synchronize()
    print(f"mode {os.environ.get('AGB_CMPT')} ts{ts_in} {c}->{c}: {e0.elapsed_time(e1)/20*1e3:.1f} us  checksum {float(y.double().abs().sum()):.10e}  nan {int(torch.isnan(y).sum())}", flush=True)
    torch.save(y.cpu(), f"/tmp/y_{os.environ.get('AGB_CMPT')}_{ts_in}.pt")
'''
for mode in ("1", "3", "0"):
    env = dict(os.environ, AGB_CMPT=mode)
    subprocess.run([sys.executable, "-c", code], env=env, cwd="/root/repo", timeout=600)
import torch
for ts in (2, 4, 8, 16):
    a, b3 = torch.load(f"/tmp/y_1_{ts}.pt"), torch.load(f"/tmp/y_3_{ts}.pt")
    print(f"ts{ts}: cma vs cmpt max abs diff {float((a - b3).abs().max()):.3e}  bit-equal {bool(torch.equal(a, b3))}")
