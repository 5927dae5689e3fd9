"""Wave timelines of the pair-compacted convolution kernel on the SENet14 pyramid of a synthetic batch: per level the shader
clock the chip holds while the kernel runs, the clocks a wave spends per 16-pair group (2048 = pure MFMA issue), how busy the
resident-wave slots are, and the spread of the work per tile — the clock x slots x in-wave decomposition of DESIGN.md section 5.

Needs the instrumented build (cd dpcr-agb_amd/csrc && make timeline); run on a GPU box:
    AGB_LIBRARY=dpcr-agb_amd/libagbhip_timeline.so python tools/wave_timeline.py [--no-balanced]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AGB_LIBRARY", os.path.join(ROOT, "dpcr-agb_amd", "libagbhip_timeline.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-balanced", action="store_true", help="fixed interleave instead of the work-balanced tiles")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from dpcr_agb_amd import _lib, sparse_ops, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw
    L = _lib.load()
    if not hasattr(L, "agb_debug_cmp_timeline"):
        raise SystemExit(f"{_lib.LIB_PATH} is not the instrumented build: cd dpcr-agb_amd/csrc && make timeline")
    sparse_ops.DEFAULTS.balanced_tiles = not args.no_balanced
    dev = torch.device("cuda", 0)
    b = synthetic.make_sparse_batch(list(range(args.batch)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=args.batch, bounds=b.coord_bounds)
    ts = 1
    slots = 8192
    for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
        while ts < ts_in:
            cm.stride(ts, 2)
            ts *= 2
        n = cm.level(ts_in).n
        nbr = cm.kernel_map(ts_in, 3, 1)
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27 * c, c, device=dev) * 0.05
        for _ in range(3):
            spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
        torch.cuda.synchronize()
        L.agb_debug_cmp_timeline(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
        e1.record()
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * (4 * slots))()
        L.agb_debug_cmp_timeline(out, 0)
        a = np.array(out[:], dtype=np.uint64).reshape(slots, 4)
        a = a[a[:, 1] > 0]
        t0, t1 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64)
        ticks, g = a[:, 2].astype(np.float64), a[:, 3].astype(np.float64)
        base = t0.min()
        t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0
        dur, span = t1 - t0, t1.max()
        resident = int(((t0 <= span / 4) & (t1 > span / 4)).sum())
        print(f"ts {ts_in:2d} {c:3d}->{c:3d} rows {n:6d}: {e0.elapsed_time(e1) * 1e3:5.0f} us; {len(a)} waves ({resident} resident), "
              f"shader clock {np.mean(ticks / dur) / 1e3:.2f} GHz, {np.mean(ticks / np.maximum(g, 1)):.0f} clocks per group "
              f"(in-wave {2048 / np.mean(ticks / np.maximum(g, 1)):.2f}), slots busy {dur.sum() / (resident * span):.3f}, "
              f"groups per wave max/mean {g.max() / g.mean():.3f}, corr(duration, groups) {np.corrcoef(dur, g)[0, 1]:.2f}", flush=True)


if __name__ == "__main__":
    main()
