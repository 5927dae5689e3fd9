"""Would the weight gradient of a layer overlap with its data gradient if they ran on two streams?  Per level of the SENet14
pyramid (synthetic batch of 32 plots): data gradient (k_spconv_cma, flipped offsets) + weight gradient (k_spconv_dw_cmp) issued
back to back on one stream vs on two streams, interleaved repetitions.
   python tools/overlap_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from dpcr_agb_amd import sparse_ops, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw, weight_grad_raw
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    b = synthetic.make_sparse_batch(list(range(32)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=32, bounds=b.coord_bounds)
    side = torch.cuda.Stream()
    ts = 1
    for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
        while ts < ts_in:
            cm.stride(ts, 2)
            ts *= 2
        n = cm.level(ts_in).n
        nbr = cm.kernel_map(ts_in, 3, 1)
        x = torch.randn(n, c, device=dev)
        dy = torch.randn(n, c, device=dev)
        wt = torch.randn(27 * c, c, device=dev) * 0.05
        dw = torch.zeros(27, c, c, device=dev)
        opts = sparse_ops.KernelOptions()

        def seq():
            spconv_forward_raw(dy, wt, nbr, 1, None, n, 27, c, c)
            weight_grad_raw(x, dy, nbr, dw, n, 27, c, c, opts)

        def conc():
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                weight_grad_raw(x, dy, nbr, dw, n, 27, c, c, opts)
                done = side.record_event()
            spconv_forward_raw(dy, wt, nbr, 1, None, n, 27, c, c)
            torch.cuda.current_stream().wait_event(done)

        tot = {"seq": 0.0, "conc": 0.0}
        for rep in range(6):
            for name, fn in (("seq", seq), ("conc", conc)):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    tot[name] += e0.elapsed_time(e1) / 5 * 1e3
        print(f"ts{ts_in:2d} {c:3d}->{c:3d}: one stream {tot['seq'] / 5:7.1f} us   two streams {tot['conc'] / 5:7.1f} us "
              f"({100 * (tot['conc'] / tot['seq'] - 1):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
