"""A/B of the fp32 weight-gradient kernels in ONE process, interleaved repetitions per shape (the four 3^3 stride-1 levels of the
SENet14 pyramid of a synthetic batch of 32 plots, plus the strided 2^3 maps with --strided).
   python tools/dw_ab.py          persistent = k_spconv_dwa (csrc/dwa.hip, the product path), staged = k_spconv_dw_cmp<0>
                                  (KernelOptions.dw_variant 1), reg = k_spconv_dw_reg with atomics (dw_variant 2)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from dpcr_agb_amd import sparse_ops, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    b = synthetic.make_sparse_batch(list(range(32)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=32, bounds=b.coord_bounds)
    ts = 1
    variants = (("auto", {}), ("persistent", dict(dw_variant=3)), ("staged", dict(dw_variant=1)), ("staged2048", dict(dw_variant=4)),
                ("reg", dict(dw_variant=2)))
    cases = []
    for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
        while ts < ts_in:
            cm.stride(ts, 2)
            ts *= 2
        n = cm.level(ts_in).n
        cases.append((f"ts{ts_in:2d} 3^3 {c}->{c}", cm.kernel_map(ts_in, 3, 1), n, n, 27, c, c))
    if "--strided" in sys.argv:
        for ts_in, c in ((2, 64), (4, 128), (8, 256)):
            n_in, n_out = cm.level(ts_in).n, cm.level(2 * ts_in).n
            cases.append((f"ts{ts_in:2d} 2^3 s2 {c}->{2 * c}", cm.kernel_map(ts_in, 2, 2), n_in, n_out, 8, c, 2 * c))
            cases.append((f"ts{ts_in:2d} 3^3 s2 {c}->{2 * c}", cm.kernel_map(ts_in, 3, 2), n_in, n_out, 27, c, 2 * c))
    for name, nbr, n_in, n_out, K3, cin, cout in cases:
        x = torch.randn(n_in, cin, device=dev)
        dy = torch.randn(n_out, cout, device=dev)
        pairs = int((nbr[:, :n_out] >= 0).sum())
        # fp64 reference
        ref = torch.zeros(K3, cin, cout, dtype=torch.float64, device=dev)
        for k in range(K3):
            idx = nbr[k, :n_out].long()
            pres = idx >= 0
            ref[k] = x[idx[pres]].double().t() @ dy[pres].double()
        tot = {v[0]: 0.0 for v in variants}
        err = {}
        for rep in range(5):
            for tag, kw in variants:
                opts = sparse_ops.KernelOptions(**kw)
                dw = torch.zeros(K3, cin, cout, device=dev)
                sparse_ops.weight_grad_raw(x, dy, nbr, dw, n_out, K3, cin, cout, opts)
                torch.cuda.synchronize()
                if rep == 0:
                    err[tag] = float((dw.double() - ref).abs().max() / ref.abs().max())
                    if tag == "persistent":      # (auto = the product's choice: staged 64 x 64 or 128 x 64 tiles, atomics)
                        dw2 = torch.zeros(K3, cin, cout, device=dev)
                        sparse_ops.weight_grad_raw(x, dy, nbr, dw2, n_out, K3, cin, cout, opts)
                        err["repeat"] = bool(torch.equal(dw, dw2))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    sparse_ops.weight_grad_raw(x, dy, nbr, dw, n_out, K3, cin, cout, opts)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    tot[tag] += e0.elapsed_time(e1) / 5 * 1e3
        print(f"{name} ({pairs / 1e6:.2f} M pairs): " + "  ".join(
            f"[{t}] {tot[t] / 4:6.1f} us {2.0 * pairs * cin * cout / (tot[t] / 4) / 1e6:5.1f} TF ({err[t]:.0e})" for t, _ in variants)
            + f"  persistent bitwise repeatable: {err['repeat']}", flush=True)


if __name__ == "__main__":
    main()
