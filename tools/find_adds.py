"""Which autograd nodes the element-wise additions / fills / copies of one training step come from: one step of a model under
torch.profiler, every `aten::add` / `aten::add_` / `aten::fill_` / `aten::copy_` listed under the chain of its enclosing
operators (the autograd engine names the backward node it is evaluating).
`python tools/find_adds.py [--model SENet50] [--precision bf16] [--bf16-rows] [--batch 8] [--ops add,add_]`."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="SENet50")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--bf16-rows", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--ops", default="add,add_")
    args = ap.parse_args()
    import dpcr_agb_amd
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    dpcr_agb_amd.limit_host_threads()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_064))
    if args.model == "KPConv":
        from dpcr_agb_amd.instance import KPConvModel
        model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds)
        pool = [synthetic.make_point_batch(list(range(i * args.batch, (i + 1) * args.batch)), n_points=args.points)
                for i in range(2)]
        for b in pool:
            b.pos, b.x = b.pos.to(dev), b.x.to(dev)
        model.to(dev).train()
    else:
        model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS[args.model]), "minkowski", ds)
        model.to(dev).train()
        model.set_kernel_options(precision=args.precision, bf16_activations=args.bf16_rows)
        pool = [synthetic.make_sparse_batch(list(range(i * args.batch, (i + 1) * args.batch)), n_points=args.points).to(dev)
                for i in range(2)]
    model.init_train_objects(TRAINING_NFI)

    def step(i):
        model.set_input(pool[i % 2], dev)
        model.optimize_parameters(epoch=0, batch_size=args.batch, num_batches=133)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        step(3)
        torch.cuda.synchronize()
    want = {"aten::" + o for o in args.ops.split(",")}
    seen = collections.Counter()
    for e in prof.events():
        if args.ops == "all":
            if not e.name.startswith("aten::") or (e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::")):
                continue
            if e.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::reshape", "aten::slice", "aten::select",
                          "aten::detach", "aten::empty_strided", "aten::as_strided", "aten::t", "aten::transpose",
                          "aten::unsqueeze", "aten::squeeze", "aten::view_as", "aten::alias", "aten::contiguous"):
                continue      # (no kernel behind them, or counted through the copy they make)
        elif e.name not in want:
            continue
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        shapes = ""
        seen[(e.name, " <- ".join(chain) or "(top level)")] += 1
    for (name, chain), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"{n:5d}  {name:14s} {chain}")


if __name__ == "__main__":
    main()
