"""Host time of one KPConv training step (BASELINE config 3) on a drained device queue, by phase, and the cProfile view of
the enqueuing thread.  `python tools/host_phases_kpconv.py [--steps N] [--cprofile]`."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--cprofile", action="store_true")
    args = ap.parse_args()
    import dpcr_agb_amd
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import KPConvModel
    dpcr_agb_amd.limit_host_threads()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    B = args.batch
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds)
    pool = [synthetic.make_point_batch(list(range(i * B, (i + 1) * B)), n_points=args.points) for i in range(2)]
    for b in pool:
        b.pos, b.x = b.pos.to(dev), b.x.to(dev)
    model.to(dev).train()
    model.init_train_objects(TRAINING_NFI)
    model.reserve_workspace(dev, main_bytes=12 << 30, side_bytes=6 << 30)
    phases = dict(set_input=[], prefetch=[], forward=[], backward=[], optimiser=[])

    def step(i, record):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.set_input(pool[i % 2], dev)
        t1 = time.perf_counter()
        model.prefetch_input(pool[(i + 1) % 2], dev)
        t2 = time.perf_counter()
        model.forward()
        t3 = time.perf_counter()
        model.optimizer.zero_grad(set_to_none=True)
        model.loss.backward()
        t4 = time.perf_counter()
        model.optimizer.step()
        t5 = time.perf_counter()
        if record:
            for k, v in zip(phases, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                phases[k].append(v * 1e3)

    model.prefetch_input(pool[0], dev)
    for i in range(4):
        step(i, False)
    if args.cprofile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(4, 4 + args.steps):
            step(i, True)
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(30)
        st.sort_stats("cumulative").print_stats(45)
    else:
        for i in range(4, 4 + args.steps):
            step(i, True)
    tot = 0.0
    for k, v in phases.items():
        v = sorted(v)
        tot += v[len(v) // 2]
        print(f"[host_phases_kpconv] {k:10s} median {v[len(v) // 2]:7.3f} ms")
    print(f"[host_phases_kpconv] sum of medians {tot:.3f} ms")


if __name__ == "__main__":
    main()
