"""Host time of one MSENet14 training step by phase (set_input / forward / backward / optimiser / prefetch_input), measured on
an EMPTY device queue (the device is drained before every step, so no enqueue ever waits for a queue slot): what the
enqueuing thread costs, independent of the device.  `python tools/host_phases.py [--fused 0|1] [--steps N]`."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--cprofile", action="store_true")
    args = ap.parse_args()
    import dpcr_agb_amd
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    dpcr_agb_amd.limit_host_threads()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_064))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds).to(dev).train()
    model.set_kernel_options(fused_blocks=bool(args.fused))
    model.init_train_objects(TRAINING_NFI)
    pool = [synthetic.make_sparse_batch(list(range(i * 100, i * 100 + args.batch)), n_points=args.points).to(dev)
            for i in range(4)]
    model.reserve_workspace(dev, main_bytes=16 << 30, side_bytes=8 << 30)
    model.prefetch_input(pool[0], dev)
    model.prefetch_input(pool[1], dev)
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    model.PACE_DEPTH = 0
    phases = {k: [] for k in ("set_input", "forward", "backward", "optim", "prefetch", "total")}
    prof = None
    if args.cprofile:
        import cProfile
        prof = cProfile.Profile()
    for i in range(args.steps):
        torch.cuda.synchronize()
        if prof is not None and i >= 8:
            prof.enable()
        t0 = time.perf_counter()
        model.set_input(pool[i % 4], dev)
        t1 = time.perf_counter()
        model(epoch=0)
        t2 = time.perf_counter()
        model._optimizer.zero_grad(set_to_none=True)
        model.loss.backward()
        t3 = time.perf_counter()
        model._optimizer.step()
        model._step_scheduler(0, args.batch, 133)
        model._num_batches += 1
        t4 = time.perf_counter()
        model.prefetch_input(pool[(i + 2) % 4], dev)
        t5 = time.perf_counter()
        if prof is not None and i >= 8:
            prof.disable()
        for k, v in zip(phases, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
            phases[k].append(v * 1e3)
    torch.cuda.synchronize()
    print(f"fused_blocks={args.fused}: host ms per step on an empty queue (median of the last {args.steps - 8} steps)")
    for k, v in phases.items():
        v = sorted(v[8:])
        print(f"  {k:10s} {v[len(v) // 2]:7.3f}   min {v[0]:7.3f}")
    if prof is not None:
        import pstats
        st = pstats.Stats(prof)
        st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
