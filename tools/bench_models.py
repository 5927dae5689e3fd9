"""Throughput of the other BASELINE.json configurations (parity-test cases; bench.py stays on the MSENet14 metric):
  config 2  MPointNet fp32, B=64 x 16k-pt plots          config 3  KPConv rigid, B=32 x 6144 / 16000-pt plots
  config 5' MSENet50 fp32, B=32 (single GPU part of config 5)
Usage: python tools/bench_models.py [pointnet] [kpconv] [kpconv16k] [senet50] [--steps K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dpcr_agb_amd  # noqa: E402

dpcr_agb_amd.limit_host_threads()


def run(name, steps):
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import KPConvModel, MinkowskiBaselineModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
    if name in ("pointnet", "senet50"):
        key, B = ("MPointNet", 64) if name == "pointnet" else ("SENet50", 32)
        model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS[key]), "minkowski", ds)
        pool = [synthetic.make_sparse_batch(list(range(i * B, (i + 1) * B)), n_points=16000).to(dev) for i in range(2)]
        prefetch = True
    else:
        B, npts = 32, (16000 if name == "kpconv16k" else 6144)
        model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds)
        pool = [synthetic.make_point_batch(list(range(i * B, (i + 1) * B)), n_points=npts) for i in range(2)]
        for b in pool:
            b.pos, b.x = b.pos.to(dev), b.x.to(dev)
        prefetch = True
    model.to(dev).train()
    model.init_train_objects(TRAINING_NFI)

    def step(i):
        model.set_input(pool[i % 2], dev)
        model.optimize_parameters(epoch=0, batch_size=B, num_batches=133)
        if prefetch:
            model.prefetch_input(pool[(i + 1) % 2], dev)

    import gc
    gc.unfreeze()
    gc.collect()
    torch.cuda.empty_cache()
    gc.freeze()
    gc.disable()   # manual GC as in bench.py (a gen-2 sweep inside a step stalls the device queue)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(3 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    print(f"{name:10s} B={B:3d}  {dt / steps * 1e3:8.2f} ms/step  {B * steps / dt:9.1f} plots/s  "
          f"loss={float(model.loss.detach()):.4f}  params={sum(p.numel() for p in model.parameters()) / 1e6:.2f}M",
          flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = 10
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1])
        args = [a for a in args if a != str(steps)]
    names = args or ["pointnet", "kpconv", "kpconv16k", "senet50"]
    if len(names) == 1:
        run(names[0], steps)
    else:   # one process per model: allocator / stream state left by one model must not colour the next one's number
        import subprocess
        for n in names:
            subprocess.run([sys.executable, os.path.abspath(__file__), n, "--steps", str(steps)], check=False)
