"""cProfile of the host side of the headline training step (where do the ~5.5 ms of Python per step go?).
Usage (GPU box): python tools/host_profile.py [--steps 30]"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    import dpcr_agb_amd
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    dpcr_agb_amd.limit_host_threads()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds).to(dev).train()
    model.init_train_objects(TRAINING_NFI)
    pool = [synthetic.make_sparse_batch(list(range(i * 32, (i + 1) * 32)), n_points=16000).to(dev) for i in range(2)]

    def step(i):
        model.set_input(pool[i % 2], dev)
        model.optimize_parameters(epoch=0, batch_size=32, num_batches=133)

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(a.steps):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
