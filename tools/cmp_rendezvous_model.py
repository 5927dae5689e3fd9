"""What would a per-offset rendezvous of four waves cost k_spconv_cmp<128>?  (VERDICT round 3, item 5 i: a 4-wave workgroup
that stages W[k] once per CU in LDS.)  CPU model on the real kernel map of the 64 -> 64 level (tensor stride 2) of synthetic
16 000-point plots: tiles of 128 consecutive rows, per (tile, offset) the number of 16-pair MFMA groups; four tiles that
share W[k] must all have finished offset k before W[k + 2] may overwrite a double buffer, i.e. a wave is never more than one
offset ahead.  Reported: sum_k max_4(groups) / mean_4 sum_k groups for lockstep (one barrier per offset) and for a one-step
slack, over consecutive and over work-sorted quadruples of tiles; and the LDS budget of the form.  numpy only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dpcr_agb_amd import synthetic
    from oracle import sparse_ref as R
    b = synthetic.make_sparse_batch(list(range(4)), n_points=16000)
    coords = np.concatenate([b.batch.numpy()[:, None], b.coords.numpy()], 1).astype(np.int64)
    ref = R.Coords(coords, 4)
    ref.level(1, 2)
    nbr = ref.map(2, 3, 1).numpy()                      # [27, n] kernel map of the stride-2 level (64 -> 64 layers)
    n = nbr.shape[1]
    T = n // 128
    pairs = (nbr[:, :T * 128] >= 0).reshape(27, T, 128).sum(2)            # [27, tiles]
    groups = (pairs + 15) // 16
    print(f"{n} rows, {T} tiles of 128 rows, density {pairs.sum() / (27 * T * 128):.2f}, groups per (tile, offset): mean "
          f"{groups.mean():.2f}, sd {groups.std():.2f}; MFMA slots filled {pairs.sum() / (16 * groups.sum()):.2f}")

    def loss(order, slack):
        tot_lock, tot_free = 0.0, 0.0
        for q in range(0, T - 3, 4):
            g = groups[:, order[q:q + 4]].astype(float)                    # [27, 4]
            t = np.zeros(4)                                                # finish time of each wave
            done = np.zeros((28, 4))
            for k in range(27):
                # a wave may start offset k once every wave has finished offset k - 1 - slack
                gate = done[max(k - slack, 0)].max() if k - slack > 0 else 0.0
                t = np.maximum(t, gate) + g[k]
                done[k + 1] = t
            tot_lock += t.max()
            tot_free += g.sum(0).mean()
        return tot_lock / tot_free

    cons = np.arange(T)
    by_work = np.argsort(groups.sum(0))
    for name, order in (("consecutive tiles", cons), ("tiles sorted by total work (best case for a quadruple)", by_work)):
        print(f"{name}: lockstep per offset x{loss(order, 0):.3f}, one offset of slack x{loss(order, 1):.3f}, "
              f"two x{loss(order, 2):.3f}  (time of the quadruple / mean work of its tiles)")
    ys = 129 * 68 * 4
    print(f"LDS: running sums of one 128-row tile {ys / 1024:.1f} KB + lists 2.3 KB; four tiles + a double W buffer (2 x 16 KB) = "
          f"{(4 * (ys + 2400) + 32768) / 1024:.0f} KB of 160 KB; with 96-row tiles {(4 * (97 * 68 * 4 + 1900) + 32768) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
