"""From a file on disk to a prediction (SURVEY.md 8(f)3 + f4): hand-made LAS bytes -> plots.read_pt (the reference's
las_dataset.py:32-71 reader interface) -> the NFI TEST transform chain on the device (sparse-xy.yaml test_transform:
transforms.SparsePlotPipeline) -> MSENet14 whose weights come through checkpoint.load_reference_weights from a file in the
reference's Checkpoint layout (metrics/model_checkpoint.py:43-58: ``models`` {"latest": state_dict}, DataParallel's ``module.``
prefix, ME's [Cin, Cout] kernel of a 1x1 layer) -> ``get_reg_output()``.  The same plots go through the oracle: per-sample
transform chain (oracle/transforms_ref.py), GridSampling3D(last) (oracle/voxelize_ref.py), network (oracle/sparse_ref.py) in
fp64.  The readers are exercised against bytes assembled field by field from the public LAS 1.2 layout (laspy is absent)."""
import struct

import numpy as np
import pytest
import torch

from oracle import sparse_ref as R
from oracle import transforms_ref as T
from oracle import voxelize_ref as V

pytestmark = pytest.mark.gpu


def _handmade_las(path, pts):
    """LAS 1.2, point format 0 (20-byte records), scale 1 mm, written with struct only."""
    n = len(pts)
    scale, off = (0.001, 0.001, 0.001), (float(pts[:, 0].min()), float(pts[:, 1].min()), float(pts[:, 2].min()))
    head = bytearray(227)
    head[0:4] = b"LASF"
    head[24:26] = bytes([1, 2])
    struct.pack_into("<H", head, 94, 227)
    struct.pack_into("<I", head, 96, 227)
    head[104] = 0
    struct.pack_into("<H", head, 105, 20)
    struct.pack_into("<I", head, 107, n)
    struct.pack_into("<6d", head, 131, *scale, *off)
    body = bytearray()
    for i in range(n):
        q = [int(round((float(pts[i][d]) - off[d]) / scale[d])) for d in range(3)]
        body += struct.pack("<iiiHBBbBH", q[0], q[1], q[2], 100, 9, 2, 0, 0, 1)
    with open(path, "wb") as f:
        f.write(bytes(head) + bytes(body))


def test_las_file_to_prediction_matches_oracle(device, tmp_path):
    from dpcr_agb_amd import checkpoint, plots, synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    from dpcr_agb_amd.transforms import SparsePlotPipeline, nfi_test_transform

    # ---- three plots as LAS files (the raw frame of the reference: metres around the plot centre, heights above a datum)
    raws = []
    for i, n in enumerate((5000, 3000, 4000)):
        pos, _, _ = synthetic.make_plot(700 + i, n)
        raw = np.stack([(pos[:, 0] - 0.5) * 30.0, (pos[:, 1] - 0.5) * 30.0, pos[:, 2] * 40.0 + 2.5], 1)
        f = str(tmp_path / f"plot{i}.las")
        _handmade_las(f, raw)
        got, _feats, _crs = plots.read_pt(f)
        assert got.shape == raw.shape and np.abs(got - raw).max() <= 0.0005 + 1e-9       # (1 mm records)
        raws.append(np.asarray(got, dtype=np.float32))

    # ---- a checkpoint file in the reference's layout, written the way the reference's trainer leaves it
    torch.manual_seed(5)
    ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_016))
    opt = Opt(MODEL_OPTIONS["SENet14"])
    donor = MinkowskiBaselineModel(opt, "minkowski", ds)
    with torch.no_grad():       # running statistics / affine parameters that are not the initial ones
        for k, v in donor.state_dict().items():
            if k.endswith("running_mean"):
                v.normal_(0.0, 0.05)
            elif k.endswith("running_var"):
                v.uniform_(0.5, 1.5)
            elif k.endswith("bn.weight"):
                v.uniform_(0.8, 1.2)
    sd = {}
    for k, v in donor.state_dict().items():
        v = v.detach().clone()
        if k.endswith(".kernel") and v.dim() == 3 and v.shape[0] == 1 and ".downsample." not in k:
            v = v[0]                                   # ME keeps a stride-1 kernel_size-1 kernel as [Cin, Cout]
        sd["module." + k] = v                          # nn.DataParallel's prefix (trainer.py:149-150)
    ck = str(tmp_path / "SENet14.pt")
    torch.save({"models": {"latest": sd, "best_loss_reg": sd}, "optimizer": ("AdaBelief", {}), "schedulers": {},
                "stats": {"train": [], "test": [], "val": []}, "run_config": {}, "dataset_properties": {}}, ck)

    model = MinkowskiBaselineModel(opt, "minkowski", ds)
    unmatched = checkpoint.load_reference_weights(model, ck, weight_name="loss_reg", strict=True)
    assert unmatched == []
    model.to(device).eval()

    # ---- device test chain -> prediction
    pipe = SparsePlotPipeline(nfi_test_transform())
    rng = np.random.default_rng(3)
    # the oracle's per-sample chain first (the crop decides the lengths the voxel shuffle is drawn for)
    ops = [T.test_transform_sample(torch.from_numpy(r)) for r in raws]
    lens = [len(o[0]) for o in ops]
    perms = [rng.permutation(n) for n in lens]
    batch = pipe(raws, device, perms=torch.from_numpy(np.concatenate(perms)))
    with torch.no_grad():
        model.set_input(batch, device)
        model.forward()
        pred = model.get_reg_output().cpu().double()

    # ---- the oracle on the same points
    oc, ok, ol = V.batch_grid_sampling_last(np.concatenate([o[0].numpy() for o in ops]), lens, perms, 0.0125)
    feats = torch.cat([o[1] for o in ops])[torch.from_numpy(np.asarray(ok))].double()
    bidx = np.repeat(np.arange(len(ol)), ol)
    coords = np.concatenate([bidx[:, None], np.asarray(oc, dtype=np.int64)], 1)
    assert int(batch.coords.shape[0]) == len(coords)
    assert np.array_equal(batch.coords.cpu().numpy().astype(np.int64), coords[:, 1:])
    ref_sd = {k: v.double() if v.is_floating_point() else v for k, v in donor.model.state_dict().items()}
    out = R.resnet_forward(ref_sd, coords, feats, (1, 1, 1, 1), batch_size=len(ol), training=False)
    ref = out * donor.reg_scale_targets.double() + donor.reg_center_targets.double()
    err = float((pred - ref).abs().max() / ref.abs().max())
    print(f"file -> prediction: max rel err vs the oracle {err:.2e}; predictions {pred.tolist()}")
    assert err < 1e-4
