"""Reference checkpoint layout (metrics/model_checkpoint.py): round trip through a file written the way the reference's
trainer writes it (DataParallel prefix, MinkowskiEngine kernel shapes for kernel_size 1), weights and optimiser state."""
import os

import torch

from dpcr_agb_amd import synthetic
from dpcr_agb_amd.checkpoint import Checkpoint, adapt_state_dict, load_reference_weights
from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
from dpcr_agb_amd.instance import MinkowskiBaselineModel


def make_model(seed):
    torch.manual_seed(seed)
    ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_008))
    return MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds)


def test_state_dict_names_are_the_references():
    keys = set(make_model(0).state_dict())
    # torch_points3d/modules/MinkowskiEngine/SENet.py / resnet_block.py / senet_block.py attribute paths
    for k in ["model.blocks.0.0.conv.kernel", "model.blocks.0.0.conv.bias", "model.blocks.0.0.norm.bn.running_mean",
              "model.blocks.1.0.conv1.kernel", "model.blocks.1.0.norm2.bn.weight", "model.blocks.1.0.se.fc.0.linear.weight",
              "model.blocks.1.0.se.fc.2.linear.bias", "model.blocks.2.0.downsample.0.kernel",
              "model.blocks.2.0.downsample.1.bn.num_batches_tracked", "model.final.linears.0.weight",
              "model.final.linears.1.bias", "reg_scale_targets", "reg_center_targets", "reg_weights"]:
        assert k in keys, k


def test_reference_layout_round_trip(tmp_path):
    src, dst = make_model(1), make_model(2)
    src.init_train_objects(TRAINING_NFI)
    # what the reference's trainer would have written: DataParallel prefix, ME's [1, Cin, Cout] for strided 1x1 convs
    sd = {}
    for k, v in src.state_dict().items():
        if k.endswith("downsample.0.kernel"):
            assert v.dim() == 3 and v.shape[0] == 1     # kernel_size 1, stride 2: ME keeps the offset axis
            v = v.reshape(v.shape[1], v.shape[2])       # ... and a stride-1 file would drop it
        sd["module." + k] = v.clone()
    ck = Checkpoint(os.path.join(tmp_path, "SENet14.pt"))
    ck.save_objects({"latest": sd, "best_loss_reg": sd}, "train", {"epoch": 3, "loss_reg": 0.5}, src.optimizer,
                    {"lr_scheduler": src._lr_scheduler}, None, run_config={"models": {"SENet14": {}}})
    raw = torch.load(ck.path, map_location="cpu", weights_only=False)
    assert set(raw) >= {"models", "optimizer", "schedulers", "grad_scale", "stats", "run_config", "dataset_properties"}
    assert raw["optimizer"][0] == "AdaBelief" and raw["stats"]["train"][-1]["epoch"] == 3

    loaded = Checkpoint.load(str(tmp_path), "SENet14")
    assert not loaded.is_empty and loaded.get_state_dict("loss_reg") is not None
    ok, unmatched = adapt_state_dict(loaded.get_state_dict("latest"), dst)
    assert not unmatched and set(ok) == set(dst.state_dict())
    assert load_reference_weights(dst, ck.path, "loss_reg", strict=True) == []
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    dst.init_train_objects(TRAINING_NFI)
    loaded.load_optim_sched(dst)
    assert dst.optimizer.state_dict()["param_groups"][0]["lr"] == src.optimizer.state_dict()["param_groups"][0]["lr"]


def test_missing_checkpoint_is_reported(tmp_path):
    import pytest
    with pytest.raises(ValueError):
        Checkpoint.load(str(tmp_path), "nope", strict=True)
    assert Checkpoint.load(str(tmp_path), "nope", strict=False, resume=False).is_empty
