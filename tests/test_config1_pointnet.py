"""BASELINE.json config 1: "PointNet biomass regression, batch=2, env_cpu.yml CPU path (plumbing, no GPU)".

The reference's CPU path for this model is MinkowskiEngine on the CPU, which cannot be installed here;
the CPU restatement (oracle/sparse_ref.py:pointnet_forward, following PointNet.py:16-49 and models/instance/base.py)
stands in for it.  CPU test: the restatement trains on a batch of 2 synthetic plots with the reference recipe and the
model <-> trainer plumbing (target statistics, loss, AdaBelief, scheduler) behaves — the loss falls, state_dict keys are
the reference's.  GPU test: the HIP path reproduces the same three steps."""
import numpy as np
import pytest
import torch


def _setup():
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    torch.manual_seed(0)
    ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_032))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["MPointNet"]), "minkowski", ds)
    batch = synthetic.make_sparse_batch([3, 4], n_points=2000)
    return model, batch


def _oracle_steps(model, batch, steps, dtype=torch.float64):
    from oracle import sparse_ref as R
    from dpcr_agb_amd.optim import AdaBelief
    sd = {k: v.detach().clone().to(dtype).requires_grad_(v.is_floating_point() and "running" not in k)
          if v.is_floating_point() else v.clone() for k, v in model.model.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = AdaBelief(params, lr=0.005, weight_decay=1e-2)
    feats = torch.cat([batch.pos, batch.x], 1).to(dtype)
    losses = []
    for _ in range(steps):
        upd = {}
        out = R.pointnet_forward(sd, batch.batch, feats, len(batch), update=upd)
        loss = R.reg_loss(out, batch.y_reg.to(dtype), model.reg_center_targets.to(dtype),
                          model.reg_scale_targets.to(dtype), model.reg_weights.to(dtype))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(params, 100)
        opt.step()
        sd.update(upd)
        losses.append(float(loss.detach()))
    return losses


def test_cpu_restatement_trains_batch_of_two():
    model, batch = _setup()
    keys = list(model.model.state_dict().keys())
    # the reference's parameter names (PointNet.py:16-41; final replaced by SeparateLinear, minkowski.py:39-46)
    for k in ("blocks.0.linear.weight", "blocks.7.bn.running_var", "mlp.3.linear.weight", "final.linears.1.bias"):
        assert k in keys
    assert model.model.state_dict()["blocks.0.linear.weight"].shape == (64, 6)       # add_pos: xyz + 3 features
    losses = _oracle_steps(model, batch, 6, torch.float32)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_hip_path_matches_cpu_restatement_batch_of_two(device):
    from dpcr_agb_amd.config import TRAINING_NFI
    model, batch = _setup()
    ref = _oracle_steps(model, batch, 3)
    model.to(device).train()
    model.init_train_objects(TRAINING_NFI)
    for step in range(3):
        model.set_input(batch, device)
        model.optimize_parameters(epoch=0, batch_size=2, num_batches=100)
        got = float(model.loss.detach())
        # The head's BatchNorms run over B = 2 rows: every normalised value is +-1 up to eps, gradients through them are
        # ~eps-sized and fp32 rounding is amplified with every parameter update (measured: exact to 1e-4 before the first
        # update, 1e-3 after one, 1.5e-2 after two) — the config is plumbing, as BASELINE.json says.
        tol = (1e-4, 1e-3, 5e-2)[step]
        assert abs(got - ref[step]) < tol * max(1.0, abs(ref[step])), (step, got, ref[step])
