"""Minimal attribute/``.get`` config object standing in for the reference's OmegaConf DictConfig
(hydra/omegaconf are not part of the hot path; values mirror the YAML quoted in SURVEY.md Appendix B)."""


class Opt(dict):
    """dict with attribute access, nested conversion and ``.get(key, default)`` like a DictConfig."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Opt):
            v = Opt(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


# torch-points3d/conf/models/instance/minkowski_baseline.yaml
MODEL_OPTIONS = {
    "MPointNet": Opt(model_name="MinkowskiPointNet", D=3, activation="gelu", first_stride=1, dropout=0.0,
                     global_pool="sum", add_pos=True, conv_type="SPARSE"),
    **{name: Opt(model_name=mn, D=3, activation="gelu", first_stride=1, dropout=0.0, drop_path=0.01,
                 global_pool="sum", conv_type="SPARSE")
       for name, mn in [("ResNet14", "ResNet14_"), ("ResNet18", "ResNet18_"), ("ResNet34", "ResNet34_"),
                        ("ResNet50", "ResNet50_"), ("ResNet101", "ResNet101_"), ("SENet14", "SENet14"),
                        ("SENet18", "SENet18"), ("SENet34", "SENet34"), ("SENet50", "SENet50"),
                        ("SENet101", "SENet101")]},
}

# torch-points3d/conf/training/nfi/minkowski.yaml + conf/lr_scheduler/cosineawr.yaml
TRAINING_NFI = Opt(
    epochs=310, batch_size=32, grad_clip=100, enable_mixed=False,
    optim=Opt(base_lr=0.005, optimizer=Opt(name="AdaBelief", params=Opt(lr=0.005, weight_decay=1e-2))),
    lr_scheduler=Opt(name="CosineAnnealingWarmRestarts", params=Opt(T_0=10, T_mult=2),
                     update_scheduler_on="on_num_batch"),
)

# torch-points3d/conf/data/instance/NFI/reg.yaml:21-24
NFI_TARGETS = Opt(
    BMag_ha=Opt(task="regression", weight=0.5),
    V_ha=Opt(task="regression", weight=0.5),
)

FIRST_SUBSAMPLING = 0.0125  # conf/data/instance/NFI/default.yaml:23 (normalised units)


# torch-points3d/conf/models/instance/kpconv.yaml:15-75 (FEAT = dataset feature dimension)
def kpconv_config(in_features_dim=3, first_subsampling_dl=FIRST_SUBSAMPLING):
    return Opt(
        in_points_dim=3, in_features_dim=in_features_dim, in_radius=1.0,
        architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided", "resnetb",
                      "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided", "resnetb", "resnetb",
                      "global_sum"],
        first_features_dim=64, use_batch_norm=True, batch_norm_momentum=0.02, activation="relu",
        num_kernel_points=15, first_subsampling_dl=first_subsampling_dl, conv_radius=2.5, deform_radius=5.0,
        KP_extent=1.0, KP_influence="linear", aggregation_mode="sum", fixed_kernel_points="center", modulated=False,
        deform_fitting_mode="point2point", deform_fitting_power=1.0, deform_lr_factor=0.1, repulse_extent=1.2)


MODEL_OPTIONS["KPConv"] = Opt(conv_type="PARTIAL_DENSE", config=kpconv_config())
