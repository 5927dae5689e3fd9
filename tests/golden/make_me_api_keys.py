"""Generates tests/golden/me_api_keys.json: proof that the REFERENCE's own sparse-backbone model files run on this
repo's MinkowskiEngine look-alike unchanged (SURVEY.md §8(b) "Minkowski model API").

Build container only (needs /root/reference).  The script binds ``sys.modules["MinkowskiEngine"]`` to
``dpcr_agb_amd.me_compat`` and imports, from where they lie,
    torch_points3d/modules/MinkowskiEngine/{common,resnet_block,senet_block,SENet,PointNet}.py
(the package ``__init__`` itself pulls in ``networks.py -> modules.py -> torch_points3d.utils.config`` = omegaconf, absent
here, so the five files are loaded under a hand-made package object; ``initialize_minkowski_unet`` — ``__init__.py:11-17``,
``getattr(module, model_name)(in_channels=, out_channels=, D=, conv1_kernel_size=, **kwargs)`` — is restated below).
Each model is constructed with the keyword arguments of ``models/instance/minkowski.py:32-38`` and the values of
``conf/models/instance/minkowski_baseline.yaml``; what is stored is DATA about the constructed modules: state_dict
key -> shape, parameter order, and the (name, class name) tree.  No reference source text is stored.

Run:  python tests/golden/make_me_api_keys.py
"""
import importlib.util
import json
import os
import sys
import types
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/torch-points3d/torch_points3d/modules/MinkowskiEngine"
FILES = ("common", "resnet_block", "senet_block", "SENet", "PointNet")
OUT = os.path.join(ROOT, "tests", "golden", "me_api_keys.json")

# (fixture name, class, in_channels, kwargs) — kwargs as MinkowskiBaselineModel passes them (minkowski.py:32-38)
BASE = dict(activation="gelu", first_stride=1, global_pool="sum", bias=True, bn_momentum=0.1, norm_type="bn",
            dropout=0.0)
CASES = [
    ("SENet14", "SENet14", 3, dict(BASE, drop_path=0.01)),
    ("SENet50", "SENet50", 3, dict(BASE, drop_path=0.01)),
    ("ResNet14_", "ResNet14_", 3, dict(BASE, drop_path=0.01)),
    ("SENet18_ln", "SENet18", 4, dict(BASE, drop_path=0.0, norm_type="ln")),
    ("SENet14_in_relu_mean", "SENet14", 3, dict(BASE, drop_path=0.0, norm_type="in", activation="relu",
                                                global_pool="mean", first_stride=2, dropout=0.1)),
    ("MinkowskiPointNet", "MinkowskiPointNet", 3, dict(BASE, drop_path=0.0)),
]


def load_reference_modules():
    """The reference's five model files, imported with MinkowskiEngine := dpcr_agb_amd.me_compat."""
    import dpcr_agb_amd.me_compat as ME
    sys.modules["MinkowskiEngine"] = ME
    for name in ("torch_points3d", "torch_points3d.modules", "torch_points3d.modules.MinkowskiEngine"):
        pkg = types.ModuleType(name)
        pkg.__path__ = [REF] if name.endswith("MinkowskiEngine") else []
        sys.modules[name] = pkg
    mods = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)
        for f in FILES:
            full = "torch_points3d.modules.MinkowskiEngine." + f
            spec = importlib.util.spec_from_file_location(full, os.path.join(REF, f + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            spec.loader.exec_module(mod)
            mods[f] = mod
    return mods


def describe(model):
    return {
        "state_dict": [[k, list(v.shape)] for k, v in model.state_dict().items()],
        "parameters": [k for k, _ in model.named_parameters()],
        "modules": [[n, type(m).__name__] for n, m in model.named_modules()],
    }


def initialize_minkowski_unet(mods, model_name, in_channels, out_channels, D=3, conv1_kernel_size=3, **kwargs):
    # __init__.py:11-17 (the registry there is the package namespace = SENet.py's classes + MinkowskiPointNet)
    cls = getattr(mods["SENet"], model_name, None) or getattr(mods["PointNet"], model_name)
    return cls(in_channels=in_channels, out_channels=out_channels, D=D, conv1_kernel_size=conv1_kernel_size, **kwargs)


def main():
    mods = load_reference_modules()
    out = {"_generator": "tests/golden/make_me_api_keys.py", "_reference_files": [f + ".py" for f in FILES], "cases": {}}
    for name, cls, cin, kw in CASES:
        model = initialize_minkowski_unet(mods, cls, cin, 2, **kw)
        d = describe(model)
        d["model_name"], d["in_channels"], d["kwargs"] = cls, cin, kw
        out["cases"][name] = d
        print(f"{name}: {len(d['state_dict'])} state_dict entries, {len(d['modules'])} modules "
              f"(reference classes on me_compat)")
    with open(OUT, "w") as f:
        json.dump(out, f, sort_keys=True, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
