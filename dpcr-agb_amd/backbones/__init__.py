"""Backbones of the hot path, behind the reference's factory name
``initialize_minkowski_unet(model_name, in_channels, out_channels, D=3, conv1_kernel_size=3, **kwargs)``
(torch_points3d/modules/MinkowskiEngine/__init__.py:11-17)."""
from . import sparse as _sparse
from .sparse import *  # noqa: F401,F403
from .pointnet import MinkowskiPointNet  # noqa: F401
from .kpconv import KPCNN, KPConv  # noqa: F401

_REGISTRY = {name: getattr(_sparse, name) for name in _sparse.__all__
             if isinstance(getattr(_sparse, name), type) and issubclass(getattr(_sparse, name), _sparse.ResNetBase)
             and getattr(_sparse, name) is not _sparse.ResNetBase}
_REGISTRY["MinkowskiPointNet"] = MinkowskiPointNet


def initialize_minkowski_unet(model_name, in_channels, out_channels, D=3, conv1_kernel_size=3, **kwargs):
    try:
        net_cls = _REGISTRY[model_name]
    except KeyError:
        raise AttributeError(f"unknown sparse backbone '{model_name}' (known: {sorted(_REGISTRY)})")
    return net_cls(in_channels=in_channels, out_channels=out_channels, D=D, conv1_kernel_size=conv1_kernel_size,
                   **kwargs)
