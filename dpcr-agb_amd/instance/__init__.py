from .minkowski import MinkowskiBaselineModel  # noqa: F401
from .kpconv import KPConvModel  # noqa: F401
