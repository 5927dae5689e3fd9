from .minkowski import MinkowskiBaselineModel  # noqa: F401
