from .auto_regressive import sample_auto_regressive  # noqa: F401
