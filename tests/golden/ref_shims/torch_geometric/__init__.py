from . import data, nn  # noqa: F401
