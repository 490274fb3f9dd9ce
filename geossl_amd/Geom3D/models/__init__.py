from .painn import PaiNN  # noqa: F401
from .schnet import SchNet  # noqa: F401
