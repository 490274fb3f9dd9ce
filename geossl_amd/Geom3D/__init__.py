"""Drop-in mirror of the reference's ``Geom3D`` package surface for the hot path
(Geom3D/models/__init__.py:1-2 exports PaiNN and SchNet)."""
from . import models  # noqa: F401
