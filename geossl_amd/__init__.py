"""geossl_amd — MI355X (gfx950) native hot path of chao1224/GeoSSL: SchNet message passing + the
denoising-distance-matching (DDM) step, behind the reference's own Python interface.

    from geossl_amd.Geom3D.models import SchNet, PaiNN
    from geossl_amd.NCSN import NCSN_version_03
    from geossl_amd.pretrain_GeoSSL import do_DDM, perturb

or ``geossl_amd.install_as_reference()`` to make ``import Geom3D`` / ``import NCSN`` resolve here.
"""
import sys

__version__ = "0.1.0"


def install_as_reference():
    """Alias this package's mirrors under the reference's top-level import names
    (``from Geom3D.models import SchNet, PaiNN``; ``from NCSN import NCSN_version_03``)."""
    from . import Geom3D, NCSN
    sys.modules.setdefault("Geom3D", Geom3D)
    sys.modules.setdefault("Geom3D.models", Geom3D.models)
    from .Geom3D import dataloaders
    sys.modules.setdefault("Geom3D.dataloaders", dataloaders)
    sys.modules.setdefault("NCSN", NCSN)
    return Geom3D, NCSN
