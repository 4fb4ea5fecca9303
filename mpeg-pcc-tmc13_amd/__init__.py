"""mpeg-pcc-tmc13_amd -- MI355X-native attribute-transform hot path of TMC13.

The directory name carries the reference's name (with hyphens), so it is
loaded under the importable alias ``mpeg_pcc_tmc13_amd`` by
``__graft_entry__.load_package()`` / ``tests/conftest.py``.
"""
from . import params, synth  # noqa: F401
from .params import (LiftParams, LodParams, PredParams, RahtInterParams, RahtParams, RecolourParams, lift_params,  # noqa: F401
                     lod_params, pred_params, raht_params, recolour_params)


def context(device=0, stream=None):
    """Open a device context (raises if the HIP library / a gfx950 GPU is missing)."""
    from .raht import Context
    return Context(device, stream)
