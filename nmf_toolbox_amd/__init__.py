"""nmf_toolbox_amd -- MI355X-native drop-in for the multiplicative-update hot path of colinvaz/nmf-toolbox.

Only the path named by BASELINE.json's north_star lives here: nmf / cnmf / nmfsc (+ the two helpers
they call), behind the C ABI of include/nmfx.h (libnmfx.so, hand-written HIP for gfx950).
"""
from .toolbox import ReconstructFromDecomposition, SortDictionary, cnmf, constrainednmf, cnmfsc, lnmf, nmf, nmfsc, projfunc, reconstruct_from_decomposition  # noqa: F401
from ._lib import NmfxError, device_count  # noqa: F401

__all__ = ["nmf", "cnmf", "nmfsc", "cnmfsc", "lnmf", "constrainednmf", "SortDictionary", "ReconstructFromDecomposition", "reconstruct_from_decomposition", "projfunc", "NmfxError", "device_count"]
