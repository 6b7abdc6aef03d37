"""Dataset side of the hot path: the reference's FluidNPZSequenceDataset / augmentation / normalisation API
(ml/dataset/) and the HBM-resident store + batch gather built on it."""
from .augmentations import apply_augmentation, apply_horizontal_flip, apply_rollout_augmentation  # noqa: F401
from .normalization import load_normalization_scales  # noqa: F401
from .npz_sequence import FluidNPZSequenceDataset  # noqa: F401


def __getattr__(name):
    # the HBM-resident classes bind the HIP library on import; keep `import fluidnet_amd.dataset` light for host-only users
    if name in ("HbmSequenceStore", "HbmBatchLoader"):
        from . import hbm_resident
        return getattr(hbm_resident, name)
    raise AttributeError(name)
