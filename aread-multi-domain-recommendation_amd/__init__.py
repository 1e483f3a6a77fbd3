"""aread_amd -- MI355X-native (gfx950) implementation of AREAD's CTR forward/backward hot path.

The package mirrors the reference's module interface for that path only (model/layer.py,
model/aread.py); the compute lives in libaread_hip.so behind the C ABI of include/aread_hip.h."""
from . import _lib                                  # noqa: F401
from .layer import FeaturesEmbedding, MultiLayerPerceptron   # noqa: F401
from .plan import RowPlan                           # noqa: F401
from .aread import AREAD, pack_masks                # noqa: F401
from . import dist                                  # noqa: F401
from .optim import FusedAdam, Adam                  # noqa: F401

__all__ = ["FeaturesEmbedding", "MultiLayerPerceptron", "RowPlan", "AREAD", "pack_masks", "FusedAdam", "Adam"]
