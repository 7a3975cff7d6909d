"""occformer_amd -- MI355X-native (gfx950) OccFormer forward hot path.

Importing the package registers the reference's registry names
(ViewTransformerLiftSplatShootVoxel, OccupancyEncoder, MSDeformAttnPixelDecoder3D,
Mask2FormerNuscOccHead, Mask2FormerOccHead, OccupancyFormer, ...) so that the reference's
``projects/configs/*.py`` build unchanged through ``occformer_amd.registry``.
"""
from . import registry  # noqa: F401
from .registry import (ATTENTION, BACKBONES, DETECTORS, HEADS, MODELS, NECKS, Config, ConfigDict,  # noqa: F401
                       build_model)
from . import view_transformer  # noqa: F401
from . import encoder, pixel_decoder, head, detector, efficientnet  # noqa: F401,E402
