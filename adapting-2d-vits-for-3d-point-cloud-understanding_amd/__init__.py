"""MI355X-native Pix4Point hot path (tokenizer + ViT encoder) behind the reference's Python API.

The directory name follows the build contract and is not a Python identifier; import it with
``importlib.import_module('adapting-2d-vits-for-3d-point-cloud-understanding_amd')`` or through the
``p4p_amd`` alias module at the repository root.
"""
from . import _lib, ops, functional, functional_apf, vit, pix4point, apf, augment, ddp, trainer, apf_trainer, scanobjectnn, sampler  # noqa: F401
from .trainer import Pix4PointTrainer  # noqa: F401
from .pix4point import (ClsHead, P3Embed, Pix4Point, PointViT, farthest_point_sampling,  # noqa: F401
                        group_knn)
from .apf import AdaptPointFormer  # noqa: F401
from .apf_trainer import APFTrainer  # noqa: F401
from .scanobjectnn import ScanObjectNN  # noqa: F401
from .sampler import (fps_sampling_with_knn, fps_sampling_with_knn_optimized, fps_sampling_with_knn_topk,  # noqa: F401
                      fps_sampling_with_knn_cuda_optimized)          # what the reference's src/data/__init__.py exports
from .functional import invalidate_operand_caches  # noqa: F401
