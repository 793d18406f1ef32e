"""drakes_amd — MI355X-native reverse-sampling hot path of DRAKES (DNA).

Public surface mirrors the reference: `drakes_amd.Diffusion` for
drakes_dna/diffusion_gosai_update.py::Diffusion, `drakes_amd.models.dnaconv.CNNModel` for
drakes_dna/models/dnaconv.py::CNNModel.  Importing the package does not load the HIP library;
the first kernel call does, and raises if it has not been built.
"""
from .diffusion import Diffusion  # noqa: F401
from .config import default_config  # noqa: F401
from .finetune_loop import fine_tune  # noqa: F401
from ._lib import DrakesHipError  # noqa: F401

__all__ = ['Diffusion', 'default_config', 'fine_tune', 'DrakesHipError']
