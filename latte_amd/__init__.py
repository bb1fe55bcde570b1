"""latte_amd — MI355X-native Latte denoising engine (hand-written HIP for gfx950 behind a C-ABI).

Public surface = the reference entry points for the sampling hot path:
``Latte_models`` / ``get_models`` / ``find_model`` (models/latte.py, models/__init__.py, utils.py),
``create_diffusion`` (diffusion/__init__.py), ``AutoencoderKL`` (diffusers, decode only), ``LatteT2V`` / ``LattePipeline`` (models/latte_t2v.py, sample/pipeline_latte.py:
the Latte-1 text-to-video denoiser and its sampling loop), ``load_config`` (OmegaConf.load stand-in for the YAMLs).
"""
from ._lib import LatteError, load_library  # noqa: F401
from .config import Config, load_config  # noqa: F401
from .diffusion import SpacedDiffusion, create_diffusion  # noqa: F401
from .models import Latte, Latte_models, find_model, get_models  # noqa: F401
from .pipeline import LattePipeline  # noqa: F401
from .t2v import LatteT2V  # noqa: F401
from .training import LatteTrainer  # noqa: F401
from .vae import AutoencoderKL, AutoencoderKLTemporalDecoder  # noqa: F401
from .video_io import read_avi, read_mp4, write_avi, write_mp4  # noqa: F401
from . import parallel  # noqa: F401

__version__ = "0.1.0"
