"""twingan_amd -- MI355X-native TwinGAN G+D training hot path.

Host-side mirror (Python, like the reference) of nets/pggan.py, nets/pggan_utils.py, libs/ops.py and
the training-step parts of twingan.py / image_generation.py / model_deploy.py, running on the
hand-written gfx950 kernels of ``libtwingan_hip.so`` (include/twingan_hip.h).  There is no CPU path.
"""
from .config import Config  # noqa: F401
