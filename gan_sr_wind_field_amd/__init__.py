"""MI355X-native (gfx950) implementation of the GAN_SR_wind_field 3D-conv GAN
train-step hot path, behind the reference's own Python API.

Sub-modules mirror the reference tree (``CNN_models``, ``GAN_models``, ``config``,
``tools``) so that ``run.py``/``train.py`` style callers are a drop-in; the
arithmetic runs in hand-written HIP kernels reached through the C ABI declared in
``include/windsr_hip.h`` (``_lib`` / ``hip_ops``).
"""
__version__ = "0.1.0"
