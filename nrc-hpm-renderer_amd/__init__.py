"""nrc-hpm-renderer_amd: MI355X-native Neural Radiance Cache path for heterogeneous participating media.

Holds only what the hot path needs: csrc/ (HIP kernels + the C ABI of include/nrc_hpm.h),
the host-side mirror of the reference's NeuralRadianceCache / NrcHpmRenderer / McHpmRenderer
interface (api.py) and scene inputs (scene.py).
"""
from . import scene  # noqa: F401
