"""howl_amd -- MI355X-native implementation of Howl's audio hot path (frontend + res8/LSTM fwd/bwd).

Python host code mirrors the reference's own module surface (``howl.model`` registry,
``StandardAudioTransform`` / ``ZmuvTransform``, ``InferenceContext``, the inference engines and the
``training.run`` entry points); the arithmetic runs in hand-written gfx950 kernels behind the C ABI in
``include/howl_hip.h`` (``howl_amd/libhowl_hip.so``).  There is no CPU fallback: ops raise if the library
is missing or a tensor is not on a HIP device.
"""
__version__ = "0.1.0"
