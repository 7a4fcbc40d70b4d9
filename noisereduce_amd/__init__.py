"""noisereduce_amd -- MI355X-native spectral gating, drop-in for the
``reduce_noise()`` / ``TorchGate`` API of timsainb/noisereduce (hot path only:
STFT -> noise statistics -> mask -> 2-D smoothing -> masked multiply -> ISTFT, as HIP
kernels behind the C ABI in include/mi355gate.h).  Mirrors
/root/reference/noisereduce/__init__.py:1."""
from noisereduce_amd.noisereduce import reduce_noise  # noqa: F401

__all__ = ["reduce_noise"]
