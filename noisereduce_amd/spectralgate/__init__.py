from .base import SpectralGate  # noqa: F401
from .nonstationary import SpectralGateNonStationary  # noqa: F401
from .stationary import SpectralGateStationary  # noqa: F401
