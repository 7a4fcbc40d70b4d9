"""TorchGate: an nn.Module drop-in for /root/reference/noisereduce/torchgate
(torchgate/__init__.py:12)."""
from .torchgate import TorchGate  # noqa: F401
