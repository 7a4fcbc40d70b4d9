"""wav scaling helpers with the names of /root/reference/noisereduce/utils.py:4-15.  Not on the hot
path: reduce_noise accepts int16 directly (converted on the device, result truncated like
ndarray.astype)."""
import numpy as np


def int16_to_float32(data):
    """int16 PCM -> float32 in [-1, 1] (utils.py:4-9)."""
    if np.max(np.abs(data)) > 32768:
        raise ValueError("Data has values above 32768")
    return (data / 32768.0).astype("float32")


def float32_to_int16(data):
    """float32 in [-1, 1] -> int16 PCM (utils.py:12-15)."""
    if np.max(data) > 1:
        data = data / np.max(np.abs(data))
    return np.array(data * 32767).astype("int16")
