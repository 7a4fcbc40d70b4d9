"""PCM scaling helpers (the reference exposes them as noisereduce.utils, utils.py:4-15).  They are not
on the hot path: reduce_noise takes int16 arrays as they are (converted on the device, result
truncated like ndarray.astype)."""
import numpy as np

_FULL_SCALE = 32768.0      # |int16| range used for the float conversion
_INT16_PEAK = 32767        # largest positive PCM count


def int16_to_float32(data):
    """PCM counts -> float32 samples in [-1, 1]; counts beyond the int16 range are an error."""
    pcm = np.asarray(data)
    if pcm.size and float(np.abs(pcm).max()) > _FULL_SCALE:
        raise ValueError("Data has values above 32768")
    return np.asarray(pcm / _FULL_SCALE, dtype=np.float32)


def float32_to_int16(data):
    """float samples -> int16 PCM counts (truncating); a signal that exceeds +1 is peak-normalised first."""
    x = np.asarray(data)
    if x.size and x.max() > 1:
        x = x / np.abs(x).max()
    return (x * _INT16_PEAK).astype(np.int16)
