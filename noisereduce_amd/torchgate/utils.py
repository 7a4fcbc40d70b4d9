"""Small helpers with the names of /root/reference/noisereduce/torchgate/utils.py.  The
dB conversion and the sigmoid are computed inside the HIP kernels; these host versions
exist for callers that import them."""
import torch


@torch.no_grad()
def amp_to_db(x, eps=torch.finfo(torch.float64).eps, top_db=40):
    """20*log10(|x| + eps) floored at (max over the last axis) - top_db (utils.py:5-23)."""
    x_db = 20 * torch.log10(x.abs() + eps)
    return torch.max(x_db, (x_db.max(-1).values - top_db).unsqueeze(-1))


@torch.no_grad()
def temperature_sigmoid(x, x0, temp_coeff):
    """sigmoid((x - x0) / temp_coeff) (utils.py:26-39)."""
    return torch.sigmoid((x - x0) / temp_coeff)


@torch.no_grad()
def linspace(start, stop, num=50, endpoint=True, **kwargs):
    """torch.linspace with numpy's endpoint=False option (utils.py:42-66)."""
    if endpoint:
        return torch.linspace(start, stop, num, **kwargs)
    return torch.linspace(start, stop, num + 1, **kwargs)[:-1]
