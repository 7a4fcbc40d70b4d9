import sys, os, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd.torchgate import TorchGate
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = (0.1 * torch.randn(256, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * torch.arange(16000, device=dev) / 16000)).float()
for ns in (False, True):
    tg = TorchGate(sr=16000, nonstationary=ns).to(dev)
    for _ in range(3): tg(x)
    g = tg._gate_for(dev)
    g.profile_read(reset=True); g.profile_enable(True)
    for _ in range(10): tg(x)
    p = g.profile_read(reset=True); g.profile_enable(False)
    print("nonstationary" if ns else "stationary", {k: round(v[0] / 10, 4) for k, v in p.items()})
