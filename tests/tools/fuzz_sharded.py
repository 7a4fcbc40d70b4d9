"""Random time-sharded configurations on 2 or 3 ranks sharing cuda:0 (gloo), HIP engine, against the
single-process oracle.  usage: python tests/tools/fuzz_sharded.py [world] [cases]"""
import os, socket, sys
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import spectralgate_oracle as O


def cases(k):
    r = np.random.default_rng(4242 + k)
    cs = int(r.integers(8000, 40000))
    pad = int(r.integers(0, min(cs, 6000)))
    nch = int(r.integers(2, 9))
    n = nch * cs - int(r.integers(0, cs - 1100))      # last chunk partial
    C = int(r.choice([1, 2]))
    return dict(cs=cs, pad=pad, n=n, C=C, seed=k, dtype=str(r.choice(["float32", "float64"])))


def worker(rank, world, port, ncases, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from noisereduce_amd.sharded import HipStationaryBackend, TimeShardedStationary, alloc_shard, shard_bounds
    errs = []
    for k in range(ncases):
        c = cases(k)
        y = np.stack([O.synth_signal(c["n"], seed=c["seed"] * 3 + ch, tone_hz=400.0 * (ch + 1)).astype(np.float64)
                      for ch in range(c["C"])]).astype(c["dtype"])
        want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, chunk_size=c["cs"], padding=c["pad"])
        bounds = [shard_bounds(c["n"], c["cs"], world, r_) for r_ in range(world)]
        s0, s1 = bounds[rank]
        # every rank takes the same decision: skip cases with an empty shard or one shorter than the padding
        if any(b1 - b0 < max(c["pad"], 1) for b0, b1 in bounds):
            errs.append(0.0)
            continue
        backend = HipStationaryBackend(48000, dev, chunk_size=c["cs"], padding=c["pad"], n_fft=1024)
        dt = torch.float32 if c["dtype"] == "float32" else torch.float64
        ext, shard = alloc_shard(c["C"], s1 - s0, c["pad"], dt, dev)
        shard.copy_(torch.from_numpy(y[:, s0:s1]))
        out = TimeShardedStationary(backend, 513).run(shard, ext=ext if c["pad"] else None)
        errs.append(float(np.max(np.abs(out.double().cpu().numpy() - want[:, s0:s1])) / np.max(np.abs(want))))
        dist.barrier()
    ret[rank] = errs
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, ncases, ret), nprocs=world, join=True)
    worst = max(max(v) for v in ret.values())
    print("world", world, "cases", ncases, "worst rel err %.2e" % worst, {r: ["%.1e" % e for e in v] for r, v in ret.items()})
    assert worst < 3e-4
