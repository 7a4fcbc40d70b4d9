"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the
header declares, the Python mirror resolves parameters like the reference, and the product
path refuses to run without a GPU (no silent CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HAS_GPU = torch.cuda.is_available()


def test_cabi_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from noisereduce_amd import _ffi
    # product ABI + the development surface (stage taps, A/B options, per-kernel timing): same library
    header = open(os.path.join(ROOT, "include", "mi355gate.h")).read()
    debug_header = open(os.path.join(ROOT, "include", "mi355gate_debug.h")).read()
    product = set(re.findall(r"\b(sg_[a-z_0-9]+)\s*\(", header))
    # the product header carries no development switches, taps or profiling entry points
    assert not [n for n in product if n.startswith(("sg_debug_", "sg_profile_", "sg_stage_"))], product
    assert set(re.findall(r"#define (SG_OPT_\w+)", header)) == {"SG_OPT_FAST_INTEGER", "SG_OPT_FORCE_EXACT"}
    assert "SG_STAGE_" not in header
    declared = product | set(re.findall(r"\b(sg_[a-z_0-9]+)\s*\(", debug_header))
    lib = _ffi.load_library()
    assert declared == set(_ffi.exported_symbols()), declared ^ set(_ffi.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sg_version() == 100
    assert lib.sg_stage_name(8) == b"k_apply_istft"
    # and nothing else: the library is built with -fvisibility=hidden (no mangled kernel stubs leak)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True)
    exported = {ln.split()[-1] for ln in out.stdout.splitlines() if ln.strip() and ln.split()[-2] in "TDBRW"}
    assert exported == declared, sorted(exported ^ declared)


def test_sg_params_struct_layout_matches_header():
    """ctypes mirror of struct sg_params: same field order as the header."""
    from noisereduce_amd import _ffi
    header = open(os.path.join(ROOT, "include", "mi355gate.h")).read()
    body = header[header.index("typedef struct sg_params {"):header.index("} sg_params;")]
    names = re.findall(r"^\s*(?:int32_t|int64_t|double)\s+(\w+);", body, flags=re.M)
    assert names == [f[0] for f in _ffi.SgParams._fields_]


def test_filter_design_matches_oracle():
    from noisereduce_amd.spectralgate.base import _smoothing_filter, _triangle
    for nf, nt in [(5, 9), (16, 3), (1, 4), (7, 1)]:
        assert np.allclose(_smoothing_filter(nf, nt), O.smoothing_filter(nf, nt), rtol=0, atol=1e-16)
    assert np.allclose(_triangle(3), O.triangle(3))


def test_iir_coefficient():
    from noisereduce_amd.spectralgate.nonstationary import iir_coefficient
    assert iir_coefficient(2.0, 48000, 256) == O.iir_coefficient(2.0, 48000, 256)
    assert abs(iir_coefficient(2.0, 48000, 256) - 0.0026631) < 1e-7


def test_input_validation_happens_before_any_device_work():
    """The reference's ValueErrors (base.py:60,105-123) are raised by the mirror too."""
    from noisereduce_amd import reduce_noise
    with pytest.raises(ValueError):
        reduce_noise(np.zeros((2, 2, 100)), 48000)
    with pytest.raises(ValueError):
        reduce_noise(np.zeros(5000), 48000, freq_mask_smooth_hz=10)
    with pytest.raises(ValueError):
        reduce_noise(np.zeros(5000), 48000, time_mask_smooth_ms=1)
    with pytest.raises(ValueError):
        reduce_noise(np.zeros(5000), 48000, use_torch=True, n_jobs=2)


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    from noisereduce_amd import reduce_noise
    from noisereduce_amd.torchgate import TorchGate
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        reduce_noise(np.zeros(5000), 48000, stationary=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TorchGate(sr=16000)(torch.zeros(2, 4000))


def test_torchgate_module_surface():
    from noisereduce_amd.torchgate import TorchGate
    tg = TorchGate(sr=16000)
    assert (tg.n_fft, tg.win_length, tg.hop_length) == (1024, 1024, 256)
    assert tuple(tg.smoothing_filter.shape) == (1, 1, 33, 7)
    K = O.smoothing_filter(16, 3)
    assert np.allclose(tg.smoothing_filter[0, 0].numpy(), K, atol=1e-7)
    assert TorchGate(sr=16000, freq_mask_smooth_hz=None, time_mask_smooth_ms=None).smoothing_filter is None
    with pytest.raises(ValueError):
        TorchGate(sr=48000, freq_mask_smooth_hz=10)
    with pytest.raises(AssertionError):
        TorchGate(sr=16000, prop_decrease=1.5)


def _build_c_example(tmp_path):
    """gcc-compile tests/c_abi/example.c (plain C99, no torch, no C++) and link it against the built
    library and the HIP runtime.  Returns the path of the binary."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("gcc / HIP runtime not available")
    import __graft_entry__
    __graft_entry__.build()
    libdir = os.path.join(root, "noisereduce_amd")
    exe = str(tmp_path / "c_abi_example")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", os.path.join(root, "tests", "c_abi", "example.c"),
                    "-I" + os.path.join(root, "include"), "-L" + libdir, "-lmi355gate", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_header_is_plain_c_and_c_example_links(tmp_path):
    """include/mi355gate.h must be valid C99 (and C++), and a plain-C caller must link against the
    library with nothing but the HIP runtime next to it."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    for name in ("mi355gate.h", "mi355gate_debug.h"):
        hdr = os.path.join(root, "include", name)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                       check=True)
        subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", hdr], check=True)
    assert os.path.exists(_build_c_example(tmp_path))


def test_committed_bench_line_has_the_contract_fields():
    """The bench line committed under profiles/ (produced by `python bench.py` on an MI355X) carries every
    field of the driver's contract, incl. the roofline and cpu_baseline objects."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "profiles", "r*_bench.json"))
    assert files, "no bench line committed under profiles/"

    def version(path):  # r01_v11_bench.json -> (1, 11)
        m = re.match(r"r(\d+)_v(\d+)", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2)))
    line = json.loads(open(max(files, key=version)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "Msamples/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value is consistent with ms_per_step and the workload size
    assert abs(line["value"] - line["config"]["samples_per_gpu"] * line["n_gpus"] / line["ms_per_step"] / 1e3) \
        < 0.01 * line["value"]


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE) re-executes itself through torch.distributed.run with
    one rank per GPU on 127.0.0.1 (the driver's first multi-GPU run must measure something)."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", lambda c, env=None: calls.append((c, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    bench.main()
    assert len(calls) == 1 and calls[0][0][-6:] == ["--gpus", "2", "--steps", "1", "--warmup", "0"]
    assert calls[0][1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher with another world size: a clear error, no silent mismatch
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit):
        bench.main()


def test_bench_gpu_state_reader_degrades_without_a_device():
    """bench.py reads sclk / mclk / power through librocm_smi64 in-process; on a host without an AMD GPU (this build
    container) the reader reports why instead of raising, and the bench line simply carries that."""
    import importlib
    bench = importlib.import_module("bench")
    st = bench.GpuState(0).read()
    assert isinstance(st, dict)
    if not HAS_GPU:
        assert "error" in st or all(v is None for v in st.values())
    else:
        assert st.get("sclk_mhz") is None or st["sclk_mhz"] > 0
    assert bench.gpu_state() == {} or isinstance(bench.gpu_state(), dict)


def test_bench_dry_nccl_is_forwarded_to_the_ranks():
    import importlib
    import sys
    bench = importlib.import_module("bench")
    cmd = bench.launcher_command(8, ["--gpus", "8", "--dry-nccl"], port=29998)
    assert cmd[-3:] == ["--gpus", "8", "--dry-nccl"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[0] == sys.executable


def test_channel_bounds_cover_every_channel_once():
    from noisereduce_amd.sharded import channel_bounds
    for c_total in (1, 2, 7, 8, 63, 64, 65):
        for world in (1, 2, 3, 8):
            b = [channel_bounds(c_total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == c_total
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_pipeline_pieces_refuses_views_the_piecewise_upload_cannot_take():
    """Host logic only (no GPU): the pipelined upload slices the caller's array per piece with torch.from_numpy, which
    rejects negative strides; zero channels / samples have nothing to pipeline."""
    from noisereduce_amd.spectralgate import base
    n = 600000 * 8
    y = np.zeros((2, n), dtype=np.float32)

    def pieces(arr, c=2, frames=n, cs=600000):
        sg = base.SpectralGate.__new__(base.SpectralGate)
        sg._tensor_io, sg._chunk_size, sg._dtype, sg.y, sg.n_channels, sg.n_frames = False, cs, np.float32, arr, c, frames
        return sg._pipeline_pieces()

    ok = pieces(y)
    assert ok is not None and ok[0][0] == 0 and ok[-1][1] == n and all(a < b for a, b in ok)
    assert all(ok[i][1] == ok[i + 1][0] for i in range(len(ok) - 1))
    assert pieces(y[:, ::-1]) is None          # time-reversed view
    assert pieces(y[::-1]) is None             # channel-reversed view
    assert pieces(np.asfortranarray(y)) is not None   # strided but positive: torch.from_numpy takes it
    one = np.zeros(n, np.float32)
    assert pieces(np.expand_dims(one, 0), c=1) is not None                       # mono: whatever stride the unit axis got
    assert pieces(np.lib.stride_tricks.as_strided(one, (1, n), (0, 4)), c=1) is not None
    assert pieces(np.expand_dims(one[::-1], 0), c=1) is None
    assert pieces(np.zeros((0, n), np.float32), c=0) is None
    assert pieces(np.zeros((2, 0), np.float32), frames=0) is None
    assert pieces(y, cs=None) is None
