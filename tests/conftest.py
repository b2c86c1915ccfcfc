import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Test batches are tiny, where the engine's size heuristic would always pick the direct convolution kernel:
# force the Winograd kernel wherever it is legal so whole-network tests cover it (individual tests switch to
# "0" = direct / "1" = heuristic with monkeypatch.setenv).
os.environ.setdefault("SSDE_WINOGRAD", "2")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    # The CPU suite (`-m "not gpu"` in the build container) spends its time in the kernel emulator, one lane-fiber at a time:
    # without a GPU in sight and without an explicit -n, fan the tests out over pytest-xdist workers (each test is
    # self-contained; the emulator library is built once, before the workers start).  SSDE_TEST_WORKERS=0 switches it off,
    # =N picks the count.  On the GPU box the suite stays one process: the tests time kernels and share one device.
    want = os.environ.get("SSDE_TEST_WORKERS", "")
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return None                                  # (a worker: it must not fan out again)
    if want == "0" or not hasattr(config.option, "numprocesses") or config.option.numprocesses is not None:
        return None
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return None
    try:
        import torch
        if torch.cuda.is_available():
            return None
    except Exception:
        return None
    n = int(want) if want.isdigit() else max(1, min(6, (os.cpu_count() or 2) - 2))
    if n > 1:
        try:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import emu
            if emu.available():
                emu.build_emu.build()          # once, here: the workers find it built
        except Exception:
            pass
        config.option.numprocesses = n
        if getattr(config.option, "dist", "no") == "no":
            config.option.dist = "load"
        os.environ["SSDE_TEST_THREADS"] = str(max(1, (os.cpu_count() or n) // n))
    return None


def pytest_configure(config):
    # the CPU oracle (small torch ops) crawls when oneDNN fans out over the 256 hardware threads of the GPU box
    import torch
    torch.set_num_threads(int(os.environ.get("SSDE_TEST_THREADS", min(16, os.cpu_count() or 1))))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# the 3-way bf16 split of the GEMM-shaped contractions (SSDE_MATRIX=bf16x6 -> SSDE_CONVF_BF16X6) is the mode
# bench.py's headline runs in, so EVERY GPU test runs in both matrix modes inside one `pytest -m gpu` invocation -- forwards,
# trajectories, training steps, the 571 gradients, bench sizes, C-host plans -- at unchanged tolerances.  (The mode reaches the
# library as a flag of the launch arguments the host builds, _lib.conv_route_flags; tests that pin a mode themselves override it.)
MATRIX_MODES = ("f32", "bf16x6")


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is not None and "ssde_matrix_mode" in metafunc.fixturenames:
        metafunc.parametrize("ssde_matrix_mode", MATRIX_MODES, indirect=True)


@pytest.fixture(autouse=True)
def ssde_matrix_mode(request, monkeypatch):
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("SSDE_MATRIX", mode)
    yield mode


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
