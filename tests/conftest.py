"""pytest configuration: markers, package loading, native test helpers."""
import importlib.util
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import ray-tracer_amd/ (hyphenated directory) as module `ray_tracer_amd`."""
    if "ray_tracer_amd" in sys.modules:
        return sys.modules["ray_tracer_amd"]
    pkg_dir = ROOT / "ray-tracer_amd"
    spec = importlib.util.spec_from_file_location("ray_tracer_amd", pkg_dir / "__init__.py",
                                                  submodule_search_locations=[str(pkg_dir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ray_tracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def _make(directory: Path, target_file: Path):
    """Always run make: the Makefiles track their sources, so this is a no-op when the helper is current and a rebuild
    when an edited kernel / lane header / oracle would otherwise be tested through a stale .so."""
    r = subprocess.run(["make", "-C", str(directory)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"make -C {directory} failed:\n{r.stdout}\n{r.stderr}")
    return target_file


@pytest.fixture(scope="session")
def rt():
    mod = load_package()
    import os
    if not os.environ.get("RT_MI355X_LIB"):
        _make(ROOT / "ray-tracer_amd" / "csrc", mod.LIB_PATH)
    mod.lib()
    if not os.environ.get("RT_MI355X_LIB"):
        assert mod.build_hash() == mod.source_hash(), "librt_mi355x.so was not built from the sources in this tree"
    return mod


@pytest.fixture(scope="session")
def scenes(rt):
    import importlib
    return importlib.import_module("ray_tracer_amd.scenes")


@pytest.fixture(scope="session")
def oracle():
    _make(ROOT / "oracle", ROOT / "oracle" / "_build" / "librt_oracle.so")
    import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def lane_emul(rt):
    _make(ROOT / "tests", ROOT / "tests" / "_build" / "liblane_emul.so")
    import lane_emul_binding
    return lane_emul_binding


@pytest.fixture(scope="session")
def lane_devmath(lane_emul):
    """the same harness compiled with the DEVICE's arithmetic forms (tests/Makefile, -DRT_EMULATE_DEVICE_MATH)"""
    ns = lane_emul.load("liblane_emul_devmath.so")
    assert ns.device_math and not lane_emul.device_math
    return ns


@pytest.fixture(scope="session")
def gpu_device(rt):
    n = rt.device_count()
    if n < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box (there is no CPU fallback)")
    return 0
