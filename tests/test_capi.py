"""CPU tests of the C-ABI boundary: the library loads, exports every function include/se2gpu.h declares, fails
loudly without a device (no CPU fallback), and the C++ adapters (include/se2lam_amd/*.h) compile and link."""
import os
import re
import shutil
import subprocess
import ctypes as C

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "se2gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(se2gpu_[A-Za-z0-9_]+)\s*\(", txt))
    names -= {"se2gpu_allreduce_fn"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from se2lam_amd import capi
    lib = capi.lib()
    declared = _declared_functions()
    assert len(declared) > 50
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table covers the header exactly
    assert sorted(capi.SYMBOLS) == declared


def test_struct_layouts():
    from se2lam_amd import capi
    assert C.sizeof(capi.Keypoint) == 28          # cv::KeyPoint
    assert C.sizeof(capi.OrbParams) == 32
    assert C.sizeof(capi.FrameBounds) == 16
    assert C.sizeof(capi.BaStats) == 4 * 4 + 3 * 8 + 64 * 8 * 2 + 64 * 4


def test_no_cpu_fallback_without_device():
    from se2lam_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    from se2lam_amd.optimizer import SlamOptimizer
    from se2lam_amd.orb import ORBextractor
    from se2lam_amd.matcher import ORBmatcher
    for ctor in (SlamOptimizer, ORBextractor, ORBmatcher):
        with pytest.raises(capi.Se2GpuError) as e:
            ctor()
        assert e.value.code == capi.ERR_NO_DEVICE
    assert capi.lib().se2gpu_hamming(None, None) == capi.ERR_INVALID


def test_product_never_imports_oracle():
    """The product path (se2lam_amd/, include/) never imports, links or executes anything under oracle/."""
    pat_py = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[/.]_build|liboracle|orb_ref|ba_ref|match_ref", re.M)
    for top in ("se2lam_amd", "include"):
        for base, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                    continue
                txt = open(os.path.join(base, f)).read()
                assert not pat_py.search(txt), os.path.join(base, f)


def _build_adapter_binary(tmp_path):
    cxx = shutil.which("g++")
    assert cxx
    out = str(tmp_path / "cpp_adapters")
    libdir = os.path.join(ROOT, "se2lam_amd", "lib")
    subprocess.check_call([cxx, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp_adapters_compile.cpp"), "-o", out, "-L", libdir, "-lse2gpu",
                           "-Wl,-rpath," + libdir])
    return out


def test_cpp_adapters_compile_and_link(tmp_path):
    exe = _build_adapter_binary(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout or "adapters" in r.stdout


@pytest.mark.gpu
def test_cpp_adapters_run_on_gpu(tmp_path):
    exe = _build_adapter_binary(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "localBA adapters" in r.stdout and "ORBextractor adapter" in r.stdout and "ORBmatcher adapter" in r.stdout
