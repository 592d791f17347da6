"""tools/chol64_probe.hip - the sixteen-wave, 64-wide block column of DESIGN.md 8.1 (ii), an experiment for the next round - has its
wave protocol (LDS progress counters, who reads which pivot row when, the T waves one hop behind the D waves) run on the CPU by
tools/waveemu under several seeded interleavings of the waves, and its result compared with a plain elimination.  This checks
logic only; what the experiment is for - the cost of a pivot with sixteen waves on one LDS - needs the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("catch", [2, 4])
def test_sixteen_wave_block_column_under_the_wave_emulator(tmp_path, catch):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "chol64_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-DWAVEEMU", f"-DCATCH={catch}", "-I", os.path.join(ROOT, "tools", "waveemu"), "-x", "c++",
                        os.path.join(ROOT, "tools", "chol64_probe.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "2", "2", "5"], capture_output=True, text=True, timeout=300)       # 2 blocks, 2 repetitions, 5 interleavings
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("(ok)") == 5 and "MISMATCH" not in r.stdout
