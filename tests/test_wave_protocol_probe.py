"""tools/chol64_probe.hip and tools/chol64_solve.hip - the sixteen-wave, 64-wide block column of DESIGN.md 8.1 (ii): the elimination
alone, and the whole dense pose solve as a standalone prototype of k_chol_tiles at that tile size - are experiments for the next
round, written without a GPU.  Their LOGIC runs here on the CPU under tools/waveemu (every work-item a fibre, the waves of a
task interleaved at random from a seed, wave collectives and the MFMA through an exchange buffer): LDS progress counters, who
reads which pivot row when, the T waves one hop behind the D waves, the plan, the flags, the staging counters, the MFMA operand
mapping, the publish layout, the x tasks.  What the experiments are for - the cost of a pivot and of a block column with
sixteen waves on one LDS - needs the GPU."""
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


@pytest.mark.parametrize("n", [("63",), ("128",), ("130",), ("nd", "1", "2")])
def test_64_wide_solve_prototype_under_the_wave_emulator(tmp_path, n):
    """n = 63: one tile, the rhs row inside it (the diagonal task hands y over in LDS); 128: the rhs row opens a tile row of its own;
    130: three block columns, the last a 2-column panel (pad pivots), R tiles, x tasks over three tile rows; nd 1 2: two uncoupled
    arcs of one tile and a separator of two - tasks without a tile in some block column, R tiles that do not exist."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "chol64_solve_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-DWAVEEMU", "-Wno-psabi", "-I", os.path.join(ROOT, "tools", "waveemu"), "-x", "c++",
                        os.path.join(ROOT, "tools", "chol64_solve.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, *n, "1", "2"], capture_output=True, text=True, timeout=600)       # 1 repetition, 2 interleavings
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("(ok)") == 2 and "MISMATCH" not in r.stdout
