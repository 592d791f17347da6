"""experiments/chol64_probe.hip and experiments/chol64_solve.hip - the sixteen-wave, 64-wide block column of docs/history/DESIGN_rounds_1-5.md 8.1 (ii): the elimination
alone, and the whole dense pose solve as a standalone prototype of k_chol_tiles at that tile size - are experiments for the next
round, written without a GPU.  Their LOGIC runs here on the CPU under experiments/waveemu (every work-item a fibre, the waves of a
task interleaved at random from a seed, wave collectives and the MFMA through an exchange buffer): LDS progress counters, who
reads which pivot row when, the T waves one hop behind the D waves, the plan, the flags, the staging counters, the MFMA operand
mapping, the publish layout, the x tasks.  What the experiments are for - the cost of a pivot and of a block column with
sixteen waves on one LDS - needs the GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("catch", [2, 4])
def test_sixteen_wave_block_column_under_the_wave_emulator(tmp_path, catch):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "chol64_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-DWAVEEMU", f"-DCATCH={catch}", "-I", os.path.join(ROOT, "experiments", "waveemu"), "-x", "c++",
                        os.path.join(ROOT, "experiments", "chol64_probe.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "2", "2", "6"], capture_output=True, text=True, timeout=300)       # 2 blocks, 2 repetitions, 6 interleavings
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("(ok)") == 6 and "MISMATCH" not in r.stdout


@pytest.mark.parametrize("n", [("63",), ("128",), ("130",), ("nd", "1", "2")])
def test_64_wide_solve_prototype_under_the_wave_emulator(tmp_path, n):
    """n = 63: one tile, the rhs row inside it (the diagonal task hands y over in LDS); 128: the rhs row opens a tile row of its own;
    130: three block columns, the last a 2-column panel (pad pivots), R tiles, x tasks over three tile rows; nd 1 2: two uncoupled
    arcs of one tile and a separator of two - tasks without a tile in some block column, R tiles that do not exist."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "chol64_solve_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-DWAVEEMU", "-Wno-psabi", "-I", os.path.join(ROOT, "experiments", "waveemu"), "-x", "c++",
                        os.path.join(ROOT, "experiments", "chol64_solve.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, *n, "1", "3"], capture_output=True, text=True, timeout=600)       # 1 repetition, 3 interleavings
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("(ok)") == 3 and "MISMATCH" not in r.stdout and "2 of 2 bit-identical to the first" in r.stdout
    if n == ("130",):       # the launch side by side: three workgroups in flight on OS threads, polling each other's flags
        r = subprocess.run([exe, *n, "1", "3"], capture_output=True, text=True, timeout=600, env=dict(os.environ, SE2_EMU_RESIDENT="3"))
        assert r.returncode == 0 and r.stdout.count("(ok)") == 3 and "2 of 2 bit-identical to the first" in r.stdout, r.stdout + r.stderr


def test_shipped_dense_solve_kernel_under_the_wave_emulator(tmp_path):
    """The product's own d_chol_tiles (se2lam_amd/csrc/ba.hip) - cut out of the source at test time by
    experiments/waveemu/extract_chol_tiles.py, which fails if one of its anchors has moved - solves dense and arcs + separator systems
    on the CPU with its four waves interleaved adversarially (seeded weights starve a wave while the others run ahead).  A logic
    race inside a workgroup - slab counters against the staging tiles, the multiplier columns overlaid on Tc / Ta, `ready_s` against
    COLV / MRC - shows as a wrong solution or a hang.  (Removing the kernel's `loaded_s` wait is caught in 4 of 100 interleavings:
    profiles/r04_chol32_emulated.txt; the long campaign is there too.)"""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    inc = tmp_path / "chol32_body.inc"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "waveemu", "extract_chol_tiles.py"), str(inc)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = str(tmp_path / "chol32_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "experiments", "waveemu"), "-pthread", "-I", str(tmp_path), os.path.join(ROOT, "experiments", "chol32_emu.cpp"),
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for args, n in ((("100", "12"), 12), (("63", "4"), 4), (("nd", "1", "2", "1", "4"), 4)):
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.count("(ok)") == n and "MISMATCH" not in r.stdout, r.stdout + r.stderr
        assert f"{n - 1} of {n - 1} bit-identical to the first" in r.stdout      # the result does not depend on the schedule
    # the whole launch side by side: three workgroups in flight on OS threads, dispatched in index order, polling each other's flags
    # (with the flag test of the staging step removed this mode fails at once, the one-after-the-other mode above cannot see it)
    r = subprocess.run([exe, "100", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, SE2_EMU_INDEFINITE="1"))
    assert r.returncode == 0 and "the failure flag is raised" in r.stdout, r.stdout + r.stderr       # a negated diagonal entry is reported
    r = subprocess.run([exe, "100", "6"], capture_output=True, text=True, timeout=900, env=dict(os.environ, SE2_EMU_RESIDENT="3"))
    assert r.returncode == 0 and r.stdout.count("(ok)") == 6 and "5 of 5 bit-identical to the first" in r.stdout, r.stdout + r.stderr


def test_staged_patches_still_apply():
    """experiments/patches holds changes to the product that were written and checked as far as a machine without a GPU allows; they must
    keep applying to the tree they were written against (a change of the kernel underneath is noticed here, not next round)."""
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    pdir = os.path.join(ROOT, "experiments", "patches")
    patches = sorted(f for f in os.listdir(pdir) if f.endswith(".patch"))
    if not patches:
        pytest.skip("nothing is staged (round 5 applied one of round 4's two patches and dropped the other: experiments/patches/README.md)")
    for f in patches:
        r = subprocess.run(["git", "apply", "--check", os.path.join(pdir, f)], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, f + ": " + r.stderr


@pytest.mark.parametrize("P", [21, 33, 64])
def test_experiments_plan_equals_the_products_plan(tmp_path, P):
    """experiments/waveemu/chol_host.h builds the task plan the experiments run on (dense, arcs + separator); the product's is
    solve_plan_build in csrc/ba.hip.  On a dense system of P poses the two must be the same lists - the emulated kernel then runs on
    exactly what the device kernel is given."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import numpy as np
    from test_solve_plan import _plan
    inc = tmp_path / "chol32_body.inc"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "waveemu", "extract_chol_tiles.py"), str(inc)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = str(tmp_path / "chol32_emu")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "experiments", "waveemu"), "-I", str(tmp_path),
                        os.path.join(ROOT, "experiments", "chol32_emu.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, str(3 * P)], capture_output=True, text=True, env=dict(os.environ, SE2_EMU_PRINT_PLAN="1")).stdout.split("\n")
    tasks = np.array([[int(v) for v in l.split()[1:]] for l in out if l.startswith("T ")], np.int32)
    deps = np.array([int(l.split()[1]) for l in out if l.startswith("D ")], np.int32)
    want = _plan(P, 3, None, allow_nd=False)
    assert np.array_equal(tasks, want["tasks"]) and np.array_equal(deps, want["deps"])


def test_64_wide_prototype_on_the_products_own_tile_64_plan(tmp_path):
    """The product's plan code at tile 64 (se2gpu_ba_debug_solve_plan_tile: nested dissection of an open band of 120 key frames, the
    partitions padded to tile boundaries in the middle of the system) handed to the prototype kernel as it is - task list, dependency
    lists, the padded system - under the emulator, one task after the other and three in flight.  (The ring of 200 key frames, 704
    columns and 121 tasks, takes a minute per interleaving: by hand, `chol64_solve_emu file <path>`.)"""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import numpy as np
    import test_solve_plan as T
    P, D, tile = 120, 3, 64
    pattern = T._band(P, 12, False)
    rng = np.random.default_rng(5)
    S = T._random_spd(rng, P, D, pattern)
    b = rng.normal(size=D * P)
    plan = T._plan(P, D, pattern, True, tile=tile)
    assert plan["nsys"] % tile == 0 and plan["depth"] < plan["nbc"]          # re-ordered and padded
    nsys, off = plan["nsys"], plan["off"]
    ld = -(-(nsys + 1) // tile) * tile
    A = np.zeros((ld, ld))
    cols = np.concatenate([off[p] + np.arange(D) for p in range(P)])
    A[np.ix_(cols, cols)] = S
    pad = np.setdiff1d(np.arange(nsys), cols)
    A[pad, pad] = 1.0
    A[nsys, cols] = b
    path = str(tmp_path / "system.bin")
    with open(path, "wb") as f:
        f.write(np.array([nsys, ld, len(plan["tasks"]), len(plan["deps"])], np.int32).tobytes())
        f.write(np.ascontiguousarray(plan["tasks"], np.int32).tobytes())
        f.write(np.ascontiguousarray(plan["deps"], np.int32).tobytes())
        f.write(np.ascontiguousarray(A, np.float64).tobytes())
    exe = str(tmp_path / "chol64_solve_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-DWAVEEMU", "-I", os.path.join(ROOT, "experiments", "waveemu"), "-x", "c++",
                        os.path.join(ROOT, "experiments", "chol64_solve.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for env in ({}, {"SE2_EMU_RESIDENT": "3"}):
        r = subprocess.run([exe, "file", path, "1", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.count("(ok)") == 1 and "MISMATCH" not in r.stdout, r.stdout + r.stderr


def _write_system(path, P, pattern, tile, seed=5, D=3):
    """a random SPD system with the given pose pattern, laid out by the PRODUCT'S plan (nested dissection, padded partitions), and that plan"""
    import numpy as np
    import test_solve_plan as T
    rng = np.random.default_rng(seed)
    S = T._random_spd(rng, P, D, pattern)
    b = rng.normal(size=D * P)
    plan = T._plan(P, D, pattern, True, tile=tile)
    nsys, off = plan["nsys"], plan["off"]
    ld = -(-(nsys + 1) // tile) * tile
    A = np.zeros((ld, ld))
    cols = np.concatenate([off[p] + np.arange(D) for p in range(P)])
    A[np.ix_(cols, cols)] = S
    pad = np.setdiff1d(np.arange(nsys), cols)
    A[pad, pad] = 1.0
    A[nsys, cols] = b
    with open(path, "wb") as f:
        f.write(np.array([nsys, ld, len(plan["tasks"]), len(plan["deps"])], np.int32).tobytes())
        f.write(np.ascontiguousarray(plan["tasks"], np.int32).tobytes())
        f.write(np.ascontiguousarray(plan["deps"], np.int32).tobytes())
        f.write(np.ascontiguousarray(A, np.float64).tobytes())
    return plan


def test_shipped_kernel_on_the_products_own_plan(tmp_path):
    """The shipped d_chol_tiles on what the product hands it for an open band of 120 key frames - nested dissection, partitions padded in
    the middle of the system, 125 tasks - under the emulator, one task after the other and four in flight; the 200 key-frame ring of
    the bench (608 columns, 333 tasks) by hand: profiles/r04_chol32_emulated.txt."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import test_solve_plan as T
    path = str(tmp_path / "system.bin")
    plan = _write_system(path, 120, T._band(120, 12, False), 32)
    assert plan["nsys"] % 32 == 0 and plan["depth"] < plan["nbc"]
    inc = tmp_path / "chol32_body.inc"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "waveemu", "extract_chol_tiles.py"), str(inc)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = str(tmp_path / "chol32_emu")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "experiments", "waveemu"), "-I", str(tmp_path),
                        os.path.join(ROOT, "experiments", "chol32_emu.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for env in ({}, {"SE2_EMU_RESIDENT": "4"}):
        r = subprocess.run([exe, "file", path, "1", "2"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.count("(ok)") == 2 and "1 of 1 bit-identical to the first" in r.stdout, r.stdout + r.stderr
