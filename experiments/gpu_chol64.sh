#!/bin/bash
# The two 64-wide block column experiments of docs/history/DESIGN_rounds_1-5.md 8.1 on the GPU box (each run under its own timeout; the kernels give up on a
# flag after 2 s by themselves):   gpurun --timeout 300 -- 'bash tools/gpu_chol64.sh'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out tools/bin
hipcc --offload-arch=gfx950 -O3 tools/chol64_probe.hip -o tools/bin/chol64_probe || exit 1
hipcc --offload-arch=gfx950 -O3 -DCATCH=4 tools/chol64_probe.hip -o tools/bin/chol64_probe4 || exit 1
hipcc --offload-arch=gfx950 -O3 -DPIVCHK=1 tools/chol64_probe.hip -o tools/bin/chol64_probe_s || exit 1
hipcc --offload-arch=gfx950 -O3 -I tools/waveemu tools/chol64_solve.hip -o tools/bin/chol64_solve || exit 1
{
    echo "== elimination alone, catch-up 2 columns per poll"; timeout 60 tools/bin/chol64_probe 1 20
    echo "== elimination alone, pivot test in scalar registers (2 VALU FP64 instructions per pivot fewer: would carry over to k_chol_tiles)"; timeout 60 tools/bin/chol64_probe_s 1 20
    echo "== elimination alone, catch-up 4 columns per poll (6 VGPRs spilled)"; timeout 60 tools/bin/chol64_probe4 1 20
    echo "== the solve, two arcs of 3 tiles + a separator of 5 (the 200 key-frame shape: 8 block columns on the chain)"; timeout 60 tools/bin/chol64_solve nd 3 5 20
    echo "== the solve, dense 600 columns (10 block columns on the chain)"; timeout 60 tools/bin/chol64_solve 600 20
} 2>&1 | tee gpurun_out/chol64.txt
