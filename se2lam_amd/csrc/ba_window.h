// One workgroup per local window (csrc/ba_window.hip): the argument pack of a window and the launcher.  csrc/ba.hip fills the packs
// from its handles (se2gpu_ba_optimize_batch) - the kernel knows nothing about handles, streams or pools.
#pragma once
#include "ba_device.h"
#include "common.h"

namespace se2gpu {

struct WindowArgs {
    badev::CamDev cam;
    int P, L, E, O, iters, mode;
    const int* lm_ptr;        // L + 1: the observation edges are sorted by landmark
    const int* e_kf;          // E: pose index of an edge
    const double* e_uv;       // E x 2
    const double* e_info;     // E x 3 (xx, xy, yy)
    double* poses_a;          // P x 3, the two estimate buffers (BaCtl::sel says which one holds the estimate)
    double* poses_b;
    double* lms_a;            // L x 3
    double* lms_b;
    const uint8_t* fixed;     // P
    const int* o_i;           // O: PreEdgeSE2 (this key frame, next key frame)
    const int* o_j;
    const double* o_meas;     // O x 3
    const double* o_info;     // O x 9
    badev::BaCtl* ctl;        // the window's controller block (device)
    double* mail;             // device address of the window's mapped mailbox, or NULL
    const int* stop;          // device address of the mapped force-stop word, or NULL
    int4* desc;               // scratch: the kernel lists the landmarks by their observation counts here (L x 16 B) and copies the
                              // observations into that order behind the list (E x 44 B)
    double* ainv;             // L x 6 of scratch: the build pass leaves every landmark's factor A here (list order) for the update pass
    int debug;                // (unused: the run-time debug switch is gone, the field keeps the pack's layout)
    long long* stamps;        // debug: 16 phase time stamps (100 MHz wall clock) of the LAST trial, or NULL
};

constexpr int kWindowMaxDegree = 64;     // observations of one landmark the kernel takes (a wave per landmark beyond 16)

// dynamic LDS a window of P poses, nfree of them free, needs with `threads` threads per workgroup (0: does not fit 160 KiB)
size_t ba_window_lds_bytes(int P, int nfree, int threads);
// count workgroups of `threads` (128, 256 or 512) threads, one per pack; asynchronous on st
int ba_window_launch(const WindowArgs* d_args, int count, int threads, size_t lds_bytes, hipStream_t st);

}  // namespace se2gpu
