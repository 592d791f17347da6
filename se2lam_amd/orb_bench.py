"""ORB leg of bench.py: batched extract + MatchByWindow on frames resident in HBM.

A "step" is one batch of B synthetic 640x480 frames through the extractor (8 levels, 1000 features) and the
window matcher (frame t vs t+1 inside the batch, B pairs with wrap-around), i.e. B frames of extract+match.
N>1 ranks: frame-parallel replicas, no collective - weak scaling.
"""
from __future__ import annotations

import time

import numpy as np

from . import capi, synth
from .matcher import ORBmatcher
from .orb import ORBextractor

HBM_PEAK_GBS = 8000.0
B_ORB = 5_742_474      # algorithmic bytes per frame, SURVEY.md §8(d)
B_MATCH = 132_000      # algorithmic bytes per frame pair


MIN_TIMED_S = 2.0      # floor of the timed region (as bench.py's BA leg)


def run(rank, world, batch, steps, sync_all, dist, torch, warmup=2, cap=1024, traffic=None, inflight=2, streaming_leg=True):
    B = batch
    imgs = synth.frames(B, start=1000 * rank)
    # `inflight` extractor handles (own streams, own pyramids) take turns, so that batch k+1 (and k+2) is already queued while batch k runs:
    # the device never waits for the host's sync / launch turnaround between batches, and the latency-bound tail of one
    # batch (cell lists, level selection, orientation) overlaps the VALU-bound head of the next (pyramid, FAST score).
    nex = max(1, int(inflight))
    exs = [ORBextractor(max_batch=B) for _ in range(nex)]
    ex = exs[0]
    mt = ORBmatcher(0.9, max_features=cap, max_batch=B)
    d_img = capi.DeviceArray.from_numpy(imgs)
    # three output sets: the matcher of batch k (its own stream; 1 wave per pair, latency-bound) runs while the
    # extractors already work on batches k+1 and k+2
    bufs = []
    for _ in range(nex + 1):
        bufs.append(dict(kps=capi.DeviceArray(B * cap * 28), desc=capi.DeviceArray(B * cap * 32),
                         cnt=capi.DeviceArray(B * 4), m=capi.DeviceArray(B * cap * 4), nm=capi.DeviceArray(B * 4),
                         done=capi.Timer(), used=False))
    pa = np.arange(B, dtype=np.int32)
    pb = (pa + 1) % B
    d_pa = capi.DeviceArray.from_numpy(pa)
    d_pb = capi.DeviceArray.from_numpy(pb)
    state = {"k": 0, "pending": []}     # pending: batches whose extraction is queued but not yet matched

    def finish(k):
        b = bufs[k % len(bufs)]
        exs[k % nex].sync()                  # extraction of batch k complete (+ capacity check)
        b["done"].start(mt.stream())
        mt.match_window_batch_device(b["kps"].ptr, b["desc"].ptr, b["cnt"].ptr, cap, d_pa.ptr, d_pb.ptr, B, 20,
                                     b["m"].ptr, b["nm"].ptr)
        b["done"].stop(mt.stream())
        b["used"] = True

    def step():
        k = state["k"]
        state["k"] += 1
        b = bufs[k % len(bufs)]
        if b["used"]:
            b["done"].elapsed_ms()          # host-waits for the match that last read this buffer set (event sync)
        exs[k % nex].extract_batch_device(d_img.ptr, B, 480, 640, b["kps"].ptr, b["desc"].ptr, b["cnt"].ptr, cap)
        state["pending"].append(k)
        while len(state["pending"]) >= nex:  # (one handle: finish this batch now; two: the previous one)
            finish(state["pending"].pop(0))

    def drain():
        while state["pending"]:
            finish(state["pending"].pop(0))
        mt.sync()

    for _ in range(warmup):
        step()
    drain()
    # The timed region is never shorter than MIN_TIMED_S: `steps` batches are a probe when they take less (10 batches are
    # 15 ms, a quarter of which is the pipeline of two batches in flight filling and draining), and the sample is taken
    # again with as many batches as the floor needs.  `steps` in the result is what ran, `steps_requested` what was asked.
    steps_requested = steps
    while True:
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = dist.allreduce_max(dt)
        if dt >= MIN_TIMED_S:
            break
        steps = max(steps + 1, int(np.ceil(1.25 * steps * MIN_TIMED_S / max(dt, 1e-6))))
    fps = world * B * steps / dt
    last = bufs[(state["k"] - 1) % len(bufs)]
    cnt = last["cnt"].to_numpy(np.int32, (B,))
    nm = last["nm"].to_numpy(np.int32, (B,))
    d_kps, d_desc, d_cnt = last["kps"], last["desc"], last["cnt"]

    # per-kernel durations (HIP events around every extractor launch, separate pass)
    ex.profile(True)
    for _ in range(min(steps, 5)):
        ex.extract_batch_device(d_img.ptr, B, 480, 640, d_kps.ptr, d_desc.ptr, d_cnt.ptr, cap)
        ex.sync()
    prof = ex.profile_report()
    ex.profile(False)
    kern = {k: {"avg_us": 1e3 * ms / max(n, 1), "launches": n, "total_ms": ms} for k, (ms, n) in prof.items()}
    dom = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
    roof = None
    if dom:
        # one launch of an extractor kernel processes the B frames of the batch; its algorithmic bytes are its stage's
        # share of SURVEY.md section 8(d)'s pass-per-stage accounting (sum over the 8 pyramid levels = 950,532 px)
        stage = {"k_level0": 307_200 + 307_200, "k_resize": (926_546 + 643_332 - 307_200) / 7.0,
                 "k_fast_score": 950_532, "k_cell_collect": 60_000, "k_cell_retain": 60_000, "k_blur": 1_901_064,
                 "k_orientation": 749_000, "k_describe": 512_000 + 60_000, "k_level_select": 60_000}
        b_launch = stage.get(dom, B_ORB) * B
        ach = b_launch / (kern[dom]["avg_us"] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": (traffic or {}).get(dom, {}).get("traffic_bytes"),
                "traffic_stale": dom in (traffic or {}).get("_meta", {}).get("stale_kernels", []),
                "algorithmic_bytes_per_launch": b_launch,
                "avg_launch_us": kern[dom]["avg_us"],
                "whole_step_achieved": (B_ORB + B_MATCH) * fps / world / 1e9,
                "whole_step_frac": (B_ORB + B_MATCH) * fps / world / 1e9 / HBM_PEAK_GBS,
                "kernels_us": {k: round(v["avg_us"], 2) for k, v in kern.items()}}
        if dom == "k_fast_score":
            # The kernel is not bound by bytes but by VALU issue: its instruction count per frame from the SQ counters
            # of profiles/r05c_orb_sq_counters.json (63,488 waves x 4,438.6 VALU instructions per 256 frames; the dense kernel's
            # count does not depend on the image), 4 clk per wave64 instruction on a 16-lane SIMD, 1,024 SIMDs at the 2.4 GHz peak
            insts = 63_488 * 4_438.6 / 256.0 * B
            min_us = insts * 4.0 / 1024.0 / 2.4e3
            roof["valu"] = {"wave_instructions_per_launch": insts, "clk_per_instruction": 4, "simds": 1024, "clock_ghz_assumed": 2.4,
                            "min_launch_us": min_us, "frac": min_us / kern[dom]["avg_us"],
                            "source": "profiles/r05c_orb_sq_counters.json (SQ_INSTS_VALU per wave x waves)"}
            roof["bound_by"] = {"what": "VALU issue", "frac_of_valu_issue": min_us / kern[dom]["avg_us"]}
    import sys
    print(f"[orb_bench] resident {fps:.0f} frames/s with {nex} batches in flight", file=sys.stderr, flush=True)
    streaming = None
    if streaming_leg:
        try:   # (at most two handles: see streaming_child)
            streaming = run_streaming(rank, world, B, min(steps, 40), sync_all, dist, exs[:2], mt, cap)
            if nex > 2:
                streaming["note_handles"] = ("measured beside the resident leg's third extractor handle, whose streams take the hardware "
                                             "queue the upload stream would have had: a caller that streams keeps two handles")
        except Exception as exc:   # the streaming leg must never take the headline numbers down with it
            streaming = {"error": repr(exc)}
    return {"metric": "ORB extract+match frames/s @640x480", "value": fps, "unit": "frames/s", "n_gpus": world,
            "batch": B, "steps": steps, "steps_requested": steps_requested, "timed_s": dt, "ms_per_batch": 1e3 * dt / steps,
            "scaling": "weak", "batches_in_flight": nex, "gpu_max_hw_queues": _hw_queues(),
            "config": {"workload": "640x480 u8, 8-level pyramid, 1000 features/frame, MatchByWindow(win 20, ratio 0.9), "
                                   "frames t vs t+1", "features_per_frame": float(cnt.mean()),
                       "matches_per_pair": float(nm.mean())},
            "roofline": roof, "cpu_baseline": None, "streaming": streaming}


def _hw_queues():
    """GPU_MAX_HW_QUEUES of this process (the HIP runtime's default is 4): the resident and the streaming leg run with different
    settings, and a reader of the bench line should see which"""
    import os
    return os.environ.get("GPU_MAX_HW_QUEUES", "default (4)")


def streaming_child(B, steps, cap=1024):
    """The PCIe-inclusive leg in a process of its own, as a streaming caller is one: two extractor handles, the matcher, the
    upload stream - and GPU_MAX_HW_QUEUES=32 (bench.py sets it for this child), so that every stream has a hardware queue to
    itself.  With the default of four queues, dealt in the order the streams are created, the upload stream lands behind an
    extractor's and the leg measures 83-119 k frames/s depending on how many streams the process happened to create before
    (tools/orb_stream_probe.py); with 32 it is 155 k, the rate of the PCIe link.  The resident leg wants the opposite (202 k
    frames/s with four queues, 185 k with eight or more), which is why the two legs are two processes since round 5."""
    exs = [ORBextractor(max_batch=B) for _ in range(2)]
    mt = ORBmatcher(0.9, max_features=cap, max_batch=B)
    d_img = capi.DeviceArray.from_numpy(synth.frames(B))
    kps, desc, cnt = capi.DeviceArray(B * cap * 28), capi.DeviceArray(B * cap * 32), capi.DeviceArray(B * 4)
    # resident batches on each handle first: their pyramid streams exist before the copy stream does, the score-kernel choice
    # has settled and the device is at its clocks (a fresh process that streams at once measures 100 k frames/s)
    for _ in range(24):
        for e in exs:
            e.extract_batch_device(d_img.ptr, B, 480, 640, kps.ptr, desc.ptr, cnt.ptr, cap)
        for e in exs:
            e.sync()

    def sync_all():
        capi.check(capi.lib().se2gpu_device_synchronize())

    out = run_streaming(0, 1, B, steps, sync_all, None, exs, mt, cap, warmup=8)
    out["process"] = "own process, two extractor handles"
    out["gpu_max_hw_queues"] = _hw_queues()
    return out


def run_streaming(rank, world, B, steps, sync_all, dist, exs, mt, cap, nhost=4, warmup=2):
    """The same extract + match step with the camera-side transfers IN the loop ("GPU numbers
    include H2D/D2H"): every batch is a FRESH set of B frames uploaded from pinned host memory (nhost distinct batches,
    round-robin) on a copy stream of its own, and the key points, descriptors and match lists of every batch are
    downloaded to pinned memory behind the match, on the matcher's stream.  The extractor handles alternate as in run();
    ordering is by events only (the timers' stop events): upload k+1 and match / download k-1 overlap extraction k; the
    host waits only where the library does (se2gpu_orb_sync).  Streams: one per extractor handle, the matcher's, the upload
    stream - one hardware queue each on a device with four."""
    frame_bytes = 480 * 640
    host = capi.PinnedArray((nhost, B, 480, 640), np.uint8)
    for i in range(nhost):
        host.array[i] = synth.frames(B, start=1000 * rank + 37 * i)
    up = capi.Stream()
    L = capi.lib()
    nex = len(exs)
    sets = []
    for _ in range(nex + 1):
        sets.append(dict(img=capi.DeviceArray(B * frame_bytes), kps=capi.DeviceArray(B * cap * 28),
                         desc=capi.DeviceArray(B * cap * 32), cnt=capi.DeviceArray(B * 4), m=capi.DeviceArray(B * cap * 4),
                         nm=capi.DeviceArray(B * 4),
                         h_kps=capi.PinnedArray((B * cap * 28,), np.uint8), h_desc=capi.PinnedArray((B * cap * 32,), np.uint8),
                         h_cnt=capi.PinnedArray((B,), np.int32), h_m=capi.PinnedArray((B * cap,), np.int32),
                         h_nm=capi.PinnedArray((B,), np.int32),
                         uploaded=capi.Timer(), extracted=capi.Timer(), downloaded=capi.Timer(), used=False))
    pa = np.arange(B, dtype=np.int32)
    d_pa = capi.DeviceArray.from_numpy(pa)
    d_pb = capi.DeviceArray.from_numpy((pa + 1) % B)
    state = {"k": 0, "pending": []}

    def upload(k):
        b = sets[k % len(sets)]
        if b["used"]:
            b["extracted"].make_wait(up.h)       # the extraction that last read this image buffer
        b["uploaded"].start(up.h)
        capi.check(L.se2gpu_memcpy_h2d_async(b["img"].ptr, host.array[k % nhost].ctypes.data, B * frame_bytes, up.h))
        b["uploaded"].stop(up.h)

    def finish(k):
        b = sets[k % len(sets)]
        exs[k % nex].sync()
        mt.match_window_batch_device(b["kps"].ptr, b["desc"].ptr, b["cnt"].ptr, cap, d_pa.ptr, d_pb.ptr, B, 20,
                                     b["m"].ptr, b["nm"].ptr)
        b["downloaded"].start(mt.stream())
        for dst, src, n in ((b["h_kps"], b["kps"], B * cap * 28), (b["h_desc"], b["desc"], B * cap * 32),
                            (b["h_cnt"], b["cnt"], B * 4), (b["h_m"], b["m"], B * cap * 4), (b["h_nm"], b["nm"], B * 4)):
            capi.check(L.se2gpu_memcpy_d2h_async(dst.array.ctypes.data, src.ptr, n, mt.stream()))
        b["downloaded"].stop(mt.stream())
        b["used"] = True

    def step():
        k = state["k"]
        state["k"] += 1
        b = sets[k % len(sets)]
        ex = exs[k % nex]
        upload(k + 1)                            # next batch's frames travel while this one is extracted
        b["uploaded"].make_wait(ex.stream())
        if b["used"]:
            b["downloaded"].make_wait(ex.stream())   # the results of the batch that last used these buffers have left
        b["extracted"].start(ex.stream())
        ex.extract_batch_device(b["img"].ptr, B, 480, 640, b["kps"].ptr, b["desc"].ptr, b["cnt"].ptr, cap)
        b["extracted"].stop(ex.stream())
        state["pending"].append(k)
        while len(state["pending"]) >= nex:
            finish(state["pending"].pop(0))

    def drain():
        while state["pending"]:
            finish(state["pending"].pop(0))
        mt.sync()

    upload(0)
    for _ in range(warmup):
        step()
    drain()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = dist.allreduce_max(dt)
    last = sets[(state["k"] - 1) % len(sets)]
    per_batch_up = B * frame_bytes
    per_batch_down = B * cap * (28 + 32 + 4) + 8 * B
    return {"value": world * B * steps / dt, "unit": "frames/s", "ms_per_batch": 1e3 * dt / steps,
            "h2d_bytes_per_batch": per_batch_up, "d2h_bytes_per_batch": per_batch_down,
            "h2d_gbs": per_batch_up * steps / dt / 1e9, "d2h_gbs": per_batch_down * steps / dt / 1e9,
            "features_per_frame": float(last["h_cnt"].array.mean()), "matches_per_pair": float(last["h_nm"].array.mean()),
            "batches_in_flight": nex,
            "note": "fresh frames uploaded from pinned memory every batch (own stream), key points + descriptors + match "
                    "lists downloaded every batch behind the match; PCIe-inclusive, never the headline `value`"}
