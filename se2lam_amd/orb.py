"""Python mirror of se2lam::ORBextractor over the C ABI (harness view; C++ twin: include/se2lam_amd/ORBextractor.h).

Reference interface: /root/reference/include/se2lam/ORBextractor.h:38-83
    ORBextractor(nfeatures=1000, scaleFactor=1.2f, nlevels=8, scoreType=FAST_SCORE, fastTh=20)
    operator()(image, mask, keypoints, descriptors);  GetLevels();  GetScaleFactor()
All compute happens in libse2gpu.so (HIP); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

HARRIS_SCORE, FAST_SCORE = 0, 1
KP_DTYPE = capi.KP_DTYPE


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=FAST_SCORE, fastTh=20,
                 max_rows=480, max_cols=640, max_batch=1):
        p = capi.OrbParams(nfeatures, scaleFactor, nlevels, scoreType, fastTh, max_rows, max_cols, max_batch)
        self._h = C.c_void_p()
        capi.check(capi.lib().se2gpu_orb_create(C.byref(p), C.byref(self._h)))
        self.nfeatures = nfeatures
        self.max_batch = max_batch

    def GetLevels(self) -> int:
        return int(capi.lib().se2gpu_orb_levels(self._h))

    def GetScaleFactor(self) -> float:
        return float(capi.lib().se2gpu_orb_scale_factor(self._h))

    def __call__(self, image, mask=None, cap=None):
        """-> (keypoints (n,) structured cv::KeyPoint array, descriptors (n,32) uint8)"""
        cap = cap or 2 * self.nfeatures
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        if image is None or image.size == 0:
            img_ptr, rows, cols, step = None, 0, 0, 0
        else:
            image = np.ascontiguousarray(image, np.uint8)
            assert image.ndim == 2
            img_ptr, (rows, cols), step = image.ctypes.data, image.shape, image.strides[0]
        mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)   # kept in a name until the call returns
        mask_ptr = None if mask is None else mask.ctypes.data
        capi.check(capi.lib().se2gpu_orb_extract(self._h, img_ptr, rows, cols, step, mask_ptr, kps.ctypes.data,
                                                 desc.ctypes.data, cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    # -- batched, device resident -------------------------------------------------------------
    def extract_batch_device(self, d_imgs, nframes, rows, cols, d_kps, d_desc, d_counts, cap):
        capi.check(capi.lib().se2gpu_orb_extract_batch_device(self._h, d_imgs, nframes, rows, cols, d_kps, d_desc,
                                                              d_counts, cap))

    def score_kernel(self):
        """-> (kernel the next batch would be scored with: 'dense' | 'sparse', last measured FAST candidate density or None)"""
        info = np.zeros(2, np.int32)
        capi.check(capi.lib().se2gpu_orb_score_kernel(self._h, info.ctypes.data))
        return ("sparse" if info[0] else "dense"), (None if info[1] < 0 else info[1] * 1e-6)

    def sync(self):
        capi.check(capi.lib().se2gpu_orb_sync(self._h))

    def extract_batch(self, images, cap=None):
        """images (B, rows, cols) u8 host array -> list of (kps, desc); uses the batched device path."""
        images = np.ascontiguousarray(images, np.uint8)
        B, rows, cols = images.shape
        cap = cap or 2 * self.nfeatures
        d_img = capi.DeviceArray.from_numpy(images)
        d_kps = capi.DeviceArray(B * cap * 28)
        d_desc = capi.DeviceArray(B * cap * 32)
        d_cnt = capi.DeviceArray(B * 4)
        self.extract_batch_device(d_img.ptr, B, rows, cols, d_kps.ptr, d_desc.ptr, d_cnt.ptr, cap)
        self.sync()
        cnt = d_cnt.to_numpy(np.int32, (B,))
        kps = d_kps.to_numpy(KP_DTYPE, (B, cap))
        desc = d_desc.to_numpy(np.uint8, (B, cap, 32))
        return [(kps[b, :cnt[b]].copy(), desc[b, :cnt[b]].copy()) for b in range(B)]

    def debug_level(self, frame, level, blurred=False, bordered=False):
        out = np.zeros(4096 * 4096 // 4, np.uint8)
        r = C.c_int(); c = C.c_int()
        capi.check(capi.lib().se2gpu_orb_debug_level(self._h, frame, level, int(blurred) | (2 if bordered else 0),
                                                     out.ctypes.data, out.size,
                                                     C.byref(r), C.byref(c)))
        return out[:r.value * c.value].reshape(r.value, c.value).copy()

    def debug_score(self, frame, level):
        out = np.zeros(4096 * 4096 // 4, np.uint8)
        r = C.c_int(); c = C.c_int()
        capi.check(capi.lib().se2gpu_orb_debug_score(self._h, frame, level, out.ctypes.data, out.size,
                                                     C.byref(r), C.byref(c)))
        return out[:r.value * c.value].reshape(r.value, c.value).copy()

    def stream(self):
        return capi.lib().se2gpu_orb_stream(self._h)

    def profile(self, enable: bool):
        capi.check(capi.lib().se2gpu_orb_profile(self._h, int(enable)))

    def profile_report(self):
        out = {}
        i = 0
        while True:
            name = C.c_char_p(); ms = C.c_double(); n = C.c_int64()
            if capi.lib().se2gpu_orb_profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(n)) != 0:
                break
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    def __del__(self):
        try:
            if self._h:
                capi.lib().se2gpu_orb_destroy(self._h)
                self._h = None
        except Exception:
            pass

