"""Batch-level drivers of the C-ABI kernels.

torch is used here only as plumbing: device / pinned allocations, the current HIP stream and
`torch.Tensor` as the hand-off type.  Every byte of arithmetic happens inside
libdali_amd_kernels.so (device) or libdali_amd_host.so (host-side entropy decode, random crop
generation); if the kernel library is missing the calls raise -- there is no CPU fallback.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _capi as capi

_TORCH_DTYPE = {capi.UINT8: torch.uint8, capi.FLOAT16: torch.float16, capi.FLOAT: torch.float32,
                capi.INT8: torch.int8}


def _align(v, a):
    return (v + a - 1) // a * a


def current_stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _DescUploader:
    """Pinned staging ring for descriptor tables (host -> device on the current stream).

    A staging buffer is reused only after the event recorded behind its copy has completed, so the
    launches stay asynchronous (no host synchronisation in the steady state)."""

    def __init__(self):
        self._free = []   # (pinned tensor, event)

    def upload(self, ctypes_array, device):
        nbytes = C.sizeof(ctypes_array)
        slot = None
        for i, (buf, ev) in enumerate(self._free):
            if buf.numel() >= nbytes and ev.query():
                slot = self._free.pop(i)
                break
        if slot is None:
            buf = torch.empty(max(nbytes, 4096), dtype=torch.uint8, pin_memory=True)
            ev = torch.cuda.Event()
        else:
            buf, ev = slot
        C.memmove(buf.data_ptr(), C.addressof(ctypes_array), nbytes)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dev.copy_(buf[:nbytes], non_blocking=True)
        ev.record()
        self._free.append((buf, ev))
        return dev


_uploader = _DescUploader()

_pool = None


def _thread_pool(num_threads=None):
    global _pool
    n = num_threads or os.cpu_count() or 1
    if _pool is None or _pool._max_workers != n:
        _pool = ThreadPoolExecutor(max_workers=n)
    return _pool


# =====================================================================================
# JPEG
# =====================================================================================
class JpegBatchPlan:
    """Geometry + buffer layout of one batch of JPEG streams (host side, no device work)."""

    def __init__(self, encoded, out_pitch_align=16):
        host = capi.host()
        self.n = len(encoded)
        self.encoded = [np.frombuffer(e, dtype=np.uint8) if not isinstance(e, np.ndarray) else
                        np.ascontiguousarray(e, dtype=np.uint8).reshape(-1) for e in encoded]
        self.infos = (capi.JpegInfo * max(self.n, 1))()
        for i, e in enumerate(self.encoded):
            capi.check_host(host.daliamdJpegParse(e.ctypes.data_as(C.c_void_p), C.c_size_t(e.size),
                                                  C.byref(self.infos[i])))
            inf = self.infos[i]
            if inf.num_components not in (1, 3):
                raise capi.DaliAmdError(
                    f"sample {i}: JPEG with {inf.num_components} components is not supported")
        # layout: coefficient elements (int16), plane bytes, output bytes
        self.coef_off = np.zeros((self.n, 3), np.int64)
        self.plane_off = np.zeros((self.n, 3), np.int64)
        self.out_off = np.zeros(self.n, np.int64)
        self.out_pitch = np.zeros(self.n, np.int64)
        co = po = oo = 0
        for i in range(self.n):
            inf = self.infos[i]
            for c in range(inf.num_components):
                self.coef_off[i, c] = co
                co += inf.coef_elems[c]
                self.plane_off[i, c] = po
                po += inf.coef_elems[c]          # one byte per coefficient
            self.out_off[i] = oo
            self.out_pitch[i] = _align(3 * inf.width, out_pitch_align)
            oo += _align(self.out_pitch[i] * inf.height, 256)
        self.coef_elems, self.plane_bytes, self.out_bytes = int(co), int(po), int(oo)
        self.quant = np.zeros((self.n, 3, 64), np.uint16)

    def shapes(self):
        return [(self.infos[i].height, self.infos[i].width, 3) for i in range(self.n)]

    def entropy_decode(self, coef_host, num_threads=None):
        """Huffman-decodes every stream into `coef_host` (int16 host tensor/array of
        self.coef_elems elements, ideally pinned).  Runs on a thread pool; the C call releases
        the GIL."""
        host = capi.host()
        base = coef_host.data_ptr() if isinstance(coef_host, torch.Tensor) else coef_host.ctypes.data

        def one(i):
            inf = self.infos[i]
            ptrs = (C.c_void_p * 4)()
            for c in range(inf.num_components):
                ptrs[c] = base + 2 * int(self.coef_off[i, c])
            e = self.encoded[i]
            rc = host.daliamdJpegDecodeCoefficients(e.ctypes.data_as(C.c_void_p), C.c_size_t(e.size),
                                                    C.byref(inf), ptrs,
                                                    self.quant[i].ctypes.data_as(C.c_void_p))
            if rc:
                msg = host.daliamdHostGetLastErrorMessage()
                raise capi.DaliAmdError(f"sample {i}: {msg.decode() if msg else 'decode failed'}")

        if self.n <= 1 or (num_threads is not None and num_threads <= 1):
            for i in range(self.n):
                one(i)
        else:
            list(_thread_pool(num_threads).map(one, range(self.n)))

    def build_descs(self, coef_dev, planes_dev, out_dev):
        """IDCT + colour descriptor tables for device buffers (torch tensors)."""
        lib = capi.kernels()
        ncomp_total = sum(self.infos[i].num_components for i in range(self.n))
        idct = (capi.JpegIdctDesc * max(ncomp_total, 1))()
        color = (capi.JpegColorDesc * max(self.n, 1))()
        cb, pb, ob = coef_dev.data_ptr(), planes_dev.data_ptr(), out_dev.data_ptr()
        k = 0
        for i in range(self.n):
            inf = self.infos[i]
            cd = color[i]
            for c in range(inf.num_components):
                d = idct[k]
                d.coef = cb + 2 * int(self.coef_off[i, c])
                d.plane = pb + int(self.plane_off[i, c])
                d.blocks_x = inf.blocks_x[c]
                d.nblocks = inf.blocks_x[c] * inf.blocks_y[c]
                d.pitch = inf.blocks_x[c] * 8
                C.memmove(d.quant, self.quant[i, c].ctypes.data, 128)
                cd.plane[c] = d.plane
                cd.pitch[c] = d.pitch
                cd.h_samp[c], cd.v_samp[c] = inf.h_samp[c], inf.v_samp[c]
                cd.down_w[c], cd.down_h[c] = inf.down_w[c], inf.down_h[c]
                k += 1
            cd.width, cd.height, cd.color = inf.width, inf.height, inf.color
            cd.out = ob + int(self.out_off[i])
            cd.out_pitch = int(self.out_pitch[i])
        n_idct_wg, n_color_wg = C.c_int(0), C.c_int(0)
        capi.check(lib.daliamdJpegIdctSetup(idct, ncomp_total, C.byref(n_idct_wg)))
        capi.check(lib.daliamdJpegColorSetup(color, self.n, C.byref(n_color_wg)))
        return (idct, ncomp_total, n_idct_wg.value), (color, self.n, n_color_wg.value)

    def output_views(self, out_dev):
        views = []
        for i in range(self.n):
            inf = self.infos[i]
            v = torch.as_strided(out_dev, (inf.height, inf.width, 3), (int(self.out_pitch[i]), 3, 1),
                                 int(self.out_off[i]))
            views.append(v)
        return views


def jpeg_gpu_stage(plan, coef_dev, planes_dev, out_dev, descs=None):
    """Enqueues dequant+IDCT and upsample+colour for a planned batch on the current stream."""
    lib = capi.kernels()
    if descs is None:
        descs = plan.build_descs(coef_dev, planes_dev, out_dev)
    (idct, n_idct, wg_idct), (color, n_color, wg_color) = descs
    dev = coef_dev.device
    idct_dev = _uploader.upload(idct, dev)
    color_dev = _uploader.upload(color, dev)
    s = current_stream_ptr(dev)
    capi.check(lib.daliamdJpegIdctRun(s, C.c_void_p(idct_dev.data_ptr()), n_idct, wg_idct))
    capi.check(lib.daliamdJpegColorRun(s, C.c_void_p(color_dev.data_ptr()), n_color, wg_color))
    return idct_dev, color_dev


def decode_jpeg_batch(encoded, device="cuda", num_threads=None, out_pitch_align=16):
    """Hybrid decode of a batch of JPEG byte strings -> list of u8 HWC RGB device tensors.

    Host: header parse + Huffman (thread pool) into pinned memory; device: everything else."""
    device = torch.device(device)
    plan = JpegBatchPlan(encoded, out_pitch_align)
    coef_host = torch.empty(max(plan.coef_elems, 1), dtype=torch.int16, pin_memory=True)
    plan.entropy_decode(coef_host, num_threads)
    coef_dev = coef_host.to(device, non_blocking=True)
    planes = torch.empty(max(plan.plane_bytes, 1), dtype=torch.uint8, device=device)
    out = torch.empty(max(plan.out_bytes, 1), dtype=torch.uint8, device=device)
    keep = jpeg_gpu_stage(plan, coef_dev, planes, out)
    views = plan.output_views(out)
    # keep scratch alive until the stream has consumed it
    for t in (coef_dev, planes, coef_host) + tuple(keep):
        if t.is_cuda:
            t.record_stream(torch.cuda.current_stream(device))
    plan._keepalive = (coef_host,)
    return views, plan


# =====================================================================================
# Resample (+ fused CropMirrorNormalize)
# =====================================================================================
def _fill4(dst, src):
    for i in range(4):
        dst[i] = float(src[i]) if i < len(src) else (float(src[-1]) if len(src) == 1 else 0.0)


def resample_batch(images, out_size, rois=None, interp_min=capi.INTERP_LINEAR, interp_mag=capi.INTERP_LINEAR,
                   antialias=True, out_dtype=capi.UINT8, out_layout=capi.LAYOUT_HWC, mean=None, inv_std=None,
                   mirror=None, out=None, return_descs=False):
    """Resamples a batch of u8 HWC device tensors (possibly row-strided views) to out_size=(H, W).

    rois[i] = (y0, x0, y1, x1) in source pixels or None.  With mean/inv_std the CropMirrorNormalize
    epilogue is fused: output dtype float16/float32, layout CHW or HWC, optional per-sample mirror.
    Returns a dense tensor [N, ...]."""
    lib = capi.kernels()
    n = len(images)
    oh, ow = int(out_size[0]), int(out_size[1])
    dev = images[0].device if n else torch.device("cuda")
    ch = images[0].shape[2] if n else 3
    normalize = mean is not None
    if out is None:
        shape = (n, ch, oh, ow) if out_layout == capi.LAYOUT_CHW else (n, oh, ow, ch)
        out = torch.empty(shape, dtype=_TORCH_DTYPE[out_dtype], device=dev)
    args = (capi.ResampleArgs * max(n, 1))()
    esz = out.element_size()
    per_sample = oh * ow * ch * esz
    for i, img in enumerate(images):
        if img.dtype != torch.uint8 or img.dim() != 3 or img.stride(2) != 1 or img.stride(1) != img.shape[2]:
            raise capi.DaliAmdError("resample_batch expects u8 HWC tensors with dense pixels")
        a = args[i]
        a.in_ = img.data_ptr()
        a.in_h, a.in_w, a.channels = img.shape
        a.in_pitch = img.stride(0)
        if rois is not None and rois[i] is not None:
            a.use_roi = 1
            a.roi_y0, a.roi_x0, a.roi_y1, a.roi_x1 = [float(v) for v in rois[i]]
        a.out_h, a.out_w = oh, ow
        a.min_filter, a.mag_filter, a.antialias = interp_min, interp_mag, 1 if antialias else 0
        a.out = out.data_ptr() + i * per_sample
        a.out_dtype, a.out_layout = out_dtype, out_layout
        a.normalize = 1 if normalize else 0
        a.mirror = int(mirror[i]) if mirror is not None else 0
        if normalize:
            _fill4(a.mean, mean)
            _fill4(a.inv_std, inv_std)
    descs = (capi.ResampleDesc * max(n, 1))()
    nwg, lds = C.c_int(0), C.c_int(0)
    capi.check(lib.daliamdResampleSetup(args, n, descs, C.byref(nwg), C.byref(lds)))
    descs_dev = _uploader.upload(descs, dev)
    capi.check(lib.daliamdResampleRun(current_stream_ptr(dev), C.c_void_p(descs_dev.data_ptr()), n, nwg.value,
                                      lds.value))
    descs_dev.record_stream(torch.cuda.current_stream(dev))
    if return_descs:
        return out, descs, nwg.value, lds.value
    return out


# =====================================================================================
# stand-alone CropMirrorNormalize
# =====================================================================================
def cmn_batch(images, anchors_yx, crop_hw, mirror=None, mean=None, inv_std=None, fill_values=(0.0,),
              out_dtype=capi.FLOAT, out_layout=capi.LAYOUT_CHW, pad_output=False):
    """Crop (+mirror, normalise, pad channels, out-of-bounds fill) a batch of u8 HWC device tensors.
    All crops share crop_hw = (h, w) (uniform output) -> dense [N, C, h, w] / [N, h, w, C]."""
    lib = capi.kernels()
    n = len(images)
    ch_, cw_ = int(crop_hw[0]), int(crop_hw[1])
    dev = images[0].device if n else torch.device("cuda")
    cin = images[0].shape[2] if n else 3
    cout = cin
    if pad_output:
        cout = 1
        while cout < cin:
            cout *= 2
    shape = (n, cout, ch_, cw_) if out_layout == capi.LAYOUT_CHW else (n, ch_, cw_, cout)
    out = torch.empty(shape, dtype=_TORCH_DTYPE[out_dtype], device=dev)
    per_sample = ch_ * cw_ * cout * out.element_size()
    descs = (capi.CmnDesc * max(n, 1))()
    fv = list(fill_values) if len(fill_values) else [0.0]
    for i, img in enumerate(images):
        d = descs[i]
        d.in_ = img.data_ptr()
        d.in_h, d.in_w, d.channels = img.shape
        d.in_pitch = img.stride(0)
        d.anchor_y, d.anchor_x = int(anchors_yx[i][0]), int(anchors_yx[i][1])
        d.crop_h, d.crop_w = ch_, cw_
        d.mirror = int(mirror[i]) if mirror is not None else 0
        d.normalize = 1 if mean is not None else 0
        if mean is not None:
            _fill4(d.mean, mean)
            _fill4(d.inv_std, inv_std)
        for c in range(4):
            d.fill[c] = float(fv[0]) if len(fv) == 1 else (float(fv[c]) if c < len(fv) else 0.0)
        d.out_channels = cout
        d.out_dtype, d.out_layout = out_dtype, out_layout
        d.out = out.data_ptr() + i * per_sample
    nwg = C.c_int(0)
    capi.check(lib.daliamdCmnSetup(descs, n, C.byref(nwg)))
    descs_dev = _uploader.upload(descs, dev)
    capi.check(lib.daliamdCmnRun(current_stream_ptr(dev), C.c_void_p(descs_dev.data_ptr()), n, nwg.value))
    descs_dev.record_stream(torch.cuda.current_stream(dev))
    return out


# =====================================================================================
# host-side random helpers
# =====================================================================================
def philox_state(key, ctr_hi=0, ctr_lo=0, phase=0):
    s = capi.PhiloxState()
    s.key = key & (2 ** 64 - 1)
    s.ctr[0], s.ctr[1], s.phase = ctr_lo, ctr_hi & (2 ** 64 - 1), phase
    return s


def random_crop_batch(master, shapes_hw, aspect=(3 / 4, 4 / 3), area=(0.08, 1.0), num_attempts=10):
    host = capi.host()
    shapes = np.ascontiguousarray(shapes_hw, dtype=np.int32).reshape(-1, 2)
    n = shapes.shape[0]
    anchors = np.zeros((n, 2), np.int32)
    crops = np.zeros((n, 2), np.int32)
    capi.check_host(host.daliamdRandomCropBatch(C.byref(master), n, shapes.ctypes.data_as(C.c_void_p),
                                                C.c_float(aspect[0]), C.c_float(aspect[1]),
                                                C.c_float(area[0]), C.c_float(area[1]), int(num_attempts),
                                                anchors.ctypes.data_as(C.c_void_p),
                                                crops.ctypes.data_as(C.c_void_p)))
    return anchors, crops


def coin_flip_batch(master, batch, probability=0.5):
    host = capi.host()
    p = np.ascontiguousarray(np.broadcast_to(np.asarray(probability, np.float32), (batch,)))
    out = np.zeros(batch, np.int32)
    capi.check_host(host.daliamdCoinFlipBatch(C.byref(master), batch, p.ctypes.data_as(C.c_void_p), 1,
                                              out.ctypes.data_as(C.c_void_p)))
    return out


def cmn_norm_args(mean, std, scale=1.0, shift=0.0):
    host = capi.host()
    mean = np.atleast_1d(np.asarray(mean, np.float32))
    std = np.atleast_1d(np.asarray(std, np.float32))
    n = max(mean.size, std.size)
    mo, io = np.zeros(n, np.float32), np.zeros(n, np.float32)
    k = host.daliamdCmnNormArgs(mean.ctypes.data_as(C.c_void_p), mean.size, std.ctypes.data_as(C.c_void_p),
                                std.size, C.c_float(scale), C.c_float(shift), mo.ctypes.data_as(C.c_void_p),
                                io.ctypes.data_as(C.c_void_p))
    if k < 0:
        capi.check_host(1)
    return mo[:k].copy(), io[:k].copy()
