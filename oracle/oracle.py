"""ctypes bindings for the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package (dali_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_crop_anchor.restype = C.c_int64
        _lib.orc_crop_anchor.argtypes = [C.c_double, C.c_int64, C.c_int64, C.c_int]
        _lib.orc_float2half.restype = C.c_uint16
        _lib.orc_float2half.argtypes = [C.c_float]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---------------------------------------------------------------- RNG / crop
def philox_block(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox_block(_p(ctr, C.c_uint32), _p(key, C.c_uint32), _p(out, C.c_uint32))
    return out


class Philox(C.Structure):
    _fields_ = [("key", C.c_uint64), ("ctr", C.c_uint64 * 2), ("phase", C.c_int),
                ("out", C.c_uint32 * 4)]

    def __init__(self, key=0, ctr_hi=0, ctr_lo=0, phase=0):
        super().__init__()
        lib().orc_philox_init(C.byref(self), C.c_uint64(key & (2**64 - 1)), C.c_uint64(ctr_hi),
                              C.c_uint64(ctr_lo), C.c_int(phase))

    def next(self):
        f = lib().orc_philox_next
        f.restype = C.c_uint32
        return f(C.byref(self))

    def skipahead(self, n):
        lib().orc_philox_skipahead(C.byref(self), C.c_uint64(n))

    def skipahead_sequence(self, n):
        lib().orc_philox_skipahead_sequence(C.byref(self), C.c_uint64(n))


def rrc_batch(seed, iteration, shapes_hw, aspect=(3 / 4, 4 / 3), area=(0.08, 1.0), num_attempts=10):
    shapes = np.ascontiguousarray(shapes_hw, dtype=np.int32).reshape(-1, 2)
    n = shapes.shape[0]
    anchors = np.zeros((n, 2), np.int32)
    crops = np.zeros((n, 2), np.int32)
    lib().orc_rrc_batch(C.c_int64(seed), C.c_int64(iteration), C.c_int(n), _p(shapes, C.c_int),
                        C.c_float(aspect[0]), C.c_float(aspect[1]), C.c_float(area[0]),
                        C.c_float(area[1]), C.c_int(num_attempts), _p(anchors, C.c_int),
                        _p(crops, C.c_int))
    return anchors, crops


def coin_flip_batch(seed, iteration, batch, probability=0.5):
    out = np.zeros(batch, np.int32)
    lib().orc_coin_flip_batch(C.c_int64(seed), C.c_int64(iteration), C.c_int(batch),
                              C.c_float(probability), _p(out, C.c_int32))
    return out


def crop_anchor(norm, crop, insz, rounding="round"):
    return int(lib().orc_crop_anchor(float(np.float32(norm)), int(crop), int(insz),
                                     1 if rounding == "round" else 0))


# ---------------------------------------------------------------- resample
FILTER_NN, FILTER_LINEAR, FILTER_TRIANGULAR, FILTER_CUBIC, FILTER_LANCZOS3, FILTER_GAUSSIAN = 0, 1, 2, 3, 4, 5


def filter_table(ftype, radius):
    """(coeffs, scale, anchor, support) of a tabulated resampling filter at the given radius."""
    co = np.zeros(193, np.float32)
    sc, an, sup = C.c_float(0), C.c_float(0), C.c_int(0)
    n = lib().orc_filter_table(int(ftype), C.c_float(radius), _p(co, C.c_float), C.byref(sc), C.byref(an), C.byref(sup))
    return co[:n].copy(), sc.value, an.value, sup.value


def resample_u8(img, out_hw, roi=None, min_filter=FILTER_LINEAR, mag_filter=FILTER_LINEAR,
                antialias=True, round_mode=0, return_info=False):
    """img: u8 HWC.  roi = (y0, x0, y1, x1) floats or None."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, Cn = img.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = np.zeros((oh, ow, Cn), np.uint8)
    info = np.zeros(8, np.int32)
    r = np.asarray(roi if roi is not None else (0, 0, 0, 0), dtype=np.float32)
    rc = lib().orc_resample_u8(_p(img, C.c_uint8), H, W, Cn, 1 if roi is not None else 0,
                               _p(r, C.c_float), oh, ow, min_filter, mag_filter,
                               1 if antialias else 0, round_mode, _p(out, C.c_uint8), None,
                               _p(info, C.c_int))
    if rc:
        raise RuntimeError(f"orc_resample_u8 failed: {rc}")
    return (out, info) if return_info else out


T_U8, T_I16, T_U16, T_F32 = 0, 1, 2, 3
_NP_OF_T = {0: np.uint8, 1: np.int16, 2: np.uint16, 3: np.float32}


def resample_typed(img, out_hw, out_type=None, roi=None, min_filter=FILTER_LINEAR, mag_filter=FILTER_LINEAR, antialias=True):
    """img: u8 / i16 / u16 / f32 HWC.  out_type: T_* (default: the input's type; T_F32 = the unrounded result)."""
    img = np.ascontiguousarray(img)
    in_t = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.uint16): 2, np.dtype(np.float32): 3}[img.dtype]
    out_type = in_t if out_type is None else out_type
    H, W, Cn = img.shape
    oh, ow = out_hw
    inf = np.ascontiguousarray(img.astype(np.float32))
    out = np.zeros((oh, ow, Cn), np.float32)
    r = np.zeros(4, np.float32) if roi is None else np.asarray(roi, np.float32)
    rc = lib().orc_resample_typed(_p(inf, C.c_float), H, W, Cn, 0 if roi is None else 1, _p(r, C.c_float), oh, ow,
                                  min_filter, mag_filter, 1 if antialias else 0, out_type, _p(out, C.c_float))
    assert rc == 0, rc
    return out.astype(_NP_OF_T[out_type])


def resample_f32(img, out_hw, roi=None, min_filter=FILTER_LINEAR, mag_filter=FILTER_LINEAR, antialias=True):
    """Same as resample_u8 with the float result of the second pass (fn.resize(dtype=FLOAT))."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, Cn = img.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = np.zeros((oh, ow, Cn), np.float32)
    r = np.asarray(roi if roi is not None else (0, 0, 0, 0), dtype=np.float32)
    rc = lib().orc_resample_u8_to_f32(_p(img, C.c_uint8), H, W, Cn, 1 if roi is not None else 0, _p(r, C.c_float), oh, ow,
                                      min_filter, mag_filter, 1 if antialias else 0, _p(out, C.c_float))
    if rc:
        raise RuntimeError(f"orc_resample_u8_to_f32 failed: {rc}")
    return out


def triangular_support(radius):
    return lib().orc_triangular_support(C.c_float(radius))


def init_triangular(out_size, srcx0, scale, radius):
    sup = triangular_support(radius)
    idx = np.zeros(out_size, np.int32)
    co = np.zeros(out_size * sup, np.float32)
    lib().orc_init_triangular(out_size, C.c_float(srcx0), C.c_float(scale), C.c_float(radius),
                              _p(idx, C.c_int32), _p(co, C.c_float))
    return idx, co.reshape(out_size, sup)


# ---------------------------------------------------------------- CMN
F32, F16, U8, I8 = 0, 1, 2, 3
_NP = {F32: np.float32, F16: np.float16, U8: np.uint8, I8: np.int8}


def float2half(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.zeros(a.shape, np.uint16)
    lib().orc_float2half_array(_p(a, C.c_float), _p(out, C.c_uint16), C.c_int64(a.size))
    return out.view(np.float16)


def cmn_norm_args(mean, std, scale=1.0, shift=0.0):
    mean = np.atleast_1d(np.asarray(mean, np.float32))
    std = np.atleast_1d(np.asarray(std, np.float32))
    n = max(mean.size, std.size)
    mo = np.zeros(n, np.float32)
    io = np.zeros(n, np.float32)
    k = lib().orc_cmn_norm_args(_p(mean, C.c_float), mean.size, _p(std, C.c_float), std.size,
                                C.c_float(scale), C.c_float(shift), _p(mo, C.c_float),
                                _p(io, C.c_float))
    return mo[:k].copy(), io[:k].copy()


def cmn_u8(img, anchor_yx, crop_hw, mirror=False, mean=None, inv_std=None, layout="CHW",
           pad_output=False, pad_oob=False, fill_values=(), dtype=F32):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, Cn = img.shape
    ch, cw = int(crop_hw[0]), int(crop_hw[1])
    cout = Cn
    if pad_output:
        cout = 1
        while cout < Cn:
            cout *= 2
    shape = (cout, ch, cw) if layout == "CHW" else (ch, cw, cout)
    out = np.zeros(shape, _NP[dtype])
    mean = np.zeros(0, np.float32) if mean is None else np.ascontiguousarray(mean, np.float32)
    inv = np.zeros(0, np.float32) if inv_std is None else np.ascontiguousarray(inv_std, np.float32)
    fv = np.ascontiguousarray(fill_values, np.float32)
    rc = lib().orc_cmn_u8(_p(img, C.c_uint8), H, W, Cn, int(anchor_yx[0]), int(anchor_yx[1]), ch, cw,
                          1 if mirror else 0, _p(mean, C.c_float), _p(inv, C.c_float), mean.size,
                          1 if layout == "CHW" else 0, 1 if pad_output else 0,
                          1 if pad_oob else 0, _p(fv, C.c_float), fv.size, dtype,
                          out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("crop window out of bounds")
    return out


# ---------------------------------------------------------------- JPEG
def jpeg_info(data):
    buf = np.frombuffer(data, dtype=np.uint8)
    info = np.zeros(15, np.int32)
    rc = lib().orc_jpeg_info(_p(buf, C.c_uint8), C.c_size_t(buf.size), _p(info, C.c_int))
    if rc:
        raise RuntimeError(f"orc_jpeg_info failed: {rc}")
    return dict(width=int(info[0]), height=int(info[1]), ncomp=int(info[2]),
                progressive=bool(info[3]), hmax=int(info[4]), vmax=int(info[5]),
                orientation=int(info[6]),
                sampling=[(int(info[7 + 2 * i]), int(info[8 + 2 * i])) for i in range(int(info[2]))])


def jpeg_decode_rgb(data, return_coefs=False):
    buf = np.frombuffer(data, dtype=np.uint8)
    inf = jpeg_info(data)
    H, W = inf["height"], inf["width"]
    rgb = np.zeros((H, W, 3), np.uint8)
    coef_ptrs = (C.POINTER(C.c_int16) * 4)()
    coefs = []
    qt = np.zeros((4, 64), np.uint16)
    if return_coefs:
        mcux = -(-W // (8 * inf["hmax"]))
        mcuy = -(-H // (8 * inf["vmax"]))
        for i, (h, v) in enumerate(inf["sampling"]):
            a = np.zeros((mcuy * v, mcux * h, 64), np.int16)
            coefs.append(a)
            coef_ptrs[i] = _p(a, C.c_int16)
    rc = lib().orc_jpeg_decode_rgb(_p(buf, C.c_uint8), C.c_size_t(buf.size), _p(rgb, C.c_uint8),
                                   coef_ptrs if return_coefs else None,
                                   _p(qt, C.c_uint16) if return_coefs else None)
    if rc:
        raise RuntimeError(f"orc_jpeg_decode_rgb failed: {rc}")
    return (rgb, coefs, qt, inf) if return_coefs else rgb


# ---------------------------------------------------------------- whole path (CPU baseline)
def pipeline_batch(jpegs, rrc_seed, flip_seed, iteration, out_hw=(224, 224), mean=None, inv_std=None, nthreads=0):
    """decode -> RandomResizedCrop -> CropMirrorNormalize(fp16 CHW) for a batch on an OpenMP team."""
    n = len(jpegs)
    bufs = [np.frombuffer(j, dtype=np.uint8) for j in jpegs]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * n)(*[b.size for b in bufs])
    out = np.zeros((n, 3, out_hw[0], out_hw[1]), np.float16)
    mean = np.ascontiguousarray(mean, np.float32)
    inv = np.ascontiguousarray(inv_std, np.float32)
    failed = lib().orc_pipeline_batch(ptrs, sizes, n, C.c_int64(rrc_seed), C.c_int64(flip_seed), C.c_int64(iteration),
                                      int(out_hw[0]), int(out_hw[1]), _p(mean, C.c_float), _p(inv, C.c_float),
                                      out.ctypes.data_as(C.c_void_p), int(nthreads))
    if failed:
        raise RuntimeError(f"{failed} samples failed in the oracle pipeline")
    return out


# ---------------------------------------------------------------- heavy augmentation (configs[2])
def warp_affine_u8(img, matrix, out_hw=None, interp=1, fill=None):
    """matrix: 2x3 dst->src.  fill=None -> clamp border; else constant border (scalar or per channel)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W, Cn = img.shape
    oh, ow = out_hw if out_hw is not None else (H, W)
    out = np.zeros((oh, ow, Cn), np.uint8)
    m = np.ascontiguousarray(matrix, np.float32).reshape(6)
    fv = None
    if fill is not None:
        fv = np.ascontiguousarray(np.broadcast_to(np.asarray(fill, np.float32), (Cn,)))
    lib().orc_warp_affine_u8(_p(img, C.c_uint8), H, W, Cn, _p(m, C.c_float), int(oh), int(ow), int(interp),
                             _p(fv, C.c_float) if fv is not None else None, _p(out, C.c_uint8))
    return out


def rotate_params(angle, in_hw, size=None, keep_size=False):
    """fn.rotate: (destination->source matrix 2x3, (out_h, out_w))."""
    hw = np.array(size if size is not None else (0, 0), np.int32)
    m = np.zeros(6, np.float32)
    lib().orc_rotate_params(C.c_float(angle), int(in_hw[0]), int(in_hw[1]), 1 if keep_size else 0, _p(hw, C.c_int), _p(m, C.c_float))
    return m.reshape(2, 3), (int(hw[0]), int(hw[1]))


def rotate_u8(img, angle, size=None, keep_size=False, interp=1, fill=None):
    m, hw = rotate_params(angle, img.shape[:2], size, keep_size)
    return warp_affine_u8(img, m, out_hw=hw, interp=interp, fill=fill)


def affine_inverse(matrix):
    m = np.ascontiguousarray(matrix, np.float32).reshape(6)
    out = np.zeros(6, np.float32)
    lib().orc_affine_inverse_2x3(_p(m, C.c_float), _p(out, C.c_float))
    return out.reshape(2, 3)


def gaussian_window(sigma=0.0, window_size=0):
    if window_size == 0:
        window_size = lib().orc_gaussian_diameter(C.c_float(sigma))
    elif sigma == 0:
        f = lib().orc_gaussian_sigma_from_diameter
        f.restype = C.c_float
        sigma = f(int(window_size))
    w = np.zeros(window_size, np.float32)
    lib().orc_gaussian_window(C.c_float(sigma), int(window_size), _p(w, C.c_float))
    return w


def gaussian_blur_u8(img, window_x, window_y=None):
    img = np.ascontiguousarray(img, np.uint8)
    H, W, Cn = img.shape
    wx = np.ascontiguousarray(window_x, np.float32)
    wy = wx if window_y is None else np.ascontiguousarray(window_y, np.float32)
    out = np.zeros_like(img)
    lib().orc_gaussian_blur_u8(_p(img, C.c_uint8), H, W, Cn, _p(wx, C.c_float), wx.size, _p(wy, C.c_float), wy.size,
                               _p(out, C.c_uint8))
    return out


def color_twist_matrix(hue=0.0, saturation=1.0, value=1.0, brightness=1.0, contrast=1.0):
    m = np.zeros(9, np.float32)
    off = C.c_float(0)
    lib().orc_color_twist_matrix(C.c_float(hue), C.c_float(saturation), C.c_float(value), C.c_float(brightness),
                                 C.c_float(contrast), _p(m, C.c_float), C.byref(off))
    return m.reshape(3, 3), np.float32(off.value)


def linear_transform_u8(img, matrix, offset):
    img = np.ascontiguousarray(img, np.uint8)
    m = np.ascontiguousarray(matrix, np.float32).reshape(9)
    o = np.ascontiguousarray(np.broadcast_to(np.asarray(offset, np.float32), (3,)))
    out = np.zeros_like(img)
    lib().orc_linear_transform_u8(_p(img, C.c_uint8), C.c_int64(img.shape[0] * img.shape[1]), _p(m, C.c_float),
                                  _p(o, C.c_float), _p(out, C.c_uint8))
    return out


def erase_u8(img, anchors_yx, shapes_yx, fill=(0.0,), normalized_anchor=False, normalized_shape=False,
             centered_anchor=False):
    img = np.ascontiguousarray(img, np.uint8)
    H, W, Cn = img.shape
    a = np.ascontiguousarray(anchors_yx, np.float32).reshape(-1, 2)
    s = np.ascontiguousarray(shapes_yx, np.float32).reshape(-1, 2)
    f = np.ascontiguousarray(fill, np.float32).reshape(-1)
    out = np.zeros_like(img)
    flags = (1 if normalized_anchor else 0) | (2 if normalized_shape else 0) | (4 if centered_anchor else 0)
    lib().orc_erase_u8(_p(img, C.c_uint8), H, W, Cn, _p(a, C.c_float), _p(s, C.c_float), a.shape[0], flags,
                       _p(f, C.c_float), f.size, _p(out, C.c_uint8))
    return out


RESIZE_MODES = {"default": 0, "stretch": 1, "not_larger": 2, "not_smaller": 3}


def resize_params(in_hw, size=(0, 0), mode="default", max_size=None, subpixel_scale=True, roi=None, roi_relative=False):
    """fn.resize size/region arithmetic for one 2-D sample.  size = requested (H, W), 0 = unspecified;
    roi = (start_y, start_x, end_y, end_x).  Returns (out_hw, (y0, x0, y1, x1)) with the source region as floats."""
    inp = np.asarray(in_hw, np.int32)
    req = np.asarray(size, np.float32)
    ms = None if max_size is None else np.asarray(np.broadcast_to(max_size, 2), np.float32).copy()
    r = np.asarray(roi if roi is not None else (0, 0, 0, 0), np.float32)
    out_hw = np.zeros(2, np.int32)
    lo, hi = np.zeros(2, np.float32), np.zeros(2, np.float32)
    rc = lib().orc_resize_params(_p(inp, C.c_int32), _p(req, C.c_float), RESIZE_MODES[mode],
                                 _p(ms, C.c_float) if ms is not None else None, 1 if subpixel_scale else 0,
                                 1 if roi is not None else 0, 1 if roi_relative else 0, _p(r, C.c_float),
                                 _p(out_hw, C.c_int32), _p(lo, C.c_float), _p(hi, C.c_float))
    if rc:
        raise RuntimeError("Cannot produce non-empty output from empty input")
    return (int(out_hw[0]), int(out_hw[1])), (float(lo[0]), float(lo[1]), float(hi[0]), float(hi[1]))


def resize_crop_mirror_params(in_hw, crop=(0, 0), crop_pos=(0.5, 0.5), mirror=0, rounding="round", **resize_kw):
    """fn.resize_crop_mirror: the ResizeAttr arithmetic, then the CropAttr window of the RESIZED image projected back
    into source coordinates, then the flips as swapped region ends (resize_crop_mirror.cc:85-118; crop window
    crop_attr.cc:168-240).  crop = (H, W), non-positive = whole axis; crop_pos = (y, x) normalised; mirror bit 0 =
    horizontal, bit 1 = vertical.  Returns (out_hw, (y0, x0, y1, x1))."""
    out_hw, roi = resize_params(in_hw, **resize_kw)
    lo = [np.float32(roi[0]), np.float32(roi[1])]
    hi = [np.float32(roi[2]), np.float32(roi[3])]
    out = list(out_hw)
    for d in range(2):
        c, norm = int(crop[d]), np.float32(crop_pos[d])
        if c <= 0:
            c, norm = out_hw[d], np.float32(0.5)
        anchor = crop_anchor(float(norm), c, out_hw[d], rounding)
        ratio = (float(hi[d]) - float(lo[d])) / out_hw[d]          # double arithmetic, float storage
        offset = float(lo[d])
        lo[d] = np.float32(anchor * ratio + offset)
        hi[d] = np.float32((anchor + c) * ratio + offset)
        if mirror & (1 << (1 - d)):
            lo[d], hi[d] = hi[d], lo[d]
        out[d] = c
    return (out[0], out[1]), (float(lo[0]), float(lo[1]), float(hi[0]), float(hi[1]))
