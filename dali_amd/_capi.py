"""ctypes bindings of the two C-ABI libraries (include/dali_amd_kernels.h, include/dali_amd_host.h).

The product path fails loudly when a library is missing: there is NO CPU fallback for the
device kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")
KERNELS_LIB = os.path.join(_LIBDIR, "libdali_amd_kernels.so")
HOST_LIB = os.path.join(_LIBDIR, "libdali_amd_host.so")


class DaliAmdError(RuntimeError):
    pass


# ------------------------------------------------------------------ enums
UINT8, FLOAT16, FLOAT, INT8, INT16, UINT16, INT32, UINT32 = 0, 1, 2, 3, 4, 5, 6, 7
LAYOUT_HWC, LAYOUT_CHW = 0, 1
INTERP_NN, INTERP_LINEAR, INTERP_TRIANGULAR, INTERP_CUBIC, INTERP_LANCZOS3, INTERP_GAUSSIAN = 0, 1, 2, 3, 4, 5
JPEG_GRAY, JPEG_YCC, JPEG_RGB = 0, 1, 2


# ------------------------------------------------------------------ structs (kernels)
class JpegIdctDesc(C.Structure):
    _fields_ = [("coef", C.c_void_p), ("plane", C.c_void_p), ("blocks_x", C.c_int32),
                ("nblocks", C.c_int32), ("pitch", C.c_int32), ("wg_start", C.c_int32),
                ("quant", C.c_uint16 * 64), ("rect_x0", C.c_int32), ("rect_y0", C.c_int32),
                ("rect_w", C.c_int32), ("reserved", C.c_int32)]


class GatherDesc(C.Structure):
    """daliamdGatherDesc (include/dali_amd_kernels.h): one record of the batched device-side copy."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("bytes", C.c_uint64), ("reserved", C.c_uint64)]


class JpegHuffDesc(C.Structure):
    _fields_ = [("ecs", C.c_void_p), ("scratch", C.c_void_p), ("status", C.c_void_p), ("coef", C.c_void_p * 3),
                ("ecs_len", C.c_int32), ("blocks_per_mcu", C.c_int32), ("mcus_x", C.c_int32),
                ("total_blocks", C.c_int32), ("blocks_x", C.c_int32 * 3), ("h_samp", C.c_int32 * 3),
                ("v_samp", C.c_int32 * 3), ("tile_start", C.c_int32), ("num_tiles", C.c_int32),
                ("seg_start", C.c_int32), ("num_segments", C.c_int32), ("blk_wg_start", C.c_int32),
                ("table_owner", C.c_int32), ("comp_of_block", C.c_uint8 * 12), ("h_of_block", C.c_uint8 * 12),
                ("v_of_block", C.c_uint8 * 12), ("dc_sel", C.c_uint8 * 4), ("ac_sel", C.c_uint8 * 4),
                ("bits", (C.c_uint8 * 16) * 4), ("vals", (C.c_uint8 * 256) * 4), ("rect", (C.c_int32 * 4) * 3),
                ("plane", C.c_void_p * 3), ("plane_pitch", C.c_int32 * 3), ("restart_interval", C.c_int32),
                ("quant", (C.c_uint16 * 64) * 3), ("rgb", C.c_void_p), ("rgb_pitch", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("reserved", C.c_int32), ("index", C.c_void_p), ("index_out", C.c_void_p), ("tables", C.c_void_p)]


class JpegColorDesc(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("pitch", C.c_int32 * 3), ("h_samp", C.c_int32 * 3),
                ("v_samp", C.c_int32 * 3), ("down_w", C.c_int32 * 3), ("down_h", C.c_int32 * 3),
                ("width", C.c_int32), ("height", C.c_int32), ("color", C.c_int32),
                ("out", C.c_void_p), ("out_pitch", C.c_int32), ("wg_start", C.c_int32),
                ("orientation", C.c_int32), ("out_format", C.c_int32), ("roi_x0", C.c_int32),
                ("roi_y0", C.c_int32), ("roi_w", C.c_int32), ("roi_h", C.c_int32), ("out_x0", C.c_int32),
                ("out_y0", C.c_int32)]


class NormalizeDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("outer", C.c_int64), ("reduced", C.c_int64),
                ("inner", C.c_int64), ("sum_mean", C.c_void_p), ("sum_var", C.c_void_p), ("mean", C.c_void_p),
                ("inv_std", C.c_void_p), ("stat_count", C.c_double), ("scalar_mean", C.c_float),
                ("scalar_inv_std", C.c_float), ("use_scalar_mean", C.c_int32), ("use_scalar_inv_std", C.c_int32),
                ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("owns_stats", C.c_int32),
                ("stat_wg_start", C.c_int32), ("stat_chunks", C.c_int32), ("apply_wg_start", C.c_int32)]


class JpegRoiPlan(C.Structure):
    _fields_ = [("roi_x0", C.c_int32), ("roi_y0", C.c_int32), ("roi_w", C.c_int32), ("roi_h", C.c_int32),
                ("out_x0", C.c_int32), ("out_y0", C.c_int32), ("rect", (C.c_int32 * 4) * 3)]


class ResampleArgs(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("channels", C.c_int32), ("in_pitch", C.c_int32), ("use_roi", C.c_int32),
                ("roi_y0", C.c_float), ("roi_x0", C.c_float), ("roi_y1", C.c_float),
                ("roi_x1", C.c_float), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("min_filter", C.c_int32), ("mag_filter", C.c_int32), ("antialias", C.c_int32),
                ("out", C.c_void_p), ("out_dtype", C.c_int32), ("out_layout", C.c_int32),
                ("normalize", C.c_int32), ("mirror", C.c_int32), ("mean", C.c_float * 4),
                ("inv_std", C.c_float * 4), ("in_dtype", C.c_int32), ("unrounded", C.c_int32),
                ("full_h", C.c_int32), ("full_w", C.c_int32), ("org_y", C.c_int32), ("org_x", C.c_int32)]


class ResampleDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("channels", C.c_int32), ("in_pitch", C.c_int32), ("out_h", C.c_int32),
                ("out_w", C.c_int32), ("first_axis", C.c_int32), ("origin", C.c_float * 2),
                ("scale", C.c_float * 2), ("fscale", C.c_float * 2), ("fanchor", C.c_float * 2),
                ("support", C.c_int32 * 2), ("lo", C.c_int32 * 2), ("ext", C.c_int32 * 2),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("tiles_x", C.c_int32),
                ("tiles_y", C.c_int32), ("wg_start", C.c_int32), ("out_dtype", C.c_int32),
                ("out_layout", C.c_int32), ("normalize", C.c_int32), ("mirror", C.c_int32),
                ("mean", C.c_float * 4), ("inv_std", C.c_float * 4), ("round_lo", C.c_int32 * 4), ("round_hi", C.c_int32 * 4),
                ("lds_bytes", C.c_int32), ("staged", C.c_int32), ("table_off", C.c_int64),
                ("tab_start", C.c_int32), ("use_lut", C.c_int32), ("filter_kind", C.c_int32 * 2),
                ("in_dtype", C.c_int32), ("unrounded", C.c_int32), ("generic", C.c_int32), ("round_lanes", C.c_int32),
                ("tmp_w", C.c_int32), ("tmp_h", C.c_int32), ("tmp_off", C.c_int64), ("gen_start", C.c_int64 * 2)]


class ResamplePlan(C.Structure):
    _fields_ = [("num_tiles", C.c_int32), ("lds_bytes", C.c_int32), ("table_entries", C.c_int32), ("reserved", C.c_int32),
                ("workspace_bytes", C.c_size_t), ("generic_items", C.c_int64 * 2)]


class CmnDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("channels", C.c_int32), ("in_pitch", C.c_int32), ("anchor_y", C.c_int32),
                ("anchor_x", C.c_int32), ("crop_h", C.c_int32), ("crop_w", C.c_int32),
                ("mirror", C.c_int32), ("normalize", C.c_int32), ("mean", C.c_float * 4),
                ("inv_std", C.c_float * 4), ("fill", C.c_float * 4), ("out_channels", C.c_int32),
                ("out_dtype", C.c_int32), ("out_layout", C.c_int32), ("out", C.c_void_p),
                ("wg_start", C.c_int32)]


class WarpAffineDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("channels", C.c_int32), ("in_pitch", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("out_pitch", C.c_int32), ("matrix", C.c_float * 6), ("interp", C.c_int32),
                ("border_clamp", C.c_int32), ("fill", C.c_float * 4), ("wg_start", C.c_int32),
                ("reserved", C.c_int32)]


MAX_BLUR_WINDOW = 63


class GaussianBlurDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32),
                ("channels", C.c_int32), ("in_pitch", C.c_int32), ("out_pitch", C.c_int32),
                ("size_x", C.c_int32), ("size_y", C.c_int32), ("window_x", C.c_float * 64),
                ("window_y", C.c_float * 64), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("tiles_x", C.c_int32), ("wg_start", C.c_int32), ("lds_bytes", C.c_int32)]


MAX_ERASE_REGIONS = 8


class PointwiseDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32),
                ("channels", C.c_int32), ("in_pitch", C.c_int32), ("out_pitch", C.c_int32),
                ("transform", C.c_int32), ("matrix", C.c_float * 9), ("offset", C.c_float * 3),
                ("num_regions", C.c_int32), ("region", (C.c_int32 * 4) * 8), ("fill", C.c_float * 4),
                ("wg_start", C.c_int32)]


# ------------------------------------------------------------------ structs (host)
class JpegInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("num_components", C.c_int32),
                ("progressive", C.c_int32), ("h_samp", C.c_int32 * 4), ("v_samp", C.c_int32 * 4),
                ("hmax", C.c_int32), ("vmax", C.c_int32), ("blocks_x", C.c_int32 * 4),
                ("blocks_y", C.c_int32 * 4), ("down_w", C.c_int32 * 4), ("down_h", C.c_int32 * 4),
                ("orientation", C.c_int32), ("color", C.c_int32), ("restart_interval", C.c_int32),
                ("coef_elems", C.c_int64 * 4)]


class JpegScan(C.Structure):
    _fields_ = [("eligible", C.c_int32), ("blocks_per_mcu", C.c_int32), ("mcus_x", C.c_int32),
                ("mcus_y", C.c_int32), ("ecs_offset", C.c_int64), ("ecs_length", C.c_int64),
                ("comp_of_block", C.c_uint8 * 10), ("h_of_block", C.c_uint8 * 10),
                ("v_of_block", C.c_uint8 * 10), ("dc_sel", C.c_uint8 * 4), ("ac_sel", C.c_uint8 * 4),
                ("dc_bits", (C.c_uint8 * 16) * 4), ("dc_vals", (C.c_uint8 * 256) * 4),
                ("ac_bits", (C.c_uint8 * 16) * 4), ("ac_vals", (C.c_uint8 * 256) * 4),
                ("quant", (C.c_uint16 * 64) * 4), ("restart_interval", C.c_int32),
                ("length_is_upper_bound", C.c_int32)]


class PhiloxState(C.Structure):
    _fields_ = [("key", C.c_uint64), ("ctr", C.c_uint64 * 2), ("phase", C.c_int32)]


# ------------------------------------------------------------------ loading
_kernels = None
_host = None

_KERNEL_SYMBOLS = [
    "daliamdGetLastErrorMessage", "daliamdClearLastError", "daliamdVersion", "daliamdDeviceCount",
    "daliamdSetDevice", "daliamdDeviceInfo", "daliamdDevicePciBusId", "daliamdRangePush", "daliamdRangePop", "daliamdKernelTimingEnable", "daliamdKernelTimingReport",
    "daliamdStreamCreate", "daliamdStreamCreateWithPriority", "daliamdStreamDestroy",
    "daliamdStreamSynchronize", "daliamdStreamWaitEvent", "daliamdEventCreate",
    "daliamdEventDestroy", "daliamdEventRecord", "daliamdEventSynchronize", "daliamdEventQuery", "daliamdEventElapsedMs",
    "daliamdMalloc", "daliamdFree", "daliamdHostAlloc", "daliamdHostFree", "daliamdMemcpyH2DAsync",
    "daliamdMemcpyD2HAsync", "daliamdMemcpyD2DAsync", "daliamdMemsetAsync", "daliamdMemcpy2DD2DAsync",
    "daliamdHostRegister", "daliamdHostUnregister", "daliamdGatherCopy", "daliamdJpegHuffmanIndexBuildHost",
    "daliamdJpegIdctSetup", "daliamdJpegIdctRun", "daliamdJpegHuffmanScratchBytes", "daliamdJpegHuffmanScratchBytesRestart", "daliamdJpegHuffmanSetup",
    "daliamdJpegHuffmanRun", "daliamdJpegHuffmanRunProfiled", "daliamdJpegHuffmanSetupColor", "daliamdJpegHuffmanRunColor",
    "daliamdJpegHuffmanRunProfiledColor", "daliamdJpegHuffmanColorFusable", "daliamdJpegHuffmanIndexBytes", "daliamdJpegHuffmanRunFront", "daliamdJpegHuffmanRunBack", "daliamdJpegHuffmanTablesBytes", "daliamdJpegHuffmanTablesBuild", "daliamdJpegColorSetup", "daliamdJpegPlanRoi", "daliamdJpegColorRun",
    "daliamdResampleSetup", "daliamdResampleRun", "daliamdResampleRunTables", "daliamdResampleRunPasses", "daliamdCmnSetup", "daliamdCmnRun",
    "daliamdWarpAffineSetup", "daliamdWarpAffineRun", "daliamdGaussianWindow", "daliamdGaussianBlurSetup",
    "daliamdGaussianBlurRun", "daliamdGaussianBlurPointwiseRun", "daliamdColorTwistMatrix", "daliamdPointwiseSetup", "daliamdPointwiseRun",
    "daliamdHannWindow", "daliamdSpectrogramTwiddles", "daliamdSpectrogramSetup", "daliamdSpectrogramRun", "daliamdMelFilterBankWeights",
    "daliamdMelFilterBankBands", "daliamdMelFilterBankSetup", "daliamdMelFilterBankRun", "daliamdMelFilterBankMfmaLayout", "daliamdSpectrogramMelRun", "daliamdToDecibelsSetup", "daliamdToDecibelsRun", "daliamdDctTable", "daliamdLifterCoeffs", "daliamdDctRun",
    "daliamdAudioResampleLobes", "daliamdAudioResampleWindow", "daliamdAudioResampleSetup", "daliamdAudioResampleRun",
    "daliamdConvertNormSetup", "daliamdConvertNormRun",
    "daliamdNormalizeSetup", "daliamdNormalizeRun",
]

_HOST_SYMBOLS = [
    "daliamdHostGetLastErrorMessage", "daliamdJpegParse", "daliamdJpegDecodeCoefficients", "daliamdJpegEncodeBaselineScan", "daliamdJpegIndexedIs", "daliamdJpegIndexedParse", "daliamdJpegIndexedValidate", "daliamdJpegIndexedBuild", "daliamdJpegDecodeRgbHost", "daliamdJpegOutputChannels", "daliamdJpegDecodeHost", "daliamdConvertRgbRows",
    "daliamdJpegAnalyzeScan", "daliamdJpegAnalyzeHeader",
    "daliamdRandomCropBatch", "daliamdCoinFlipBatch", "daliamdPhiloxAdvanceSequence",
    "daliamdPhiloxStateToString", "daliamdPhiloxStateFromString", "daliamdPhiloxGenerate",
    "daliamdCmnNormArgs", "daliamdCropAnchor", "daliamdResampleRunHost", "daliamdCmnRunHost", "daliamdAudioResampleHost",
    "daliamdConvertNormHost", "daliamdSpectrogramHost", "daliamdMelFilterBankHost", "daliamdToDecibelsHost", "daliamdDctHost",
    "daliamdWarpAffineHost", "daliamdGaussianBlurHost", "daliamdPointwiseHost", "daliamdNormalizeHost",
    "daliamdImageCachePolicyCreate", "daliamdImageCachePolicyDestroy", "daliamdImageCachePolicyOnDecode",
    "daliamdImageCachePolicyFind", "daliamdImageProbe", "daliamdImageDecodeRgb",
    "daliamdFlacProbe", "daliamdFlacDecode",
]


def declared_kernel_symbols():
    return list(_KERNEL_SYMBOLS)


def declared_host_symbols():
    return list(_HOST_SYMBOLS)


def kernels():
    """libdali_amd_kernels.so (HIP).  Raises if it has not been built."""
    global _kernels
    if _kernels is None:
        if not os.path.exists(KERNELS_LIB):
            raise DaliAmdError(
                f"{KERNELS_LIB} is missing: the gfx950 kernel library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback for device operators.")
        # torch bundles its own libamdhip64 (requested by file name, soname libamdhip64.so.7).  Load it
        # FIRST so that this library's NEEDED libamdhip64.so.7 resolves to the same runtime instance;
        # the other order would put two HIP runtimes in the process.
        import torch  # noqa: F401
        lib = C.CDLL(KERNELS_LIB)
        lib.daliamdGetLastErrorMessage.restype = C.c_char_p
        _kernels = lib
    return _kernels


def host():
    """libdali_amd_host.so (C++ host side)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise DaliAmdError(f"{HOST_LIB} is missing: run __graft_entry__.build()")
        lib = C.CDLL(HOST_LIB)
        lib.daliamdHostGetLastErrorMessage.restype = C.c_char_p
        lib.daliamdCropAnchor.restype = C.c_int64
        lib.daliamdImageCachePolicyCreate.restype = C.c_void_p
        lib.daliamdImageCachePolicyCreate.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
        lib.daliamdImageCachePolicyDestroy.argtypes = [C.c_void_p]
        lib.daliamdImageCachePolicyOnDecode.restype = C.c_int64
        lib.daliamdImageCachePolicyOnDecode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64]
        lib.daliamdImageCachePolicyFind.restype = C.c_int64
        lib.daliamdImageCachePolicyFind.argtypes = [C.c_void_p, C.c_char_p]
        lib.daliamdCropAnchor.argtypes = [C.c_float, C.c_int64, C.c_int64, C.c_int]
        _host = lib
    return _host


def check(rc):
    """Raises DaliAmdError carrying the library's thread-local message when rc != 0."""
    if rc != 0:
        msg = kernels().daliamdGetLastErrorMessage()
        raise DaliAmdError(f"[dali_amd kernels error {rc}] {msg.decode() if msg else ''}")


def check_host(rc):
    if rc != 0:
        msg = host().daliamdHostGetLastErrorMessage()
        raise DaliAmdError(f"[dali_amd host error] {msg.decode() if msg else ''}")
