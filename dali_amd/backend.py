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
                capi.INT8: torch.int8, capi.INT16: torch.int16, capi.UINT16: torch.uint16}
_KERNEL_DTYPE = {torch.uint8: capi.UINT8, torch.int16: capi.INT16, torch.uint16: capi.UINT16, torch.float32: capi.FLOAT}


def _align(v, a):
    return (v + a - 1) // a * a


_stream_override = None


class use_stream:
    """`with use_stream(torch_stream):` - makes torch_stream current AND tells this module its handle, so that the many
    launches of a step do not each ask torch for the current stream (a few microseconds of Python per query)."""

    def __init__(self, stream):
        self._stream, self._ctx = stream, torch.cuda.stream(stream)

    def __enter__(self):
        global _stream_override
        self._prev = _stream_override
        self._ctx.__enter__()
        _stream_override = int(self._stream.cuda_stream)
        return self

    def __exit__(self, *exc):
        global _stream_override
        _stream_override = self._prev
        return self._ctx.__exit__(*exc)


def current_stream_ptr(device=None):
    if _stream_override is not None:
        return C.c_void_p(_stream_override)
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _DevSlice:
    """A descriptor table in the uploader's device ring."""
    __slots__ = ("_ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self._ptr, self.nbytes = ptr, nbytes

    def data_ptr(self):
        return self._ptr

    def record_stream(self, _stream):
        pass   # the ring outlives the kernels that read it (see _DescUploader)


class _DescUploader:
    """Descriptor tables host -> device on the current stream: per stream one pinned and one device ring of
    kSegments x kSegBytes.  A table is copied into the pinned ring, one hipMemcpyAsync moves it to the same offset of
    the device ring, and the launches that follow on that stream read it there.  An event is recorded on the stream
    whenever the write position leaves a segment; before a segment is written again that event is waited for (it has
    long completed: a step uploads well under a megabyte), so nothing is reused while a copy or a kernel may still
    read it and the launches stay asynchronous.  No torch call, no allocation in the steady state."""
    kSegBytes = 2 << 20
    kSegments = 8

    def __init__(self):
        self._rings = {}

    def _ring(self, stream, device):
        r = self._rings.get((stream, str(device)))
        if r is None:
            lib = capi.kernels()
            r = {"pinned": torch.empty(self.kSegBytes * self.kSegments, dtype=torch.uint8, pin_memory=True),
                 "dev": torch.empty(self.kSegBytes * self.kSegments, dtype=torch.uint8, device=device),
                 "off": 0, "events": [None] * self.kSegments}
            for k in range(self.kSegments):
                ev = C.c_void_p()
                capi.check(lib.daliamdEventCreate(C.byref(ev), 0))
                r["events"][k] = [ev, False]
            self._rings[(stream, str(device))] = r
        return r

    def upload(self, table, device):
        """table: numpy structured array or ctypes array."""
        if isinstance(table, np.ndarray):
            nbytes, src = table.nbytes, table.ctypes.data
        else:
            nbytes, src = C.sizeof(table), C.addressof(table)
        lib = capi.kernels()
        sp = current_stream_ptr(device)
        if nbytes > self.kSegBytes:    # a table larger than a segment: its own buffers, freed by torch's allocator rules
            buf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            C.memmove(buf.data_ptr(), src, nbytes)
            dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
            dev.copy_(buf, non_blocking=True)
            dev.record_stream(torch.cuda.current_stream(device))
            self._big = (buf, dev)
            return dev
        r = self._ring(sp.value or 0, device)
        size = _align(max(nbytes, 1), 256)
        off = r["off"]
        seg = off // self.kSegBytes
        if (off + size - 1) // self.kSegBytes != seg or off + size > self.kSegBytes * self.kSegments:
            # leave this segment: everything enqueued so far that reads it is in front of this event
            ev = r["events"][seg]
            capi.check(lib.daliamdEventRecord(ev[0], sp))
            ev[1] = True
            seg = (seg + 1) % self.kSegments
            off = seg * self.kSegBytes
            nxt = r["events"][seg]
            if nxt[1]:
                capi.check(lib.daliamdEventSynchronize(nxt[0]))
                nxt[1] = False
        C.memmove(r["pinned"].data_ptr() + off, src, nbytes)
        dst = r["dev"].data_ptr() + off
        capi.check(lib.daliamdMemcpyH2DAsync(C.c_void_p(dst), C.c_void_p(r["pinned"].data_ptr() + off), C.c_size_t(nbytes), sp))
        r["off"] = off + size
        return _DevSlice(dst, nbytes)


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
_DTYPES = {}


def _dtype(ctypes_struct):
    """numpy dtype mirroring a ctypes structure; the conversion walks every field, so it is done once per type."""
    dt = _DTYPES.get(ctypes_struct)
    if dt is None:
        dt = _DTYPES[ctypes_struct] = np.dtype(ctypes_struct)
    return dt


class JpegBatchPlan:
    """Geometry + buffer layout of one batch of JPEG streams (host side, no device work)."""

    def __init__(self, encoded, out_pitch_align=16, rois=None, exact_scan=True):
        """rois: optional per-sample windows (y0, x0, h, w) in image pixels (None entries = whole image): only the
        window is decoded (region-of-interest decode; EXIF orientation is not applied by this driver).
        exact_scan: see analyze_scans (False = what the decoders.image operator does)."""
        host = capi.host()
        self.exact_scan = exact_scan
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
        # per-sample geometry as arrays (vectorised descriptor construction)
        inf = np.frombuffer(self.infos, dtype=_dtype(capi.JpegInfo))[:self.n]
        self.inf = inf
        ncomp = inf["num_components"].astype(np.int64)
        elems = inf["coef_elems"][:, :3].astype(np.int64) * (np.arange(3)[None, :] < ncomp[:, None])
        flat = elems.reshape(-1)
        starts = np.concatenate([[0], np.cumsum(flat)[:-1]]).reshape(-1, 3) if self.n else np.zeros((0, 3), np.int64)
        self.coef_off = starts                     # int16 elements
        self.plane_off = starts                    # bytes: one byte per coefficient
        # region-of-interest plans (daliamdJpegPlanRoi): window in source pixels + block rectangles per component
        self.out_h, self.out_w = inf["height"].astype(np.int64).copy(), inf["width"].astype(np.int64).copy()
        self.roi_plans = None
        if rois is not None and self.n:
            lib = capi.kernels()
            self.roi_plans = np.zeros(self.n, _dtype(capi.JpegRoiPlan))
            self.has_roi = np.zeros(self.n, bool)
            for i, r in enumerate(rois):
                if r is None:
                    continue
                hs = (C.c_int32 * 3)(*[int(v) for v in inf["h_samp"][i, :3]])
                vs = (C.c_int32 * 3)(*[int(v) for v in inf["v_samp"][i, :3]])
                plan = capi.JpegRoiPlan()
                capi.check(lib.daliamdJpegPlanRoi(int(inf["width"][i]), int(inf["height"][i]), int(ncomp[i]), hs, vs, 1,
                                                  int(r[0]), int(r[1]), int(r[2]), int(r[3]), C.byref(plan)))
                self.roi_plans[i] = np.frombuffer(plan, dtype=_dtype(capi.JpegRoiPlan))[0]
                self.has_roi[i] = True
                self.out_h[i], self.out_w[i] = int(r[2]), int(r[3])
        self.out_pitch = (3 * self.out_w + out_pitch_align - 1) // out_pitch_align * out_pitch_align
        osz = (self.out_pitch * self.out_h + 255) // 256 * 256
        self.out_off = np.concatenate([[0], np.cumsum(osz)[:-1]]).astype(np.int64) if self.n else np.zeros(0, np.int64)
        self.coef_elems = int(flat.sum())
        self.plane_bytes = self.coef_elems
        self.out_bytes = int(osz.sum())
        self.comp_mask = (np.arange(3)[None, :] < ncomp[:, None])
        self.quant = np.zeros((self.n, 3, 64), np.uint16)

    def shapes(self):
        return [(int(self.out_h[i]), int(self.out_w[i]), 3) for i in range(self.n)]

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

    def analyze_scans(self, exact=None):
        """Scan analysis of every stream: eligibility for the GPU Huffman decoder, Huffman/quantisation tables,
        position of the entropy-coded segment.  exact: daliamdJpegAnalyzeScan walks the scan and reports its exact
        length; else daliamdJpegAnalyzeHeader (what decoders.image runs): headers only, the segment is "everything
        behind SOS" and the un-stuffing kernel finds its end."""
        host = capi.host()
        exact = self.exact_scan if exact is None else exact
        self.scans = (capi.JpegScan * max(self.n, 1))()
        for i, e in enumerate(self.encoded):
            if exact:
                capi.check_host(host.daliamdJpegAnalyzeScan(e.ctypes.data_as(C.c_void_p), C.c_size_t(e.size),
                                                            C.byref(self.infos[i]), C.byref(self.scans[i])))
            else:
                info = capi.JpegInfo()
                capi.check_host(host.daliamdJpegAnalyzeHeader(e.ctypes.data_as(C.c_void_p), C.c_size_t(e.size),
                                                              C.byref(info), C.byref(self.scans[i])))
                assert bytes(info) == bytes(self.infos[i]), "daliamdJpegAnalyzeHeader and daliamdJpegParse disagree"
        self.scan = np.frombuffer(self.scans, dtype=_dtype(capi.JpegScan))[:self.n]
        self.gpu_eligible = self.scan["eligible"].astype(bool) if self.n else np.zeros(0, bool)
        return self.gpu_eligible

    def upload_streams(self, device):
        """Copies the entropy-coded segments of the GPU-eligible streams to the device (16-byte aligned, one
        pinned staging buffer, one H2D copy) and allocates the decoder's scratch.  Idempotent."""
        if getattr(self, "_ecs_dev", None) is not None and self._ecs_dev.device == device:
            return
        lib = capi.kernels()
        if not hasattr(self, "scan"):
            self.analyze_scans()
        sc = self.scan
        sel = np.nonzero(self.gpu_eligible)[0]
        self._huff_sel = sel
        ecs_len = sc["ecs_length"][sel].astype(np.int64)
        self._ecs_len = ecs_len
        self._ecs_off = np.concatenate([[0], np.cumsum(_align(ecs_len, 16))[:-1]]).astype(np.int64)
        need = np.zeros(len(sel), np.int64)
        nb = C.c_size_t(0)
        total_blocks = (sc["mcus_x"][sel] * sc["mcus_y"][sel] * sc["blocks_per_mcu"][sel]).astype(np.int64)
        mcus, ri = (sc["mcus_x"][sel] * sc["mcus_y"][sel]).astype(np.int64), sc["restart_interval"][sel].astype(np.int64)
        intervals = np.where(ri > 0, (mcus + np.maximum(ri, 1) - 1) // np.maximum(ri, 1), 0)
        for j in range(len(sel)):
            capi.check(lib.daliamdJpegHuffmanScratchBytesRestart(int(ecs_len[j]), int(total_blocks[j]), int(intervals[j]),
                                                                 C.byref(nb)))
            need[j] = nb.value
        self._scratch_off = np.concatenate([[0], np.cumsum(need)[:-1]]).astype(np.int64)
        stage = torch.empty(max(int(_align(ecs_len, 16).sum()), 16), dtype=torch.uint8, pin_memory=True)
        stage_np = stage.numpy()
        for j, i in enumerate(sel):
            o, l = int(sc["ecs_offset"][i]), int(ecs_len[j])
            stage_np[self._ecs_off[j]:self._ecs_off[j] + l] = self.encoded[i][o:o + l]
        self._ecs_stage = stage
        self._ecs_dev = stage.to(device, non_blocking=True)
        self.stream_bytes = int(ecs_len.sum())
        self.huffman_scratch_bytes = int(need.sum())
        self._huff_ws = self.new_huffman_workspace(device)
        # quantisation tables of the GPU-decoded streams come from the scan analysis
        self.quant[sel] = sc["quant"][sel, :3]

    def _device_tables(self, table, device):
        """Per stream of `table`: device address of its finished code tables (one buffer per distinct set, kept by the plan)."""
        lib = capi.kernels()
        nb = C.c_size_t(0)
        capi.check(lib.daliamdJpegHuffmanTablesBytes(C.byref(nb)))
        store = self.__dict__.setdefault("_table_store", {})
        one = table.dtype.itemsize
        out = np.zeros(len(table), np.uint64)
        for j in range(len(table)):
            key = b"".join(table[f][j].tobytes() for f in ("blocks_per_mcu", "comp_of_block", "dc_sel", "ac_sel", "bits", "vals"))
            if key not in store:
                host = torch.empty(nb.value, dtype=torch.uint8)
                capi.check(lib.daliamdJpegHuffmanTablesBuild(C.c_void_p(table.ctypes.data + j * one), C.c_void_p(host.data_ptr())))
                store[key] = host.to(device)
            out[j] = store[key].data_ptr()
        return out

    def set_index_mode(self, mode, device=None):
        """None / "build" / "use" (see huffman_descs).  The entries live in one device buffer of the plan."""
        assert mode in (None, "build", "use")
        if mode and getattr(self, "_index_dev", None) is None:
            self.upload_streams(device)
            lib = capi.kernels()
            nb = C.c_size_t(0)
            sizes = []
            for l in self._ecs_len:
                capi.check(lib.daliamdJpegHuffmanIndexBytes(int(l), C.byref(nb)))
                sizes.append(nb.value)
            sizes = np.asarray(sizes, np.int64)
            self._index_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(sizes) else np.zeros(0, np.int64)
            self.index_bytes = int(sizes.sum())
            base = torch.zeros(max(self.index_bytes, 64) + 64, dtype=torch.uint8, device=device)
            self._index_base = base
            self._index_dev = base[(-base.data_ptr()) % 64:]      # 64-byte aligned entries
        self.index_mode = mode

    def host_index(self, j):
        """Index entry of GPU-eligible stream j (position in the decoder's table) built on the HOST
        (daliamdJpegHuffmanIndexBuildHost: what tools/jpeg2idx.py writes next to a file).  Returns (bytes as uint8 array,
        status)."""
        lib = capi.kernels()
        if getattr(self, "_huff_template", None) is None:
            raise RuntimeError("host_index: build the decoder's table first (huffman_descs)")
        d = self._huff_template[j:j + 1].copy()
        i = int(self._huff_sel[j])
        o, l = int(self.scan["ecs_offset"][i]), int(self._ecs_len[j])
        seg = np.ascontiguousarray(self.encoded[i][o:o + l])
        d["ecs"] = seg.ctypes.data
        nb = C.c_size_t(0)
        capi.check(lib.daliamdJpegHuffmanIndexBytes(l, C.byref(nb)))
        out = np.zeros(nb.value, np.uint8)
        status = C.c_int32(0)
        capi.check(lib.daliamdJpegHuffmanIndexBuildHost(d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(status)))
        return out, status.value

    def new_huffman_workspace(self, device):
        """Decoder scratch + status words for one batch in flight (pipelined callers keep one per slot)."""
        return {"scratch": torch.empty(max(self.huffman_scratch_bytes, 256), dtype=torch.uint8, device=device),
                "status": torch.zeros(max(len(self._huff_sel), 1), dtype=torch.int32, device=device)}

    def huffman_descs(self, coef_dev, ws=None, planes_dev=None, rgb_dev=None):
        """daliamdJpegHuffDesc table of the GPU-eligible streams (numpy structured array) + the grid sizes.
        planes_dev: fused output - the decoder dequantises and inverse-transforms the blocks itself and writes the
        component planes (the input of the colour kernel); coef_dev may be None then.
        rgb_dev: the batch's output buffer (JpegBatchPlan.out_off / out_pitch layout) - streams whose geometry allows it
        (YCbCr 4:2:0, no region of interest) leave the decoder as RGB (fused colour output; self.fused_color says which)."""
        lib = capi.kernels()
        ws = ws or self._huff_ws
        # the table only depends on the plan and on where the buffers are: a caller that decodes the same resident batch
        # into the same buffers again (benchmarks) gets the table it built the first time
        key = (ws["scratch"].data_ptr(), ws["status"].data_ptr(), None if coef_dev is None else coef_dev.data_ptr(),
               None if planes_dev is None else planes_dev.data_ptr(), self._ecs_dev.data_ptr(),
               None if rgb_dev is None else rgb_dev.data_ptr(), getattr(self, "index_mode", None),
               getattr(self, "host_tables", False))
        cache = self.__dict__.setdefault("_huff_desc_cache", {})
        if key in cache:
            return cache[key]
        sc, inf, sel = self.scan, self.inf, self._huff_sel
        m = len(sel)
        # everything that only depends on the streams (tables, geometry, quantisation) is laid out once per plan;
        # a call copies that template and fills in the buffer addresses
        tmpl = getattr(self, "_huff_template", None)
        if tmpl is None:
            t = np.zeros(max(m, 1), _dtype(capi.JpegHuffDesc))[:m]
            t["ecs_len"] = self._ecs_len
            t["blocks_per_mcu"] = sc["blocks_per_mcu"][sel]
            t["mcus_x"] = sc["mcus_x"][sel]
            t["total_blocks"] = sc["mcus_x"][sel] * sc["mcus_y"][sel] * sc["blocks_per_mcu"][sel]
            t["restart_interval"] = sc["restart_interval"][sel]
            t["blocks_x"] = inf["blocks_x"][sel, :3]
            t["h_samp"] = inf["h_samp"][sel, :3]
            t["v_samp"] = inf["v_samp"][sel, :3]
            t["comp_of_block"][:, :10] = sc["comp_of_block"][sel]
            t["h_of_block"][:, :10] = sc["h_of_block"][sel]
            t["v_of_block"][:, :10] = sc["v_of_block"][sel]
            t["dc_sel"] = sc["dc_sel"][sel]
            t["ac_sel"] = sc["ac_sel"][sel]
            t["bits"][:, 0:2] = sc["dc_bits"][sel, 0:2]
            t["bits"][:, 2:4] = sc["ac_bits"][sel, 0:2]
            t["vals"][:, 0:2] = sc["dc_vals"][sel, 0:2]
            t["vals"][:, 2:4] = sc["ac_vals"][sel, 0:2]
            t["quant"] = self.quant[sel]
            t["plane_pitch"] = np.where(self.comp_mask[sel], inf["blocks_x"][sel, :3] * 8, 0)
            if self.roi_plans is not None:
                t["rect"] = np.where(self.has_roi[sel][:, None, None], self.roi_plans["rect"][sel], 0)
            tmpl = self._huff_template = t
        d = tmpl.copy()
        d["ecs"] = self._ecs_dev.data_ptr() + self._ecs_off
        d["scratch"] = ws["scratch"].data_ptr() + self._scratch_off
        d["status"] = ws["status"].data_ptr() + 4 * np.arange(m)
        if getattr(self, "host_tables", False) and m:
            # code tables built on the host, once per distinct DHT contents + MCU structure of the batch, resident on the
            # device (daliamdJpegHuffDesc.tables): the launch builds nothing
            d["tables"] = self._device_tables(d, ws["scratch"].device)
        mode = getattr(self, "index_mode", None)
        if mode and m:
            # side information of resident streams (daliamdJpegHuffDesc.index / index_out): "build" - this decode leaves an
            # index entry per stream in self._index_dev; "use" - decode from those entries (the segments are not looked at)
            ok = sc["restart_interval"][sel] == 0
            d["index_out" if mode == "build" else "index"] = np.where(ok, self._index_dev.data_ptr() + self._index_off, 0)
            if mode == "use":
                d["ecs"] = np.where(ok, 0, d["ecs"])
        if coef_dev is not None:
            d["coef"] = np.where(self.comp_mask[sel], coef_dev.data_ptr() + 2 * self.coef_off[sel], 0)
        if planes_dev is not None:
            d["plane"] = np.where(self.comp_mask[sel], planes_dev.data_ptr() + self.plane_off[sel], 0)
        else:
            d["plane_pitch"] = 0
        fused = np.zeros(self.n, bool)
        if rgb_dev is not None and m:
            rgb_ptr = rgb_dev.data_ptr() + self.out_off[sel]
            ok = (inf["color"][sel] != capi.JPEG_RGB) & ((inf["width"][sel] > 4) | (sc["blocks_per_mcu"][sel] != 6)) & \
                (self.out_pitch[sel] % 8 == 0) & (rgb_ptr % 8 == 0)
            if self.roi_plans is not None:
                ok &= ~self.has_roi[sel]
            one = _dtype(capi.JpegHuffDesc).itemsize
            base = d.ctypes.data
            for j in np.nonzero(ok)[0]:
                ok[j] = bool(lib.daliamdJpegHuffmanColorFusable(C.c_void_p(base + int(j) * one)))
            d["rgb"] = np.where(ok, rgb_ptr, 0)
            d["rgb_pitch"] = np.where(ok, self.out_pitch[sel], 0)
            d["width"] = inf["width"][sel]
            d["height"] = inf["height"][sel]
            fused[sel] = ok
        self.fused_color = fused
        ntiles, nsegs, nbwg, kinds = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        capi.check(lib.daliamdJpegHuffmanSetupColor(d.ctypes.data_as(C.c_void_p), m, C.byref(ntiles), C.byref(nsegs),
                                                    C.byref(nbwg), C.byref(kinds)))
        if len(cache) > 8:
            cache.clear()
        cache[key] = _HuffDescs((d, ntiles.value, nsegs.value, nbwg.value), kinds.value, fused)
        return cache[key]

    def run_gpu_huffman(self, coef_dev, descs=None, events=None, ws=None, kernel_events=None, planes_dev=None, rgb_dev=None):
        """Launches the GPU entropy decoder for the uploaded streams on the current stream (every decoded block is
        written exactly once as a full line: no zero-fill).  events: optional (before, after) events for timing.
        planes_dev: fused dequantisation + IDCT; rgb_dev: fused colour output where possible (see huffman_descs)."""
        lib = capi.kernels()
        dev = (coef_dev if coef_dev is not None else planes_dev if planes_dev is not None else rgb_dev).device
        m = len(self._huff_sel)
        ws = ws or self._huff_ws
        if descs is None:
            descs = self.huffman_descs(coef_dev, ws, planes_dev, rgb_dev)
        table, ntiles, nsegs, nbwg = descs
        kinds = getattr(descs, "kinds", 1)
        self.fused_color = getattr(descs, "fused", np.zeros(self.n, bool))
        d_dev = _uploader.upload(table, dev) if m else None
        s = current_stream_ptr(dev)
        if events:
            events[0].record()
        if m and kernel_events is not None:   # ctypes array of daliamdEvent_t (see KernelEvents)
            capi.check(lib.daliamdJpegHuffmanRunProfiledColor(s, C.c_void_p(d_dev.data_ptr()), m, ntiles, nsegs, nbwg, kinds,
                                                              kernel_events))
        elif m:
            capi.check(lib.daliamdJpegHuffmanRunColor(s, C.c_void_p(d_dev.data_ptr()), m, ntiles, nsegs, nbwg, kinds))
        if events:
            events[1].record()
        self._huff_keep = [d_dev]
        return ws["status"][:m]

    def entropy_decode_gpu(self, coef_dev, num_threads=None, planes_dev=None, rgb_dev=None):
        """Entropy-decodes the batch into `coef_dev` (int16 device tensor of self.coef_elems elements):
        eligible streams on the GPU (daliamdJpegHuffmanRun), the rest (progressive,
        multi-scan) on the host.  Returns the device status tensor (one int32 per GPU-decoded stream) and the
        list of sample indices it refers to; call `check_gpu_status` once the stream is synchronised."""
        dev = coef_dev.device
        self.upload_streams(dev)
        sel = self._huff_sel
        rest = np.nonzero(~self.gpu_eligible)[0]
        status = self.run_gpu_huffman(coef_dev, planes_dev=planes_dev, rgb_dev=rgb_dev)
        keep = []
        if len(rest):
            host = capi.host()

            def one(i):
                n_el = int(self.inf["coef_elems"][i, :3][self.comp_mask[i]].sum())
                buf = torch.empty(n_el, dtype=torch.int16, pin_memory=True)
                ptrs = (C.c_void_p * 4)()
                for c in range(self.infos[i].num_components):
                    ptrs[c] = buf.data_ptr() + 2 * int(self.coef_off[i, c] - self.coef_off[i, 0])
                e = self.encoded[i]
                rc = host.daliamdJpegDecodeCoefficients(e.ctypes.data_as(C.c_void_p), C.c_size_t(e.size),
                                                        C.byref(self.infos[i]), ptrs,
                                                        self.quant[i].ctypes.data_as(C.c_void_p))
                if rc:
                    msg = host.daliamdHostGetLastErrorMessage()
                    raise capi.DaliAmdError(f"sample {i}: {msg.decode() if msg else 'decode failed'}")
                return buf

            bufs = list(_thread_pool(num_threads).map(one, rest)) if len(rest) > 1 else [one(int(rest[0]))]
            for i, buf in zip(rest, bufs):
                o = int(self.coef_off[i, 0])
                coef_dev[o:o + buf.numel()].copy_(buf, non_blocking=True)
            keep += bufs
        self._huff_keep += keep
        return status, sel

    def huffman_block_starts(self, ws=None):
        """Block starts the last GPU entropy decode of this batch found per stream (int32 [2] of the per-stream scratch
        headers, jpeg_huffman.hip MakeLayout): blocks + 1 for a complete stream.  Synchronises."""
        ws = ws or self._huff_ws
        if not len(self._huff_sel):
            return np.zeros(0, np.int64)
        words = ws["scratch"].view(torch.int32)
        idx = torch.as_tensor(self._scratch_off // 4 + 2, device=words.device)
        return words[idx].cpu().numpy().astype(np.int64)

    def check_gpu_status(self, status):
        """Raises for GPU-decoded streams whose entropy-coded segment was short of blocks (corrupt / truncated)."""
        st = status.cpu().numpy()
        bad = np.nonzero(st)[0]
        if len(bad):
            i = int(self._huff_sel[bad[0]])
            why = {3: "restart markers in a stream without a restart interval",
                   4: "a restart interval does not end where its marker is"}.get(
                       int(st[bad[0]]), "the entropy-coded segment ends before the last MCU")
            raise capi.DaliAmdError(f"sample {i}: corrupt JPEG data: {why} (GPU Huffman status {int(st[bad[0]])})")

    def build_descs(self, coef_dev, planes_dev, out_dev, fused_huffman=False):
        """IDCT + colour descriptor tables (numpy structured arrays mirroring the C structs).
        fused_huffman: the GPU entropy decoder already wrote the planes of the streams it decoded (huffman_descs with
        planes_dev): only the host-decoded streams go through the IDCT kernel."""
        lib = capi.kernels()
        skip = getattr(self, "fused_color", None) if fused_huffman else None   # left the entropy decoder as RGB
        key = (coef_dev.data_ptr(), planes_dev.data_ptr(), out_dev.data_ptr(), bool(fused_huffman),
               None if skip is None else skip.tobytes())
        cache = self.__dict__.setdefault("_stage_desc_cache", {})
        if key in cache:
            return cache[key]
        inf, m = self.inf, self.comp_mask
        color_mask = m
        if fused_huffman:
            m = m & ~self.gpu_eligible[:, None]
        cb, pb, ob = coef_dev.data_ptr(), planes_dev.data_ptr(), out_dev.data_ptr()
        ncomp_total = int(m.sum())
        idct = np.zeros(max(ncomp_total, 1), _dtype(capi.JpegIdctDesc))
        color = np.zeros(max(self.n, 1), _dtype(capi.JpegColorDesc))
        if self.n:
            bx = inf["blocks_x"][:, :3]
            by = inf["blocks_y"][:, :3]
            plane_ptr = pb + self.plane_off
            idct["coef"][:ncomp_total] = (cb + 2 * self.coef_off)[m]
            idct["plane"][:ncomp_total] = plane_ptr[m]
            idct["blocks_x"][:ncomp_total] = bx[m]
            idct["nblocks"][:ncomp_total] = (bx * by)[m]
            idct["pitch"][:ncomp_total] = (bx * 8)[m]
            idct["quant"][:ncomp_total] = self.quant[m]
            if self.roi_plans is not None:
                rect = self.roi_plans["rect"]                       # [n, 3, 4]
                roi3 = np.broadcast_to(self.has_roi[:, None], m.shape)
                rw = np.where(roi3, rect[:, :, 2] - rect[:, :, 0], 0)
                rh = rect[:, :, 3] - rect[:, :, 1]
                idct["rect_x0"][:ncomp_total] = np.where(roi3, rect[:, :, 0], 0)[m]
                idct["rect_y0"][:ncomp_total] = np.where(roi3, rect[:, :, 1], 0)[m]
                idct["rect_w"][:ncomp_total] = rw[m]
                idct["nblocks"][:ncomp_total] = np.where(roi3, rw * rh, bx * by)[m]
            color["plane"][:self.n] = np.where(color_mask, plane_ptr, 0)
            color["pitch"][:self.n] = np.where(color_mask, bx * 8, 0)
            color["h_samp"][:self.n] = np.where(color_mask, inf["h_samp"][:, :3], 1)
            color["v_samp"][:self.n] = np.where(color_mask, inf["v_samp"][:, :3], 1)
            color["down_w"][:self.n] = inf["down_w"][:, :3]
            color["down_h"][:self.n] = inf["down_h"][:, :3]
            color["width"][:self.n] = inf["width"]
            color["height"][:self.n] = inf["height"]
            color["color"][:self.n] = inf["color"]
            color["out"][:self.n] = ob + self.out_off
            color["out_pitch"][:self.n] = self.out_pitch
            if self.roi_plans is not None:
                for f in ("roi_x0", "roi_y0", "roi_w", "roi_h", "out_x0", "out_y0"):
                    color[f][:self.n] = np.where(self.has_roi, self.roi_plans[f], 0)
        n_color = self.n
        if skip is not None and skip.any():
            kept = color[:self.n][~skip]
            n_color = len(kept)
            color = np.ascontiguousarray(kept) if n_color else np.zeros(1, _dtype(capi.JpegColorDesc))
        n_idct_wg, n_color_wg, color_mask_out = C.c_int(0), C.c_int(0), C.c_int(0)
        capi.check(lib.daliamdJpegIdctSetup(idct.ctypes.data_as(C.c_void_p), ncomp_total, C.byref(n_idct_wg)))
        capi.check(lib.daliamdJpegColorSetup(color.ctypes.data_as(C.c_void_p), n_color, C.byref(n_color_wg),
                                             C.byref(color_mask_out)))
        if len(cache) > 8:
            cache.clear()
        cache[key] = ((idct, ncomp_total, n_idct_wg.value), (color, n_color, (n_color_wg.value, color_mask_out.value)))
        return cache[key]

    def output_views(self, out_dev):
        views = []
        for i in range(self.n):
            inf = self.infos[i]
            v = torch.as_strided(out_dev, (int(self.out_h[i]), int(self.out_w[i]), 3), (int(self.out_pitch[i]), 3, 1),
                                 int(self.out_off[i]))
            views.append(v)
        return views


def jpeg_gpu_stage(plan, coef_dev, planes_dev, out_dev, descs=None, split_events=None, start_event=None,
                   fused_huffman=False):
    """Enqueues dequant+IDCT and upsample+colour for a planned batch on the current stream.
    split_events: optional (event_before_color,) recorded between the two kernels (bench timing).
    fused_huffman: see JpegBatchPlan.build_descs."""
    lib = capi.kernels()
    if descs is None:
        descs = plan.build_descs(coef_dev, planes_dev, out_dev, fused_huffman=fused_huffman)
    (idct, n_idct, wg_idct), (color, n_color, wg_color) = descs
    dev = coef_dev.device
    idct_dev = _uploader.upload(idct, dev) if n_idct else None
    color_dev = _uploader.upload(color, dev)
    s = current_stream_ptr(dev)
    if start_event is not None:
        start_event.record()   # after the descriptor uploads: brackets the kernels only
    if n_idct:
        capi.check(lib.daliamdJpegIdctRun(s, C.c_void_p(idct_dev.data_ptr()), n_idct, wg_idct))
    if split_events:
        split_events[0].record()
    capi.check(lib.daliamdJpegColorRun(s, C.c_void_p(color_dev.data_ptr()), n_color, wg_color[0], wg_color[1]))
    return idct_dev, color_dev


class _HuffDescs(tuple):
    """(table, tiles, segments, block workgroups) of JpegBatchPlan.huffman_descs + which block kernel instances the table
    needs (`kinds`: bit 0 planes / coefficients, bit 1 fused colour output) and which samples leave as RGB (`fused`)."""

    def __new__(cls, items, kinds, fused):
        self = super().__new__(cls, items)
        self.kinds, self.fused = kinds, fused
        return self


def decode_jpeg_batch(encoded, device="cuda", num_threads=None, out_pitch_align=16, huffman="gpu", rois=None, exact_scan=True,
                      fuse_color=False, index=None, index_from=None, host_tables=False):
    """Decodes a batch of JPEG byte strings -> list of u8 HWC RGB device tensors.
    rois: optional per-sample windows (y0, x0, h, w): region-of-interest decode (decoders.image_crop & co.).

    huffman="gpu": entropy decoding on the device for baseline single-scan streams (host for the rest);
    huffman="host": header parse + Huffman on the host thread pool into pinned memory (the hybrid path).
    fuse_color: 4:2:0 / 4:4:4 / grayscale streams leave the GPU entropy decoder as RGB (daliamdJpegHuffDesc.rgb; same bits,
    fewer bytes moved, measured slower on MI355X - see HISTORY.md section 9 - hence opt-in).
    Dequantisation, IDCT, upsampling and colour conversion always run on the device."""
    device = torch.device(device)
    plan = JpegBatchPlan(encoded, out_pitch_align, rois=rois, exact_scan=exact_scan)
    plan.host_tables = host_tables
    if index:
        # index="build": the decode leaves the side information of every stream in the plan; index="use" with
        # index_from=<the plan of a "build" decode of the same streams>: decode from it (other windows are fine)
        if index == "use":
            plan.upload_streams(device)
            plan._index_dev, plan._index_off, plan.index_bytes = index_from._index_dev, index_from._index_off, index_from.index_bytes
            assert np.array_equal(plan._ecs_len, index_from._ecs_len)
        plan.set_index_mode(index, device)
    status = None
    planes = torch.empty(max(plan.plane_bytes, 1), dtype=torch.uint8, device=device)
    out = torch.empty(max(plan.out_bytes, 1), dtype=torch.uint8, device=device)
    if huffman == "gpu":
        # the GPU entropy decoder writes the component planes itself (fused dequantisation + IDCT); the coefficient
        # buffer only carries the host-decoded streams (progressive, restart markers, ...) to the IDCT kernel
        coef_host = None
        coef_dev = torch.empty(max(plan.coef_elems, 1), dtype=torch.int16, device=device)
        # ... and, fuse_color, the RGB image of every YCbCr 4:2:0 stream (no colour launch for those)
        status, _ = plan.entropy_decode_gpu(coef_dev, num_threads, planes_dev=planes, rgb_dev=out if fuse_color else None)
    elif huffman == "host":
        coef_host = torch.empty(max(plan.coef_elems, 1), dtype=torch.int16, pin_memory=True)
        plan.entropy_decode(coef_host, num_threads)
        coef_dev = coef_host.to(device, non_blocking=True)
    else:
        raise ValueError(f"huffman must be 'gpu' or 'host', got {huffman!r}")
    keep = jpeg_gpu_stage(plan, coef_dev, planes, out,
                          descs=plan.build_descs(coef_dev, planes, out, fused_huffman=huffman == "gpu"))
    views = plan.output_views(out)
    if status is not None and status.numel():
        plan.check_gpu_status(status)   # synchronises
    # keep scratch alive until the stream has consumed it
    for t in (coef_dev, planes) + tuple(keep):
        if getattr(t, "is_cuda", False):
            t.record_stream(torch.cuda.current_stream(device))
    plan._keepalive = (coef_host, coef_dev)
    return views, plan


# =====================================================================================
# Resample (+ fused CropMirrorNormalize)
# =====================================================================================
_workspaces = {}


def _stream_workspace(dev, stream_ptr, nbytes):
    """Device scratch of a launch sequence, one buffer per (device, stream): kernels of one stream are ordered, so the
    next call's writes come after this call's reads."""
    key = (str(dev), int(getattr(stream_ptr, "value", stream_ptr) or 0))
    t = _workspaces.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes * 3 // 2), 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = t
    return t


def _fill4(dst, src):
    for i in range(4):
        dst[i] = float(src[i]) if i < len(src) else (float(src[-1]) if len(src) == 1 else 0.0)


HUFFMAN_KERNELS = ("PrepareKernel", "UnstuffScatterKernel", "SyncKernel", "PropagateKernel", "DcKernel", "BlockKernel")
# names the launches of daliamdJpegHuffmanRunColor are timed under (daliamdKernelTimingReport): the six stages above, the
# block kernel's instance with the fused colour output and the seam launch behind it
HUFFMAN_KERNEL_NAMES = HUFFMAN_KERNELS + ("BlockColorKernel", "SeamKernel", "IndexedSyncKernel", "IndexBuildKernel")


def huffman_algorithmic_bytes(stream_bytes, coef_elems, num_streams, fused):
    """Algorithmic HBM bytes per launch of each kernel of daliamdJpegHuffmanRun for a batch of
    `num_streams` streams with `stream_bytes` entropy-coded bytes and `coef_elems` coefficients (64 per block)."""
    blocks = coef_elems / 64
    return {"PrepareKernel": stream_bytes + 60 * 1024 * num_streams,   # stream read once + 60 KB of code tables per stream
            "UnstuffScatterKernel": 2 * stream_bytes,
            "SyncKernel": stream_bytes + 4 * blocks,       # + the block starts out
            "PropagateKernel": 0,
            "DcKernel": (4 + 10) * blocks,                  # start in; position, level and segment of the block out
            # stream + per-block position / level in; 8-bit samples (fused dequantisation + IDCT) or coefficients out
            "BlockKernel": stream_bytes + 10 * blocks + (coef_elems if fused else 2 * coef_elems)}


def count_huffman_symbols(coef, coef_elems):
    """Huffman symbols of a batch from its decoded coefficients (int16 tensor, 64 per block, column-major blocks): per
    block the DC symbol, one symbol per non-zero AC coefficient, a ZRL per full run of 16 zeros in front of one, and the
    end-of-block unless the last coefficient of the scan is non-zero.  Exact; runs where the tensor lives."""
    a = torch.as_tensor(coef)[:coef_elems].reshape(-1, 64)
    scan = torch.tensor([0, 8, 1, 2, 9, 16, 24, 17, 10, 3, 4, 11, 18, 25, 32, 40, 33, 26, 19, 12, 5, 6, 13, 20, 27, 34, 41,
                         48, 56, 49, 42, 35, 28, 21, 14, 7, 15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45,
                         52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63], device=a.device)   # column-major block -> scan order
    pos = torch.arange(1, 64, device=a.device, dtype=torch.int32)[None, :]
    total = 0
    for lo in range(0, a.shape[0], 1 << 17):
        nz = a[lo:lo + (1 << 17)][:, scan][:, 1:] != 0
        total += nz.shape[0] + int(nz.sum()) + int((~nz[:, -1]).sum())
        idx = torch.where(nz, pos, torch.zeros_like(pos))
        prev = torch.cummax(idx, dim=1).values
        prev = torch.cat([torch.zeros_like(prev[:, :1]), prev[:, :-1]], dim=1)   # scan index of the previous non-zero
        total += int(torch.where(nz, (pos - prev - 1) // 16, torch.zeros_like(pos)).sum())
    return total


class KernelEvents:
    """n+1 timing events of the kernel library (daliamdEvent*) bracketing n consecutive kernels of one launch call."""

    def __init__(self, n):
        lib = capi.kernels()
        self.n = n
        self.handles = (C.c_void_p * (n + 1))()
        for i in range(n + 1):
            capi.check(lib.daliamdEventCreate(C.byref(self.handles, i * C.sizeof(C.c_void_p)), 1))

    def elapsed_ms(self):
        """Per-kernel durations; call after the stream has been synchronised."""
        lib = capi.kernels()
        out, ms = [], C.c_float(0)
        for i in range(self.n):
            capi.check(lib.daliamdEventElapsedMs(C.c_void_p(self.handles[i]), C.c_void_p(self.handles[i + 1]), C.byref(ms)))
            out.append(ms.value)
        return out

    def __del__(self):
        try:
            lib = capi.kernels()
            for h in self.handles:
                if h:
                    lib.daliamdEventDestroy(C.c_void_p(h))
        except Exception:
            pass


class ImageTable:
    """Addresses and geometry of a batch of u8 HWC device images as numpy columns (validated once), so that the
    per-batch descriptor construction is vectorised instead of touching every tensor from Python."""

    def __init__(self, images):
        for img in images:
            if img.dtype not in _KERNEL_DTYPE or img.dim() != 3 or img.stride(2) != 1 or img.stride(1) != img.shape[2]:
                raise capi.DaliAmdError("expected u8 (or i16 / u16 / f32) HWC tensors with dense pixels")
        self.n = len(images)
        self.dtype = _KERNEL_DTYPE[images[0].dtype] if images else capi.UINT8
        if any(_KERNEL_DTYPE[img.dtype] != self.dtype for img in images):
            raise capi.DaliAmdError("all images of a batch must have the same element type")
        self.esize = images[0].element_size() if images else 1
        self.device = images[0].device if self.n else torch.device("cuda")
        self.ptr = np.array([img.data_ptr() for img in images], np.uint64)
        shp = np.array([tuple(img.shape) for img in images], np.int32).reshape(-1, 3)
        self.h, self.w, self.c = shp[:, 0], shp[:, 1], shp[:, 2]
        self.pitch = np.array([img.stride(0) * img.element_size() for img in images], np.int32)   # bytes
        self.images = images   # keeps the storage alive


def resample_batch(images, out_size, rois=None, interp_min=capi.INTERP_LINEAR, interp_mag=capi.INTERP_LINEAR,
                   antialias=True, out_dtype=capi.UINT8, out_layout=capi.LAYOUT_HWC, mean=None, inv_std=None,
                   mirror=None, out=None, return_descs=False, start_event=None, unrounded=False):
    """Resamples a batch of u8 HWC device tensors (possibly row-strided views) to out_size=(H, W).

    rois[i] = (y0, x0, y1, x1) in source pixels or None.  With mean/inv_std the CropMirrorNormalize
    epilogue is fused: output dtype float16/float32, layout CHW or HWC, optional per-sample mirror.
    i16 / u16 / f32 images resample to their own type (pass out_dtype=None); unrounded=True returns the float result of
    the second pass as it is (fn.resize(dtype=FLOAT)).  Returns a dense tensor [N, ...]."""
    lib = capi.kernels()
    tab = images if isinstance(images, ImageTable) else ImageTable(images)
    n = tab.n
    if unrounded:
        out_dtype = capi.FLOAT
    elif out_dtype is None or (tab.dtype != capi.UINT8 and out_dtype == capi.UINT8 and mean is None):
        out_dtype = tab.dtype
    oh, ow = int(out_size[0]), int(out_size[1])
    dev = tab.device
    ch = int(tab.c[0]) if n else 3
    normalize = mean is not None
    if out is None:
        shape = (n, ch, oh, ow) if out_layout == capi.LAYOUT_CHW else (n, oh, ow, ch)
        out = torch.empty(shape, dtype=_TORCH_DTYPE[out_dtype], device=dev)
    esz = out.element_size()
    per_sample = oh * ow * ch * esz
    args = np.zeros(max(n, 1), _dtype(capi.ResampleArgs))
    if n:
        a = args[:n]
        a["in_"] = tab.ptr
        a["in_h"], a["in_w"], a["channels"] = tab.h, tab.w, tab.c
        a["in_pitch"] = tab.pitch
        if rois is not None:
            if isinstance(rois, np.ndarray):
                r = rois.astype(np.float32)
                a["use_roi"] = 1
            else:
                r = np.array([rr if rr is not None else (0, 0, 0, 0) for rr in rois], np.float32)
                a["use_roi"] = [rr is not None for rr in rois]
            a["roi_y0"], a["roi_x0"], a["roi_y1"], a["roi_x1"] = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
        a["out_h"], a["out_w"] = oh, ow
        a["min_filter"], a["mag_filter"], a["antialias"] = interp_min, interp_mag, 1 if antialias else 0
        a["out"] = out.data_ptr() + per_sample * np.arange(n, dtype=np.int64)
        a["out_dtype"], a["out_layout"] = out_dtype, out_layout
        a["normalize"] = 1 if normalize else 0
        a["in_dtype"], a["unrounded"] = tab.dtype, 1 if unrounded else 0
        if mirror is not None:
            a["mirror"] = np.asarray(mirror, np.int32)
        if normalize:
            m4, i4 = np.zeros(4, np.float32), np.zeros(4, np.float32)
            _fill4(m4, mean)
            _fill4(i4, inv_std)
            a["mean"], a["inv_std"] = m4, i4
    descs = np.zeros(max(n, 1), _dtype(capi.ResampleDesc))
    plan = capi.ResamplePlan()
    capi.check(lib.daliamdResampleSetup(args.ctypes.data_as(C.c_void_p), n, descs.ctypes.data_as(C.c_void_p), C.byref(plan)))
    descs_dev = _uploader.upload(descs, dev)
    stream_ptr = current_stream_ptr(dev)
    ws = _stream_workspace(dev, stream_ptr, plan.workspace_bytes)
    if start_event is not None:
        start_event.record()
    capi.check(lib.daliamdResampleRun(stream_ptr, C.c_void_p(descs_dev.data_ptr()), n, C.byref(plan), C.c_void_p(ws.data_ptr())))
    if return_descs:
        return out, descs, plan.num_tiles, plan.lds_bytes
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


# =====================================================================================
# heavy augmentation kernels (configs[2]): thin batch drivers over the C ABI
# =====================================================================================
def _img_fields(d, img, out):
    d["in_"], d["out"] = img.data_ptr(), out.data_ptr()


def warp_affine_batch(images, matrices, out_size=None, interp=capi.INTERP_LINEAR, fill_value=None):
    """matrices[i]: 2x3 dst->src.  fill_value None -> clamp border."""
    lib = capi.kernels()
    n = len(images)
    dev = images[0].device
    descs = np.zeros(n, _dtype(capi.WarpAffineDesc))
    outs = []
    for i, img in enumerate(images):
        h, w, c = img.shape
        oh, ow = (h, w) if out_size is None else out_size
        out = torch.empty((oh, ow, c), dtype=torch.uint8, device=dev)
        outs.append(out)
        d = descs[i]
        d["in_"], d["out"] = img.data_ptr(), out.data_ptr()
        d["in_h"], d["in_w"], d["channels"], d["in_pitch"] = h, w, c, img.stride(0)
        d["out_h"], d["out_w"], d["out_pitch"] = oh, ow, ow * c
        d["matrix"] = np.asarray(matrices[i], np.float32).reshape(6)
        d["interp"] = interp
        d["border_clamp"] = 1 if fill_value is None else 0
        if fill_value is not None:
            d["fill"] = np.broadcast_to(np.asarray(fill_value, np.float32), (4,)) if np.ndim(fill_value) == 0 else \
                np.pad(np.asarray(fill_value, np.float32), (0, 4 - len(fill_value)))
    nwg = C.c_int(0)
    capi.check(lib.daliamdWarpAffineSetup(descs.ctypes.data_as(C.c_void_p), n, C.byref(nwg)))
    dd = _uploader.upload(descs, dev)
    capi.check(lib.daliamdWarpAffineRun(current_stream_ptr(dev), C.c_void_p(dd.data_ptr()), n, nwg.value))
    return outs


def gaussian_window(sigma=0.0, window_size=0):
    lib = capi.kernels()
    w = np.zeros(64, np.float32)
    s = C.c_float(0)
    d = lib.daliamdGaussianWindow(C.c_float(sigma), int(window_size), w.ctypes.data_as(C.c_void_p), C.byref(s))
    if d < 0:
        capi.check(1)
    return w[:d].copy()


def gaussian_blur_batch(images, sigma=0.0, window_size=0):
    lib = capi.kernels()
    n = len(images)
    dev = images[0].device
    descs = np.zeros(n, _dtype(capi.GaussianBlurDesc))
    sig = np.broadcast_to(np.asarray(sigma, np.float32), (n,))
    wsz = np.broadcast_to(np.asarray(window_size, np.int32), (n,))
    outs = []
    for i, img in enumerate(images):
        h, w, c = img.shape
        out = torch.empty((h, w, c), dtype=torch.uint8, device=dev)
        outs.append(out)
        win = gaussian_window(float(sig[i]), int(wsz[i]))
        d = descs[i]
        d["in_"], d["out"] = img.data_ptr(), out.data_ptr()
        d["h"], d["w"], d["channels"], d["in_pitch"], d["out_pitch"] = h, w, c, img.stride(0), w * c
        d["size_x"] = d["size_y"] = win.size
        d["window_x"][:win.size] = win
        d["window_y"][:win.size] = win
    nwg, lds = C.c_int(0), C.c_int(0)
    capi.check(lib.daliamdGaussianBlurSetup(descs.ctypes.data_as(C.c_void_p), n, C.byref(nwg), C.byref(lds)))
    dd = _uploader.upload(descs, dev)
    capi.check(lib.daliamdGaussianBlurRun(current_stream_ptr(dev), C.c_void_p(dd.data_ptr()), n, nwg.value, lds.value))
    return outs


def color_twist_matrix(hue=0.0, saturation=1.0, value=1.0, brightness=1.0, contrast=1.0):
    lib = capi.kernels()
    m = np.zeros(9, np.float32)
    off = C.c_float(0)
    lib.daliamdColorTwistMatrix(C.c_float(hue), C.c_float(saturation), C.c_float(value), C.c_float(brightness),
                                C.c_float(contrast), m.ctypes.data_as(C.c_void_p), C.byref(off))
    return m.reshape(3, 3), np.float32(off.value)


def pointwise_batch(images, matrices=None, offsets=None, regions=None, fill=(0.0,)):
    """Colour transform (matrices[i] 3x3 + offsets[i]) and/or erase (regions[i] = list of (y0, x0, y1, x1))."""
    lib = capi.kernels()
    n = len(images)
    dev = images[0].device
    descs = np.zeros(n, _dtype(capi.PointwiseDesc))
    outs = []
    for i, img in enumerate(images):
        h, w, c = img.shape
        out = torch.empty((h, w, c), dtype=torch.uint8, device=dev)
        outs.append(out)
        d = descs[i]
        d["in_"], d["out"] = img.data_ptr(), out.data_ptr()
        d["h"], d["w"], d["channels"], d["in_pitch"], d["out_pitch"] = h, w, c, img.stride(0), w * c
        if matrices is not None:
            d["transform"] = 1
            d["matrix"] = np.asarray(matrices[i], np.float32).reshape(9)
            d["offset"] = np.broadcast_to(np.asarray(offsets[i], np.float32), (3,))
        if regions is not None:
            regs = regions[i]
            d["num_regions"] = len(regs)
            for k, r in enumerate(regs):
                y0, x0, y1, x1 = r
                d["region"][k] = (max(0, y0), max(0, x0), min(h, y1), min(w, x1))
        f = list(fill)
        d["fill"] = [f[0]] * 4 if len(f) == 1 else (f + [0.0] * 4)[:4]
    nwg = C.c_int(0)
    capi.check(lib.daliamdPointwiseSetup(descs.ctypes.data_as(C.c_void_p), n, C.byref(nwg)))
    dd = _uploader.upload(descs, dev)
    capi.check(lib.daliamdPointwiseRun(current_stream_ptr(dev), C.c_void_p(dd.data_ptr()), n, nwg.value))
    return outs
