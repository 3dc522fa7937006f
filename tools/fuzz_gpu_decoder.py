"""Mutation fuzzing of the DEVICE side of the JPEG decoder (TEST TOOL): valid baseline streams (4:2:0 / 4:2:2 / 4:4:4 /
grey, optimised tables, restart intervals) are truncated, bit-flipped, overwritten and spliced BEHIND their headers (and
sometimes inside the DHT / DQT / SOF / SOS segments), then decoded through the GPU entropy decoder
(backend.decode_jpeg_batch(huffman="gpu")) on the CPU model of the kernels built with AddressSanitizer
(tools/hipemu, make SAN=address) with zero slack behind the allocations.  A decode either succeeds or raises
DaliAmdError; a sanitizer report, a crash or a decode that does not come back is a defect of a kernel (an out-of-bounds
access that corrupts HBM or a loop that hangs the GPU on the real device).

    tools/fuzz_gpu_decoder.sh [iterations] [seed]
"""
import io
import os
import signal
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
from PIL import Image


def seeds(rng):
    from tests.util import synth_image
    out = []
    for (h, w), kw in [((61, 83), dict(subsampling="4:2:0")), ((48, 64), dict(subsampling="4:4:4")),
                       ((40, 72), dict(subsampling="4:2:2")), ((64, 64), dict(subsampling="4:2:0", optimize=True)),
                       ((72, 96), dict(subsampling="4:2:0", restart_marker_blocks=2)),
                       ((50, 70), dict(subsampling="4:2:0", restart_marker_rows=1)),
                       ((120, 160), dict(subsampling="4:2:0", quality=95))]:
        b = io.BytesIO()
        Image.fromarray(synth_image(rng, h, w)).save(b, "JPEG", **dict(dict(quality=80), **kw))
        out.append(b.getvalue())
    b = io.BytesIO()
    Image.fromarray(synth_image(rng, 40, 56)).convert("L").save(b, "JPEG", quality=75)
    out.append(b.getvalue())
    # flat and banded content: hundreds of block starts per 256-byte slice - the lanes whose start list outgrows its LDS slots
    # and decode again from its last entry (round 5)
    for sub, rst in (("4:2:0", {}), ("4:4:4", {}), ("4:2:0", dict(restart_marker_blocks=5))):
        img = synth_image(rng, 160, 240).copy()
        img[24:120] = (90, 200, 30)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=sub, **rst)
        out.append(b.getvalue())
    if os.environ.get("FUZZ_BIG") == "1":   # streams of several 61 KB segments (the serial repair pass over segments) and tiles
        out = []
        for (h, w), kw in [((600, 800), dict(subsampling="4:2:0", quality=97)), ((500, 700), dict(subsampling="4:4:4", quality=95)),
                           ((640, 640), dict(subsampling="4:2:0", quality=96, restart_marker_rows=4))]:
            b = io.BytesIO()
            noisy = np.clip(synth_image(rng, h, w).astype(np.int16) + rng.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(noisy).save(b, "JPEG", **kw)
            out.append(b.getvalue())
    return out


def sos_offset(d):
    i = d.find(b"\xff\xda")
    return i if i > 0 else len(d) // 2


def mutate(rng, data, others):
    d = bytearray(data)
    body = sos_offset(d) + 14
    # nine times in ten the damage is in the entropy-coded segment (what only the device looks at), else anywhere
    lo = body if rng.random() < 0.9 and body < len(d) - 2 else 2
    kind = rng.integers(0, 7)
    if kind == 0:
        del d[rng.integers(lo, len(d)):]
        if rng.random() < 0.5:
            d += b"\xff\xd9"
    elif kind == 1:
        for _ in range(rng.integers(1, 8)):
            i = rng.integers(lo, len(d)); d[i] ^= 1 << rng.integers(0, 8)
    elif kind == 2:
        i = rng.integers(lo, len(d)); n = min(len(d) - i, int(rng.integers(1, 64)))
        d[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 3:
        i = rng.integers(lo, len(d)); n = min(len(d) - i, int(rng.integers(1, 40)))
        d[i:i + n] = bytes([0xFF if rng.integers(0, 2) else 0x00]) * n
    elif kind == 4:   # a stray marker (RSTn, EOI, DNL ...) in the middle of the segment
        i = rng.integers(lo, len(d))
        d[i:i + 2] = bytes([0xFF, int(rng.choice([0xD0, 0xD3, 0xD7, 0xD9, 0xDC, 0xC4, 0x01, 0xFF]))])
    elif kind == 5:   # the entropy-coded segment of another stream behind this one's headers
        o = others[rng.integers(0, len(others))]
        d[body:] = o[sos_offset(o) + 14:]
    else:             # delete a run (every later restart marker arrives early)
        i = rng.integers(lo, len(d)); n = min(len(d) - i, int(rng.integers(1, 200)))
        del d[i:i + n]
    return bytes(d)


def containers(base):
    """Indexed JPEG containers (tools/jpeg2idx.py) of the seeds that can have one."""
    import ctypes as C
    from dali_amd import _capi as capi
    host = capi.host()
    out = []
    for e in base:
        data = np.frombuffer(e, np.uint8)
        n = C.c_size_t(0)
        if host.daliamdJpegIndexedBuild(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), None, C.c_size_t(0), C.byref(n)) != 0:
            continue
        box = np.zeros(n.value, np.uint8)
        if host.daliamdJpegIndexedBuild(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), box.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(box.size), C.byref(n)) == 0:
            out.append(box.tobytes())
    return out


def mutate_container(rng, box):
    """Damage anywhere in a container: its own header, the JPEG headers it carries, the clean stream, the slice entries (ordinals,
    entry positions, block indices, DC predictors), the entry's header (lengths, block-start count); truncation."""
    d = bytearray(box)
    hlen = int.from_bytes(d[8:12], "little")
    idx = 64 + ((hlen + 63) & ~63)
    ecs_len = int.from_bytes(d[12:16], "little")
    entries = idx + 64 + ((ecs_len + 256 + 63) & ~63)
    kind = rng.integers(0, 7)
    if kind == 0:     # bit flips in the slice entries
        for _ in range(rng.integers(1, 12)):
            i = rng.integers(min(entries, len(d) - 1), len(d)); d[i] ^= 1 << rng.integers(0, 8)
    elif kind == 1:   # the entry's header: clean length / block starts / slices
        i = idx + 4 * int(rng.integers(0, 3))
        d[i:i + 4] = int(rng.integers(0, 1 << 31)).to_bytes(4, "little") if rng.random() < 0.5 else \
            int(max(0, int.from_bytes(d[i:i + 4], "little") + int(rng.integers(-3, 4)))).to_bytes(4, "little")
    elif kind == 2:   # the clean stream
        for _ in range(rng.integers(1, 10)):
            i = rng.integers(idx + 64, max(idx + 65, entries)); d[min(i, len(d) - 1)] ^= 1 << rng.integers(0, 8)
    elif kind == 3:   # ordinals shifted / entries overwritten wholesale
        i = entries + 12 * int(rng.integers(0, max(1, (len(d) - entries) // 12)))
        n = min(len(d) - i, 12 * int(rng.integers(1, 6)))
        d[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 4:   # the container's own header or the JPEG headers inside it
        i = rng.integers(0, idx); d[i] ^= 1 << rng.integers(0, 8)
    elif kind == 5:
        del d[rng.integers(16, len(d)):]
    else:             # entries of another position of the same file (plausible but wrong)
        a = entries + 12 * int(rng.integers(0, max(1, (len(d) - entries) // 12 - 2)))
        b = entries + 12 * int(rng.integers(0, max(1, (len(d) - entries) // 12 - 2)))
        d[a:a + 12], d[b:b + 12] = d[b:b + 12], d[a:a + 12]
    return bytes(d)


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from tests import hipemu_env
    hipemu_env.activate("address")
    from dali_amd import backend as B
    from dali_amd._capi import DaliAmdError
    rng = np.random.default_rng(seed)
    base = seeds(rng)
    for e in base:   # the seeds themselves decode
        B.decode_jpeg_batch([np.frombuffer(e, np.uint8)], device="cuda", huffman="gpu")
    ok = rejected = on_device = 0
    # every third batch goes through the PRODUCT pipeline instead (decoders.image(device="mixed") behind an external
    # source: the operator's header-only parse and the upload of "everything behind SOS", the end of the scan found on
    # the device, the status word read back when the outputs are requested)
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipes = {}

    def through_pipeline(batch):
        n = len(batch)
        if n not in pipes:
            pipe = Pipeline(batch_size=n, num_threads=2, device_id=0, prefetch_queue_depth=1, exec_async=False, exec_pipelined=False)
            with pipe:
                pipe.set_outputs(fn.decoders.image(fn.external_source(name="enc"), device="mixed"))
            pipe.build()
            pipes[n] = pipe
        pipes[n].feed_input("enc", [np.frombuffer(b, np.uint8) for b in batch])
        try:
            pipes[n].run()
        except RuntimeError:
            del pipes[n]    # an iteration that failed leaves the pipeline unusable, as in the reference
            raise

    def hung(*_):
        raise SystemExit(f"fuzz_gpu_decoder: a decode did not come back within 120 s (iteration {it}, seed {seed})")
    signal.signal(signal.SIGALRM, hung)
    boxes = containers(base) if os.environ.get("FUZZ_CONTAINERS") == "1" else []
    if os.environ.get("FUZZ_CONTAINERS") == "1":
        assert len(boxes) >= 6
        through_pipeline(boxes[:3])     # the sound containers decode
    for it in range(iterations):
        if boxes:
            # FUZZ_CONTAINERS=1: indexed containers with damage anywhere - the index entries reach the DEVICE as they are, so a
            # forged entry must end in wrong pixels or an error, never outside the decode's buffers
            batch = [mutate_container(rng, boxes[rng.integers(0, len(boxes))]) for _ in range(int(rng.integers(1, 4)))]
            if rng.random() < 0.3:
                batch.insert(int(rng.integers(0, len(batch) + 1)), boxes[rng.integers(0, len(boxes))])
            signal.alarm(120)
            try:
                through_pipeline(batch)
                ok += 1
            except (DaliAmdError, RuntimeError):
                rejected += 1
            finally:
                signal.alarm(0)
            if (it + 1) % 50 == 0:
                print(f"{it + 1}: {ok} decoded, {rejected} rejected", flush=True)
            continue
        batch = [mutate(rng, base[rng.integers(0, len(base))], base) for _ in range(int(rng.integers(1, 4)))]
        if rng.random() < 0.3:   # a sound stream next to the damaged ones
            batch.insert(int(rng.integers(0, len(batch) + 1)), base[rng.integers(0, len(base))])
        signal.alarm(120)
        try:
            try:
                plan = B.JpegBatchPlan([np.frombuffer(b, np.uint8) for b in batch]) if hasattr(B, "JpegBatchPlan") else None
                if plan is not None:
                    on_device += int(np.sum(plan.analyze_scans()))
            except DaliAmdError:
                pass
            if it % 3 == 2:
                through_pipeline(batch)
            else:
                B.decode_jpeg_batch([np.frombuffer(b, np.uint8) for b in batch], device="cuda", huffman="gpu")
            ok += 1
        except (DaliAmdError, RuntimeError):
            rejected += 1
        finally:
            signal.alarm(0)
        if (it + 1) % 50 == 0:
            print(f"{it + 1}: {ok} decoded, {rejected} rejected", flush=True)
    print(f"fuzz_gpu_decoder OK: {iterations} batches, {ok} decoded, {rejected} rejected, {on_device} streams took the device path")


if __name__ == "__main__":
    main()
