#!/usr/bin/env python
"""Headline benchmark: images/s on the JPEG -> RandomResizedCrop -> CropMirrorNormalize pipeline
(224x224, batch 256 per GPU, fp16 CHW output) on N MI355X, plus the roofline of the dominant
kernel and the CPU baseline (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Data set: 1024 synthetic ImageNet-like JPEGs per GPU = 4 distinct batches of 256 which the steps
rotate through (the entropy decoder's time depends on the content, one batch would be one sample of it).
One "step" = one iteration of the PRODUCT pipeline (dali_amd.Pipeline: the C++ executor, its stage threads, its ring of
HIP streams) over one batch whose inputs are already resident in HBM:
  readers.file(skip_cached_images=True)          the files are resident: the reader emits empty samples (loader.h:466-480)
  decoders.image(device="mixed", cache_type="encoded")
      JPEG entropy-coded segments in HBM  ->  un-stuff + Huffman decode + fused dequant / IDCT   [6 kernels]
                                          ->  chroma upsample + YCbCr->RGB                        [JpegColorKernel]
  random_resized_crop -> crop_mirror_normalize(mirror=coin_flip)      host Philox windows + mirror bits, ONE fused
                                                                      resample + CMN launch        [ResampleKernel]
Everything a step costs is inside the timed region: the stage threads' descriptor construction and upload, the random
numbers, every launch, and the Python call that hands the batch out.  What is NOT in it is what the metric excludes by
definition (inputs resident in HBM): the file reads, header parse / scan analysis and the H2D of the JPEG bytes happen in
the set-up epoch that makes the data set resident; the end-to-end figures WITH them are the `e2e_pipeline*` entries.
`--driver python` runs the round-1/2 bench instead (kernel library driven from Python, dali_amd/backend.py): kept for the
kernel experiments (`--huffman host`, `--no-fused-idct`), never the headline.

Multi-GPU: sample sharding exactly like readers.file(shard_id, num_shards): rank r owns the contiguous shard
[r*1024, (r+1)*1024) of a world*1024-image data set (loader.cc:78-87); no collective on the data path ("scaling":
"weak").  For N > 1 the line also carries `e2e_pipeline_sharded`: BASELINE configs[4] through dali_amd.Pipeline on
every rank (readers.file(shard_id=rank, num_shards=N) over the shared data set directory, batch 512 per GPU, host
threads pinned to the GPU's NUMA node), timed with the same barrier / max-over-ranks rule.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

# multi-process GPU work on this pool needs dmabuf IPC (RCCL's communicator set-up fails with `hipIpcGetMemHandle: invalid
# argument` otherwise); the driver exports it - kept here for a launch from a bare shell
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling


def effective_cpu_count():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


# Pipeline(set_affinity=True) - the reference's argument (pipeline.py: "set_affinity: set CPU affinity mask to the closest
# one to the GPU"): the executor's threads and the first touch of their page-locked buffers stay on the GPU's NUMA node.
# On the two-socket bench host: 20-step `value` 430 000 against 416 000, end to end from files 355-387 000 against
# 319-364 000 (gpurun_out/r04o, r04p).  BENCH_AFFINITY=0 switches it off.
AFFINITY = os.environ.get("BENCH_AFFINITY", "1") == "1"


def device_local_cpus(device_index):
    """CPUs of the GPU's NUMA node that this process may use (what Pipeline(set_affinity=True) binds to); [] when unknown."""
    try:
        import ctypes as C
        from dali_amd import _capi as capi
        buf = C.create_string_buffer(64)
        if capi.kernels().daliamdDevicePciBusId(int(device_index), buf, 64) != 0:
            return []
        cpus = set()
        for part in open(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return sorted(cpus & set(os.sched_getaffinity(0)))
    except (OSError, ValueError, AttributeError):
        return []


def measured_traffic(kernel, workload=""):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, produced by
    tools/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH_SIZE correction).  bench.py
    cannot run rocprofv3 around itself, so this is the figure of the profiled run of this same command."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", f"*{workload}_traffic.json"))
                   if workload or not any(w in os.path.basename(f) for w in ("heavy_aug", "audio", "indexed", "normalize", "fused_colour")))
    if not files:
        return None, None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        def norm(n):
            # rocprofv3 reports the instantiation, the KernelTimer scopes of the library a role: SpectrogramFastKernel<9, 4, 0>
            # is the SpectrogramKernel launch, <9, 4, 1 / 2> (MEL != 0) the fused SpectrogramMel(Mfma)Kernel launch
            base, _, targs = n.partition("<")
            if base == "SpectrogramFastKernel":
                mel = targs.rstrip(">").split(",")[2:3]
                return "SpectrogramMelKernel" if mel and mel[0].strip() != "0" else "SpectrogramKernel"
            return base.replace("MelMfma", "Mel")
        k = ks.get(kernel) or next((v for n, v in ks.items() if norm(n) == norm(kernel)), {})
        return k.get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def measured_copy_ceiling(device, mib=1024, iters=10):
    """Device-to-device copy of `mib` MiB (hipMemcpyDtoD through torch): read + write bytes per second.  The honest
    denominator next to the 8 TB/s vendor figure."""
    import torch
    src = torch.empty(mib << 20, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 2.0 * (mib << 20) * iters / (a.elapsed_time(b) * 1e-3) / 1e9


def measured_h2d_ceiling(device, mib=256, iters=4):
    """Host->device rate of this box (GB/s): pinned source, one stream, 256 MiB copies - the ceiling of anything that has
    to cross the bus once per step."""
    import torch
    src = torch.empty(mib * 2**20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(mib * 2**20, dtype=torch.uint8, device=device)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):      # the best of three trains (one disturbed train made a leg look faster than the bus: frac 1.3)
        t0 = time.perf_counter()
        for _ in range(iters):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, iters * mib * 2**20 / (time.perf_counter() - t0) / 1e9)
    del src, dst
    return best


def make_dataset(first_index, count, workers=0, variant="baseline"):
    """Synthetic ImageNet-like JPEGs.  Image i depends only on its global index.  Forks generator
    processes: call before torch / the HIP runtime are initialised."""
    from dali_amd.testing import synth_dataset
    return synth_dataset(first_index, count, seed=1234, workers=workers, variant=variant)


class HotPath:
    """One device-resident batch + the per-step launch sequence (all launches on `stream`)."""

    def __init__(self, enc, device, stream, seed=1234, huffman="gpu", fused_idct=True, first_iteration=0):
        import torch
        from dali_amd import backend as B
        self.torch, self.B = torch, B
        self.device, self.stream = device, stream
        self.n = len(enc)
        self.huffman = huffman
        self.fused_idct = fused_idct and huffman == "gpu"   # the entropy decoder writes the planes itself
        self.plan = B.JpegBatchPlan(enc, out_pitch_align=16)
        self.coef_host = torch.empty(self.plan.coef_elems, dtype=torch.int16, pin_memory=True)
        t0 = time.perf_counter()
        self.plan.entropy_decode(self.coef_host, num_threads=effective_cpu_count())
        self.huffman_s = time.perf_counter() - t0
        if huffman == "gpu":
            if not self.plan.analyze_scans().all():
                raise SystemExit("bench: the synthetic batch must be baseline single-scan JPEG")
            self.plan.upload_streams(device)
            self.coef = torch.empty(self.plan.coef_elems, dtype=torch.int16, device=device)
            # one-time self check: the GPU entropy decoder reproduces the host decoder's coefficients exactly
            status = self.plan.run_gpu_huffman(self.coef)
            torch.cuda.synchronize()
            self.plan.check_gpu_status(status)
            if not torch.equal(self.coef.cpu(), self.coef_host) and not os.environ.get("BENCH_SKIP_SELF_CHECK"):
                raise SystemExit("bench: GPU Huffman output differs from the host entropy decoder")
            self.symbols = B.count_huffman_symbols(self.coef, self.plan.coef_elems)   # exact, for the symbol rate
            if self.fused_idct:   # ... and its fused dequantisation + IDCT the stand-alone IDCT kernel's planes
                ref_planes = torch.zeros(self.plan.plane_bytes, dtype=torch.uint8, device=device)
                got_planes = torch.zeros_like(ref_planes)
                scratch_rgb = torch.empty(self.plan.out_bytes, dtype=torch.uint8, device=device)
                B.jpeg_gpu_stage(self.plan, self.coef, ref_planes, scratch_rgb)
                status = self.plan.run_gpu_huffman(None, planes_dev=got_planes)
                torch.cuda.synchronize()
                self.plan.check_gpu_status(status)
                if not torch.equal(ref_planes, got_planes) and not os.environ.get("BENCH_SKIP_SELF_CHECK"):
                    raise SystemExit("bench: fused Huffman+IDCT planes differ from the IDCT kernel's")
                del ref_planes, got_planes, scratch_rgb
        else:
            self.coef = self.coef_host.to(device)
        self.coef_host = None
        self.planes = torch.empty(self.plan.plane_bytes, dtype=torch.uint8, device=device)
        self.rgb = torch.empty(self.plan.out_bytes, dtype=torch.uint8, device=device)
        self.out = torch.empty((self.n, 3, 224, 224), dtype=torch.float16, device=device)
        self.ws = self.plan.new_huffman_workspace(device) if huffman == "gpu" else None
        self.views = self.plan.output_views(self.rgb)
        self.image_table = B.ImageTable(self.views)
        self.host_s = 0.0
        self.kernel_events = []
        self.shapes = np.array([v.shape[:2] for v in self.views], np.int32)
        # one RandomResizedCrop / CoinFlip operator pair runs over the whole data set: the generator of a batch is the
        # master advanced by the batches before it (OperatorWithRng::Advance)
        self.rrc_master = B.philox_state(seed)
        self.flip_master = B.philox_state(seed + 1)
        self.rrc_master.ctr[1] += first_iteration * self.n
        self.flip_master.ctr[1] += first_iteration * self.n
        self.mean, self.inv_std = B.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255],
                                                  [0.229 * 255, 0.224 * 255, 0.225 * 255])
        self.last_rois = None
        # algorithmic bytes per launch
        P = sum(int(s[0]) * int(s[1]) for s in self.shapes)
        self.pixels = P
        self.bytes_idct = 3 * self.plan.coef_elems            # 2 B coefficient in + 1 B sample out
        self.bytes_color = self.plan.plane_bytes + 3 * P       # planes in + RGB out

    def step(self, record=None, kernel_events=None, advance=1):
        """Enqueues one pass of the hot path over the batch.  `advance`: iterations of the operator pair between two
        passes over THIS batch (= the number of distinct batches the steps rotate through)."""
        torch, B = self.torch, self.B
        from dali_amd import _capi as capi
        ev = record
        t0 = time.perf_counter()
        with B.use_stream(self.stream):
            if self.huffman == "gpu":
                self.plan.run_gpu_huffman(self.coef, events=ev[5:7] if ev else None, ws=self.ws,
                                          kernel_events=kernel_events.handles if kernel_events else None,
                                          planes_dev=self.planes if self.fused_idct else None)
            B.jpeg_gpu_stage(self.plan, self.coef, self.planes, self.rgb, split_events=ev[1:2] if ev else None,
                             start_event=ev[0] if ev else None, fused_huffman=self.fused_idct)
            anchors, crops = B.random_crop_batch(self.rrc_master, self.shapes)
            mirror = B.coin_flip_batch(self.flip_master, self.n, 0.5)
            self.rrc_master.ctr[1] += advance * self.n   # OperatorWithRng::Advance
            self.flip_master.ctr[1] += advance * self.n
            rois = np.concatenate([anchors, anchors + crops], 1).astype(np.float32)
            self.last_rois = (anchors, crops)
            if ev:
                ev[2].record()
            B.resample_batch(self.image_table, (224, 224), rois=rois, out_dtype=capi.FLOAT16, out_layout=capi.LAYOUT_CHW,
                             mean=self.mean, inv_std=self.inv_std, mirror=mirror, out=self.out,
                             start_event=ev[3] if ev else None)
            if ev:
                ev[4].record()
        self.host_s += time.perf_counter() - t0
        return anchors, crops

    def resample_bytes(self, crops):
        return int(3 * (crops[:, 0].astype(np.int64) * crops[:, 1]).sum() + 6 * 224 * 224 * self.n)


def cpu_baseline(enc, seconds_budget=20.0):
    """The oracle (CPU restatement of DALI's CPU backend: decode + RRC + CMN) on the same workload, one task per
    sample on an OpenMP team spanning all usable host cores (resize_op_impl_cpu.h:84-107).  Bounded sample: whole
    passes over the batch until ~`seconds_budget` core-seconds of CPU work have been spent."""
    from oracle import oracle as O
    cores = effective_cpu_count()
    mean, inv = O.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255])
    O.pipeline_batch(enc[:min(len(enc), 2 * cores)], 1234, 1235, 0, mean=mean, inv_std=inv, nthreads=cores)  # warm-up
    done, it = 0, 0
    t0 = time.perf_counter()
    while True:
        O.pipeline_batch(enc, 1234, 1235, it, mean=mean, inv_std=inv, nthreads=cores)
        done += len(enc)
        it += 1
        el = time.perf_counter() - t0
        if el * cores >= seconds_budget or el > 30:
            break
    return {"value": done / el, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{done} images = {it} pass(es) over the {len(enc)}-image batch; decode+RRC+CMN per image on the C "
                      f"oracle (-O3 -msse2, OpenMP, one task per sample), {cores} threads, {el:.2f} s wall"}


def pillow_baseline(enc, seconds_budget=20.0):
    """Secondary, informational CPU baseline: Pillow / libjpeg-turbo (SIMD IDCT and
    colour conversion) decode + Image.resize(box=crop window, BILINEAR) + numpy CropMirrorNormalize, one task per
    sample in a ThreadPoolExecutor over all usable host cores (Pillow releases the GIL in decode and resize).  Same
    crop windows and mirror bits as the product (oracle Philox); the resize filter is Pillow's, not DALI's - this
    is a speed reference of a tuned CPU decoder, the parity reference is `cpu_baseline`."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from oracle import oracle as O
    cores = effective_cpu_count()
    mean = np.array([0.485 * 255, 0.456 * 255, 0.406 * 255], np.float32)
    inv = (1.0 / np.array([0.229 * 255, 0.224 * 255, 0.225 * 255])).astype(np.float32)
    shapes = []
    for e in enc:
        w, h = Image.open(io.BytesIO(e)).size
        shapes.append((h, w))

    def one(job):
        e, (ay, ax), (ch, cw), flip = job
        im = Image.open(io.BytesIO(e)).convert("RGB")
        r = im.resize((224, 224), Image.BILINEAR, box=(ax, ay, ax + cw, ay + ch))
        a = np.asarray(r, np.float32)
        if flip:
            a = a[:, ::-1]
        return ((a - mean) * inv).transpose(2, 0, 1).astype(np.float16)

    done, it = 0, 0
    with ThreadPoolExecutor(cores) as pool:
        t0 = time.perf_counter()
        while True:
            anchors, crops = O.rrc_batch(1234, it, shapes)
            flips = O.coin_flip_batch(1235, it, len(enc), 0.5)
            out = list(pool.map(one, zip(enc, anchors, crops, flips)))
            assert out[0].shape == (3, 224, 224)
            done += len(enc)
            it += 1
            el = time.perf_counter() - t0
            if el * cores >= seconds_budget or el > 30:
                break
    from PIL import features
    return {"value": done / el, "unit": "images/s", "cores": cores, "kind": "pillow",
            "libjpeg_turbo": features.version("libjpeg_turbo") or features.version("jpg"),
            "sample": f"{done} images = {it} pass(es) over the first {len(enc)}-image batch; Pillow decode + resize(box, "
                      f"BILINEAR) + numpy CMN, ThreadPoolExecutor({cores}), {el:.2f} s wall"}


class quiet_gc:
    """Timed regions run with the cyclic garbage collector off (collected right before): a generation-2 pass over this
    process's objects takes milliseconds on the MAIN thread - the consumer -, which drains the five batches in flight and leaves
    the GPU idle; inside the driver's 20-step region (9 ms) one such pause halved a run's `value` (306 k among 547-567 k,
    gpurun_out/r06_five).  timeit does the same."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        if os.environ.get("BENCH_QUIET_GC", "1") == "0":
            return
        # (no collection HERE: it takes tens of milliseconds during which the device idles and clocks down - 3 % of a 20-step
        # region, measured A/B on one box; the legs collect before their warm-up iterations instead)
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def thread_cpu_seconds():
    """CPU seconds (user + system) every thread of this process has used so far, summed by thread name with the
    trailing index stripped (the product names its threads: dali-rd<i> = file reader, dali-cpupool<i> / dali-devpool<i> =
    the two operator thread pools, dali-cpustage / dali-devstage = the executor's stage threads)."""
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/comm") as f:
                name = f.read().strip().rstrip("0123456789")
            if not name.startswith("dali-"):
                # the interpreter's main thread, or a thread somebody else made (HIP / ROCr runtime helpers, torch pools)
                name = "main" if int(tid) == os.getpid() else f"other({name})"
            with open(f"/proc/self/task/{tid}/schedstat") as f:
                ns = int(f.read().split()[0])          # time spent on a CPU, nanoseconds
        except (OSError, ValueError, IndexError):
            continue
        out[name] = out.get(name, 0.0) + ns * 1e-9
    return out


def write_dataset(root, enc, first_index=0):
    """Data set directory for readers.file: root/<class 0..9>/img_<global index>.jpg (sorted order = index order
    inside a class; labels = class directory)."""
    for c in range(10):
        os.makedirs(os.path.join(root, f"{c:02d}"), exist_ok=True)
    for i, e in enumerate(enc):
        g = first_index + i
        with open(os.path.join(root, f"{g % 10:02d}", f"img_{g:07d}.jpg"), "wb") as f:
            f.write(e)
    # the files' dirty pages go to the disk NOW, not when the kernel's flusher wakes up in the middle of a 9 ms timed region
    # (the legs that follow a freshly written data set showed one-off dips of 10-25 %, gpurun_out/r06_final3)
    os.sync()


def e2e_pipeline(root, batch, device_id, iters=400, threads=None, roi_decode=False, cache_mb=0, shard_id=0, num_shards=1,
                 depth=5, sync=None, set_affinity=False, reader_depth=2, index_path=None):
    """The same hot path through the product's DALI-style pipeline (C++ host framework): readers.file (page cache)
    -> decoders.image(mixed: header parse + scan analysis on the host thread pool, H2D of the entropy-coded
    segments, GPU Huffman/IDCT/colour) -> random_resized_crop + crop_mirror_normalize (fused).  PCIe-inclusive and
    host-inclusive, therefore reported next to `value`, never as `value`.  `sync`: barrier callable bracketing the
    timed region (multi-rank runs); returns this rank's elapsed seconds in "elapsed_s"."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    # the executor runs two thread pools of this size side by side (host stage | device stage): three quarters of the
    # usable cores each keeps their sum near the CPU quota instead of far above it (16 -> 12: +25 % on the bench box)
    threads = threads or max(2, effective_cpu_count() * 3 // 4)
    # four sequential stages (file reads | parse + staging | H2D | kernels) need several batches in flight to overlap
    set_affinity = set_affinity or AFFINITY
    pipe = Pipeline(batch_size=batch, num_threads=threads, device_id=device_id, seed=1234, prefetch_queue_depth=depth,
                    set_affinity=set_affinity)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=root, name="Reader", shard_id=shard_id, num_shards=num_shards,
                                        prefetch_queue_depth=reader_depth, **({"index_path": index_path} if index_path else {}))
        if roi_decode:   # the variant NVIDIA's own benchmark uses (hw_decoder_bench.py:178-188): ROI decode + resize
            images = fn.decoders.image_random_crop(jpegs, device="mixed", output_type=types.RGB)
            crops = fn.resize(images, size=[224, 224])
        else:
            cache = dict(cache_size=cache_mb, cache_type="threshold") if cache_mb else {}
            images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB, **cache)
            crops = fn.random_resized_crop(images, size=[224, 224])
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW",
                                       mean=[0.485 * 255, 0.456 * 255, 0.406 * 255],
                                       std=[0.229 * 255, 0.224 * 255, 0.225 * 255],
                                       mirror=fn.random.coin_flip(probability=0.5))
        pipe.set_outputs(out, labels)
    pipe.build()
    # every ring slot allocates its pinned / device buffers on first use and grows them until it has seen the largest
    # batch of the data set: (ring slots) x (distinct batches) iterations before the clock starts
    for _ in range(max(24, (depth + 1) * 5)):
        pipe.run()
    if sync:
        sync()
    pipe.operator_host_times()                        # opens the host-time window
    cpu0 = thread_cpu_seconds()
    with quiet_gc():
        t0 = time.perf_counter()
        for _ in range(iters):
            pipe.run()
        pipe._backend.wait_enqueued()                      # (see run_resident_pipeline: the clock stops when all of them are done)
        if sync:
            sync()
        el = time.perf_counter() - t0
    cpu1 = thread_cpu_seconds()
    host_times = pipe.operator_host_times()
    cpu = {k: round(1e3 * (v - cpu0.get(k, 0.0)) / iters, 3) for k, v in cpu1.items() if v - cpu0.get(k, 0.0) > 0}
    return {"value": iters * batch / el, "unit": "images/s", "ms_per_batch": 1e3 * el / iters, "elapsed_s": el,
            "batch": batch, "iters": iters, "host_threads": threads, "prefetch_queue_depth": depth,
            "reader_prefetch_queue_depth": reader_depth,
            "host_ms_per_operator": {k: round(v, 4) for k, v in host_times.items() if not k.startswith("<")},
            "host_stage_ms_per_batch": host_times.get("<host stage>"),
            "device_stage_ms_per_batch": host_times.get("<device stage>"),
            "slot_wait_ms_per_batch": host_times.get("<slot wait>"),
            "cpu_ms_per_batch_by_thread_group": cpu, "cpu_ms_per_batch": round(sum(cpu.values()), 3),
            "cpus_busy": round(sum(cpu.values()) / (1e3 * el / iters), 2),
            "shard_id": shard_id, "num_shards": num_shards, "kernels": pipe.executed_kernels(),
            # readers.file handed out its registered file mappings and the device fetched the bytes itself (round 5; the
            # reader's default: only when the process may use fewer than four CPUs - DESIGN.md section 4)
            "reader_zero_copy": "gather_encoded" in pipe.executed_kernels(),
            "note": "dali_amd.Pipeline end to end from encoded files in the page cache (file read, header parse, "
                    "H2D of the JPEG bytes on a copy stream, all device stages, fp16 CHW batch on the device)"}


def batch_statistics(enc_batches, device, count_symbols=True):
    """Per distinct batch: what the algorithmic-byte formulas need - entropy-coded bytes,
    coefficient count, pixels, image shapes - from the product's own parser; Huffman symbols counted exactly from the
    host decoder's coefficients (set-up only)."""
    import torch
    from dali_amd import backend as B
    stats = []
    for enc in enc_batches:
        plan = B.JpegBatchPlan(enc, out_pitch_align=16)
        if not plan.analyze_scans().all():
            raise SystemExit("bench: the synthetic batch must be baseline single-scan JPEG")
        shapes = np.array([(int(plan.inf["height"][i]), int(plan.inf["width"][i])) for i in range(plan.n)], np.int32)
        st = {"stream_bytes": float(plan.scan["ecs_length"].sum()), "coef_elems": float(plan.coef_elems),
              "pixels": float((shapes[:, 0].astype(np.int64) * shapes[:, 1]).sum()), "shapes": shapes, "symbols": None}
        if count_symbols:
            coef = torch.empty(plan.coef_elems, dtype=torch.int16, pin_memory=True)
            plan.entropy_decode(coef, num_threads=effective_cpu_count())
            st["symbols"] = float(B.count_huffman_symbols(coef.to(device), plan.coef_elems))
            del coef
        stats.append(st)
    return stats


def resident_pipeline(root, batch, device_id, depth, threads, shard_id=0, num_shards=1, cache_mb=4096, roi_decode=False,
                      crop_seed=None, flip_seed=None, roi_fusion=True, cache_type="encoded", random_shuffle=False):
    """The headline pipeline: configs[1] with the data set resident in HBM as encoded streams.  roi_decode: the fused
    variant decoders.image_random_crop -> resize -> crop_mirror_normalize (only the crop window is dequantised,
    transformed and colour-converted), on the same resident streams.  crop_seed / flip_seed: explicit operator seeds
    (tests/test_gpu_headline.py holds exactly this graph to the oracle); default: from the pipeline's seed.
    cache_type="indexed": the streams are resident WITH the side information their first decode found (un-stuffed bytes +
    the decoder state in front of every 256-byte slice): epochs >= 2 skip the position passes' relaxation."""
    seeded = lambda seed: {} if seed is None else {"seed": seed}  # noqa: E731
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=threads, device_id=device_id, seed=1234, prefetch_queue_depth=depth,
                    set_affinity=AFFINITY)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=root, name="Reader", shard_id=shard_id, num_shards=num_shards,
                                        stick_to_shard=True, skip_cached_images=True, random_shuffle=random_shuffle,
                                        **({"initial_fill": 4096} if random_shuffle else {}))
        if roi_decode:
            images = fn.decoders.image_random_crop(jpegs, device="mixed", output_type=types.RGB, cache_size=cache_mb,
                                                   cache_type=cache_type)
            crops = fn.resize(images, size=[224, 224])
        else:
            images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB, cache_size=cache_mb, cache_type=cache_type)
            crops = fn.random_resized_crop(images, size=[224, 224], **seeded(crop_seed))
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW",
                                       mean=[0.485 * 255, 0.456 * 255, 0.406 * 255],
                                       std=[0.229 * 255, 0.224 * 255, 0.225 * 255],
                                       mirror=fn.random.coin_flip(probability=0.5, **seeded(flip_seed)))
        pipe.set_outputs(out, labels)
    # roi_fusion=False: the executor's graph-level fusion "a decoders.image that feeds only a random_resized_crop decodes the
    # windows that operator draws" is switched off for this pipeline (read when the pipeline is built)
    saved = os.environ.get("DALI_AMD_ROI_FUSION")
    if not roi_fusion:
        os.environ["DALI_AMD_ROI_FUSION"] = "0"
    try:
        pipe.build()
    finally:
        if not roi_fusion:
            if saved is None:
                os.environ.pop("DALI_AMD_ROI_FUSION", None)
            else:
                os.environ["DALI_AMD_ROI_FUSION"] = saved
    return pipe


def generate_variant_dataset(variant, root, count):
    """Writes `count` files of a data set variant under `root` (write_dataset layout) from a CHILD process - the generator forks
    workers, which a process with the GPU runtime up must not, and its memory goes away with it - and returns the files' sizes."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "enc = bench.make_dataset(0, %d, workers=bench.effective_cpu_count(), variant=%r); bench.write_dataset(%r, enc); "
            "print(json.dumps([len(e) for e in enc]))" % (ROOT, count, variant, root))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(f"generating the {variant} data set failed: {out.stderr[-300:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])


def write_linked_dataset(root, src_root, enc, copies):
    """`copies` hard links of every file of the data set under `src_root` (write_dataset layout): a data set of
    copies * len(enc) FILES - every one its own resident stream in the encoded cache, which is keyed by path - that costs
    the file system and the page cache one copy."""
    for c in range(10):
        os.makedirs(os.path.join(root, f"{c:02d}"), exist_ok=True)
    for i in range(len(enc)):
        src = os.path.join(src_root, f"{i % 10:02d}", f"img_{i:07d}.jpg")
        for k in range(copies):
            g = k * len(enc) + i
            os.link(src, os.path.join(root, f"{g % 10:02d}", f"img_{g:07d}.jpg"))


def resident_variant_leg(args, root, n_files, file_bytes, dev_index, steps, random_shuffle=False, cache_type="encoded",
                         time_kernels=False):
    """The headline graph (bench.resident_pipeline) on another data set, timed like `value` on one GPU: set-up epochs until
    the encoded cache stops growing (streams it cannot keep - progressive, four-component - stay file-fed, as they would
    in production), then every ring slot warm, then `steps` iterations.  Returns {value, ms_per_step, cache stats}."""
    import torch
    from dali_amd import _backend
    B, depth = args.batch, max(1, args.inflight)
    threads = max(2, effective_cpu_count() * 3 // 4)
    per_epoch = max(1, n_files // B)
    pipe = resident_pipeline(root, B, dev_index, depth, threads, cache_mb=max(64, int(2.2 * file_bytes / 2**20)),
                             random_shuffle=random_shuffle, cache_type=cache_type)
    t_setup = time.perf_counter()
    before = _backend.encoded_cache_stats(dev_index)
    seen, done = -1, 0
    for _ in range(3):                                 # epochs until nothing new becomes resident
        for _ in range(per_epoch):
            pipe.run()
            done += 1
        now = _backend.encoded_cache_stats(dev_index)["streams"]
        if now == seen:
            break
        seen = now
    for _ in range((depth + 2) * min(per_epoch, 8) + args.warmup):
        pipe.run()
    pipe._backend.wait_enqueued()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    mid = _backend.encoded_cache_stats(dev_index)
    pipe.operator_host_times()
    if time_kernels:
        kernel_timing()
        kernel_timing(True)
    with quiet_gc():
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.run()
        pipe._backend.wait_enqueued()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    ktimes = None
    if time_kernels:
        kernel_timing(False)
        ktimes = {k: {"launches": c, "avg_ms": ms} for k, (c, ms) in kernel_timing().items()}
    host = pipe.operator_host_times()
    after = _backend.encoded_cache_stats(dev_index)
    out = {"value": B * steps / el, "unit": "images/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "files": n_files,
           "file_MB": file_bytes / 1e6, "resident_streams": after["streams"] - before["streams"],
           "resident_MB": (after["bytes_used"] - before["bytes_used"]) / 1e6,
           "hits_per_step": (after["hits"] - mid["hits"]) / steps, "misses_per_step": (after["misses"] - mid["misses"]) / steps,
           "random_shuffle": random_shuffle, "setup_s": t_setup, "setup_iterations": done,
           "host_stage_ms_per_step": host.get("<host stage>"), "device_stage_ms_per_step": host.get("<device stage>"),
           "host_ms_per_operator": {k: round(v, 4) for k, v in host.items() if not k.startswith("<")},
           "kernels": pipe.executed_kernels()}
    if ktimes:
        out["kernel_ms_in_schedule"] = ktimes
    del pipe
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def iterator_leg(args, root, enc_all, dev_index, steps):
    """The headline pipeline behind DALIGenericIterator (plugin/pytorch.py: every output is copied into a torch tensor of
    the consumer on a side stream and the copy has completed when __next__ returns): the rate a PyTorch training loop
    sees.  Same resident data set, same depth; one extra 77 MB device-to-device copy per batch."""
    import torch
    from dali_amd import _backend
    from dali_amd.plugin.pytorch import DALIGenericIterator
    B, nb = args.batch, max(1, args.batches)
    depth = max(1, args.inflight)
    threads = max(2, effective_cpu_count() * 3 // 4)
    pipe = resident_pipeline(root, B, dev_index, depth, threads, cache_mb=max(64, int(2 * sum(len(e) for e in enc_all) / 2**20)))
    it = DALIGenericIterator([pipe], ["data", "label"])
    done = 0
    while _backend.encoded_cache_stats(dev_index)["streams"] < nb * B and done < 64 * nb:
        next(it)
        done += 1
    for _ in range((depth + 2) * nb + args.warmup):
        next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = next(it)
    pipe._backend.wait_enqueued()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    shape = tuple(out[0]["data"].shape)
    del it, pipe, out
    return {"value": steps * B / el, "unit": "images/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "output": f"torch.float16 {shape} on the consumer's stream + labels",
            "note": "bench.resident_pipeline consumed through dali_amd.plugin.pytorch.DALIGenericIterator"}


def run_resident_pipeline(args, root, enc_all, device, dev_index, rank, world, local_world, barrier, dist):
    """Times K iterations of the product pipeline on the resident data set.  Returns what the JSON line is built from."""
    import torch
    from dali_amd import _backend, backend as Bk
    B, nb = args.batch, max(1, args.batches)
    per_rank = nb * B
    depth = max(1, args.inflight)
    threads = max(2, effective_cpu_count() // max(1, local_world) * 3 // 4)
    stats = batch_statistics([enc_all[b * B:(b + 1) * B] for b in range(nb)], device)
    # (one cache per device AND process: twice the shard's file bytes is room to spare for its entropy-coded segments)
    pipe = resident_pipeline(root, B, dev_index, depth, threads, shard_id=rank, num_shards=world,
                             cache_mb=max(64, int(2.2 * sum(len(e) for e in enc_all) / 2**20)), cache_type=args.cache_type)
    # ---- set-up, untimed: epochs until the shard is resident, then until the reader (which runs ahead of the decoder)
    # has stopped reading files and every ring slot has seen every distinct batch
    done = 0
    t_setup = time.perf_counter()
    while _backend.encoded_cache_stats(dev_index)["streams"] < per_rank and done < 64 * nb:
        pipe.run()
        done += 1
    for _ in range((depth + 2) * nb):
        pipe.run()
        done += 1
    before = _backend.encoded_cache_stats(dev_index)
    t_setup = time.perf_counter() - t_setup
    import gc
    gc.collect()                                      # (before the warm-up, not between it and the clock: see quiet_gc)
    kernel_timing(12 * (args.steps + depth + 2))      # events for every launch of the timed region, created now
    for _ in range(args.warmup):
        pipe.run()
        done += 1
    kernel_timing(False)
    kernel_timing()                                   # (drop whatever the set-up recorded)
    barrier()
    pipe.operator_host_times()                        # opens the host-time window
    kernel_timing(True)
    stamps = []
    with quiet_gc():
        t0 = time.perf_counter()
        if os.environ.get("BENCH_STEP_TIMES") == "1":     # debugging: when each step of the timed region returned
            for _ in range(args.steps):
                pipe.run()
                stamps.append(time.perf_counter() - t0)
        else:
            for _ in range(args.steps):
                pipe.run()
        # K iterations were scheduled since t0 (the `depth` batches handed out first were complete before it: barrier above);
        # the clock stops when all K are done - the last ones may still be with the executor's threads, so wait until they
        # are on the streams before synchronising the device (no batch is credited that the region did not produce)
        pipe._backend.wait_enqueued()
        barrier()
        elapsed = time.perf_counter() - t0
    if stamps:
        print("step return times (ms):", " ".join(f"{1e3 * x:.2f}" for x in stamps), file=sys.stderr)
    kernel_timing(False)
    host_times = pipe.operator_host_times()
    after = _backend.encoded_cache_stats(dev_index)
    times = kernel_timing()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if os.environ.get("BENCH_TEST_SINGLE_DEVICE") == "1" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if after["hits"] - before["hits"] < B * (args.steps - depth - 1) or after["misses"] != before["misses"]:
        raise SystemExit(f"bench: the timed region was not served from the resident streams ({before} -> {after})")
    # batches the timed steps covered: the reader is sequential over the shard, iteration i = batch i mod nb
    covered = [stats[(done + k) % nb] for k in range(args.steps)]
    mean = lambda key: float(np.mean([c[key] for c in covered]))  # noqa: E731
    # crop windows: drawn by the product's generator over the same image shapes (bench seed; the operator inside the
    # pipeline has its own seed from the pipeline's seed sequence - same distribution, other sample)
    master = Bk.philox_state(1234)
    res_bytes = []
    # ... and, when the decoder decodes only the windows (graph-level fusion): what share of a batch's blocks / MCU rows /
    # pixels the value passes and the colour kernel serve - the crop window + the filter's reach (ceil(scale) + 2 pixels
    # for the triangular filter, RandomResizedCropGpu::DrawWindows) on the 16 x 16 MCU grid of 4:2:0 streams
    roi_fused = "windows_of_the_consumer" in pipe.executed_kernels()
    win_px, rect_blocks, row_blocks, all_blocks = [], [], [], []
    for k in range(args.steps):
        anchors, crops = Bk.random_crop_batch(master, covered[k]["shapes"])
        master.ctr[1] += B
        res_bytes.append(float(3 * (crops[:, 0].astype(np.int64) * crops[:, 1]).sum() + 6 * 224 * 224 * B))
        if roi_fused:
            shp = covered[k]["shapes"].astype(np.int64)
            a, c = np.asarray(anchors, np.int64), np.asarray(crops, np.int64)
            reach = np.ceil(np.maximum(c / 224.0, 1.0)).astype(np.int64) + 2
            lo, hi = np.maximum(a - reach, 0), np.minimum(a + c + reach, shp)
            lo = lo // np.array([2, 8]) * np.array([2, 8])      # (windows start on the colour kernel's grid: DrawWindows)
            win_px.append(float(((hi - lo)[:, 0] * (hi - lo)[:, 1]).sum()))
            m_lo, m_hi, m_all = lo // 16, -(-hi // 16), -(-shp // 16)
            rect_blocks.append(float(((m_hi - m_lo)[:, 0] * (m_hi - m_lo)[:, 1]).sum()))
            row_blocks.append(float((m_hi[:, 0] * m_all[:, 1]).sum()))
            all_blocks.append(float((m_all[:, 0] * m_all[:, 1]).sum()))
    roi = None
    if roi_fused:
        roi = {"window_pixels": float(np.mean(win_px)), "rect_share": float(np.sum(rect_blocks) / np.sum(all_blocks)),
               "row_share": float(np.sum(row_blocks) / np.sum(all_blocks))}
    return {"roi": roi, "pipe": pipe, "elapsed": elapsed, "times": times, "host_times": host_times, "mean": mean, "depth": depth,
            "resample_bytes": float(np.mean(res_bytes)), "threads": threads, "setup_s": t_setup,
            "setup_iterations": done - args.warmup, "cache": after, "symbols": float(np.mean([c["symbols"] for c in covered])),
            "stream_bytes_total": float(sum(st["stream_bytes"] for st in stats))}


def kernel_timing(enable=None):
    """Library-side kernel timing (include/dali_amd_kernels.h: daliamdKernelTimingEnable / Report): every launch is
    bracketed by HIP events on its own stream.  enable=True/False switches it; None reads {kernel: (launches, avg_ms)}."""
    from dali_amd import _capi as capi
    lib = capi.kernels()
    if enable is not None:
        # True / False switch it; an int > 1 switches it on and creates the events of that many launches up front
        lib.daliamdKernelTimingEnable(int(enable) if not isinstance(enable, bool) else (1 if enable else 0))
        return None
    need = lib.daliamdKernelTimingReport(None, 0)
    buf = C.create_string_buffer(need + 1)
    lib.daliamdKernelTimingReport(buf, need + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split("\t")
        out[name] = (int(calls), float(ms))
    return out


def _oracle_rate(work, items, seconds_budget, unit):
    """Bounded CPU-baseline sample: `work(i)` (oracle code, one item) on every usable core until about `seconds_budget`
    seconds of wall time have passed.  Returns the cpu_baseline object (kind "port")."""
    from concurrent.futures import ThreadPoolExecutor
    cores = effective_cpu_count()
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while True:
            list(ex.map(work, range(items)))
            done += items
            el = time.perf_counter() - t0
            if el * (done + items) / done > seconds_budget:
                break
    return {"value": done / el, "unit": unit, "cores": cores, "kind": "port", "wall_s": round(el, 2), "items": done}


def bench_heavy_aug(args, device, steps=None, cpu_seconds=8.0):
    """configs[2]: warp_affine + gaussian_blur(sigma=3) + color_twist + erase on 512x512 u8 images, batch 128, through
    the PRODUCT pipeline with the images resident in HBM: readers.file(skip_cached_images) -> decoders.image(mixed,
    decoded-image cache) hands out the 128 cached images in place (no file read, no decode, no copy after the first
    epoch), the per-sample matrices come from an external source, the colour / erase arguments from the random operators.
    Colour twist + erase are one launch (graph-level fusion); fusing them into the blur's write-out as well is opt-in
    (DALI_AMD_BLUR_FUSION=1): measured slower.
    Returns the JSON object; per-kernel times from HIP events around each launch (algorithmic bytes: 3 * 512 * 512 in +
    out per launch); cpu_baseline = the same four operations of the C oracle on the host cores."""
    import shutil
    import tempfile
    import torch
    from PIL import Image
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    from dali_amd.testing import synth_image
    steps = steps or args.steps
    n = 128
    rng = np.random.default_rng(1234)
    root = tempfile.mkdtemp(prefix="dali_amd_bench_aug_")
    os.makedirs(os.path.join(root, "0"))
    images = [synth_image(rng, 512, 512) for _ in range(n)]
    for i in range(n):      # (JPEG: the decoded-image cache keeps what the GPU decoder produces)
        Image.fromarray(images[i]).save(os.path.join(root, "0", f"img_{i:04d}.jpg"), quality=95)

    def matrices():
        mats = []
        for _ in range(n):
            t, s = np.deg2rad(rng.uniform(-30, 30)), rng.uniform(0.8, 1.2)
            c, sn = np.cos(t) / s, np.sin(t) / s
            m = np.array([[c, -sn, 0], [sn, c, 0]], np.float32)
            m[0, 2] = 256 - m[0, 0] * 256 - m[0, 1] * 256
            m[1, 2] = 256 - m[1, 0] * 256 - m[1, 1] * 256
            mats.append(m.reshape(6))
        return mats

    mat_sets = [matrices() for _ in range(4)]
    depth = max(1, min(args.inflight, 4))

    def build(depth):
        pipe = Pipeline(batch_size=n, num_threads=max(2, effective_cpu_count() * 3 // 4), device_id=device.index or 0, seed=1234,
                        prefetch_queue_depth=depth, set_affinity=AFFINITY)
        with pipe:
            enc, _ = fn.readers.file(file_root=root, skip_cached_images=True)
            x = fn.decoders.image(enc, device="mixed", cache_size=256, cache_type="threshold")
            m = fn.external_source(name="matrix")
            y = fn.warp_affine(x, matrix=m, fill_value=0.0)
            y = fn.gaussian_blur(y, sigma=3.0)
            y = fn.color_twist(y, hue=fn.random.uniform(range=[-30.0, 30.0]), saturation=fn.random.uniform(range=[0.7, 1.3]),
                               brightness=fn.random.uniform(range=[0.8, 1.2]), contrast=fn.random.uniform(range=[0.8, 1.2]))
            y = fn.erase(y, anchor=fn.random.uniform(range=[0.0, 0.7], shape=[2]), shape=fn.random.uniform(range=[0.1, 0.3], shape=[2]),
                         normalized=True, fill_value=0.0)
            pipe.set_outputs(y)
        pipe.build()
        fed = [0]

        def run():
            # the external source is consumed one batch per scheduled iteration: keep its queue as deep as the prefetch
            while fed[0] < pipe._scheduled + depth + 1:
                pipe.feed_input("matrix", mat_sets[fed[0] % len(mat_sets)])
                fed[0] += 1
            return pipe.run()
        return pipe, run

    try:
        pipe, run = build(depth)
        for _ in range(4 * (depth + 1) + args.warmup):     # epoch 1 decodes the files into the cache; then every slot is warm
            run()
        torch.cuda.synchronize()
        kernel_timing(12 * (steps + depth + 2))
        kernel_timing(False)
        kernel_timing()
        pipe.operator_host_times()
        kernel_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        pipe._backend.wait_enqueued()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kernel_timing(False)
        times = kernel_timing()
        host = pipe.operator_host_times()
        kernels = pipe.executed_kernels()
        del pipe, run
        # ONE batch in flight: what a launch COSTS (the figures above are how long launches LAST next to the other
        # streams' kernels - with `depth` batches in flight a launch can last longer than the step it is part of)
        alone = {}
        if depth > 1:
            pipe1, run1 = build(1)
            for _ in range(6):
                run1()
            torch.cuda.synchronize()
            kernel_timing(True)
            for _ in range(max(6, min(steps, 20))):
                run1()
            pipe1._backend.wait_enqueued()
            torch.cuda.synchronize()
            kernel_timing(False)
            alone = {k: ms for k, (calls, ms) in kernel_timing().items()}
            del pipe1, run1
    finally:
        shutil.rmtree(root, ignore_errors=True)
    per = {}
    bytes_per = 2 * 3 * 512 * 512 * n
    step_ms = 1e3 * el / steps
    for nm, (calls, ms) in times.items():
        one = alone.get(nm, ms)      # one batch in flight (depth 1: the timed region itself)
        per[nm] = {"algorithmic_bytes": bytes_per, "launches": calls, "avg_ms": one, "in_schedule_ms": ms,
                   "achieved_GBps": bytes_per / (one * 1e-3) / 1e9}
    dom = max(per, key=lambda k: per[k]["avg_ms"])
    # a launch's cost (alone, whole chip) cannot exceed the step it is part of (15 % for run-to-run spread)
    assert per[dom]["avg_ms"] <= step_ms * 1.15, (dom, per[dom], step_ms)
    traffic, traffic_src = measured_traffic(dom, "heavy_aug")
    kern_ms = sum(v["avg_ms"] for v in per.values())
    out = {"metric": "images/sec heavy-aug 512^2 b128 (warp_affine+gaussian_blur(sigma=3)+color_twist+erase)",
           "value": n * steps / el, "unit": "images/s", "n_gpus": 1, "steps": steps,
           "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8 in/out, f32 arithmetic", "data": "synthetic",
           "config": {"workload": "configs[2]: 128 x 512x512x3 u8 resident in HBM (decoded-image cache), through "
                                  "dali_amd.Pipeline", "kernels": kernels, "prefetch_queue_depth": depth,
                      "blur": ("matrix cores (v_mfma_f32_16x16x4_f32, banded Toeplitz): the fmaf chain over the taps, <= 1 LSB on "
                               "< 0.1 % of the elements against the CPU order of roundings (tests/test_gpu_augment.py); "
                               "DALI_AMD_BLUR_MFMA=0: the VALU kernel, bit-exact"
                               if "GaussianBlurMfmaKernel" in per else "VALU kernel, bit-exact against the oracle"),
                      "kernels_ms_per_step": kern_ms, "images_per_s_kernels_only": n / (kern_ms * 1e-3),
                      "kernels_ms_note": "sum of the per-launch durations with ONE batch in flight (cost); "
                                         "roofline.per_kernel[*].in_schedule_ms = durations inside the overlapped timed region",
                      "host_ms_per_step": host.get("<device stage>"), "host_stage_ms_per_step": host.get("<host stage>"),
                      "host_ms_per_operator": {k: v for k, v in host.items() if not k.startswith("<")}},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": per[dom]["achieved_GBps"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per[dom]["achieved_GBps"] / HBM_PEAK_GBS,
                        "duration": "one batch in flight (HIP events around every launch of a prefetch_queue_depth=1 pipeline, same run)",
                        "traffic": traffic, "traffic_source": traffic_src, "per_kernel": per}}
    if cpu_seconds and not args.no_cpu_baseline:
        from oracle import oracle as O
        win = O.gaussian_window(3.0)
        twist = O.color_twist_matrix(hue=15.0, saturation=1.1, brightness=1.1, contrast=0.9)
        mats = mat_sets[0]

        def one(i):
            y = O.warp_affine_u8(images[i], mats[i].reshape(2, 3), fill=(0.0,))
            y = O.gaussian_blur_u8(y, win)
            y = O.linear_transform_u8(y, *twist)
            O.erase_u8(y, [(0.3, 0.2)], [(0.2, 0.25)], normalized_anchor=True, normalized_shape=True)

        base = _oracle_rate(one, 32, cpu_seconds, "images/s")
        base["sample"] = (f"{base['items']} images = {base['items'] // 32} pass(es) over the first 32 of the 128 images: warp_affine + "
                          f"gaussian_blur(sigma=3) + colour twist + erase per image on the C oracle, one task per image, "
                          f"{base['cores']} threads, {base['wall_s']} s wall")
        out["cpu_baseline"] = base
    return out


def _wav16(x, rate):
    import struct
    pcm = np.round(np.clip(x, -1, 1) * 32767).astype("<i2")
    return (b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes())


def bench_audio(args, device, steps=None, cpu_seconds=8.0):
    """configs[3] as BASELINE.json names it: decoders.audio -> spectrogram(nfft=1024, step 256) -> mel_filter_bank(80) ->
    to_decibels on 64 utterances of 8-16 s at 16 kHz (LibriSpeech shape), from 16-bit WAV FILES: readers.file (loader
    threads, page cache) -> decoders.audio on the host thread pool (audio_decoder_op.cc:36-100) -> H2D -> one fused
    launch (+ the dB pair), asynchronous executor, two batches in flight.  Returns the JSON object; cpu_baseline = the
    numpy oracle (decode + float64 FFT spectrogram + mel + dB per utterance) on the host cores."""
    import shutil
    import tempfile
    import torch
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    steps = steps or args.steps
    rng = np.random.default_rng(1234)
    n = 64
    sigs = [rng.normal(0, 0.1, int(rng.uniform(8, 16) * 16000)).astype(np.float32) for _ in range(n)]
    wavs = [_wav16(s, 16000) for s in sigs]
    root = tempfile.mkdtemp(prefix="dali_amd_bench_audio_")
    os.makedirs(os.path.join(root, "0"))
    for i, w in enumerate(wavs):
        with open(os.path.join(root, "0", f"utt_{i:04d}.wav"), "wb") as f:
            f.write(w)
    depth = int(os.environ.get("BENCH_AUDIO_DEPTH", "2"))
    threads = max(2, effective_cpu_count() * 3 // 4)
    try:
        pipe = Pipeline(batch_size=n, num_threads=threads, device_id=device.index or 0, prefetch_queue_depth=depth,
                        exec_async=True, seed=1234, set_affinity=AFFINITY)
        with pipe:
            enc, _ = fn.readers.file(file_root=root, file_filters=["*.wav"], prefetch_queue_depth=2)
            audio, _rate = fn.decoders.audio(enc, downmix=True)
            spec = fn.spectrogram(audio.gpu(), nfft=1024, window_length=1024, window_step=256)
            mel = fn.mel_filter_bank(spec, nfilter=80, sample_rate=16000.0, freq_high=8000.0)
            pipe.set_outputs(fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0))
        pipe.enable_operator_timing()
        pipe.build()
        frames = sum(len(s) // 256 + 1 for s in sigs)
        for _ in range(args.warmup + 2 * (depth + 1)):
            pipe.run()
        torch.cuda.synchronize()
        kernel_timing(8 * (steps + depth + 2))
        kernel_timing(False)
        kernel_timing()
        pipe.operator_host_times()
        kernel_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.run()
        pipe._backend.wait_enqueued()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kernel_timing(False)
        ktimes = kernel_timing()               # HIP events around every launch, on the stream it is launched on
        host = pipe.operator_host_times()
        times = pipe.operator_device_times()   # ... and around each operator (descriptor upload + launches)
        kernels = pipe.executed_kernels()
        del pipe
    finally:
        shutil.rmtree(root, ignore_errors=True)
    samples = sum(len(s) for s in sigs)
    pcm16 = os.environ.get("DALI_AMD_NO_PCM16_FUSION", "0") in ("", "0")
    in_bytes = (2 if pcm16 else 4) * samples     # what the spectrogram kernel reads per sample: the 16-bit PCM as it crossed the bus
    # algorithmic bytes per launch: every signal sample read once, every output element written once
    algo = {"SpectrogramKernel": in_bytes + 4 * 513 * frames, "MelKernel": 4 * 513 * frames + 4 * 80 * frames,
            "DecibelMaxKernel": 4 * 80 * frames, "DecibelKernel": 8 * 80 * frames}
    # the fused launch (graph-level fusion of the chain): signal in, mel energies out - the spectrogram stays in LDS
    algo["SpectrogramMelMfmaKernel"] = algo["SpectrogramMelKernel"] = algo["SpectrogramFastKernel"] = in_bytes + 4 * 80 * frames
    per = {}
    step_ms = 1e3 * el / steps
    for kern, (calls, ms) in ktimes.items():
        if kern in algo:
            per[kern] = {"algorithmic_bytes": algo[kern], "launches": calls, "avg_ms": ms,
                         "achieved_GBps": algo[kern] / (ms * 1e-3) / 1e9}
    dom = max(per, key=lambda k: per[k]["avg_ms"]) if per else None
    if dom:
        assert per[dom]["avg_ms"] <= step_ms * 1.15, (dom, per[dom], step_ms)
    traffic, traffic_src = measured_traffic(dom, "audio") if dom else (None, None)
    ach = per[dom]["achieved_GBps"] if dom else None
    # What bounds the step is the bus: the PCM of a step crosses it once (files -> pinned reader block -> HBM) and that
    # transfer alone is most of ms_per_step.  Peak = the H2D rate measured in this run (pinned, 256 MiB copies).
    h2d_peak = measured_h2d_ceiling(device)
    h2d_ach = in_bytes / (step_ms * 1e-3) / 1e9
    out = {"metric": "utterances/sec decoders.audio->spectrogram(1024)->mel(80)->dB b64, from 16-bit WAV files",
           "value": n * steps / el, "unit": "utterances/s", "n_gpus": 1, "steps": steps,
           "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[3]: 64 mono utterances, 16 kHz, 8-16 s, 16-bit WAV files in the page cache -> "
                                  "readers.file -> decoders.audio (host: header parse) -> H2D -> spectrogram -> mel -> dB (device)",
                      "frames": frames, "samples": samples, "kernels": kernels, "prefetch_queue_depth": depth,
                      "exec_async": True, "host_threads": threads,
                      "kernels_ms_per_step": sum(v["avg_ms"] for v in per.values()),
                      "utterances_per_s_kernels_only": n / (1e-3 * sum(v["avg_ms"] for v in per.values())) if per else None,
                      "algorithmic_MB_per_step": sum(v["algorithmic_bytes"] for v in per.values()) / 1e6,
                      "h2d_MB_per_step": in_bytes / 1e6,
                      "mel_variant": "valu" if os.environ.get("DALI_AMD_MEL_VALU") == "1" else "mfma",
                      "host_ms_per_operator": {k: v for k, v in host.items() if not k.startswith("<")},
                      "host_stage_ms_per_step": host.get("<host stage>"), "device_stage_ms_per_step": host.get("<device stage>"),
                      "pcm16_fusion": pcm16,
                      "note": "value is end to end from files.  Round 4: the decoded audio feeds only the copy in front of the "
                              "gpu spectrogram, so the 25 MB of 16-bit PCM of a step cross the bus as they are (decoders.audio hands "
                              "out a view of the files' data chunks, one H2D transfer of the reader's block) and the spectrogram "
                              "kernel's load divides by 32768 - the same bits as converting to 50 MB of float on the host "
                              "(DALI_AMD_NO_PCM16_FUSION=1: that path, tests/test_gpu_audio.py holds the two to each other)"},
           "roofline": {"bound": "pcie", "kernel": "host->device transfer of the step's PCM (hipMemcpyAsync, pinned)",
                        "achieved": h2d_ach, "peak": h2d_peak, "unit": "GB/s", "frac": h2d_ach / h2d_peak if h2d_peak else None,
                        "traffic": None,
                        "note": "the step is bus-bound: its PCM bytes / ms_per_step against the H2D rate measured in this run; "
                                "`dominant_kernel` is the HBM roofline of the largest launch (what the step would be bound by "
                                "with the samples already in HBM: config.utterances_per_s_kernels_only)",
                        "dominant_kernel": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": traffic,
                                            "traffic_source": traffic_src},
                        "per_kernel": per, "operator_device_ms": times}}
    if cpu_seconds and not args.no_cpu_baseline:
        from oracle import audio as A

        def one(i):
            x, _rate = A.decode_wav(wavs[i])
            spec = A.spectrogram(x, nfft=1024, window_length=1024, window_step=256)
            A.to_decibels(A.mel_filter_bank(spec, nfilter=80, sample_rate=16000.0, freq_high=8000.0), multiplier=10.0, cutoff_db=-80.0)

        base = _oracle_rate(one, n, cpu_seconds, "utterances/s")
        base["sample"] = (f"{base['items']} utterances = {base['items'] // n} pass(es) over the 64 files: WAV decode + spectrogram "
                          f"(float64 FFT) + mel + dB per utterance on the numpy oracle, one task per utterance, "
                          f"{base['cores']} threads, {base['wall_s']} s wall")
        out["cpu_baseline"] = base
    return out


def bench_cpu_backend(args):
    """BASELINE configs[0]: the ImageNet train pipe on the product's CPU backend (decoders.image -> random_resized_crop
    -> crop_mirror_normalize, 224 x 224, batch 32) through dali_amd.Pipeline - plumbing, no GPU.  One JSON line."""
    import shutil
    import tempfile
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    B, threads = 32, effective_cpu_count()
    enc = make_dataset(0, 256, workers=threads)
    root = tempfile.mkdtemp(prefix="dali_amd_bench_cpu_")
    try:
        write_dataset(root, enc)
        pipe = Pipeline(batch_size=B, num_threads=threads, device_id=None, seed=1234, prefetch_queue_depth=2)
        with pipe:
            jpegs, labels = fn.readers.file(file_root=root, name="Reader")
            images = fn.decoders.image(jpegs, device="cpu", output_type=types.RGB)
            crops = fn.random_resized_crop(images, size=[224, 224])
            out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW",
                                           mean=[0.485 * 255, 0.456 * 255, 0.406 * 255],
                                           std=[0.229 * 255, 0.224 * 255, 0.225 * 255],
                                           mirror=fn.random.coin_flip(probability=0.5))
            pipe.set_outputs(out, labels)
        pipe.build()
        for _ in range(max(1, args.warmup)):
            pipe.run()
        steps = min(args.steps, 64)
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.run()
        el = time.perf_counter() - t0
    finally:
        shutil.rmtree(root, ignore_errors=True)
    extra = {}
    try:   # the other configurations on the CPU backend (side figures: the same host kernels the tests pin to the oracle)
        extra = cpu_backend_side_configs(threads)
    except Exception as e:   # never fatal for the configs[0] line
        extra = {"error": str(e)[:200]}
    print(json.dumps({"metric": "images/sec JPEG->RRC->CMN 224^2 b32, CPU backend", "value": steps * B / el,
                      "unit": "images/s", "n_gpus": 0, "steps": steps, "warmup": max(1, args.warmup),
                      "ms_per_step": 1e3 * el / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "u8/i32 decode, f32 resample, f16 out", "data": "synthetic",
                      "config": {"workload": "configs[0]: ImageNet train pipe on the CPU backend of dali_amd.Pipeline, "
                                             "batch 32, 224x224 fp16 CHW", "host_threads": threads,
                                 "kernels": pipe.executed_kernels()},
                      "cpu_backend_other_configs": extra}))


def cpu_backend_side_configs(threads):
    """configs[2] (512 x 512 heavy augmentation, batch 16 here) and configs[3] (spectrogram -> mel -> dB of 64 utterances)
    through dali_amd.Pipeline on the CPU backend: images / utterances per second on this host."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(1234)
    out = {}
    # ---- configs[2]
    bs = 16
    base = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    imgs = [np.ascontiguousarray(np.kron(np.roll(base, i, 0), np.ones((8, 8, 1), np.uint8))) for i in range(bs)]
    mats = []
    for _ in range(bs):
        t, sc = np.deg2rad(rng.uniform(-30, 30)), rng.uniform(0.8, 1.2)
        c, sn = np.cos(t) / sc, np.sin(t) / sc
        m = np.array([[c, -sn, 0], [sn, c, 0]], np.float32)
        m[0, 2] = 256 - m[0, 0] * 256 - m[0, 1] * 256
        m[1, 2] = 256 - m[1, 0] * 256 - m[1, 1] * 256
        mats.append(m.reshape(6))
    pipe = Pipeline(batch_size=bs, num_threads=threads, device_id=None, seed=17, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="images", layout="HWC")
        m = fn.external_source(name="matrix")
        y = fn.warp_affine(x, matrix=m, fill_value=0.0, interp_type=types.INTERP_LINEAR)
        y = fn.gaussian_blur(y, sigma=3.0)
        y = fn.color_twist(y, hue=fn.random.uniform(range=[-30.0, 30.0], seed=1), saturation=fn.random.uniform(range=[0.7, 1.3], seed=2),
                           brightness=fn.random.uniform(range=[0.8, 1.2], seed=3), contrast=fn.random.uniform(range=[0.8, 1.2], seed=4))
        y = fn.erase(y, anchor=fn.random.uniform(range=[0.0, 0.7], shape=[2], seed=5),
                     shape=fn.random.uniform(range=[0.1, 0.3], shape=[2], seed=6), normalized=True, fill_value=0.0)
        pipe.set_outputs(y)
    pipe.build()
    reps = 3
    t0 = None
    for r in range(reps + 1):
        if r == 1:
            t0 = time.perf_counter()
        pipe.feed_input("images", imgs, layout="HWC")
        pipe.feed_input("matrix", mats)
        pipe.run()
    out["configs[2] heavy_aug 512x512"] = {"value": reps * bs / (time.perf_counter() - t0), "unit": "images/s", "batch": bs,
                                           "kernels": pipe.executed_kernels()}
    # ---- configs[3]
    bs = 64
    sigs = []
    for _ in range(bs):
        n = int(rng.uniform(8, 16) * 16000)
        t = np.arange(n) / 16000.0
        sigs.append((0.3 * np.sin(2 * np.pi * (200 + 50 * t) * t) + 0.05 * rng.standard_normal(n)).astype(np.float32))
    pipe = Pipeline(batch_size=bs, num_threads=threads, device_id=None, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        spec = fn.spectrogram(x, nfft=1024, window_length=1024, window_step=256)
        mel = fn.mel_filter_bank(spec, nfilter=80, sample_rate=16000.0, freq_high=8000.0)
        pipe.set_outputs(fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0))
    pipe.build()
    for r in range(reps + 1):
        if r == 1:
            t0 = time.perf_counter()
        pipe.feed_input("x", sigs)
        pipe.run()
    out["configs[3] audio b64"] = {"value": reps * bs / (time.perf_counter() - t0), "unit": "utterances/s", "batch": bs,
                                   "kernels": pipe.executed_kernels()}
    return out


COMPACT_LINE_LIMIT = 4000   # bytes; the driver keeps an 8 KB tail of stdout and parses its last line (tests/test_bench_line.py)


def compact_line(full, details_path=None):
    """The ONE line the driver parses: the contract's keys, `config` = the workload string + the scalars the round's story
    rests on, `roofline` of the dominant kernel only, `cpu_baseline` without its prose.  Everything else (per-kernel tables,
    every leg's host account, notes) lives in the details file.  The reference's own benchmark prints one number
    (internal_tools/hw_decoder_bench.py:651)."""
    def num(x, nd=6):
        return float(f"{x:.{nd}g}") if isinstance(x, float) else x
    cfg_full = full.get("config", {})
    keep = ("global_batch", "parallelism", "batches_in_flight", "distinct_batches", "dataset_images_per_gpu", "cache_type",
            "resident_set_MB", "jpeg_bytes_per_batch", "host_ms_per_step", "decoder_single_stream_ms",
            "sync_kernel_single_stream_ms", "whole_step_frac", "e2e_images_per_s", "e2e_pcie_frac",
            "e2e_images_per_s_at_pcie_peak", "e2e_indexed_images_per_s", "e2e_sharded_images_per_s",
            "resident_indexed_images_per_s", "resident_full_decode_images_per_s", "resident_roi_decode_images_per_s",
            "iterator_images_per_s", "e2e_local_world8_images_per_s", "heavy_aug_images_per_s", "audio_utterances_per_s",
            "value_distinct_dht", "value_mixed", "value_large_images", "value_resident_4GB", "resident_4GB_set_MB",
            "graph_replay", "commands_per_iteration")
    cfg = {"workload": str(cfg_full.get("workload", ""))[:420]}
    for k in keep:
        v = cfg_full.get(k)
        if isinstance(v, (int, float, bool, str)):
            cfg[k] = num(v)
    if details_path:
        cfg["details"] = details_path
    # (the contract's own keys keep their full precision: value * ms_per_step must reproduce the batch exactly)
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = cfg
    rf = full.get("roofline") or {}
    per = (rf.get("per_kernel") or {}).get(rf.get("kernel"), {})
    out["roofline"] = {k: num(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")}
    out["roofline"]["algorithmic_bytes"] = num(per.get("algorithmic_bytes"))
    out["roofline"]["avg_ms"] = num(per.get("avg_ms"))
    out["roofline"]["alone_ms"] = num(((cfg_full.get("pipeline") or {}).get("single_stream_kernel_ms") or {}).get(rf.get("kernel")))
    out["roofline"]["whole_step_frac"] = num((rf.get("whole_step") or {}).get("frac"))
    out["roofline"]["traffic_source"] = rf.get("traffic_source")
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {"value": num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                               "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:200]}
    text = json.dumps(out)
    if len(text) > COMPACT_LINE_LIMIT:      # never hand the driver a line it cannot keep: drop the prose first
        out["config"]["workload"] = out["config"]["workload"][:120]
        if "cpu_baseline" in out:
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:80]
    assert len(json.dumps(out)) <= COMPACT_LINE_LIMIT, len(json.dumps(out))
    return out


def emit(line, full=False):
    """Writes the details object next to the bench (bench_details.json in the repo root and, on a gpurun box, under
    gpurun_out/ so that it travels back), then prints the compact line as the LAST line of stdout."""
    paths = [os.environ.get("BENCH_DETAILS") or os.path.join(ROOT, "bench_details.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not os.environ.get("BENCH_DETAILS"):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_details.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(line, f, indent=1)
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    sys.stdout.flush()
    print(json.dumps(line if full else compact_line(line, written)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--batches", type=int, default=4,
                    help="distinct batches the steps rotate through (data set = batches * batch images per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end dali_amd.Pipeline legs")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip the one-batch-in-flight leg as well (profiling: every launch of the process then belongs to "
                         "the set-up or the timed region, both at the full depth)")
    ap.add_argument("--side-legs", action="store_true",
                    help="also run the informational legs the default (driver) command leaves out: ROI-decoder and full-decode "
                         "variants of the resident graph, DALIGenericIterator, decoded-image cache, the rank-of-eight host share, "
                         "the Pillow baseline, configs[2] heavy_aug and configs[3] audio (tools/collect_profiles.sh passes it)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the realistic-mix legs (value_distinct_dht / value_mixed / value_large_images / value_resident_4GB)")
    ap.add_argument("--resident-copies", type=int, default=40,
                    help="value_resident_4GB: the data set linked this many times (40 x 1024 files of 99 KB = 4 GB of resident "
                         "streams, beyond the 256 MB Infinity Cache), read with random_shuffle=True")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole details object as the last stdout line instead of the compact line (tools/*.sh that "
                         "read roofline.per_kernel); the details are written to bench_details.json either way")
    ap.add_argument("--e2e-batch", type=int, default=512, help="batch per GPU of the sharded end-to-end leg (configs[4])")
    ap.add_argument("--emulate-local-world", type=int, default=8,
                    help="single-GPU runs: one more end-to-end leg with the CPUs one rank of a node with this many GPUs would "
                         "get (0 / 1: skip)")
    ap.add_argument("--inflight", type=int, default=5,
                    help="batches in flight on the GPU = the executor's prefetch_queue_depth (iterations rotate over its three compute streams)")
    ap.add_argument("--huffman", default="gpu", choices=["gpu", "host"],
                    help="gpu: the step starts from JPEG bytes in HBM (default); host: from host-decoded coefficient blocks")
    ap.add_argument("--no-fused-idct", action="store_true",
                    help="store the coefficients and run the stand-alone IDCT kernel (the GPU entropy decoder's default is "
                         "to dequantise + inverse-transform the blocks itself)")
    ap.add_argument("--driver", default="pipeline", choices=["pipeline", "python"],
                    help="pipeline (default): the headline is timed through dali_amd.Pipeline, the product's C++ executor, "
                         "on a data set resident in HBM as encoded streams; python: the kernel library driven from "
                         "dali_amd/backend.py (kernel experiments: --huffman host, --no-fused-idct)")
    ap.add_argument("--cache-type", default="encoded", choices=["encoded", "indexed"],
                    help="how the timed streams are resident (decoders.image cache_type): encoded = as they are in the file, every "
                         "epoch parses them anew - the meaning of `value` in every round; indexed = with the side information of "
                         "their first decode (profiling aid: the default run reports that rate as config.resident_indexed_images_per_s)")
    ap.add_argument("--workload", default="imagenet", choices=["imagenet", "heavy_aug", "audio", "cpu"],
                    help="imagenet = the headline metric (default); heavy_aug / audio = configs[2] / configs[3] side benches; "
                         "cpu = configs[0], the train pipe on the CPU backend (no GPU needed)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.workload == "cpu":
        return bench_cpu_backend(args)
    if args.driver == "pipeline" and (args.huffman != "gpu" or args.no_fused_idct):
        args.driver = "python"     # the kernel experiments exist in the Python driver only
    B = args.batch
    nb = max(1, args.batches)
    inflight = max(1, min(args.inflight, nb))
    while nb % inflight:       # a batch always runs on the same stream
        inflight -= 1
    per_rank = nb * B
    enc_all = None
    if args.workload == "imagenet":
        # shard `rank` of `world` (contiguous, like loader.cc:78-87) of a world * per_rank image data set; generated in
        # forked workers BEFORE the HIP runtime comes up in this process
        t_gen = time.perf_counter()
        enc_all = make_dataset(rank * per_rank, per_rank, workers=max(1, effective_cpu_count() // max(1, local_world)))
        t_gen = time.perf_counter() - t_gen
    # The realistic-mix data sets are generated LATER, in a child process, right before their legs (generate_variant_dataset):
    # made here - three more data sets, twenty 12-megapixel frames among them, in sixteen forked workers - they left the parent
    # with a few hundred MB of freshly freed memory in front of the headline's 9 ms region; the two runs of ~90 whose `value` came
    # out halved (306 k, 292 k: every kernel at its usual duration, the step twice as long) were both runs of that form, none of the
    # 48 headline-only runs was (tools/outlier_hunt.sh).
    variants = ()
    if (args.workload == "imagenet" and world == 1 and args.driver == "pipeline" and not args.no_variants and not args.no_e2e
            and args.cache_type == "encoded"):
        variants = ("distinct_dht", "mixed", "large")

    import torch
    import torch.distributed as dist

    # BENCH_TEST_SINGLE_DEVICE=1: every rank on cuda:0 with a gloo group - exercises the N > 1 code path (sharding, barrier,
    # max over ranks, sharded end-to-end leg) on a one-GPU box.  Its numbers mean nothing; it is a smoke test.
    single_device_test = world > 1 and os.environ.get("BENCH_TEST_SINGLE_DEVICE") == "1"
    dev_index = 0 if single_device_test else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_device_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.workload == "heavy_aug":
        return print(json.dumps(bench_heavy_aug(args, device)))
    if args.workload == "audio":
        return print(json.dumps(bench_audio(args, device)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from dali_amd.backend import HUFFMAN_KERNELS, HUFFMAN_KERNEL_NAMES, huffman_algorithmic_bytes
    root = None
    pipe_info = None
    if args.driver == "pipeline":
        import tempfile
        fused = True
        # the data set directory of readers.file: one per job, every rank writes its shard (configs[4] layout)
        root = (tempfile.mkdtemp(prefix="dali_amd_bench_") if world == 1 else
                os.path.join(tempfile.gettempdir(), f"dali_amd_bench_shared_{os.environ.get('MASTER_PORT', '0')}"))
        write_dataset(root, enc_all, first_index=rank * per_rank)
        barrier()
        r = run_resident_pipeline(args, root, enc_all, device, dev_index, rank, world, local_world, barrier, dist)
        elapsed, inflight = r["elapsed"], r["depth"]
        stream_bytes_total = r["stream_bytes_total"]
        stream_bytes, coef_elems_mean, pixels_mean = r["mean"]("stream_bytes"), r["mean"]("coef_elems"), r["mean"]("pixels")
        resample_bytes, symbols = [r["resample_bytes"]], r["symbols"]
        algo = dict(huffman_algorithmic_bytes(stream_bytes, coef_elems_mean, B, True))
        algo["JpegColorKernel"] = coef_elems_mean + 3 * pixels_mean          # planes in + RGB out
        # the block kernel with the fused colour output: stream + per-block position / level in, RGB out (no planes)
        algo["BlockColorKernel"] = algo["BlockKernel"] - coef_elems_mean + 3 * pixels_mean
        algo["SeamKernel"] = 0.0
        algo["PrepareKernel"] = stream_bytes   # (round 5: the code tables come from the host, built once per DHT set - the launch only counts the tiles' kept bytes)
        algo["IndexedSyncKernel"] = stream_bytes + 8 * coef_elems_mean / 64   # stream in (the needed slices at most), 8 B per block out
        if args.cache_type == "encoded":
            # (under cache_type="encoded" only FLAT streams - fewer than 64 bits per block, 1 of the 1 024 bench files, 0.1 % of the
            # bytes - are kept with their index and take this kernel: its bytes are not the batch's)
            algo["IndexedSyncKernel"] = 0.0
        algo["ResampleKernel"] = r["resample_bytes"]
        algo["ResampleTablesKernel"] = 0.0
        if r["roi"]:
            # the decoder decodes windows: the position passes still see every byte (the stream is serial), the DC pass stops
            # behind the window's last MCU row, the block kernel serves the window's MCU rectangle, the colour kernel its pixels
            blocks = coef_elems_mean / 64
            rect, rows = r["roi"]["rect_share"], r["roi"]["row_share"]
            algo["DcKernel"] = (4 + 10) * blocks * rows
            algo["BlockKernel"] = rect * (stream_bytes + 10 * blocks + coef_elems_mean)
            algo["JpegColorKernel"] = (coef_elems_mean / pixels_mean + 3) * r["roi"]["window_pixels"]
            algo["BlockColorKernel"] = rect * (stream_bytes + 10 * blocks) + 3 * r["roi"]["window_pixels"]
        kern = {k: (float(algo.get(k, 0.0)), ms) for k, (calls, ms) in r["times"].items()}
        launches = {k: calls for k, (calls, ms) in r["times"].items()}
        huffman_total_ms = float(sum(kern[k][1] for k in HUFFMAN_KERNEL_NAMES if k in kern))
        host_ms_per_step = r["host_times"].get("<device stage>", 0.0)
        pipe_info = {"driver": "dali_amd.Pipeline (C++ executor), data set resident in HBM as encoded streams "
                               "(decoders.image cache_type='encoded' + readers.file skip_cached_images)",
                     "host_threads": r["threads"], "prefetch_queue_depth": r["depth"],
                     "host_stage_ms_per_step": r["host_times"].get("<host stage>"),
                     "device_stage_ms_per_step": r["host_times"].get("<device stage>"),
                     "host_ms_per_operator": {k: v for k, v in r["host_times"].items() if not k.startswith("<")},
                     "setup_s": r["setup_s"], "setup_iterations": r["setup_iterations"], "encoded_cache": r["cache"],
                     "launches_timed": launches, "kernels": r["pipe"].executed_kernels(),
                     "roi_decode_fusion": r["roi"] and dict(r["roi"], note=(
                         "graph-level fusion of the executor: this decoders.image feeds only the random_resized_crop, so that "
                         "operator's windows (+ the resampling filter's reach) are what is decoded; the batch is bit-identical "
                         "to the one a full decode gives (tests/test_gpu_roi_fusion.py, tests/test_gpu_headline.py); "
                         "`resident_full_decode` below is the same pipeline with DALI_AMD_ROI_FUSION=0"))}
        huffman_single_ms = None
        single_stream = None
        if r["depth"] > 1 and not args.no_side_legs:
            # ONE batch in flight: per-kernel durations without another batch's kernels on the same CUs - what a kernel
            # costs, as opposed to how long it lasts inside the overlapped schedule of the timed region above
            pipe1 = resident_pipeline(root, B, dev_index, 1, r["threads"], shard_id=rank, num_shards=world,
                                      cache_mb=max(64, int(2.2 * sum(len(e) for e in enc_all) / 2**20)), cache_type=args.cache_type)
            for _ in range(3 * nb):
                pipe1.run()
            torch.cuda.synchronize()
            kernel_timing(True)
            for _ in range(3 * nb):
                pipe1.run()
            torch.cuda.synchronize()
            kernel_timing(False)
            single_stream = {k: ms for k, (calls, ms) in kernel_timing().items()}
            del pipe1
        if world == 1 and not args.no_e2e and args.side_legs:
            # the same resident streams through the fused ROI decoder (what NVIDIA's own decoder benchmark times,
            # hw_decoder_bench.py:178-188): informational, never `value`
            pipe2 = resident_pipeline(root, B, dev_index, r["depth"], r["threads"],
                                      cache_mb=max(64, int(2.2 * sum(len(e) for e in enc_all) / 2**20)), roi_decode=True,
                                      cache_type=args.cache_type)
            for _ in range((r["depth"] + 2) * nb + args.warmup):
                pipe2.run()
            torch.cuda.synchronize()
            t_roi = time.perf_counter()
            for _ in range(args.steps):
                pipe2.run()
            pipe2._backend.wait_enqueued()
            torch.cuda.synchronize()
            t_roi = time.perf_counter() - t_roi
            pipe_info["resident_roi_decode"] = {
                "value": B * args.steps / t_roi, "unit": "images/s", "ms_per_step": 1e3 * t_roi / args.steps,
                "kernels": pipe2.executed_kernels(),
                "note": "decoders.image_random_crop -> resize -> crop_mirror_normalize on the same resident streams: the "
                        "entropy decoder stops at the last MCU row of the crop window, only the window's blocks are "
                        "transformed and colour-converted"}
            del pipe2
            if r["roi"]:
                # ... and the headline graph with the fusion switched off: every image decoded whole, as in rounds 1-3
                pipe3 = resident_pipeline(root, B, dev_index, r["depth"], r["threads"],
                                          cache_mb=max(64, int(2.2 * sum(len(e) for e in enc_all) / 2**20)), roi_fusion=False,
                                          cache_type=args.cache_type)
                for _ in range((r["depth"] + 2) * nb + args.warmup):
                    pipe3.run()
                torch.cuda.synchronize()
                t_full = time.perf_counter()
                for _ in range(args.steps):
                    pipe3.run()
                pipe3._backend.wait_enqueued()
                torch.cuda.synchronize()
                t_full = time.perf_counter() - t_full
                pipe_info["resident_full_decode"] = {
                    "value": B * args.steps / t_full, "unit": "images/s", "ms_per_step": 1e3 * t_full / args.steps,
                    "kernels": pipe3.executed_kernels(),
                    "note": "the headline graph with DALI_AMD_ROI_FUSION=0: every image is decoded whole and the crop window is "
                            "taken by the resampling kernel (the form `value` was measured in up to round 4's collection)"}
                del pipe3
        if single_stream:
            huffman_single_ms = float(sum(single_stream.get(k, 0.0) for k in HUFFMAN_KERNEL_NAMES))
            pipe_info["single_stream_kernel_ms"] = single_stream
            pipe_info["single_stream_note"] = ("per-kernel durations with ONE batch in flight (prefetch_queue_depth=1, same "
                                               "resident streams, HIP events around every launch): kernel cost; the per_kernel "
                                               "figures of `roofline` are durations inside the overlapped schedule of the timed region")
        del r
    else:
        huffman_single_ms = None
        streams = [torch.cuda.Stream(device=device) for _ in range(inflight)]
        paths = [HotPath(enc_all[b * B:(b + 1) * B], device, streams[b % inflight], huffman=args.huffman,
                         fused_idct=not args.no_fused_idct, first_iteration=b) for b in range(nb)]
        fused = paths[0].fused_idct

        for w in range(args.warmup):
            paths[w % nb].step(advance=nb)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(args.steps)]
        kev = None
        if args.huffman == "gpu":
            from dali_amd import backend as _backend
            kev = [_backend.KernelEvents(len(_backend.HUFFMAN_KERNELS)) for _ in range(args.steps)]
        resample_bytes = []
        barrier()
        for hp in paths:
            hp.host_s = 0.0
        t0 = time.perf_counter()
        for k in range(args.steps):
            hp = paths[(args.warmup + k) % nb]
            _, crops = hp.step(record=ev[k], kernel_events=kev[k] if kev else None, advance=nb)
            resample_bytes.append(hp.resample_bytes(crops))
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if single_device_test else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # per-kernel average durations from the events recorded inside the timed region; algorithmic bytes averaged over
        # the same steps (the batches differ)
        step_paths = [paths[(args.warmup + k) % nb] for k in range(args.steps)]
        mean_of = lambda f: float(np.mean([f(hp) for hp in step_paths]))  # noqa: E731
        ms_idct = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        ms_color = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        ms_resample = float(np.mean([e[3].elapsed_time(e[4]) for e in ev]))
        bytes_idct, bytes_color = mean_of(lambda hp: hp.bytes_idct), mean_of(lambda hp: hp.bytes_color)
        stream_bytes = mean_of(lambda hp: hp.plan.stream_bytes) if args.huffman == "gpu" else None
        kern = {
            "JpegColorKernel": (bytes_color, ms_color),
            "ResampleKernel": (float(np.mean(resample_bytes)), ms_resample),
        }
        if not fused:
            kern["JpegIdctKernel"] = (bytes_idct, ms_idct)
        if args.huffman == "gpu":
            # per-kernel durations of the entropy decoder (events recorded between its launches, same stream)
            per = np.array([ke.elapsed_ms() for ke in kev]).mean(0)
            sb, ce = stream_bytes, mean_of(lambda hp: hp.plan.coef_elems)
            symbols = float(np.mean([hp.symbols for hp in paths]))
            huff_bytes = huffman_algorithmic_bytes(sb, ce, B, fused)
            for name, ms in zip(HUFFMAN_KERNELS, per):
                kern[name] = (huff_bytes[name], float(ms))
            huffman_total_ms = float(np.mean([e[5].elapsed_time(e[6]) for e in ev]))
        coef_elems_mean, pixels_mean = mean_of(lambda hp: hp.plan.coef_elems), mean_of(lambda hp: hp.pixels)
        host_ms_per_step = 1e3 * sum(hp.host_s for hp in paths) / args.steps
        huffman_s = float(np.mean([hp.huffman_s for hp in paths]))
        del paths, step_paths, hp
    dominant = max(kern, key=lambda k: kern[k][1])
    ach = kern[dominant][0] / (kern[dominant][1] * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(dominant, "indexed" if args.cache_type == "indexed" else "")

    line = None
    if rank == 0:
        value = world * B * args.steps / elapsed
        step_ms = 1e3 * elapsed / args.steps
        step_bytes = float(sum(v[0] for v in kern.values()))
        copy_ceiling = measured_copy_ceiling(device)
        # SURVEY.md 8(d) end-to-end formula: 6 P (coefficients in, RGB out) + 3 s P + 6 O (fused resample + CMN)
        post_entropy = 2 * coef_elems_mean + 3 * pixels_mean + float(np.mean(resample_bytes))
        nk = len(HUFFMAN_KERNELS) if args.huffman == "gpu" else 0
        line = {
            "metric": "images/sec JPEG->RRC->CMN 224^2 b256 per GPU",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 decode, f32 resample, f16 out", "data": "synthetic",
            "config": {"workload": "configs[1]: HIP JPEG " + ("Huffman decode -> " if args.huffman == "gpu" else "") +
                                   ("(fused) " if fused else "") + "dequant+IDCT -> upsample+YCbCr->RGB -> fused "
                                   f"RandomResizedCrop+CropMirrorNormalize, 224x224, batch={B}/GPU, fp16 CHW out; "
                                   f"{per_rank} ImageNet-like synthetic JPEGs per GPU (seed 1234) = {nb} distinct batches "
                                   "rotated across the steps, inputs = " +
                                   ("JPEG entropy-coded segments (bytes) resident in HBM" if args.huffman == "gpu" else
                                    "host-entropy-decoded coefficient blocks resident in HBM"),
                       "huffman": args.huffman, "fused_dequant_idct": fused, "batches_in_flight": inflight,
                       "distinct_batches": nb, "dataset_images_per_gpu": per_rank, "dataset_generation_s": t_gen,
                       f"huffman_ms_per_batch({nk} kernels)": huffman_total_ms if args.huffman == "gpu" else None,
                       "host_ms_per_step": host_ms_per_step,
                       "jpeg_bytes_per_batch": stream_bytes,
                       "global_batch": world * B, "parallelism": f"shard{world} (shard_id/num_shards, no collective)",
                       "pixels_per_batch": pixels_mean, "driver": args.driver,
                       # the executor decodes only the windows the random_resized_crop draws (bit-identical batch; details
                       # and the full-decode rate of the same graph under config.pipeline)
                       "roi_decode_fusion": bool(pipe_info and pipe_info.get("roi_decode_fusion")),
                       "cache_type": args.cache_type},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "measured_copy_ceiling_GBps": copy_ceiling, "frac_of_measured_ceiling": ach / copy_ceiling,
                         "note": "the entropy decoder's kernels are serial-bit decode loops, not HBM streams (SURVEY.md "
                                 "8(d)): the dominant one's fraction of the HBM peak says how far a latency chain is from a "
                                 "stream, not that bytes are wasted (traffic = counter bytes per launch of the profiled run)",
                         # whole_step (round 6): the algorithmic bytes of the work the step DOES - the sum of
                         # its kernels' figures, i.e. with the window decode only the blocks / pixels the crop keeps - over the
                         # step's duration.  The SURVEY 8(d) formula 6P + 3sP + 6O prices whole images and stays next to it.
                         "whole_step": {"algorithmic_bytes": step_bytes,
                                        "achieved_GBps": step_bytes / (step_ms * 1e-3) / 1e9,
                                        "frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "frac_of_measured_ceiling": step_bytes / (step_ms * 1e-3) / 1e9 / copy_ceiling,
                                        "survey_formula_bytes": post_entropy + (stream_bytes or 0),
                                        "survey_formula_frac": (post_entropy + (stream_bytes or 0)) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "per_kernel": {k: {"algorithmic_bytes": v[0], "avg_ms": v[1],
                                            "achieved_GBps": v[0] / (v[1] * 1e-3) / 1e9} for k, v in kern.items()}},
            "ceilings": {
                "hbm_post_entropy_images_per_s": B / (post_entropy / (HBM_PEAK_GBS * 1e9)),
                "pcie_gen5_x16_images_per_s": 64e9 / (stream_bytes / B) if args.huffman == "gpu" else None,
                "entropy_decode_images_per_s": (B / ((huffman_single_ms or huffman_total_ms) * 1e-3)
                                                if args.huffman == "gpu" else None),
                "entropy_decode_ms_per_batch_single_stream": huffman_single_ms,
                "note": "SURVEY.md 8(d): HBM bound of everything after the entropy decoder, H2D bound of the JPEG bytes "
                        "(64 GB/s), and the rate of the GPU entropy decoder alone = batch / (sum of its six kernels with ONE "
                        "batch in flight, this run); several batches in flight overlap its latency-bound synchronisation "
                        "pass with the other batches' kernels, which is how `value` can exceed it"},
            "entropy_decode": ({"symbols_per_batch": symbols, "symbols_per_s": symbols / (huffman_total_ms * 1e-3),
                                "bitstream_GBps": stream_bytes / (huffman_total_ms * 1e-3) / 1e9}
                               if args.huffman == "gpu" else None),
        }
        if pipe_info:
            line["config"]["pipeline"] = pipe_info
        else:
            line["e2e_host_huffman"] = {"huffman_s_per_batch": huffman_s, "host_threads": effective_cpu_count(),
                                        "note": "host entropy decoder on the same batches (one pass, thread pool): the "
                                                "CPU half of the hybrid variant (--huffman host); not part of `value`"}
    have_pipe_info = pipe_info is not None
    pipe_info = None
    import gc
    gc.collect()          # the headline pipeline (and with it the encoded-stream cache of the device) goes away here
    torch.cuda.empty_cache()
    if have_pipe_info and world == 1 and not args.no_e2e and rank == 0 and args.cache_type == "encoded":
        # Round 5: the same graph, the same streams, resident WITH their side information (cache_type="indexed": un-stuffed
        # bytes + 12 bytes of decoder state per 256-byte slice, left behind by the first decode).  `value` keeps its meaning
        # (streams resident as they are in the file, every epoch parses them anew); this is the rate when the epoch-invariant
        # half of the parse is not repeated.  A fresh cache: the one above held the streams without an index.
        from dali_amd import _backend
        threads = max(2, effective_cpu_count() * 3 // 4)
        cache_mb = max(64, int(2.2 * sum(len(e) for e in enc_all) / 2**20))
        pipe4 = resident_pipeline(root, B, dev_index, inflight, threads, cache_mb=cache_mb, cache_type="indexed")
        done = 0
        while _backend.encoded_cache_stats(dev_index)["streams"] < per_rank and done < 64 * nb:
            pipe4.run()
            done += 1
        for _ in range((inflight + 2) * nb + args.warmup):
            pipe4.run()
        used = _backend.encoded_cache_stats(dev_index)
        torch.cuda.synchronize()
        kernel_timing(12 * (args.steps + inflight + 2))
        kernel_timing(False)
        kernel_timing()
        kernel_timing(True)
        with quiet_gc():
            t_idx = time.perf_counter()
            for _ in range(args.steps):
                pipe4.run()
            pipe4._backend.wait_enqueued()
            torch.cuda.synchronize()
            t_idx = time.perf_counter() - t_idx
        kernel_timing(False)
        idx_times = kernel_timing()
        assert "jpeg_huffman_indexed" in pipe4.executed_kernels(), pipe4.executed_kernels()
        idx_alone = None
        if not args.no_side_legs:     # ... and what its launches cost with ONE batch in flight (same cache: pipe4 holds it)
            pipe5 = resident_pipeline(root, B, dev_index, 1, threads, cache_mb=cache_mb, cache_type="indexed")
            for _ in range(3 * nb):
                pipe5.run()
            torch.cuda.synchronize()
            kernel_timing(True)
            for _ in range(3 * nb):
                pipe5.run()
            torch.cuda.synchronize()
            kernel_timing(False)
            idx_alone = {k: ms for k, (calls, ms) in kernel_timing().items()}
            del pipe5
        ecs_total = stream_bytes_total             # entropy-coded bytes of the resident streams (scan analysis, exact)
        line["config"]["pipeline"]["resident_indexed"] = {
            "value": B * args.steps / t_idx, "unit": "images/s", "ms_per_step": 1e3 * t_idx / args.steps,
            "kernels": pipe4.executed_kernels(),
            "kernel_ms_in_schedule": {k: ms for k, (calls, ms) in idx_times.items()},
            "kernel_ms_single_stream": idx_alone,
            "resident_bytes_per_image": used["bytes_used"] / max(1, used["streams"]),
            "index_bytes_per_image": (used["bytes_used"] - ecs_total) / max(1, used["streams"]),
            "note": "decoders.image(cache_type='indexed'): epoch >= 2 launches PrepareKernel (code tables only), "
                    "IndexedSyncKernel (one decode per slice from its recorded entry state, DC symbols on the way, slices "
                    "outside the crop window skipped), BlockKernel, the colour kernel and the fused resample - no "
                    "un-stuffing, no relaxation, no hand-over check, no DC pass.  Bit-identical batches "
                    "(tests/test_gpu_jpeg_index.py, test_gpu_encoded_cache.py, test_gpu_headline.py).  "
                    "index_bytes_per_image = resident bytes beyond the entropy-coded segment itself (header, 64-byte "
                    "alignment, 12 bytes per 256-byte slice)"}
        del pipe4
        gc.collect()
        torch.cuda.empty_cache()

    if not args.no_e2e:
        import shutil
        import tempfile
        if world == 1:
            if root is None:
                root = tempfile.mkdtemp(prefix="dali_amd_bench_")
                write_dataset(root, enc_all)
            try:
                # what an end-to-end leg cannot exceed on this box: every encoded byte crosses the bus once per image
                h2d_peak = measured_h2d_ceiling(torch.device("cuda", dev_index))
                mean_file = float(np.mean([len(e) for e in enc_all]))

                def with_pcie(res):
                    ach = res["value"] * mean_file / 1e9
                    res["pcie"] = {"bound": "pcie", "achieved": ach, "peak": h2d_peak, "unit": "GB/s",
                                   "frac": ach / h2d_peak if h2d_peak else None, "bytes_per_image": mean_file,
                                   "images_per_s_at_peak": h2d_peak * 1e9 / mean_file if h2d_peak else None,
                                   "peak_source": "pinned host -> device copies of 256 MiB on one stream, measured in this run"}
                    return res
                line["e2e_pipeline"] = with_pcie(e2e_pipeline(root, B, local_rank))
                # ... and from the data set indexed offline (tools/jpeg2idx.py -> readers.file(index_path=...)): every file is read
                # as its container - headers + index entry - and decoded FROM the entry in this first sighting already: the
                # decoder's un-stuffing, relaxation, hand-over and DC passes never run; 5 % more bytes cross the bus
                idx_root = tempfile.mkdtemp(prefix="dali_amd_bench_idx_")
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import jpeg2idx
                    t_idx0 = time.perf_counter()
                    made, left = jpeg2idx.index_tree(root, idx_root, workers=effective_cpu_count(), quiet=True, threads=True)
                    t_idx0 = time.perf_counter() - t_idx0
                    box_bytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(idx_root) for f in fs)
                    res = e2e_pipeline(root, B, local_rank, index_path=idx_root)
                    mean_box = box_bytes / max(1, made)
                    ach = res["value"] * mean_box / 1e9
                    res["pcie"] = {"bound": "pcie", "achieved": ach, "peak": h2d_peak, "unit": "GB/s", "frac": ach / h2d_peak if h2d_peak else None,
                                   "bytes_per_image": mean_box, "images_per_s_at_peak": h2d_peak * 1e9 / mean_box if h2d_peak else None}
                    res["index"] = {"containers": made, "files_without": left, "container_MB": box_bytes / 1e6,
                                    "file_MB": sum(len(e) for e in enc_all) / 1e6, "build_s": t_idx0,
                                    "build_files_per_s_per_core": made / t_idx0 / effective_cpu_count() if t_idx0 else None}
                    res["note"] = ("e2e_pipeline over the same files with readers.file(index_path=...): indexed JPEG containers made by "
                                   "tools/jpeg2idx.py (host restatement of the position pass, byte-identical to the device-built index)")
                    line["e2e_pipeline_indexed"] = res
                    line["config"]["e2e_indexed_images_per_s"] = res["value"]
                except Exception as e:  # noqa: BLE001 - a side figure must not take the headline line with it
                    line["e2e_pipeline_indexed"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                finally:
                    shutil.rmtree(idx_root, ignore_errors=True)
                if variants:
                    # What real collections hold that the baseline set does not (VERDICT r05 missing 3): the headline graph,
                    # timed like `value`, on (i) per-file Huffman tables, (ii) a progressive / CMYK share, (iii) 12-megapixel
                    # outliers, (iv) a resident set beyond the 256 MB Infinity Cache, drawn with random_shuffle
                    vsteps = max(args.steps, 100)    # (a 20-step region is 9 ms: these side figures get 46 ms)
                    line["realistic"] = {}
                    for v in variants:
                        vroot = tempfile.mkdtemp(prefix=f"dali_amd_bench_{v}_")
                        try:
                            sizes = generate_variant_dataset(v, vroot, min(per_rank, 1024))
                            line["realistic"][v] = resident_variant_leg(args, vroot, len(sizes), sum(sizes), dev_index, vsteps)
                        except Exception as e:  # noqa: BLE001 - a side figure must not take the headline line with it
                            line["realistic"][v] = {"error": f"{type(e).__name__}: {e}"[:300]}
                        finally:
                            shutil.rmtree(vroot, ignore_errors=True)
                    if args.resident_copies > 1:
                        vroot = tempfile.mkdtemp(prefix="dali_amd_bench_4GB_", dir=os.path.dirname(root))
                        try:
                            write_linked_dataset(vroot, root, enc_all, args.resident_copies)
                            line["realistic"]["resident_4GB"] = resident_variant_leg(
                                args, vroot, args.resident_copies * len(enc_all), args.resident_copies * sum(len(e) for e in enc_all),
                                dev_index, max(args.steps, 200), random_shuffle=True)
                        except Exception as e:  # noqa: BLE001
                            line["realistic"]["resident_4GB"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                        finally:
                            shutil.rmtree(vroot, ignore_errors=True)
                    for key, v in (("value_distinct_dht", "distinct_dht"), ("value_mixed", "mixed"),
                                   ("value_large_images", "large"), ("value_resident_4GB", "resident_4GB")):
                        line["config"][key] = (line["realistic"].get(v) or {}).get("value")
                    line["config"]["resident_4GB_set_MB"] = (line["realistic"].get("resident_4GB") or {}).get("resident_MB")
                if args.side_legs:
                    line["e2e_pipeline_roi_decode"] = with_pcie(e2e_pipeline(root, B, local_rank, roi_decode=True))
                    line["e2e_pipeline_roi_decode"]["note"] = ("same, with decoders.image_random_crop -> resize -> "
                                                               "crop_mirror_normalize: only the crop window is decoded")
                    line["iterator"] = iterator_leg(args, root, enc_all, dev_index, max(20, min(args.steps, 200)))
                    line["iterator"]["fraction_of_value"] = line["iterator"]["value"] / line["value"]
                    gc.collect()
                    line["e2e_pipeline_decoder_cache"] = e2e_pipeline(root, B, local_rank, cache_mb=1024)
                    line["e2e_pipeline_decoder_cache"]["note"] = (
                        "same as e2e_pipeline with decoders.image(cache_size=1024, cache_type='threshold'): epoch >= 2 of a "
                        "data set whose decoded images fit in HBM.  The files are still read; decoded images are handed to "
                        "the fused resample kernel in place from the cache blob")
                if args.side_legs and args.emulate_local_world > 1:
                    # 8-GPU readiness without 8 GPUs: this rank with the share of the host an N-rank node would leave
                    # it - the process (and every thread it creates from here on) bound to usable / N CPUs, thread pools
                    # sized for them - so that the per-rank end-to-end rate under that split is a measured number
                    K = args.emulate_local_world
                    share = max(1, effective_cpu_count() // K)
                    allowed = sorted(os.sched_getaffinity(0))
                    near = device_local_cpus(dev_index) or allowed      # a rank's share lies on its GPU's NUMA node
                    os.sched_setaffinity(0, set(near[:share]))
                    try:
                        res = e2e_pipeline(root, args.e2e_batch, local_rank, iters=100, threads=max(2, share))
                    finally:
                        os.sched_setaffinity(0, set(allowed))
                    res["emulated_local_world"] = K
                    res["cpus"] = share
                    res["note"] = (f"e2e_pipeline at batch {args.e2e_batch} with the host share of one rank of a {K}-GPU node: "
                                   f"{share} of the {effective_cpu_count()} usable CPUs (affinity mask), thread pools of "
                                   f"{max(2, share)}; the GPU is not shared - a host-side bound for configs[4], not a scaling "
                                   "measurement")
                    line[f"e2e_pipeline_local_world{K}"] = res
            finally:
                shutil.rmtree(root, ignore_errors=True)
        else:
            # BASELINE configs[4]: every rank runs the whole pipeline on its shard of the SHARED data set directory
            # (readers.file(shard_id=rank, num_shards=world)), batch 512 per GPU, host threads split between the
            # ranks of the node and pinned to their GPU's NUMA node (set_affinity)
            if root is None:
                root = os.path.join(tempfile.gettempdir(), f"dali_amd_bench_shared_{os.environ.get('MASTER_PORT', '0')}")
                write_dataset(root, enc_all, first_index=rank * per_rank)
                barrier()
            threads = max(2, effective_cpu_count() // max(1, local_world))
            if rank == 0:
                print(f"bench: {local_world} ranks share {effective_cpu_count()} usable host cores: {threads} worker threads "
                      f"per rank" + (" - fewer than 4: the end-to-end leg will be host-bound" if threads < 4 else ""),
                      file=sys.stderr)
            res = e2e_pipeline(root, args.e2e_batch, dev_index, iters=100, threads=threads, shard_id=rank,
                               num_shards=world, sync=barrier, set_affinity=True)
            t = torch.tensor([res["elapsed_s"]], dtype=torch.float64, device="cpu" if single_device_test else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            barrier()
            if rank == 0:
                res["elapsed_s_max_over_ranks"] = float(t.item())
                res["value"] = world * res["batch"] * res["iters"] / float(t.item())
                res["note"] = ("configs[4]: whole-job images/s of dali_amd.Pipeline on every rank, readers.file(shard_id=rank, "
                               f"num_shards={world}) over one shared data set of {world * per_rank} files, batch "
                               f"{res['batch']}/GPU, {threads} host threads per rank, set_affinity=True; barrier + max over ranks")
                line["e2e_pipeline_sharded"] = res
                shutil.rmtree(root, ignore_errors=True)
    if args.no_e2e and root is not None:
        import shutil
        barrier()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)
    if rank == 0:
        # the numbers the round's story rests on, as top-level scalars of `config` (a parse that keeps scalars keeps them)
        cfg, pl = line["config"], line["config"].get("pipeline") or {}
        cfg["whole_step_frac"] = line["roofline"]["whole_step"]["frac"]
        cfg["decoder_single_stream_ms"] = line["ceilings"]["entropy_decode_ms_per_batch_single_stream"]
        for key, src in (("e2e_images_per_s", line.get("e2e_pipeline")), ("iterator_images_per_s", line.get("iterator")),
                         ("e2e_local_world8_images_per_s", line.get("e2e_pipeline_local_world8")),
                         ("e2e_sharded_images_per_s", line.get("e2e_pipeline_sharded")),
                         ("resident_full_decode_images_per_s", pl.get("resident_full_decode")),
                         ("resident_roi_decode_images_per_s", pl.get("resident_roi_decode")),
                         ("resident_indexed_images_per_s", pl.get("resident_indexed"))):
            cfg[key] = src["value"] if isinstance(src, dict) and "value" in src else None
        if isinstance(pl.get("resident_indexed"), dict):
            cfg["index_bytes_per_image"] = pl["resident_indexed"].get("index_bytes_per_image")
        if isinstance(line.get("e2e_pipeline"), dict) and "pcie" in line["e2e_pipeline"]:
            cfg["e2e_pcie_frac"] = line["e2e_pipeline"]["pcie"]["frac"]
            cfg["e2e_images_per_s_at_pcie_peak"] = line["e2e_pipeline"]["pcie"]["images_per_s_at_peak"]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(enc_all[:B])
            if args.side_legs:
                line["cpu_baseline_pillow"] = pillow_baseline(enc_all[:B])
        if world == 1 and args.side_legs:
            # BASELINE.json configs[2] and configs[3], compact: the same legs `--workload heavy_aug` / `--workload audio`
            # run, each with its own roofline and oracle cpu_baseline (same cores, same run)
            import gc
            gc.collect()
            side_steps = max(60, min(args.steps, 100))   # (a 20-step region from an idle GPU is half ramp-up for these legs: 155 k against 240 k)
            for key, leg in (("heavy_aug", bench_heavy_aug), ("audio", bench_audio)):
                try:
                    line[key] = leg(args, device, steps=side_steps, cpu_seconds=6.0)
                except Exception as e:  # noqa: BLE001 - a side leg must not take the headline line with it
                    line[key] = {"error": f"{type(e).__name__}: {e}"}
                gc.collect()
        # (more scalars for a reader that keeps `config`'s scalars only: the side legs' rates, the host cost of the file-fed
        # legs, which reader fed the rank-of-eight leg, the dominant kernel's duration with one batch in flight)
        def scalar(obj, *path):
            for k in path:
                obj = obj.get(k) if isinstance(obj, dict) else None
            return obj if isinstance(obj, (int, float, bool)) else None
        cfg["heavy_aug_images_per_s"] = scalar(line, "heavy_aug", "value")
        cfg["audio_utterances_per_s"] = scalar(line, "audio", "value")
        cfg["e2e_cpu_ms_per_batch"] = scalar(line, "e2e_pipeline", "cpu_ms_per_batch")
        cfg["e2e_local_world8_cpu_ms_per_batch"] = scalar(line, "e2e_pipeline_local_world8", "cpu_ms_per_batch")
        cfg["e2e_local_world8_reader_zero_copy"] = scalar(line, "e2e_pipeline_local_world8", "reader_zero_copy")
        cfg["sync_kernel_single_stream_ms"] = scalar(pl, "single_stream_kernel_ms", "SyncKernel")
        if args.driver == "pipeline":
            cfg["resident_set_MB"] = stream_bytes_total / 1e6
        emit(line, full=args.full_line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
