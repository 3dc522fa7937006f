#!/usr/bin/env python
"""Headline benchmark: images/s on the JPEG -> RandomResizedCrop -> CropMirrorNormalize pipeline
(224x224, batch 256 per GPU, fp16 CHW output) on N MI355X, plus the roofline of the dominant
kernel and the CPU baseline (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch whose inputs are already resident in HBM:
  JPEG entropy-coded segments (bytes)  ->  zero-fill + un-stuff + Huffman decode  [UnstuffKernel, HuffmanDecodeKernel]
                                       ->  dequant + IDCT                         [JpegIdctKernel]
                                       ->  chroma upsample + YCbCr->RGB           [JpegColorKernel]
  host Philox crop windows + mirror bits (in the timed region, host side)
                                       ->  fused resample + CMN                   [ResampleKernel]
Descriptor-table construction and upload are inside the timed region; the header parse / scan analysis of the
(static) synthetic batch is done once at start-up.  `--huffman host` benchmarks the hybrid variant instead: the
coefficient blocks are produced once by the host entropy decoder and are the HBM-resident input (the host Huffman
time is then reported as `e2e_host_huffman`, never inside `value`).

Multi-GPU: sample sharding exactly like readers.file(shard_id, num_shards): rank r owns images
[r*B, (r+1)*B) of the synthetic dataset; no collective on the data path ("scaling": "weak").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

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


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, produced by
    tools/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH_SIZE correction).  bench.py
    cannot run rocprofv3 around itself, so this is the figure of the profiled run of this same command."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel, {})
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


def make_dataset(first_index, count):
    """Synthetic ImageNet-like JPEGs.  Image i depends only on its global index."""
    from tests.util import synth_jpeg_batch
    out = []
    for i in range(first_index, first_index + count):
        rng = np.random.default_rng([1234, i])
        out.extend(synth_jpeg_batch(rng, 1))
    return out


class HotPath:
    """Device-resident batch + the per-step launch sequence."""

    def __init__(self, enc, device, seed=1234, huffman="gpu", inflight=2, fused_idct=True):
        import torch
        from dali_amd import backend as B
        self.torch, self.B = torch, B
        self.device = device
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
            self.coef_dev = torch.empty(self.plan.coef_elems, dtype=torch.int16, device=device)
            # one-time self check: the GPU entropy decoder reproduces the host decoder's coefficients exactly
            status = self.plan.run_gpu_huffman(self.coef_dev)
            torch.cuda.synchronize()
            self.plan.check_gpu_status(status)
            if not torch.equal(self.coef_dev.cpu(), self.coef_host):
                raise SystemExit("bench: GPU Huffman output differs from the host entropy decoder")
            if self.fused_idct:   # ... and its fused dequantisation + IDCT the stand-alone IDCT kernel's planes
                ref_planes = torch.zeros(self.plan.plane_bytes, dtype=torch.uint8, device=device)
                got_planes = torch.zeros_like(ref_planes)
                scratch_rgb = torch.empty(self.plan.out_bytes, dtype=torch.uint8, device=device)
                B.jpeg_gpu_stage(self.plan, self.coef_dev, ref_planes, scratch_rgb)
                status = self.plan.run_gpu_huffman(None, planes_dev=got_planes)
                torch.cuda.synchronize()
                self.plan.check_gpu_status(status)
                if not torch.equal(ref_planes, got_planes):
                    raise SystemExit("bench: fused Huffman+IDCT planes differ from the IDCT kernel's")
                del ref_planes, got_planes, scratch_rgb
        else:
            self.coef_dev = self.coef_host.to(device)
        # `inflight` batches are processed concurrently, each on its own HIP stream with its own buffers (what the
        # reference's executor does with prefetch_queue_depth=2: the decode of batch i+1 overlaps the resize of batch i)
        self.slots = []
        for k in range(inflight):
            slot = {"stream": torch.cuda.Stream(device=device),
                    "coef": self.coef_dev if k == 0 else torch.empty_like(self.coef_dev),
                    "planes": torch.empty(self.plan.plane_bytes, dtype=torch.uint8, device=device),
                    "rgb": torch.empty(self.plan.out_bytes, dtype=torch.uint8, device=device),
                    "out": torch.empty((self.n, 3, 224, 224), dtype=torch.float16, device=device),
                    "ws": self.plan.new_huffman_workspace(device) if huffman == "gpu" else None}
            if huffman != "gpu" and k > 0:
                slot["coef"].copy_(self.coef_dev)
            slot["views"] = self.plan.output_views(slot["rgb"])
            slot["image_table"] = B.ImageTable(slot["views"])
            self.slots.append(slot)
        self.views = self.slots[0]["views"]
        self.out = self.slots[0]["out"]
        self.rgb = self.slots[0]["rgb"]
        self.host_s = 0.0
        self.kernel_events = []
        self.shapes = np.array([v.shape[:2] for v in self.views], np.int32)
        self.rrc_master = B.philox_state(seed)
        self.flip_master = B.philox_state(seed + 1)
        self.mean, self.inv_std = B.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255],
                                                  [0.229 * 255, 0.224 * 255, 0.225 * 255])
        self.events = None
        self.last_rois = None
        # algorithmic bytes per launch
        P = sum(int(s[0]) * int(s[1]) for s in self.shapes)
        self.pixels = P
        self.bytes_idct = 3 * self.plan.coef_elems            # 2 B coefficient in + 1 B sample out
        self.bytes_color = self.plan.plane_bytes + 3 * P       # planes in + RGB out

    def step(self, record=None, index=0):
        """Enqueues one pass of the hot path over the batch on the stream of slot index % inflight."""
        torch, B = self.torch, self.B
        from dali_amd import _capi as capi
        ev = record
        slot = self.slots[index % len(self.slots)]
        t0 = time.perf_counter()
        with torch.cuda.stream(slot["stream"]):
            if self.huffman == "gpu":
                ke = None
                if ev:   # per-kernel events of the entropy decoder: created before the timed region, one set per step
                    pool = getattr(self, "kernel_event_pool", ())
                    k = len(self.kernel_events)
                    ke = pool[k] if k < len(pool) else B.KernelEvents(len(B.HUFFMAN_KERNELS))
                    self.kernel_events.append(ke)
                self.plan.run_gpu_huffman(slot["coef"], events=ev[5:7] if ev else None, ws=slot["ws"],
                                          kernel_events=ke.handles if ke else None,
                                          planes_dev=slot["planes"] if self.fused_idct else None)
            B.jpeg_gpu_stage(self.plan, slot["coef"], slot["planes"], slot["rgb"], split_events=ev[1:2] if ev else None,
                             start_event=ev[0] if ev else None, fused_huffman=self.fused_idct)
            anchors, crops = B.random_crop_batch(self.rrc_master, self.shapes)
            mirror = B.coin_flip_batch(self.flip_master, self.n, 0.5)
            self.rrc_master.ctr[1] += self.n   # OperatorWithRng::Advance
            self.flip_master.ctr[1] += self.n
            rois = np.concatenate([anchors, anchors + crops], 1).astype(np.float32)
            self.last_rois = (anchors, crops)
            if ev:
                ev[2].record()
            B.resample_batch(slot["image_table"], (224, 224), rois=rois, out_dtype=capi.FLOAT16, out_layout=capi.LAYOUT_CHW,
                             mean=self.mean, inv_std=self.inv_std, mirror=mirror, out=slot["out"],
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


def e2e_pipeline(enc, device_id, iters=30, threads=None, roi_decode=False, cache_mb=0):
    """The same hot path through the product's DALI-style pipeline (C++ host framework): readers.file (page cache)
    -> decoders.image(mixed: header parse + scan analysis on the host thread pool, H2D of the entropy-coded
    segments, GPU Huffman/IDCT/colour) -> random_resized_crop + crop_mirror_normalize (fused).  PCIe-inclusive and
    host-inclusive, therefore reported next to `value`, never as `value`."""
    import shutil
    import tempfile
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    threads = threads or effective_cpu_count()
    root = tempfile.mkdtemp(prefix="dali_amd_bench_")
    try:
        os.makedirs(os.path.join(root, "c0"))
        for i, e in enumerate(enc):
            with open(os.path.join(root, "c0", f"{i:05d}.jpg"), "wb") as f:
                f.write(e)
        # four sequential stages (file reads | parse + staging | H2D | kernels) need several batches in flight to overlap
        pipe = Pipeline(batch_size=len(enc), num_threads=threads, device_id=device_id, seed=1234, prefetch_queue_depth=4)
        with pipe:
            jpegs, labels = fn.readers.file(file_root=root, name="Reader")
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
        for _ in range(10):   # every ring slot allocates its pinned / device buffers on first use
            pipe.run()
        t0 = time.perf_counter()
        for _ in range(iters):
            pipe.run()
        el = time.perf_counter() - t0
        return {"value": iters * len(enc) / el, "unit": "images/s", "ms_per_batch": 1e3 * el / iters,
                "host_threads": threads, "prefetch_queue_depth": 4, "kernels": pipe.executed_kernels(),
                "note": "dali_amd.Pipeline end to end from encoded files in the page cache (file read, header parse, "
                        "H2D of the JPEG bytes on a copy stream, all device stages, fp16 CHW batch on the device)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def bench_heavy_aug(args, device):
    """configs[2]: warp_affine + gaussian_blur(sigma=3) + color_twist + erase on 512x512 u8 images, batch 128,
    images resident in HBM.  One JSON line with per-kernel achieved GB/s (algorithmic bytes: 3*512*512 in + out)."""
    import torch
    from dali_amd import backend as B
    from tests.util import synth_image
    n = 128
    rng = np.random.default_rng(1234)
    base = [torch.from_numpy(synth_image(rng, 512, 512)).to(device) for _ in range(8)]
    imgs = [base[i % 8].clone() for i in range(n)]

    def params():
        mats = []
        for _ in range(n):
            t, s = np.deg2rad(rng.uniform(-30, 30)), rng.uniform(0.8, 1.2)
            c, sn = np.cos(t) / s, np.sin(t) / s
            m = np.array([[c, -sn, 0], [sn, c, 0]], np.float32)
            m[0, 2] = 256 - m[0, 0] * 256 - m[0, 1] * 256
            m[1, 2] = 256 - m[1, 0] * 256 - m[1, 1] * 256
            mats.append(m)
        tw = [B.color_twist_matrix(rng.uniform(-30, 30), rng.uniform(.7, 1.3), 1.0, rng.uniform(.8, 1.2),
                                   rng.uniform(.8, 1.2)) for _ in range(n)]
        regs = []
        for _ in range(n):
            a, sh = rng.uniform(0, .7, 2) * 512, rng.uniform(.1, .3, 2) * 512
            regs.append([(int(a[0]), int(a[1]), int(a[0] + sh[0]), int(a[1] + sh[1]))])
        return mats, [t[0] for t in tw], [t[1] for t in tw], regs

    names = ["WarpAffineKernel", "GaussianBlurKernel", "PointwiseKernel(color_twist)", "PointwiseKernel(erase)"]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]

    def step(e=None):
        mats, tm, to, regs = params()
        if e: e[0].record()
        x = B.warp_affine_batch(imgs, mats, fill_value=0.0)
        if e: e[1].record()
        x = B.gaussian_blur_batch(x, sigma=3.0)
        if e: e[2].record()
        x = B.pointwise_batch(x, tm, to)
        if e: e[3].record()
        x = B.pointwise_batch(x, regions=regs, fill=(0.0,))
        if e: e[4].record()
        return x

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per = {}
    bytes_per = 2 * 3 * 512 * 512 * n
    for j, nm in enumerate(names):
        ms = float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev]))
        per[nm] = {"algorithmic_bytes": bytes_per, "avg_ms_incl_desc_upload": ms, "achieved_GBps": bytes_per / (ms * 1e-3) / 1e9}
    dom = max(per, key=lambda k: per[k]["avg_ms_incl_desc_upload"])
    print(json.dumps({"metric": "images/sec heavy-aug 512^2 b128 (warp_affine+gaussian_blur(sigma=3)+color_twist+erase)",
                      "value": n * args.steps / el, "unit": "images/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "u8 in/out, f32 arithmetic", "data": "synthetic",
                      "config": {"workload": "configs[2]: 128 x 512x512x3 u8 resident in HBM"},
                      "roofline": {"bound": "hbm", "kernel": dom, "achieved": per[dom]["achieved_GBps"],
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per[dom]["achieved_GBps"] / HBM_PEAK_GBS,
                                   "traffic": None, "per_kernel": per}}))


def bench_audio(args, device):
    """configs[3]: spectrogram(nfft=1024, step 256) -> mel_filter_bank(80) -> to_decibels on 64 signals of 8-16 s at
    16 kHz, signals resident in HBM."""
    import torch
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(1234)
    n = 64
    sigs = [rng.normal(0, 0.1, int(rng.uniform(8, 16) * 16000)).astype(np.float32) for _ in range(n)]
    pipe = Pipeline(batch_size=n, num_threads=4, device_id=device.index or 0, prefetch_queue_depth=1, exec_async=False)
    with pipe:
        x = fn.external_source(name="x")
        spec = fn.spectrogram(x.gpu(), nfft=1024, window_length=1024, window_step=256)
        mel = fn.mel_filter_bank(spec, nfilter=80, sample_rate=16000.0, freq_high=8000.0)
        pipe.set_outputs(fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0))
    pipe.build()
    frames = sum(len(s) // 256 + 1 for s in sigs)
    for _ in range(args.warmup):
        pipe.feed_input("x", sigs)
        pipe.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.feed_input("x", sigs)
        pipe.run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    samples = sum(len(s) for s in sigs)
    algo = {"SpectrogramKernel": 4 * samples + 4 * 513 * frames, "MelKernel": 4 * 513 * frames + 4 * 80 * frames,
            "DecibelKernel": 8 * 80 * frames}
    print(json.dumps({"metric": "utterances/sec spectrogram(1024)->mel(80)->dB b64 (incl. H2D of the signals)",
                      "value": n * args.steps / el, "unit": "utterances/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "configs[3]: 64 mono signals, 16 kHz, 8-16 s", "frames": frames,
                                 "mel_gemm_flops": 2 * 80 * 513 * frames},
                      "roofline": {"bound": "hbm", "kernel": "see profiles/ (rocprofv3 kernel trace)", "achieved": None,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                   "algorithmic_bytes": algo}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end dali_amd.Pipeline leg")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches processed concurrently, each on its own HIP stream (executor prefetch depth)")
    ap.add_argument("--huffman", default="gpu", choices=["gpu", "host"],
                    help="gpu: the step starts from JPEG bytes in HBM (default); host: from host-decoded coefficient blocks")
    ap.add_argument("--no-fused-idct", action="store_true",
                    help="store the coefficients and run the stand-alone IDCT kernel (the GPU entropy decoder's default is "
                         "to dequantise + inverse-transform the blocks itself)")
    ap.add_argument("--workload", default="imagenet", choices=["imagenet", "heavy_aug", "audio"],
                    help="imagenet = the headline metric (default); heavy_aug / audio = configs[2] / configs[3] side benches")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.workload == "heavy_aug":
        return bench_heavy_aug(args, device)
    if args.workload == "audio":
        return bench_audio(args, device)

    B = args.batch
    enc = make_dataset(rank * B, B)  # shard `rank` of `world` (contiguous, like loader.cc:78-87)
    hp = HotPath(enc, device, huffman=args.huffman, inflight=max(1, args.inflight), fused_idct=not args.no_fused_idct)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        hp.step(index=w)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(args.steps)]
    if args.huffman == "gpu":
        from dali_amd import backend as _backend
        hp.kernel_event_pool = [_backend.KernelEvents(len(_backend.HUFFMAN_KERNELS)) for _ in range(args.steps)]
    resample_bytes = []
    barrier()
    hp.host_s = 0.0
    t0 = time.perf_counter()
    for k in range(args.steps):
        _, crops = hp.step(record=ev[k], index=k)
        resample_bytes.append(hp.resample_bytes(crops))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel average durations from the events recorded inside the timed region
    ms_idct = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_color = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ms_resample = float(np.mean([e[3].elapsed_time(e[4]) for e in ev]))
    kern = {
        "JpegColorKernel": (hp.bytes_color, ms_color),
        "ResampleKernel": (float(np.mean(resample_bytes)), ms_resample),
    }
    if not hp.fused_idct:
        kern["JpegIdctKernel"] = (hp.bytes_idct, ms_idct)
    if args.huffman == "gpu":
        # per-kernel durations of the entropy decoder (events recorded between its launches, same stream)
        per = np.array([ke.elapsed_ms() for ke in hp.kernel_events]).mean(0)
        sb, ce = hp.plan.stream_bytes, hp.plan.coef_elems
        symbols = hp.plan.huffman_symbol_count(hp.slots[0]["ws"])   # one 32-bit record per symbol
        rec = 4 * symbols
        # PrepareKernel: the stream read once (byte counts) + the 60 KB of code tables it writes per stream
        huff_bytes = {"PrepareKernel": sb + 60 * 1024 * B, "UnstuffScatterKernel": 2 * sb,
                      "SyncKernel": sb, "PropagateKernel": 0, "WriteKernel": sb + rec, "DcScanKernel": 0,
                      # records in; coefficients out, or (fused dequantisation + IDCT) the 8-bit samples
                      "ExpandKernel": rec + (ce if hp.fused_idct else 2 * ce)}
        from dali_amd.backend import HUFFMAN_KERNELS
        for name, ms in zip(HUFFMAN_KERNELS, per):
            kern[name] = (huff_bytes[name], float(ms))
        huffman_total_ms = float(np.mean([e[5].elapsed_time(e[6]) for e in ev]))
    dominant = max(kern, key=lambda k: kern[k][1])
    ach = kern[dominant][0] / (kern[dominant][1] * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(dominant)

    if rank == 0:
        value = world * B * args.steps / elapsed
        step_ms = 1e3 * elapsed / args.steps
        step_bytes = float(sum(v[0] for v in kern.values()))
        copy_ceiling = measured_copy_ceiling(device)
        post_entropy = hp.bytes_idct + hp.bytes_color + float(np.mean(resample_bytes))
        line = {
            "metric": "images/sec JPEG->RRC->CMN 224^2 b256 per GPU",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 decode, f32 resample, f16 out", "data": "synthetic",
            "config": {"workload": "configs[1]: HIP JPEG " + ("Huffman decode -> " if args.huffman == "gpu" else "") +
                                   ("(fused) " if hp.fused_idct else "") + "dequant+IDCT -> upsample+YCbCr->RGB -> fused "
                                   "RandomResizedCrop+CropMirrorNormalize, 224x224, batch=256/GPU, fp16 CHW out; "
                                   "ImageNet-like synthetic JPEGs (seed 1234), inputs = " +
                                   ("JPEG entropy-coded segments (bytes) resident in HBM" if args.huffman == "gpu" else
                                    "host-entropy-decoded coefficient blocks resident in HBM"),
                       "huffman": args.huffman, "fused_dequant_idct": hp.fused_idct, "batches_in_flight": len(hp.slots),
                       "huffman_ms_per_batch(7 kernels)": huffman_total_ms if args.huffman == "gpu" else None,
                       "host_ms_per_step": 1e3 * hp.host_s / args.steps,
                       "jpeg_bytes_per_batch": getattr(hp.plan, "stream_bytes", None),
                       "global_batch": world * B, "parallelism": f"shard{world} (shard_id/num_shards, no collective)",
                       "pixels_per_batch": hp.pixels},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "measured_copy_ceiling_GBps": copy_ceiling, "frac_of_measured_ceiling": ach / copy_ceiling,
                         "note": "the dominant kernel of the GPU entropy decoder is a serial-bit latency chain, not an "
                                 "HBM stream (SURVEY.md 8(d)); whole_step prices all kernels of one pass together",
                         "whole_step": {"algorithmic_bytes": step_bytes, "achieved_GBps": step_bytes / (step_ms * 1e-3) / 1e9,
                                        "frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "frac_of_measured_ceiling": step_bytes / (step_ms * 1e-3) / 1e9 / copy_ceiling},
                         "per_kernel": {k: {"algorithmic_bytes": v[0], "avg_ms": v[1],
                                            "achieved_GBps": v[0] / (v[1] * 1e-3) / 1e9} for k, v in kern.items()}},
            "ceilings": {
                "hbm_post_entropy_images_per_s": B / (post_entropy / (HBM_PEAK_GBS * 1e9)),
                "pcie_gen5_x16_images_per_s": 64e9 / (hp.plan.stream_bytes / B) if args.huffman == "gpu" else None,
                "entropy_decode_images_per_s": B / (huffman_total_ms * 1e-3) if args.huffman == "gpu" else None,
                "note": "SURVEY.md 8(d): HBM bound of everything after the entropy decoder, H2D bound of the JPEG bytes "
                        "(64 GB/s), and the measured rate of the GPU entropy decoder alone (all 7 kernels, this run)"},
            "entropy_decode": ({"symbols_per_batch": symbols, "symbols_per_s": symbols / (huffman_total_ms * 1e-3),
                                "bitstream_GBps": hp.plan.stream_bytes / (huffman_total_ms * 1e-3) / 1e9}
                               if args.huffman == "gpu" else None),
            "e2e_host_huffman": {"huffman_s_per_batch": hp.huffman_s,
                                 "host_threads": effective_cpu_count(),
                                 "note": "host entropy decoder on the same batch (one pass, thread pool): the CPU half "
                                         "of the hybrid variant (--huffman host); not part of `value`"},
        }
        if world == 1 and not args.no_e2e:
            del hp
            torch.cuda.empty_cache()
            line["e2e_pipeline"] = e2e_pipeline(enc, local_rank)
            line["e2e_pipeline_roi_decode"] = e2e_pipeline(enc, local_rank, roi_decode=True)
            line["e2e_pipeline_roi_decode"]["note"] = ("same, with decoders.image_random_crop -> resize -> "
                                                       "crop_mirror_normalize: only the crop window is decoded")
            line["e2e_pipeline_decoder_cache"] = e2e_pipeline(enc, local_rank, iters=100, cache_mb=512)
            line["e2e_pipeline_decoder_cache"]["note"] = (
                "same as e2e_pipeline with decoders.image(cache_size=512, cache_type='threshold'): epoch >= 2 of a data "
                "set whose decoded images fit in HBM (here: the one batch).  The files are still read; decoded images "
                "are handed to the fused resample kernel in place from the cache blob")
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(enc)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
