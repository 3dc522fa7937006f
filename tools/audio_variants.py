"""Development aid: host time of decoders.audio in the configs[3] pipeline of bench.py under different thread counts /
affinity (python tools/audio_variants.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


class A:
    inflight, warmup, steps, no_cpu_baseline = 5, 4, 40, True


dev = torch.device("cuda", 0)
for threads in (12, 6, 3, 16):
    bench.effective_cpu_count = lambda t=threads: t * 4 // 3
    r = bench.bench_audio(A, dev, steps=40, cpu_seconds=0)
    print(json.dumps({"threads": r["config"]["host_threads"], "utt_per_s": round(r["value"]), "ms_per_step": round(r["ms_per_step"], 3),
                      "host_ms": r["config"]["host_ms_per_operator"], "copy_ms": r["roofline"]["operator_device_ms"]}), flush=True)
