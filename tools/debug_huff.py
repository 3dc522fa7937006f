"""Development aid: runs the GPU entropy decoder on the streams given on the command line (or a built-in q5 case) and
dumps the decoder's scratch arrays (block starts per segment, per-block position / DC / segment) to an .npz under
gpurun_out/ so that they can be compared with tools/huff_model.cpp off-line."""
import sys

import numpy as np
import torch

from dali_amd import backend as B
from dali_amd.testing import encode_jpeg, synth_image

K_SEG_LANES, K_SLICE, K_TILE = 244, 256, 16384


def layout(ecs_len, num_tiles, num_segments, total_blocks):
    al = lambda v, a: (v + a - 1) // a * a  # noqa: E731
    o = 16
    L = {}
    L["tile_kept"] = o; o += al(4 * num_tiles, 16)
    L["clean"] = o; o += al(ecs_len + 256, 16)
    L["tables"] = o; o += 2 * 4 * 2048 + 2 * 4 * 512 + 16 + 16 + 288 + 288 + 1024 + 16
    L["sync_tables"] = o; o += 4 * 4 * 2048 + 2 * 4 * 512 + 16 + 16 + 288 + 288 + 1024 + 16
    L["lanes"] = o; o += al(24 * num_segments * K_SEG_LANES, 16)
    L["segs"] = o; o += al(32 * num_segments, 16)
    L["seg_cap"] = min(K_SEG_LANES * (K_SLICE * 8 // 4 + 32), total_blocks + 128)
    L["seg_starts"] = o; o += al(4 * num_segments * L["seg_cap"], 16)
    L["blk_pos"] = o; o += al(4 * total_blocks, 16)
    L["blk_dc"] = o; o += al(4 * total_blocks, 16)
    L["blk_seg"] = o; o += al(2 * total_blocks, 16)
    L["total"] = al(o, 256)
    return L


def main():
    if len(sys.argv) > 1:
        enc = [open(f, "rb").read() for f in sys.argv[1:]]
    else:
        rng = np.random.default_rng(7)
        enc = []
        for (h, w) in [(1, 1), (8, 8), (17, 23), (33, 47), (100, 75)]:
            for kw in [dict(subsampling="4:4:4"), dict(subsampling="4:2:2"), dict(subsampling="4:2:0"),
                       dict(subsampling="4:1:1"), dict(subsampling="4:2:0", progressive=True),
                       dict(subsampling="4:2:0", restart_marker_blocks=3), dict(subsampling="4:2:0", quality=100),
                       dict(subsampling="4:2:0", quality=5)]:
                enc.append(encode_jpeg(synth_image(rng, h, w), **({"quality": 85} | kw)))
            enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80))
            enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80, progressive=True))
        enc = [enc[47]]
        open("gpurun_out/debug_huff_input.jpg", "wb").write(enc[0])
    plan = B.JpegBatchPlan(enc)
    coef_host = torch.empty(plan.coef_elems, dtype=torch.int16)
    plan.entropy_decode(coef_host, num_threads=1)
    plan.analyze_scans()
    plan.upload_streams(torch.device("cuda:0"))
    coef = torch.zeros(plan.coef_elems, dtype=torch.int16, device="cuda:0")
    status = plan.run_gpu_huffman(coef)
    torch.cuda.synchronize()
    print("status", status.cpu().numpy(), "coef equal", bool(torch.equal(coef.cpu(), coef_host)))
    d, ntiles, nsegs, nbwg = plan.huffman_descs(coef)
    scratch = plan._huff_ws["scratch"].cpu().numpy()
    out = {}
    for j in range(len(plan._huff_sel)):
        dj = d[j]
        L = layout(int(dj["ecs_len"]), int(dj["num_tiles"]), int(dj["num_segments"]), int(dj["total_blocks"]))
        base = int(plan._scratch_off[j])
        s = scratch[base:base + L["total"]]
        hdr = s[:16].view(np.int32)
        nb, ns, cap = int(dj["total_blocks"]), int(dj["num_segments"]), L["seg_cap"]
        out[f"hdr{j}"] = hdr.copy()
        out[f"segs{j}"] = s[L["segs"]:L["segs"] + 32 * ns].view(np.int32).reshape(ns, 8).copy()
        out[f"lanes{j}"] = s[L["lanes"]:L["lanes"] + 24 * ns * K_SEG_LANES].view(np.int64).reshape(ns, K_SEG_LANES, 3).copy()
        out[f"starts{j}"] = s[L["seg_starts"]:L["seg_starts"] + 4 * ns * cap].view(np.uint32).reshape(ns, cap).copy()
        out[f"blk_pos{j}"] = s[L["blk_pos"]:L["blk_pos"] + 4 * nb].view(np.uint32).copy()
        out[f"blk_dc{j}"] = s[L["blk_dc"]:L["blk_dc"] + 4 * nb].view(np.int32).copy()
        out[f"blk_seg{j}"] = s[L["blk_seg"]:L["blk_seg"] + 2 * nb].view(np.uint16).copy()
        print(j, "hdr", hdr, "segs", out[f"segs{j}"][:, 2:7].tolist(), "blocks", nb)
        bad = np.nonzero(coef.cpu().numpy()[:plan.coef_elems] != coef_host.numpy()[:plan.coef_elems])[0]
        print("   first mismatching coefficient elements", bad[:10], "of", len(bad))
    np.savez_compressed("gpurun_out/debug_huff.npz", **out)


if __name__ == "__main__":
    main()
