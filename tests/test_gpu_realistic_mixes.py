"""What real collections hold that the synthetic baseline set does not, at batch scale and against the oracle (VERDICT r05
missing 3; the reference takes all of these in ONE batch: dali/operators/imgcodec/image_decoder.h:613-880, :826-834):
  distinct_dht  every file brings its own optimised Huffman tables (256 distinct table sets per batch)
  mixed         5 % progressive + 2 % CMYK + 5 % grayscale among baseline streams: device decode, host coefficient decode
                and host pixel decode side by side in one batch
  large         2 % 12-megapixel images among ImageNet-sized ones
The graph is bench.resident_pipeline - the one `value` is measured on - at batch 256 with five batches in flight, run for
three epochs so that the later ones come from the resident streams; every iteration is compared bit for bit with
decode -> RandomResizedCrop -> CropMirrorNormalize of the oracle.  bench.py times the same data sets
(config.value_distinct_dht / value_mixed / value_large_images)."""
import gc
import io

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


def _decode(e):
    inf = O.jpeg_info(e)
    if inf["ncomp"] == 4:     # CMYK: the pin is Pillow's conversion (tests/test_output_types.py)
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(e)).convert("RGB"))
    return O.jpeg_decode_rgb(e)


def _oracle_batch(images, it, mean, inv):
    shapes = [im.shape[:2] for im in images]
    anchors, crops = O.rrc_batch(1234, it, shapes)
    flips = O.coin_flip_batch(1235, it, len(images), 0.5)
    out = np.empty((len(images), 3, 224, 224), np.float16)
    for i, im in enumerate(images):
        (ay, ax), (ch, cw) = anchors[i], crops[i]
        u8 = O.resample_u8(im, (224, 224), roi=(ay, ax, ay + ch, ax + cw))
        out[i] = O.cmn_u8(u8, (0, 0), (224, 224), mirror=bool(flips[i]), mean=mean, inv_std=inv, dtype=O.F16)
    return out


@pytest.mark.parametrize("variant", ["distinct_dht", "mixed", "large"])
def test_headline_graph_on_realistic_mixes_equals_oracle(tmp_path, variant):
    import bench
    from dali_amd.testing import synth_dataset
    batch, epochs, depth = 256, 3, 5
    enc = synth_dataset(0, batch, seed=1234, workers=8, variant=variant)
    infos = [O.jpeg_info(e) for e in enc]
    if variant == "mixed":
        assert sum(i["progressive"] for i in infos) >= 5 and sum(i["ncomp"] == 4 for i in infos) >= 2
    if variant == "large":
        assert sum(i["height"] * i["width"] >= 12_000_000 for i in infos) >= 2
    if variant == "distinct_dht":
        dht = {e[e.index(b"\xff\xc4"):e.index(b"\xff\xda")] for e in enc}
        assert len(dht) >= batch - 2                                 # (practically) every stream has its own tables
    bench.write_dataset(str(tmp_path), enc)
    order = sorted(range(batch), key=lambda g: (g % 10, g))          # readers.file: class directories sorted, files inside
    images = [_decode(enc[g]) for g in order]
    pipe = bench.resident_pipeline(str(tmp_path), batch, 0, depth, 8, cache_mb=max(64, int(2.2 * sum(map(len, enc)) / 2**20)),
                                   crop_seed=1234, flip_seed=1235)
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for it in range(epochs):
        data, lab = pipe.run()
        got = data.as_tensor().cpu().numpy()
        assert got.shape == (batch, 3, 224, 224) and got.dtype == np.float16
        assert list(lab.as_array().reshape(-1)) == [g % 10 for g in order]
        ref = _oracle_batch(images, it, mean, inv)
        same = got.view(np.uint16) == ref.view(np.uint16)
        bad = np.nonzero(~same.reshape(batch, -1).all(1))[0]
        assert same.all(), f"{variant}, iteration {it}: samples {[order[b] for b in bad[:8]]} differ from the oracle"
    assert "jpeg_huffman" in pipe.executed_kernels() and "fused_resample_cmn" in pipe.executed_kernels()
    del pipe
    gc.collect()
