"""Shared helpers for the test-suite: seeded synthetic images / JPEG streams (they live in dali_amd/testing.py so
that bench.py does not depend on the test package)."""
from dali_amd.testing import (IMAGENET_LIKE_SIZES, encode_jpeg, synth_dataset, synth_dataset_image,  # noqa: F401
                              synth_image, synth_jpeg_batch)
