"""Decoded-image cache bookkeeping (host only), through the C ABI.  The cases are the reference's own unit tests of
its two cache policies restated: image_cache_blob_test.cc:41-90 ("threshold") and image_cache_largest_test.cc:48-135
("largest": images of 1..4 bytes named "1".."4", two passes)."""
import ctypes as C

import pytest

from dali_amd import _capi as capi


class Policy:
    def __init__(self, kind, size, threshold=0):
        self.lib = capi.host()
        self.h = self.lib.daliamdImageCachePolicyCreate(kind.encode(), size, threshold)
        if not self.h:
            raise RuntimeError(self.lib.daliamdHostGetLastErrorMessage().decode())

    def add(self, key, size, stored=None):
        return self.lib.daliamdImageCachePolicyOnDecode(self.h, str(key).encode(), size, size if stored is None else stored)

    def cached(self, key):
        return self.lib.daliamdImageCachePolicyFind(self.h, str(key).encode()) >= 0

    def offset(self, key):
        return self.lib.daliamdImageCachePolicyFind(self.h, str(key).encode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.daliamdImageCachePolicyDestroy(self.h)


# ---- "threshold" (ImageCacheBlob) ----
def test_threshold_empty_add_and_duplicate():
    p = Policy("threshold", 1 << 20)
    assert not p.cached("file1.jpg")
    assert p.add("file1.jpg", 300) == 0
    assert p.cached("file1.jpg")
    assert p.add("file1.jpg", 300) == -1            # AddExistingIgnored
    assert p.add("file2.jpg", 500) == 300           # entries are appended
    assert p.offset("file1.jpg") == 0 and p.offset("file2.jpg") == 300


def test_threshold_too_small_cache_and_threshold():
    p = Policy("threshold", 299)
    assert p.add("file1.jpg", 300) == -1 and not p.cached("file1.jpg")     # ErrorTooSmallCacheSize
    p = Policy("threshold", 1000, 400)
    assert p.add("small", 399) == -1                                          # below the threshold: never kept
    assert p.add("big", 400) == 0
    assert p.add("big2", 600) == 400
    assert p.add("big3", 400) == -1                                           # full: ignored, earlier entries stay
    assert p.cached("big") and p.cached("big2") and not p.cached("big3")
    with pytest.raises(RuntimeError, match="Cache size should fit at least one image"):
        Policy("threshold", 100, 200)
    with pytest.raises(RuntimeError, match="unexpected cache policy"):
        Policy("lru", 100)


def test_threshold_more_than_2000_mb():
    mb = 1 << 20
    size = 3 * 1024 * mb
    p = Policy("threshold", size)
    n = size // mb
    for i in range(n + 10):
        p.add(f"{i}_mb", mb)
    assert all(p.cached(f"{i}_mb") for i in range(n))
    assert not any(p.cached(f"{i}_mb") for i in range(n, n + 10))
    assert p.offset(f"{n - 1}_mb") == (n - 1) * mb                            # 64-bit offsets


def test_stored_size_is_what_fills_the_blob():
    """The decoder stores row-padded images: the threshold looks at H*W*C, the capacity at the stored bytes."""
    p = Policy("threshold", 1024, 100)
    assert p.add("a", 100, stored=512) == 0
    assert p.add("b", 99, stored=256) == -1
    assert p.add("c", 100, stored=512) == 512
    assert p.add("d", 100, stored=256) == -1


# ---- "largest" (ImageCacheLargest) ----
def two_passes(size):
    p = Policy("largest", size)
    for _ in range(2):
        for i in (1, 2, 3, 4):
            p.add(i, i)
    return [p.cached(i) for i in (1, 2, 3, 4)]


def test_largest_first_round_no_cache():
    p = Policy("largest", 1 << 9)
    for i in (1, 2, 3, 4):
        assert p.add(i, i) == -1
    assert not any(p.cached(i) for i in (1, 2, 3, 4))


@pytest.mark.parametrize("size,expected", [
    (1 << 9, [True, True, True, True]),       # SecondRoundCache
    (4, [False, False, False, True]),         # OnlySpaceForTheLast
    (3, [False, False, True, False]),         # OnlySpaceForTheOneBeforeTheLast
    (6, [False, True, False, True]),          # OnlySpaceFor4Plus2ButNot3
])
def test_largest_second_round(size, expected):
    assert two_passes(size) == expected


def test_largest_starts_on_the_first_repeated_key():
    p = Policy("largest", 4)
    assert p.add(4, 4) == -1 and not p.cached(4)
    assert p.add(4, 4) == 0 and p.cached(4)                                   # ReadWorks / GetWorks set-up
    assert p.add(3, 3) == -1                                                  # not in the chosen set any more
