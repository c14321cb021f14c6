"""DownsampleReadIndices (deepvariant/pileup_image_native.cc:153-165) pinned independently of
the local C++ library: the reference passes its generator BY VALUE, so for a pile-up of n >
max_reads reads the row order is `std::shuffle(iota(n), std::mt19937_64(random_seed =
2101079370))` -- a pure function of n.  The literal vectors below were produced by
tests/std_shuffle.py (a restatement of libstdc++-11's algorithm, checked against ISO C++'s
known 10000th MT19937-64 output); the product (libdvhip's host table builder) and the C++
oracle must reproduce them whatever libstdc++ they were compiled with."""
import ctypes as C

import numpy as np

from tests import std_shuffle

SEED = 2101079370   # pic_options.random_seed default, deepvariant/pileup_image.py:36-74
N96 = [32, 69, 31, 60, 53, 68, 49, 39, 76, 54, 18, 82, 83, 80, 9, 19, 84, 67, 52, 28, 79, 14, 29, 36, 95, 13, 92, 91, 64, 3, 38, 44, 2, 72, 87, 10, 62, 93, 66, 11, 26, 59, 17, 51, 46, 42, 48, 88, 75, 90, 34, 58, 61, 15, 6, 30, 21, 37, 65, 7, 43, 77, 20, 78, 33, 41, 63, 45, 86, 8, 23, 70, 22, 25, 0, 74, 94, 71, 85, 89, 12, 5, 47, 35, 73, 4, 50, 40, 81, 1, 27, 55, 16, 24, 57, 56]
N99 = [33, 7, 12, 61, 80, 96, 18, 70, 77, 86, 3, 83, 14, 81, 9, 88, 8, 85, 53, 21, 29, 66, 76, 54, 84, 97, 93, 31, 19, 65, 39, 32, 45, 73, 60, 27, 17, 63, 90, 98, 34, 16, 40, 11, 6, 47, 2, 49, 89, 91, 74, 35, 59, 72, 37, 0, 5, 68, 38, 46, 41, 50, 82, 56, 94, 23, 44, 10, 62, 87, 78, 4, 36, 58, 52, 22, 13, 95, 92, 28, 1, 30, 71, 64, 57, 43, 55, 51, 24, 25, 67, 75, 20, 15, 79, 48, 69, 26, 42]


def test_mt19937_64_known_answer():
  g = std_shuffle.MT19937_64()
  for _ in range(9999):
    g()
  assert g() == 9981545732273789042      # ISO C++ [rand.predef]


def test_literal_vectors_match_the_restatement():
  assert std_shuffle.std_shuffle_iota(96, SEED) == N96
  assert std_shuffle.std_shuffle_iota(99, SEED) == N99
  assert sorted(N96) == list(range(96)) and sorted(N99) == list(range(99))


def test_oracle_and_product_reproduce_the_pinned_permutations():
  from deepvariant_amd import _lib
  from oracle import oracle as O
  lib = _lib.lib()
  for n, want in ((96, N96), (99, N99)):
    assert O.downsample_indices(n, 95, SEED).tolist() == want
    out = np.zeros(n, np.int32)
    assert lib.dv_downsample_indices(n, 95, C.c_uint32(SEED), out.ctypes.data_as(C.c_void_p)) == 0
    assert out.tolist() == want
  for n in (97, 98, 100, 137, 200, 256):
    want = std_shuffle.std_shuffle_iota(n, SEED)
    assert O.downsample_indices(n, 95, SEED).tolist() == want
    out = np.zeros(n, np.int32)
    assert lib.dv_downsample_indices(n, 95, C.c_uint32(SEED), out.ctypes.data_as(C.c_void_p)) == 0
    assert out.tolist() == want
