"""Pure-Python restatement of `std::shuffle(first, last, std::mt19937_64(seed))` as
libstdc++ >= 11 implements it -- the library the reference is built with (Ubuntu 22.04,
gcc 11: /root/reference/Dockerfile, ubuntu:22.04 base) and therefore what
`DownsampleReadIndices` (deepvariant/pileup_image_native.cc:153-165) produces.

TEST INFRASTRUCTURE: pins the permutation independently of whatever libstdc++ libdvhip.so
and the C++ oracle were compiled against.  Algorithms restated (published sources):
  * MT19937-64 (Matsumoto & Nishimura; ISO C++ [rand.predef]: the 10000th value of a
    default-seeded engine is 9981545732273789042).
  * libstdc++ bits/stl_algo.h `shuffle`: two swaps per draw while range^2 fits the engine's
    range (`__gen_two_uniform_ints`), one leading single swap when the length is even.
  * libstdc++ bits/uniform_int_dist.h (gcc >= 11): Lemire's nearly-divisionless method
    (`_S_nd<unsigned __int128>`) for a 64-bit engine.
"""

M64 = (1 << 64) - 1


class MT19937_64:
  NN, MM = 312, 156
  MATRIX_A, UM, LM = 0xB5026F5AA96619E9, 0xFFFFFFFF80000000, 0x7FFFFFFF

  def __init__(self, seed=5489):
    mt = [0] * self.NN
    mt[0] = seed & M64
    for i in range(1, self.NN):
      mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & M64
    self.mt, self.i = mt, self.NN

  def __call__(self):
    if self.i >= self.NN:
      mt = self.mt
      for k in range(self.NN):
        x = (mt[k] & self.UM) | (mt[(k + 1) % self.NN] & self.LM)
        mt[k] = mt[(k + self.MM) % self.NN] ^ (x >> 1) ^ (self.MATRIX_A if x & 1 else 0)
      self.i = 0
    x = self.mt[self.i]
    self.i += 1
    x ^= (x >> 29) & 0x5555555555555555
    x ^= (x << 17) & 0x71D67FFFEDA60000
    x ^= (x << 37) & 0xFFF7EEE000000000
    x ^= x >> 43
    return x & M64


def _uniform(gen, rng):
  """uniform_int_distribution<uint64>{0, rng - 1}(gen), gcc >= 11 (_S_nd)."""
  product = gen() * rng
  low = product & M64
  if low < rng:
    threshold = ((1 << 64) - rng) % rng
    while low < threshold:
      product = gen() * rng
      low = product & M64
  return product >> 64


def std_shuffle_iota(n, seed):
  """iota(n) shuffled by std::shuffle with std::mt19937_64(seed)."""
  a = list(range(n))
  if n == 0:
    return a
  gen = MT19937_64(seed)
  assert M64 // n >= n          # the two-at-a-time branch (always true for pile-up sizes)
  i = 1
  if n % 2 == 0:
    j = _uniform(gen, 2)
    a[i], a[j] = a[j], a[i]
    i += 1
  while i < n:
    swap_range = i + 1
    x = _uniform(gen, swap_range * (swap_range + 1))
    p0, p1 = x // (swap_range + 1), x % (swap_range + 1)
    a[i], a[p0] = a[p0], a[i]
    i += 1
    a[i], a[p1] = a[p1], a[i]
    i += 1
  return a
