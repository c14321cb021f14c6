"""The synthetic workloads are part of the measurement contract: they must be
deterministic functions of the seed and have the documented shape statistics."""
import numpy as np

from deepvariant_amd import synth
from oracle import oracle as O


def test_illumina_batch_is_deterministic_and_has_deep_sites():
  a = synth.make_illumina_batch(300, seed=9)
  b = synth.make_illumina_batch(300, seed=9)
  for name in ('read_pos', 'bases', 'quals', 'cigar', 'list_read', 'list_code', 'ref_windows'):
    np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
  depth = np.diff(np.asarray(a.item_list_off))
  assert 25 < depth.mean() < 45 and depth.max() > 95 and depth.min() >= 4


def test_longread_batches_encode_on_the_oracle():
  for kind, width, channels in (('hifi', 147, 8), ('ont', 199, 7)):
    opts = synth.longread_options(kind)
    assert (opts.width, len(opts.channels)) == (width, channels)
    batch = synth.make_longread_batch(12, kind, seed=4)
    again = synth.make_longread_batch(12, kind, seed=4)
    np.testing.assert_array_equal(batch.cigar, again.cigar)
    out, rows = O.encode_packed(opts, batch, channels)
    img = out.reshape(12, 100, width, channels)
    assert (rows > 0).all() and (rows <= 95).all()
    # sort_by_haplotypes: the haplotype channel (index 6) is non-decreasing down the read rows
    hap = img[:, 5:, width // 2, 6].astype(int)
    for k in range(12):
      used = hap[k][:rows[k]]
      used = used[used > 0]
      assert (np.diff(used) >= 0).all()
