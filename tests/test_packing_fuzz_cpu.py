"""CPU twin of tests/test_hip_fuzz.py: the packed path (packing -> oracle's
packed adapter) must equal the proto-shaped oracle path on random inputs, so the
host logic (codes, ranks, HP, groups, aux pixels, sort positions) is exercised
without a GPU on the same configurations the GPU fuzz test uses."""
import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import packing
from oracle import oracle as O
from tests import fuzz_inputs as F


@pytest.mark.parametrize('name,channels,width,height,okw,ckw', F.CONFIGS)
def test_packed_equals_proto_path_fuzz(name, channels, width, height, okw, ckw):
  rng = np.random.default_rng(len(name) * 7919)
  opts = F.options(channels, width, height, **dict(okw))
  c = len(channels)
  chan_enums = packing.channel_enums(opts)
  need_aux = any(ch in channels for ch in ('read_mapping_percent', 'avg_base_quality',
                                           'identity', 'gap_compressed_identity'))
  if okw.get('sort_by_alt_allele_support'):
    pytest.skip('the packed adapter cannot rebuild allele groups; GPU fuzz covers it')
  if 'read_supports_variant_fuzzy' in channels:
    pytest.skip('the packed adapter has no candidate to recompute fuzzy support from; '
                'tests/test_fuzzy_channel_cpu.py and the GPU fuzz cover it')
  for trial in range(8):
    n_reads = int(rng.choice([0, 2, 12, height, 2 * height + 5]))
    call, ref, reads, start, combo = F.make_case(rng, width, n_reads, **dict(ckw))
    if 'avg_base_quality' in channels:
      for r in reads:
        r.aligned_quality = bytes(min(q, 93) for q in r.aligned_quality)
    mc = float(rng.integers(0, height + 5)) if 'mean_coverage' in channels else 0.0
    blank = [chan_enums[int(rng.integers(0, c))]] if trial % 3 == 0 else None
    table = packing.ReadTable.from_reads(reads, need_aux=need_aux)
    batch = packing.PackedBatch(table=table, width=width)
    idx = np.arange(len(reads), dtype=np.uint32)
    groups = (packing.allele_groups(call, table, idx)
              if okw.get('sort_by_alt_allele_support') else None)
    batch.add_item(call.variant.start, start, batch.add_ref_window(ref), idx,
                   packing.support_codes(call, combo, table, idx), height=height, out_off=0,
                   blank_mask=packing.blank_mask_for(chan_enums, blank), mean_coverage=mc,
                   groups=groups)
    got, rows = O.encode_packed(opts, batch, c)
    want, kept, _ = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height,
                                   mean_coverage=mc, channels_to_blank=blank,
                                   return_row_reads=True)
    assert rows[0] == kept, (name, trial)
    np.testing.assert_array_equal(got.reshape(height, width, c), want,
                                  err_msg='%s trial %d' % (name, trial))
