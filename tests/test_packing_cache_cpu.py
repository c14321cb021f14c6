"""ReadTable.from_reads keeps one packed record per Read object (`_dv_packed`) and
ReadTable.read_factory hands its Reads out with the record of the row they were made from: a
region's reads are packed several times per region (window selection, candidate calling, the
encoder's batch).  The cache must never change what a table holds."""
import copy
import dataclasses

import numpy as np

from deepvariant_amd import dv_types as T
from deepvariant_amd import packing
from tests import realigner_fixture as RF


def _same(a: packing.ReadTable, b: packing.ReadTable):
  for f in dataclasses.fields(packing.ReadTable):
    x, y = getattr(a, f.name), getattr(b, f.name)
    if isinstance(x, np.ndarray):
      assert isinstance(y, np.ndarray) and x.dtype == y.dtype and np.array_equal(x, y), f.name
    else:
      assert x == y, f.name


def _reads(n=600):
  _, sets = RF.load()
  return sets['wgs'][:n]


def test_cached_and_fresh_records_give_the_same_table():
  reads = _reads()
  fresh = packing.ReadTable.from_reads(reads)
  made = fresh.to_reads('chr20')                      # Reads that carry their row's record
  assert all(hasattr(r, '_dv_packed') for r in made)
  from_rows = packing.ReadTable.from_reads(made)
  for r in made:
    del r._dv_packed
  walked = packing.ReadTable.from_reads(made)         # records rebuilt from the objects
  again = packing.ReadTable.from_reads(made)          # ... and reused
  _same(fresh, from_rows)
  _same(fresh, walked)
  _same(fresh, again)
  # a subset in another order (what a window or a candidate's read list is)
  idx = [5, 3, 400, 17, 18]
  _same(packing.ReadTable.from_reads([reads[i] for i in idx]), packing.ReadTable.from_reads([made[i] for i in idx]))


def test_a_moved_read_is_packed_again():
  reads = _reads(50)
  made = packing.ReadTable.from_reads(reads).to_reads('chr20')
  packing.ReadTable.from_reads(made)
  moved = copy.copy(made[7])                          # shallow copy drags the cached record along ...
  moved.alignment = T.LinearAlignment(                # ... but the alignment is a new object: the record is stale
      position=T.Position('chr20', made[7].alignment.position.position + 3, False),
      mapping_quality=made[7].alignment.mapping_quality,
      cigar=[T.CigarUnit(T.CIGAR_M if hasattr(T, 'CIGAR_M') else 1, len(made[7].aligned_sequence))])
  table = packing.ReadTable.from_reads(made[:7] + [moved] + made[8:])
  assert int(table.read_pos[7]) == made[7].alignment.position.position + 3
  assert table.cigar[table.read_cigar_off[7]] == (len(made[7].aligned_sequence) << 4) | 1


def test_aux_columns_are_added_to_a_record_cached_without_them():
  reads = _reads(40)
  made = packing.ReadTable.from_reads(reads).to_reads('chr20')
  plain = packing.ReadTable.from_reads(made)
  assert plain.read_aux is None
  with_aux = packing.ReadTable.from_reads(made, need_aux=True)
  want = packing.ReadTable.from_reads(reads, need_aux=True)
  assert with_aux.read_aux is not None and np.array_equal(with_aux.read_aux, want.read_aux)


def test_lazy_reads_build_their_alignment_on_first_access_only():
  """ReadTable.read_factory hands out packing.LazyRead objects: everything but `alignment` is there
  at once; spans (realigner.utils.read_range) and tables (from_reads) come from the packed row
  without building it; the first access builds exactly what an eager Read holds."""
  from deepvariant_amd.realigner import utils as U
  reads = _reads(300)
  table = packing.ReadTable.from_reads(reads)
  made = table.to_reads('chr20')
  assert all(isinstance(r, packing.LazyRead) and isinstance(r, T.Read) and 'alignment' not in r.__dict__ for r in made)
  spans = [U.read_range(r) for r in made]
  again = packing.ReadTable.from_reads(made)
  assert all('alignment' not in r.__dict__ for r in made)           # neither needed the objects
  _same(table, again)
  for r, src, span in zip(made, reads, spans):
    assert span == U.read_range(src)                                # from the row == from the CIGAR
    assert r.alignment == src.alignment and 'alignment' in r.__dict__
    assert U.read_range(r) == span                                  # ... and again from the object
    assert (r.fragment_name, r.read_number, r.aligned_sequence, bytes(r.aligned_quality), r.fragment_length) == (
        src.fragment_name, src.read_number, src.aligned_sequence, bytes(bytearray(src.aligned_quality)), src.fragment_length)
  _same(table, packing.ReadTable.from_reads(made))                  # records now tied to the built alignments
  # a lazy read equals the eager Read with the same content, both ways round
  eager = T.Read(fragment_name=made[0].fragment_name, read_number=made[0].read_number, number_reads=2,
                 fragment_length=made[0].fragment_length, aligned_sequence=made[0].aligned_sequence,
                 aligned_quality=made[0].aligned_quality, alignment=made[0].alignment, info=dict(made[0].info))
  assert made[0] == eager and eager == made[0] and made[1] != eager
  # dataclasses.replace (make_examples_core copies reads before tagging them) builds the alignment and a full object
  fresh = table.to_reads('chr20')[5]
  tagged = dataclasses.replace(fresh, info={'HP': T.ListValue(values=[T.Value(int_value=1)])})
  assert tagged.alignment == reads[5].alignment and not hasattr(tagged, '_dv_packed')
  assert int(packing.ReadTable.from_reads([tagged]).read_hp[0]) == 1
