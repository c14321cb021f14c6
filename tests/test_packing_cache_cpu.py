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
