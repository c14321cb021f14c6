// bam_reader.cpp -- BAM (BGZF) -> the packed read table of dv_batch, on the host.
//
// SURVEY.md 8f row f1 ("BAM -> packed read SoA"): the step in front of the hot path.
// The reference reads through htslib + nucleus::SamReader into Read protos
// (third_party/nucleus/io/sam_reader.cc:734-840 ConvertToPb, :217-247 read
// requirements, third_party/nucleus/util/utils.cc:261-266 IsReadProperlyPlaced) and the
// per-region InMemoryReader is then built from those protos.  htslib is not in this
// image; zlib is.  This file inflates the BGZF blocks (in parallel), decodes the records
// of one region and writes them straight into the structure-of-arrays layout
// encode_items_kernel reads: no per-read objects, no strings except the name blob the
// host keeps for allele_support matching.
//
// Field semantics (all from ConvertToPb): qualities are raw phred bytes; read_number =
// 0 if FREAD1 or unpaired else 1; fragment_length = isize; HP = the integer HP aux tag;
// CIGAR ops are mapped to nucleus' CigarUnit enum (htslib op + 1).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "dv_internal.h"
#include "aux_planes.h"
#include "read_table.h"

namespace {

inline uint32_t le32(const uint8_t* p) {
  return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24);
}
inline uint16_t le16(const uint8_t* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

struct Block {
  size_t in_off, in_len;  // deflate payload
  size_t out_off, out_len;
};

// A run of BGZF members -> one inflated buffer.  Block boundaries come from the BC extra field
// (SAMv1 4.1), sizes from ISIZE, so blocks inflate independently on `n_threads` threads.  With
// `cut_tail_ok` the run may end inside a member (a byte range read ahead of an indexed query): the
// complete members are inflated, `*consumed` tells how many input bytes they span.  `members`
// (optional) receives (offset in the input, offset in the output) per member.
int inflate_members(const uint8_t* file, size_t size, int n_threads, bool cut_tail_ok, std::vector<uint8_t>* out,
                    size_t* consumed, std::vector<std::pair<size_t, size_t>>* members) {
  std::vector<Block> blocks;
  size_t pos = 0, total = 0;
  while (pos < size) {
    if (pos + 18 > size) {
      if (cut_tail_ok) break;
      return dv::fail(DV_ERR_BAD_INPUT, "not a BGZF file (bad gzip member header)");
    }
    if (file[pos] != 31 || file[pos + 1] != 139) {
      return dv::fail(DV_ERR_BAD_INPUT, "not a BGZF file (bad gzip member header)");
    }
    const unsigned xlen = le16(&file[pos + 10]);
    size_t p = pos + 12;
    const size_t xend = p + xlen;
    if (xend > size && cut_tail_ok) break;
    long bsize = -1;
    while (p + 4 <= xend && xend <= size) {
      const unsigned slen = le16(&file[p + 2]);
      if (file[p] == 66 && file[p + 1] == 67 && slen == 2) bsize = le16(&file[p + 4]) + 1L;
      p += 4 + slen;
    }
    if (bsize >= 0 && pos + bsize > size && cut_tail_ok) break;
    if (bsize < 0 || pos + bsize > size || static_cast<size_t>(bsize) < 12u + xlen + 8u) {
      return dv::fail(DV_ERR_BAD_INPUT, "not a BGZF file (no BC subfield / truncated block)");
    }
    Block b;
    b.in_off = xend;
    b.in_len = pos + bsize - 8 - xend;
    b.out_len = le32(&file[pos + bsize - 4]);
    if (b.out_len > 65536) return dv::fail(DV_ERR_BAD_INPUT, "BGZF block larger than 64 KiB (ISIZE)");
    b.out_off = total;
    if (members) members->emplace_back(pos, total);
    total += b.out_len;
    blocks.push_back(b);
    pos += bsize;
  }
  if (consumed) *consumed = pos;
  out->resize(total);
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  auto work = [&]() {
    z_stream zs;
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= blocks.size() || failed.load()) return;
      const Block& b = blocks[i];
      if (b.out_len == 0) continue;  // EOF marker block
      std::memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) {
        failed = 1;
        return;
      }
      zs.next_in = const_cast<Bytef*>(&file[b.in_off]);
      zs.avail_in = static_cast<uInt>(b.in_len);
      zs.next_out = out->data() + b.out_off;
      zs.avail_out = static_cast<uInt>(b.out_len);
      const int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END || zs.avail_out != 0) failed = 1;
    }
  };
  const int nt = std::max(1, std::min<int>(n_threads, static_cast<int>(blocks.size())));
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (failed.load()) return dv::fail(DV_ERR_BAD_INPUT, "BGZF block failed to inflate");
  return DV_OK;
}

// Whole file -> one inflated buffer.
int inflate_bgzf(const std::vector<uint8_t>& file, int n_threads, std::vector<uint8_t>* out) {
  return inflate_members(file.data(), file.size(), n_threads, false, out, nullptr, nullptr);
}

// Integer value of a 2-letter aux tag (c C s S i I), or false.
bool find_int_tag(const uint8_t* aux, const uint8_t* end, char t0, char t1, int32_t* value) {
  const uint8_t* p = aux;
  auto size_of = [](uint8_t ty) -> int {
    switch (ty) {
      case 'A': case 'c': case 'C': return 1;
      case 's': case 'S': return 2;
      case 'i': case 'I': case 'f': return 4;
      default: return 0;
    }
  };
  while (p + 3 <= end) {
    const bool hit = p[0] == static_cast<uint8_t>(t0) && p[1] == static_cast<uint8_t>(t1);
    const uint8_t ty = p[2];
    p += 3;
    const int sz = size_of(ty);
    if (sz) {
      if (p + sz > end) return false;
      if (hit) {
        switch (ty) {
          case 'c': *value = static_cast<int8_t>(p[0]); return true;
          case 'C': *value = p[0]; return true;
          case 's': *value = static_cast<int16_t>(le16(p)); return true;
          case 'S': *value = le16(p); return true;
          case 'i': case 'I': *value = static_cast<int32_t>(le32(p)); return true;
          default: return false;
        }
      }
      p += sz;
    } else if (ty == 'Z' || ty == 'H') {
      while (p < end && *p) ++p;
      ++p;
    } else if (ty == 'B') {
      if (p + 5 > end) return false;
      const int sub = size_of(p[0]);
      if (!sub) return false;
      p += 5 + static_cast<size_t>(le32(p + 1)) * sub;
    } else {
      return false;
    }
  }
  return false;
}

// The uint32 array of a `B:I` aux tag (CG: the real CIGAR of a record with more than 65535
// operations, SAMv1 4.2.2).  Returns false when the tag is absent or of another type.
bool find_u32_array_tag(const uint8_t* aux, const uint8_t* end, char t0, char t1,
                        const uint8_t** data, uint32_t* count) {
  const uint8_t* p = aux;
  auto size_of = [](uint8_t ty) -> int {
    switch (ty) {
      case 'A': case 'c': case 'C': return 1;
      case 's': case 'S': return 2;
      case 'i': case 'I': case 'f': return 4;
      default: return 0;
    }
  };
  while (p + 3 <= end) {
    const bool hit = p[0] == static_cast<uint8_t>(t0) && p[1] == static_cast<uint8_t>(t1);
    const uint8_t ty = p[2];
    p += 3;
    const int sz = size_of(ty);
    if (sz) {
      if (p + sz > end) return false;
      p += sz;
    } else if (ty == 'Z' || ty == 'H') {
      while (p < end && *p) ++p;
      ++p;
    } else if (ty == 'B') {
      if (p + 5 > end) return false;
      const int sub = size_of(p[0]);
      if (!sub) return false;
      const uint64_t n = le32(p + 1);
      if (static_cast<uint64_t>(end - (p + 5)) < n * sub) return false;
      if (hit) {
        if (p[0] != 'I') return false;
        *data = p + 5;
        *count = static_cast<uint32_t>(n);
        return true;
      }
      p += 5 + n * sub;
    } else {
      return false;
    }
  }
  return false;
}

// The bytes of a `Z` aux tag (without the NUL), or false.
bool find_string_tag(const uint8_t* aux, const uint8_t* end, char t0, char t1, const uint8_t** data,
                     uint32_t* len) {
  const uint8_t* p = aux;
  auto size_of = [](uint8_t ty) -> int {
    switch (ty) {
      case 'A': case 'c': case 'C': return 1;
      case 's': case 'S': return 2;
      case 'i': case 'I': case 'f': return 4;
      default: return 0;
    }
  };
  while (p + 3 <= end) {
    const bool hit = p[0] == static_cast<uint8_t>(t0) && p[1] == static_cast<uint8_t>(t1);
    const uint8_t ty = p[2];
    p += 3;
    const int sz = size_of(ty);
    if (sz) {
      if (p + sz > end) return false;
      p += sz;
    } else if (ty == 'Z' || ty == 'H') {
      const uint8_t* q = p;
      while (q < end && *q) ++q;
      if (hit && ty == 'Z') {
        *data = p;
        *len = static_cast<uint32_t>(q - p);
        return true;
      }
      p = q + 1;
    } else if (ty == 'B') {
      if (p + 5 > end) return false;
      const int sub = size_of(p[0]);
      if (!sub) return false;
      p += 5 + static_cast<size_t>(le32(p + 1)) * sub;
    } else {
      return false;
    }
  }
  return false;
}

// ---- record decoding ------------------------------------------------------------
struct RegionFilter {
  dv_read_requirements rq{};
  bool any_contig = true;
  int32_t want_ref = -1;
  int64_t start = 0, end = 0;
};

// Decodes one BAM record (r = first byte after block_size) into the table if it passes.
// Returns DV_OK (kept or skipped) or an error; *past_end is set when the record starts at
// or beyond the region end on the wanted contig (coordinate-sorted files can stop there).
int decode_record(const uint8_t* r, uint32_t block_size, const RegionFilter& f, dv_read_table* t,
                  bool* past_end) {
  static const char kNt16[] = "=ACMGRSVTWYHKDBN";  // htslib seq_nt16_str
  const dv_read_requirements& rq = f.rq;
  const int32_t ref_id = static_cast<int32_t>(le32(r));
  const int32_t rpos = static_cast<int32_t>(le32(r + 4));
  const unsigned l_read_name = r[8];
  const unsigned mapq = r[9];
  const unsigned n_cigar = le16(r + 12);
  const unsigned flag = le16(r + 14);
  const uint32_t l_seq = le32(r + 16);
  const int32_t next_ref = static_cast<int32_t>(le32(r + 20));
  const int32_t tlen = static_cast<int32_t>(le32(r + 28));
  // 64-bit sum: l_seq comes from an untrusted file, and 32-bit arithmetic wraps (an l_seq of
  // 0xAAAAAAAB made `need` ~33 and the decoder read gigabytes past the record)
  const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (static_cast<uint64_t>(l_seq) + 1) / 2 +
                        static_cast<uint64_t>(l_seq);
  if (need > block_size || l_seq > block_size || l_read_name == 0) {
    return dv::fail(DV_ERR_BAD_INPUT, "corrupt BAM record");
  }
  if (past_end && !f.any_contig && (ref_id > f.want_ref || (ref_id == f.want_ref && rpos >= f.end))) {
    *past_end = true;
  }
  if (ref_id < 0 || (flag & 0x4)) return DV_OK;                    // unmapped: no position
  if (!f.any_contig && ref_id != f.want_ref) return DV_OK;
  // PartialReadSatisfiesRequirements (sam_reader.cc:217-234)
  if ((!rq.keep_duplicates && (flag & 0x400)) ||
      (!rq.keep_failed_vendor_quality_checks && (flag & 0x200)) ||
      (!rq.keep_secondary_alignments && (flag & 0x100)) ||
      (!rq.keep_supplementary_alignments && (flag & 0x800))) {
    return DV_OK;
  }
  // IsReadProperlyPlaced (utils.cc:261-266): the mate position only exists for a
  // paired read whose mate is mapped with a valid reference id (sam_reader.cc:829-837)
  const bool paired = flag & 0x1;
  const bool has_mate_pos = paired && !(flag & 0x8) && next_ref >= 0;
  const bool properly_placed = !paired || (flag & 0x2) || !has_mate_pos || next_ref == ref_id;
  if (!rq.keep_improperly_placed && !properly_placed) return DV_OK;
  if (static_cast<int32_t>(mapq) < rq.min_mapping_quality) return DV_OK;
  const uint8_t* name = r + 32;
  const uint8_t* cig = name + l_read_name;
  const uint8_t* const seq = cig + 4 * n_cigar;
  const uint8_t* const qual = seq + (l_seq + 1) / 2;
  // Long CIGARs (ONT / HiFi reads with more than 65535 operations): the record carries the
  // placeholder <l_seq>S<reference span>N and the real operations sit in the CG:B:I tag.
  // htslib swaps them in on read (sam.c bam_tag2cigar), so nucleus' SamReader -- and
  // therefore the reference -- only ever sees the real CIGAR.
  uint32_t n_ops = n_cigar;   // operations of the effective CIGAR
  if (n_cigar == 2) {
    const uint32_t c0 = le32(cig), c1 = le32(cig + 4);
    if ((c0 & 0xF) == 4 && (c0 >> 4) == l_seq && (c1 & 0xF) == 3) {
      const uint8_t* cg = nullptr;
      uint32_t n_cg = 0;
      if (find_u32_array_tag(qual + l_seq, r + block_size, 'C', 'G', &cg, &n_cg) && n_cg > 0) {
        cig = cg;
        n_ops = n_cg;
      }
    }
  }
  int64_t ref_len = 0, query_len = 0;
  for (unsigned k = 0; k < n_ops; ++k) {
    const uint32_t v = le32(cig + 4 * k);
    const unsigned op = v & 0xF;
    if (op > 8) return dv::fail(DV_ERR_BAD_INPUT, "Unrecognized CIGAR op in BAM record");
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += v >> 4;
    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) query_len += v >> 4;  // M I S = X
  }
  // nucleus::ReadOverlapsRegion on [start, end) with ReadEnd = pos + reference span
  if (!(f.end > rpos && f.start < rpos + std::max<int64_t>(ref_len, 1))) return DV_OK;
  // AssignAlignedQuality (sam_reader.cc:722-760): QUAL, or the OQ tag when the original scores
  // are asked for
  const uint8_t* qsrc = qual;
  int qsub = 0;
  if (rq.use_original_base_quality_scores) {
    const uint8_t* oq = nullptr;
    uint32_t n_oq = 0;
    if (!find_string_tag(qual + l_seq, r + block_size, 'O', 'Q', &oq, &n_oq)) {
      return dv::fail(DV_ERR_BAD_INPUT, "use_original_base_quality_scores: a read has no OQ tag");
    }
    if (n_oq != l_seq) return dv::fail(DV_ERR_BAD_INPUT, "OQ tag and sequence are of different length");
    for (uint32_t i = 0; i < n_oq; ++i) {
      if (oq[i] < 33) return dv::fail(DV_ERR_BAD_INPUT, "OQ tag holds a character below '!'");
    }
    qsrc = oq;
    qsub = 33;
  } else if (l_seq && qual[0] == 0xff) {
    return dv::fail(DV_ERR_BAD_INPUT, "Could not read base quality scores");  // sam_reader.cc:752
  }
  // The encoder indexes bases / qualities by CIGAR query offsets without bounds checks (like
  // the reference, whose SAM parser rejects such records: "CIGAR and query sequence are of
  // different length"); SEQ '*' (l_seq = 0) with a query-consuming CIGAR is the common case.
  if (n_ops && query_len != static_cast<int64_t>(l_seq)) {
    return dv::fail(DV_ERR_BAD_INPUT, "CIGAR and query sequence are of different length");
  }
  t->pos.push_back(rpos);
  t->end.push_back(rpos + ref_len);
  t->mapq.push_back(static_cast<uint8_t>(mapq));
  t->flags.push_back(static_cast<uint8_t>(((flag & 0x10) ? DV_READ_REVERSE : 0) |
                                          ((flag & 0x800) ? DV_READ_SUPPLEMENTARY : 0)));
  t->read_number.push_back((flag & 0x40) || !paired ? 0 : 1);
  t->frag_len.push_back(tlen);
  int32_t hp = 0;
  t->hp.push_back(find_int_tag(qual + l_seq, r + block_size, 'H', 'P', &hp) ? hp : DV_HP_NONE);
  for (unsigned k = 0; k < n_ops; ++k) {
    const uint32_t v = le32(cig + 4 * k);
    t->cigar.push_back(((v >> 4) << 4) | ((v & 0xF) + 1));  // kHtslibCigarToProto
  }
  t->cigar_off.push_back(static_cast<uint32_t>(t->cigar.size()));
  const size_t b0 = t->bases.size();
  t->bases.resize(b0 + l_seq);
  for (uint32_t i = 0; i < l_seq; ++i) {
    const uint8_t byte = seq[i >> 1];
    t->bases[b0 + i] = static_cast<uint8_t>(kNt16[(i & 1) ? (byte & 0xF) : (byte >> 4)]);
  }
  {
    const size_t q0 = t->quals.size();
    t->quals.insert(t->quals.end(), qsrc, qsrc + l_seq);
    if (qsub) {
      for (size_t i = q0; i < t->quals.size(); ++i) t->quals[i] = static_cast<uint8_t>(t->quals[i] - qsub);
    }
  }
  t->seq_off.push_back(static_cast<uint32_t>(t->bases.size()));
  if (t->with_mods || t->with_flow) {   // per-base planes from MM / ML / MN and tp / t0 (aux_planes.h)
    dv::AuxFields aux;
    dv::scan_bam_aux(qual + l_seq, r + block_size, &aux);
    if (!dv::append_aux_planes(t, t->bases.data() + b0, l_seq, (flag & 0x10) != 0, aux)) {
      return dv::fail(DV_ERR_BAD_INPUT, "MM tag: a position that is not a number");
    }
  }
  t->name_off.push_back(static_cast<uint32_t>(t->names.size()));
  t->names.insert(t->names.end(), name, name + l_read_name);  // includes the NUL
  if (t->names.back() != '\0') t->names.back() = '\0';
  return DV_OK;
}

// Header: magic, text, reference names.  Returns the offset of the first record in `buf`,
// or 0 when `buf` does not hold the whole header yet (indexed path inflates more blocks).
int parse_header(const std::vector<uint8_t>& buf, const char* contig, size_t* first_record,
                 int32_t* want_ref, bool* complete) {
  *complete = false;
  if (buf.size() < 12) return DV_OK;
  if (std::memcmp(buf.data(), "BAM\1", 4) != 0) return dv::fail(DV_ERR_BAD_INPUT, "bad BAM magic");
  size_t p = 8 + static_cast<size_t>(le32(&buf[4]));
  if (p + 4 > buf.size()) return DV_OK;
  const int32_t n_ref = static_cast<int32_t>(le32(&buf[p]));
  p += 4;
  *want_ref = -1;
  for (int32_t i = 0; i < n_ref; ++i) {
    if (p + 4 > buf.size()) return DV_OK;
    const uint32_t l_name = le32(&buf[p]);
    if (p + 4 + l_name + 4 > buf.size()) return DV_OK;
    if (contig && std::strlen(contig) + 1 == l_name &&
        std::memcmp(&buf[p + 4], contig, l_name - 1) == 0) {
      *want_ref = i;
    }
    p += 4 + l_name + 4;
  }
  *first_record = p;
  *complete = true;
  return DV_OK;
}

// ---- .bai index (SAMv1 5.2) ---------------------------------------------------------
struct Chunk {
  uint64_t beg, end;  // virtual offsets: (compressed offset << 16) | offset in the block
};

// Chunks of reference `ref` that may hold records overlapping [start, end): the bins of
// reg2bins, cut by the 16 kb linear index, merged.
int bai_chunks(const std::string& bai_path, int32_t ref, int64_t start, int64_t end,
               std::vector<Chunk>* chunks) {
  FILE* f = std::fopen(bai_path.c_str(), "rb");
  if (!f) return dv::fail(DV_ERR_BAD_INPUT, "cannot open " + bai_path);
  std::vector<uint8_t> d;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  d.resize(n > 0 ? n : 0);
  const size_t got = d.empty() ? 0 : std::fread(d.data(), 1, d.size(), f);
  std::fclose(f);
  if (got != d.size() || d.size() < 8 || std::memcmp(d.data(), "BAI\1", 4) != 0) {
    return dv::fail(DV_ERR_BAD_INPUT, "bad BAI file: " + bai_path);
  }
  auto le64 = [&](size_t p) { return static_cast<uint64_t>(le32(&d[p])) | (static_cast<uint64_t>(le32(&d[p + 4])) << 32); };
  if (end > (1ll << 29)) end = 1ll << 29;
  if (start < 0) start = 0;
  std::vector<uint32_t> bins;
  {
    const int64_t b = start, e = end - 1;
    bins.push_back(0);
    for (int k = 1 + (b >> 26); k <= 1 + (e >> 26); ++k) bins.push_back(k);
    for (int k = 9 + (b >> 23); k <= 9 + (e >> 23); ++k) bins.push_back(k);
    for (int k = 73 + (b >> 20); k <= 73 + (e >> 20); ++k) bins.push_back(k);
    for (int k = 585 + (b >> 17); k <= 585 + (e >> 17); ++k) bins.push_back(k);
    for (int k = 4681 + (b >> 14); k <= 4681 + (e >> 14); ++k) bins.push_back(k);
  }
  size_t p = 8;
  const int32_t n_ref = static_cast<int32_t>(le32(&d[4]));
  if (ref >= n_ref) return dv::fail(DV_ERR_BAD_INPUT, "BAI has fewer references than the BAM");
  std::vector<Chunk> found;
  uint64_t min_off = 0;
  for (int32_t r = 0; r <= ref; ++r) {
    if (p + 4 > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAI");
    const int32_t n_bin = static_cast<int32_t>(le32(&d[p]));
    p += 4;
    for (int32_t b = 0; b < n_bin; ++b) {
      if (p + 8 > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAI");
      const uint32_t bin = le32(&d[p]);
      const int32_t n_chunk = static_cast<int32_t>(le32(&d[p + 4]));
      p += 8;
      if (p + 16ull * n_chunk > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAI");
      if (r == ref && bin != 37450 && std::find(bins.begin(), bins.end(), bin) != bins.end()) {
        for (int32_t c = 0; c < n_chunk; ++c) found.push_back({le64(p + 16 * c), le64(p + 16 * c + 8)});
      }
      p += 16ull * n_chunk;
    }
    if (p + 4 > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAI");
    const int32_t n_intv = static_cast<int32_t>(le32(&d[p]));
    p += 4;
    if (p + 8ull * n_intv > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAI");
    if (r == ref && n_intv > 0) {
      const int64_t w = std::min<int64_t>(start >> 14, n_intv - 1);
      min_off = le64(p + 8 * w);
    }
    p += 8ull * n_intv;
  }
  std::sort(found.begin(), found.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
  for (const Chunk& c : found) {
    if (c.end <= min_off) continue;
    if (!chunks->empty() && c.beg <= chunks->back().end) {
      chunks->back().end = std::max(chunks->back().end, c.end);
    } else {
      chunks->push_back(c);
    }
  }
  return DV_OK;
}

// Inflates the BGZF member at file offset `coff` and appends it to `out`; *csize = its
// compressed size (0 at end of file).
int inflate_member(FILE* f, uint64_t coff, std::vector<uint8_t>* out, size_t* csize) {
  uint8_t hdr[18];
  *csize = 0;
  if (fseeko(f, static_cast<off_t>(coff), SEEK_SET) != 0) return dv::fail(DV_ERR_BAD_INPUT, "seek failed");
  const size_t got = std::fread(hdr, 1, 18, f);
  if (got == 0) return DV_OK;
  if (got < 18 || hdr[0] != 31 || hdr[1] != 139) return dv::fail(DV_ERR_BAD_INPUT, "not a BGZF member");
  const unsigned xlen = le16(hdr + 10);
  std::vector<uint8_t> extra(xlen);
  std::memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
  if (xlen > 6 && std::fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6) {
    return dv::fail(DV_ERR_BAD_INPUT, "truncated BGZF member");
  }
  long bsize = -1;
  for (size_t p = 0; p + 4 <= xlen;) {
    const unsigned slen = le16(&extra[p + 2]);
    if (extra[p] == 66 && extra[p + 1] == 67 && slen == 2 && p + 6 <= xlen) bsize = le16(&extra[p + 4]) + 1L;
    p += 4 + slen;
  }
  if (bsize < static_cast<long>(12 + xlen + 8)) return dv::fail(DV_ERR_BAD_INPUT, "bad BGZF block size");
  const size_t payload = static_cast<size_t>(bsize) - 12 - xlen;  // deflate data + crc32 + isize
  std::vector<uint8_t> comp(payload);
  if (fseeko(f, static_cast<off_t>(coff + 12 + xlen), SEEK_SET) != 0 ||
      std::fread(comp.data(), 1, payload, f) != payload) {
    return dv::fail(DV_ERR_BAD_INPUT, "truncated BGZF member");
  }
  const uint32_t isize = le32(&comp[payload - 4]);
  if (isize > 65536) return dv::fail(DV_ERR_BAD_INPUT, "BGZF block larger than 64 KiB (ISIZE)");
  const size_t o0 = out->size();
  out->resize(o0 + isize);
  if (isize) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return dv::fail(DV_ERR_BAD_INPUT, "zlib init failed");
    zs.next_in = comp.data();
    zs.avail_in = static_cast<uInt>(payload - 8);
    zs.next_out = out->data() + o0;
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return dv::fail(DV_ERR_BAD_INPUT, "BGZF block failed to inflate");
  }
  *csize = static_cast<size_t>(bsize);
  return DV_OK;
}

// ---- .csi index (CSIv1, samtools/hts-specs): the .bai scheme with a configurable leaf size (min_shift) and
// depth, one `loffset` per bin instead of the 16 kb linear index, the whole file BGZF-compressed.
// htslib writes it for contigs longer than 2^29 (`samtools index -c`) and reads it wherever it reads a .bai.
int csi_chunks(const std::string& csi_path, int32_t ref, int64_t start, int64_t end, std::vector<Chunk>* chunks) {
  FILE* f = std::fopen(csi_path.c_str(), "rb");
  if (!f) return dv::fail(DV_ERR_BAD_INPUT, "cannot open " + csi_path);
  std::vector<uint8_t> d;
  {
    struct Closer {
      FILE* f;
      ~Closer() { std::fclose(f); }
    } closer{f};
    uint64_t coff = 0;
    for (;;) {
      size_t csize = 0;
      if (int rc = inflate_member(f, coff, &d, &csize)) return rc;
      if (csize == 0) break;
      coff += csize;
    }
  }
  if (d.size() < 16 || std::memcmp(d.data(), "CSI\1", 4) != 0) return dv::fail(DV_ERR_BAD_INPUT, "bad CSI file: " + csi_path);
  auto le64 = [&](size_t p) { return static_cast<uint64_t>(le32(&d[p])) | (static_cast<uint64_t>(le32(&d[p + 4])) << 32); };
  const int min_shift = static_cast<int32_t>(le32(&d[4])), depth = static_cast<int32_t>(le32(&d[8]));
  const int32_t l_aux = static_cast<int32_t>(le32(&d[12]));
  if (min_shift < 0 || min_shift > 40 || depth < 0 || depth > 12 || l_aux < 0) return dv::fail(DV_ERR_BAD_INPUT, "bad CSI header");
  size_t p = 16 + static_cast<size_t>(l_aux);
  if (p + 4 > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated CSI");
  const int32_t n_ref = static_cast<int32_t>(le32(&d[p]));
  p += 4;
  if (ref >= n_ref) return dv::fail(DV_ERR_BAD_INPUT, "CSI has fewer references than the BAM");
  const int64_t max_pos = 1ll << (min_shift + 3 * depth);
  if (end > max_pos) end = max_pos;
  if (start < 0) start = 0;
  // reg2bins at every level; and the bins on the path from the leaf of `start` to the root, whose first
  // existing loffset bounds the chunks from below (htslib hts_itr_query)
  std::vector<uint64_t> bins, path;
  for (int l = 0; l <= depth; ++l) {
    const uint64_t t = ((1ull << (3 * l)) - 1) / 7;
    const int s = min_shift + 3 * (depth - l);
    if (end > start) {
      for (uint64_t k = t + (static_cast<uint64_t>(start) >> s); k <= t + (static_cast<uint64_t>(end - 1) >> s); ++k) bins.push_back(k);
    }
    path.push_back(t + (static_cast<uint64_t>(start) >> s));
  }
  const uint64_t meta_bin = ((1ull << (3 * (depth + 1))) - 1) / 7 + 1;
  std::vector<Chunk> found;
  std::vector<uint64_t> path_off(path.size(), 0);
  std::vector<bool> path_has(path.size(), false);
  for (int32_t r = 0; r <= ref; ++r) {
    if (p + 4 > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated CSI");
    const int32_t n_bin = static_cast<int32_t>(le32(&d[p]));
    p += 4;
    for (int32_t b = 0; b < n_bin; ++b) {
      if (p + 16 > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated CSI");
      const uint64_t bin = le32(&d[p]);
      const uint64_t loffset = le64(p + 4);
      const int32_t n_chunk = static_cast<int32_t>(le32(&d[p + 12]));
      p += 16;
      if (n_chunk < 0 || p + 16ull * n_chunk > d.size()) return dv::fail(DV_ERR_BAD_INPUT, "truncated CSI");
      if (r == ref && bin != meta_bin) {
        if (std::find(bins.begin(), bins.end(), bin) != bins.end()) {
          for (int32_t c = 0; c < n_chunk; ++c) found.push_back({le64(p + 16 * c), le64(p + 16 * c + 8)});
        }
        for (size_t k = 0; k < path.size(); ++k) {
          if (path[k] == bin) {
            path_off[k] = loffset;
            path_has[k] = true;
          }
        }
      }
      p += 16ull * n_chunk;
    }
  }
  uint64_t min_off = 0;
  for (size_t k = path.size(); k-- > 0;) {   // deepest existing bin on the path
    if (path_has[k]) {
      min_off = path_off[k];
      break;
    }
  }
  std::sort(found.begin(), found.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
  for (const Chunk& c : found) {
    if (c.end <= min_off) continue;
    if (!chunks->empty() && c.beg <= chunks->back().end) {
      chunks->back().end = std::max(chunks->back().end, c.end);
    } else {
      chunks->push_back(c);
    }
  }
  return DV_OK;
}

// Indexed read: only the BGZF members the .bai (or .csi) points at are read and inflated.
int read_indexed(const char* path, const std::string& bai, const char* contig, RegionFilter f,
                 dv_read_table* t, int n_threads) {
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return dv::fail(DV_ERR_BAD_INPUT, std::string("cannot open ") + path);
  struct Closer {
    FILE* f;
    ~Closer() { std::fclose(f); }
  } closer{fp};
  // header (may span several members)
  std::vector<uint8_t> buf;
  uint64_t coff = 0;
  size_t first_record = 0;
  bool complete = false;
  while (!complete) {
    size_t csize = 0;
    if (int rc = inflate_member(fp, coff, &buf, &csize)) return rc;
    if (csize == 0) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAM header");
    coff += csize;
    if (int rc = parse_header(buf, contig, &first_record, &f.want_ref, &complete)) return rc;
  }
  if (f.want_ref < 0) return dv::fail(DV_ERR_BAD_INPUT, std::string("contig not in the BAM header: ") + contig);
  std::vector<Chunk> chunks;
  const bool is_csi = bai.size() > 4 && bai.compare(bai.size() - 4, 4, ".csi") == 0;
  if (int rc = is_csi ? csi_chunks(bai, f.want_ref, f.start, f.end, &chunks)
                      : bai_chunks(bai, f.want_ref, f.start, f.end, &chunks)) {
    return rc;
  }
  std::vector<uint8_t> range;
  for (const Chunk& c : chunks) {
    buf.clear();
    uint64_t next = c.beg >> 16;                 // file offset of the next member to inflate
    const uint64_t end_coff = c.end >> 16;
    size_t end_limit = static_cast<size_t>(-1);  // buffer offset matching the chunk's end
    size_t p = c.beg & 0xFFFF;
    bool past = false, eof = false;
    // The chunk's members -- through the one its end lies in, plus room for a record that runs on
    // into the next ones -- are read in one go and inflated on n_threads threads; the loop below
    // then finds its records in memory and only falls back to member-by-member reads for what a
    // record needs beyond that.
    if (end_coff >= next) {
      const size_t want = static_cast<size_t>(end_coff - next) + 3 * 65536;
      range.resize(want);
      size_t got = 0;
      if (fseeko(fp, static_cast<off_t>(next), SEEK_SET) == 0) got = std::fread(range.data(), 1, want, fp);
      std::vector<std::pair<size_t, size_t>> members;
      size_t consumed = 0;
      if (got > 0) {
        if (int rc = inflate_members(range.data(), got, n_threads, true, &buf, &consumed, &members)) return rc;
      }
      for (const auto& m : members) {
        if (next + m.first == end_coff) end_limit = m.second + (c.end & 0xFFFF);
      }
      if (end_limit == static_cast<size_t>(-1) && next + consumed > end_coff) end_limit = buf.size();
      next += consumed;
      if (consumed == 0 && got == 0) eof = true;
    }
    for (;;) {
      // make sure the record at p is complete in buf
      while (!eof && (buf.size() < p + 4 || buf.size() < p + 4 + le32(&buf[p]))) {
        if (next == end_coff && end_limit == static_cast<size_t>(-1)) end_limit = buf.size() + (c.end & 0xFFFF);
        size_t csize = 0;
        if (int rc = inflate_member(fp, next, &buf, &csize)) return rc;
        if (csize == 0) eof = true;
        next += csize;
      }
      if (next > end_coff && end_limit == static_cast<size_t>(-1)) {
        // the end member was inflated without passing through the branch above
        end_limit = buf.size();
      }
      if (buf.size() < p + 4 || p >= end_limit) break;
      const uint32_t block_size = le32(&buf[p]);
      if (block_size < 32) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAM record");
      if (buf.size() < p + 4 + block_size) break;  // ran off the file
      if (int rc = decode_record(&buf[p + 4], block_size, f, t, &past)) return rc;
      if (past) break;
      p += 4 + block_size;
    }
  }
  return DV_OK;
}

}  // namespace

extern "C" {

int dv_bam_read_region(const char* path, const char* contig, int64_t start, int64_t end,
                       const dv_read_requirements* req, int n_threads, dv_read_table** out) {
  if (!path || !out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_bam_read_region: null");
  RegionFilter flt;
  if (req) flt.rq = *req;
  flt.any_contig = contig == nullptr;
  flt.start = start;
  flt.end = end;
  std::unique_ptr<dv_read_table> t(new dv_read_table());
  t->with_mods = flt.rq.parse_base_modifications != 0;
  t->with_flow = flt.rq.parse_flow_tags != 0;
  t->seq_off.push_back(0);
  t->cigar_off.push_back(0);
  const auto t0 = std::chrono::steady_clock::now();
  auto t1 = t0;
  size_t inflated = 0;
  // <path>.bai or <path minus .bam>.bai: seek + inflate only what the region needs
  std::string bai;
  if (contig && getenv("DV_BAM_NO_INDEX") == nullptr) {
    const std::string stem = std::string(path).substr(0, std::strlen(path) > 4 ? std::strlen(path) - 4 : 0);
    for (const std::string& cand : {std::string(path) + ".bai", stem + ".bai", std::string(path) + ".csi", stem + ".csi"}) {
      if (FILE* f = std::fopen(cand.c_str(), "rb")) {
        std::fclose(f);
        bai = cand;
        break;
      }
    }
  }
  if (!bai.empty()) {
    if (int rc = read_indexed(path, bai, contig, flt, t.get(), n_threads)) return rc;
    t1 = std::chrono::steady_clock::now();
  } else {
    std::vector<uint8_t> file;
    {
      FILE* f = std::fopen(path, "rb");
      if (!f) return dv::fail(DV_ERR_BAD_INPUT, std::string("cannot open ") + path);
      std::fseek(f, 0, SEEK_END);
      const long n = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      file.resize(n > 0 ? static_cast<size_t>(n) : 0);
      const size_t got = file.empty() ? 0 : std::fread(file.data(), 1, file.size(), f);
      std::fclose(f);
      if (got != file.size()) return dv::fail(DV_ERR_BAD_INPUT, std::string("short read on ") + path);
    }
    std::vector<uint8_t> buf;
    if (int rc = inflate_bgzf(file, n_threads, &buf)) return rc;
    t1 = std::chrono::steady_clock::now();
    inflated = buf.size();
    file.clear();
    file.shrink_to_fit();
    size_t p = 0;
    bool complete = false;
    if (int rc = parse_header(buf, contig, &p, &flt.want_ref, &complete)) return rc;
    if (!complete) return dv::fail(DV_ERR_BAD_INPUT, "truncated BAM header");
    if (contig && flt.want_ref < 0) {
      return dv::fail(DV_ERR_BAD_INPUT, std::string("contig not in the BAM header: ") + contig);
    }
    while (p + 4 <= buf.size()) {
      const uint32_t block_size = le32(&buf[p]);
      if (block_size < 32 || p + 4 + block_size > buf.size()) {
        return dv::fail(DV_ERR_BAD_INPUT, "truncated BAM record");
      }
      if (int rc = decode_record(&buf[p + 4], block_size, flt, t.get(), nullptr)) return rc;
      p += 4 + block_size;
    }
  }
  if (t->bases.size() >= (1ull << 32) || t->cigar.size() >= (1ull << 32)) {
    return dv::fail(DV_ERR_UNSUPPORTED, "region too large: offsets are 32 bit");
  }
  const auto t2 = std::chrono::steady_clock::now();
  // dense rank under the reference's tuple<string, int> ordering (fragment_name, read_number)
  const size_t n = t->pos.size();
  dv::rank_read_names(t.get());
  if (getenv("DV_BAM_TIMING")) {
    const auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[dv-bam] %s: inflate%s %.1f ms (%zu MB), decode %.1f ms (%zu reads), rank %.1f ms\n",
            bai.empty() ? "full scan" : "indexed", bai.empty() ? "" : "+decode", ms(t0, t1),
            inflated >> 20, ms(t1, t2), n, ms(t2, t3));
  }
  *out = t.release();
  return DV_OK;
}

int dv_read_table_fill_batch(const dv_read_table* t, dv_batch* b) {
  if (!t || !b) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_read_table_fill_batch: null");
  b->memory = DV_MEM_HOST;
  b->n_reads = static_cast<int32_t>(t->pos.size());
  b->read_pos = t->pos.data();
  b->read_sort_pos = nullptr;
  b->read_seq_off = t->seq_off.data();
  b->read_cigar_off = t->cigar_off.data();
  b->read_mapq = t->mapq.data();
  b->read_flags = t->flags.data();
  b->read_frag_len = t->frag_len.data();
  b->read_hp = t->hp.data();
  b->read_name_rank = t->name_rank.data();
  b->read_aux = nullptr;
  b->bases = t->bases.data();
  b->quals = t->quals.data();
  b->mod_5mc = t->with_mods ? t->mod_5mc.data() : nullptr;   // (read_flags carry DV_READ_HAS_5MC / _6MA per read)
  b->mod_6ma = t->with_mods ? t->mod_6ma.data() : nullptr;
  b->cigar = t->cigar.data();
  b->n_bases = static_cast<uint32_t>(t->bases.size());
  b->n_cigar = static_cast<uint32_t>(t->cigar.size());
  return DV_OK;
}

const char* dv_read_table_name(const dv_read_table* t, int32_t i, int32_t* read_number) {
  if (!t || i < 0 || static_cast<size_t>(i) >= t->pos.size()) return nullptr;
  if (read_number) *read_number = t->read_number[i];
  return &t->names[t->name_off[i]];
}

const int64_t* dv_read_table_ends(const dv_read_table* t) { return t ? t->end.data() : nullptr; }

int dv_read_table_names(const dv_read_table* t, const char** blob, const uint32_t** offsets,
                        const uint8_t** read_numbers, uint64_t* blob_bytes) {
  if (!t) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_read_table_names: null");
  if (blob) *blob = t->names.data();
  if (offsets) *offsets = t->name_off.data();
  if (read_numbers) *read_numbers = t->read_number.data();
  if (blob_bytes) *blob_bytes = t->names.size();
  return DV_OK;
}

int dv_read_table_aux_planes(const dv_read_table* t, const uint8_t** mod_5mc, const uint8_t** mod_6ma,
                             const int8_t** tp, const uint8_t** t0, const uint8_t** flow_present) {
  if (!t) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_read_table_aux_planes: null");
  if (mod_5mc) *mod_5mc = t->with_mods ? t->mod_5mc.data() : nullptr;
  if (mod_6ma) *mod_6ma = t->with_mods ? t->mod_6ma.data() : nullptr;
  if (tp) *tp = t->with_flow ? t->tp.data() : nullptr;
  if (t0) *t0 = t->with_flow ? t->t0.data() : nullptr;
  if (flow_present) *flow_present = t->with_flow ? t->flow_present.data() : nullptr;
  return DV_OK;
}

void dv_read_table_free(dv_read_table* t) { delete t; }

}  // extern "C"
