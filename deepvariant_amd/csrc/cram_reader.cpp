// cram_reader.cpp -- CRAM 3.0 -> the packed read table of dv_batch, on the host.
//
// SURVEY.md 8f row f1 (alignment file -> packed read SoA), CRAM side.  The reference opens CRAM
// through htslib (third_party/nucleus/io/sam_reader.cc:560-640: hts_open, --use_ref_for_cram ->
// hts_set_opt(CRAM_OPT_REFERENCE), :1022-1035) and converts every record to a Read proto
// (:734-840 ConvertToPb).  htslib is not in this image, so this file restates the published format
// (CRAM format specification v3.0, samtools/hts-specs) at the level make_examples needs -- the reads of
// one contig interval -- and writes them straight into the structure-of-arrays layout the encoder
// and the allele counter read (the same dv_read_table bam_reader.cpp fills; no per-read objects).
//
// Decoded: file definition, container / slice headers, the compression header (preservation map,
// data-series and tag encodings), blocks in raw / gzip / bzip2 / lzma / rANS 4x8 (order 0 and 1), the
// encodings EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN, BYTE_ARRAY_STOP, BETA, SUBEXP, GAMMA, every read
// feature, reference-based sequence reconstruction (callback for the external FASTA, or a slice's
// embedded reference; the slice's reference MD5 is checked), mate links inside a slice with htslib's
// template-length rule (cram/cram_decode.c cram_decode_slice_xref), the .crai index.  The slices a
// query touches are decoded on `n_threads` host threads.  Not supported (DV_ERR_UNSUPPORTED): CRAM 2.x /
// 3.1 codecs, GOLOMB / GOLOMB_RICE encodings.
//
// deepvariant_amd/cram_reader.py is the same decoder in Python (it also yields Read objects for the
// object path); tests/test_cram_native_cpu.py holds the two against each other and against the BAM of
// the same alignments.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <array>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dv_internal.h"
#include "aux_planes.h"
#include "read_table.h"

namespace {

struct CramError : std::runtime_error {
  int status;
  CramError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};
[[noreturn]] void bad(const std::string& m) { throw CramError(DV_ERR_BAD_INPUT, m); }
[[noreturn]] void unsupported(const std::string& m) { throw CramError(DV_ERR_UNSUPPORTED, m); }

// ---- bounds-checked byte cursor --------------------------------------------------------
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  uint8_t u8() {
    if (p >= end) bad("truncated CRAM data");
    return *p++;
  }
  void need(size_t n) const {
    if (static_cast<size_t>(end - p) < n) bad("truncated CRAM data");
  }
  const uint8_t* take(size_t n) {
    need(n);
    const uint8_t* q = p;
    p += n;
    return q;
  }
  uint32_t le32() {
    const uint8_t* q = take(4);
    return q[0] | (q[1] << 8) | (q[2] << 16) | (static_cast<uint32_t>(q[3]) << 24);
  }
};

int32_t itf8(Cursor& c) {
  const uint32_t v = c.u8();
  if (v < 0x80) return static_cast<int32_t>(v);
  if (v < 0xC0) return static_cast<int32_t>(((v & 0x3F) << 8) | c.u8());
  if (v < 0xE0) {
    const uint32_t a = c.u8(), b = c.u8();
    return static_cast<int32_t>(((v & 0x1F) << 16) | (a << 8) | b);
  }
  if (v < 0xF0) {
    const uint32_t a = c.u8(), b = c.u8(), d = c.u8();
    return static_cast<int32_t>(((v & 0x0F) << 24) | (a << 16) | (b << 8) | d);
  }
  const uint32_t a = c.u8(), b = c.u8(), d = c.u8(), e = c.u8();
  return static_cast<int32_t>(((v & 0x0F) << 28) | (a << 20) | (b << 12) | (d << 4) | (e & 0x0F));
}

int64_t ltf8(Cursor& c) {
  const uint8_t v = c.u8();
  int n = 0;
  while (n < 8 && ((v << n) & 0x80)) ++n;
  if (n == 0) return v;
  uint64_t x = n == 8 ? 0 : (v & (0xFFu >> (n + 1)));
  for (int k = 0; k < n; ++k) x = (x << 8) | c.u8();
  return static_cast<int64_t>(x);
}

std::vector<int32_t> itf8_array(Cursor& c) {
  const int32_t n = itf8(c);
  if (n < 0) bad("negative array length in a CRAM header");
  std::vector<int32_t> out;
  out.reserve(static_cast<size_t>(n));
  for (int32_t i = 0; i < n; ++i) out.push_back(itf8(c));
  return out;
}

// ---- MD5 (RFC 1321): the slice header's checksum of the reference stretch -----------------
void md5(const uint8_t* data, size_t len, uint8_t out[16]) {
  static const uint32_t K[64] = {
      0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8,
      0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340,
      0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87,
      0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
      0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039,
      0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92,
      0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb,
      0xeb86d391};
  static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                            14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                            4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
  uint32_t h[4] = {0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476};
  auto block = [&](const uint8_t* b) {
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) {
      m[i] = b[4 * i] | (b[4 * i + 1] << 8) | (b[4 * i + 2] << 16) | (static_cast<uint32_t>(b[4 * i + 3]) << 24);
    }
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3];
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) {
        f = (bb & c) | (~bb & d);
        g = i;
      } else if (i < 32) {
        f = (d & bb) | (~d & c);
        g = (5 * i + 1) & 15;
      } else if (i < 48) {
        f = bb ^ c ^ d;
        g = (3 * i + 5) & 15;
      } else {
        f = c ^ (bb | ~d);
        g = (7 * i) & 15;
      }
      const uint32_t t = a + f + K[i] + m[g];
      a = d;
      d = c;
      c = bb;
      bb += (t << S[i]) | (t >> (32 - S[i]));
    }
    h[0] += a;
    h[1] += bb;
    h[2] += c;
    h[3] += d;
  };
  size_t i = 0;
  for (; i + 64 <= len; i += 64) block(data + i);
  uint8_t tail[128] = {0};
  const size_t rest = len - i;
  std::memcpy(tail, data + i, rest);
  tail[rest] = 0x80;
  const size_t padded = rest + 1 + 8 <= 64 ? 64 : 128;
  const uint64_t bits = static_cast<uint64_t>(len) * 8;
  for (int k = 0; k < 8; ++k) tail[padded - 8 + k] = static_cast<uint8_t>(bits >> (8 * k));
  block(tail);
  if (padded == 128) block(tail + 64);
  for (int k = 0; k < 4; ++k) {
    for (int j = 0; j < 4; ++j) out[4 * k + j] = static_cast<uint8_t>(h[k] >> (8 * j));
  }
}

// ---- block codecs ------------------------------------------------------------------------
void inflate_gzip(const uint8_t* in, size_t n_in, std::vector<uint8_t>* out, size_t size_hint) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 32) != Z_OK) bad("zlib init failed");
  out->resize(size_hint ? size_hint : std::max<size_t>(4 * n_in, 4096));
  zs.next_in = const_cast<Bytef*>(in);
  zs.avail_in = static_cast<uInt>(n_in);
  size_t produced = 0;
  bool ok = true;
  for (;;) {
    if (produced == out->size()) out->resize(out->size() * 2 + 4096);
    zs.next_out = out->data() + produced;
    zs.avail_out = static_cast<uInt>(std::min<size_t>(out->size() - produced, 1u << 30));
    const int rc = inflate(&zs, Z_NO_FLUSH);
    produced = static_cast<size_t>(zs.next_out - out->data());
    if (rc == Z_STREAM_END) {
      if (zs.avail_in == 0) break;
      if (inflateReset(&zs) != Z_OK) {   // a further gzip member follows
        ok = false;
        break;
      }
      continue;
    }
    if (rc == Z_BUF_ERROR && zs.avail_out == 0) continue;   // output full: grow and go on
    if (rc != Z_OK || (zs.avail_in == 0 && zs.avail_out != 0)) {
      ok = false;   // corrupt or truncated
      break;
    }
  }
  inflateEnd(&zs);
  if (!ok) bad("gzip data failed to inflate");
  out->resize(produced);
}

// bzip2 / lzma: the shared libraries are on every box, their headers are not in this image
using Bz2Fn = int (*)(char*, unsigned*, char*, unsigned, int, int);
using LzmaFn = unsigned (*)(uint64_t*, uint32_t, const void*, const uint8_t*, size_t*, size_t, uint8_t*, size_t*,
                            size_t);
void inflate_bz2(const uint8_t* in, size_t n_in, std::vector<uint8_t>* out, size_t raw_size) {
  static Bz2Fn fn = []() -> Bz2Fn {
    for (const char* name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) {
      if (void* h = dlopen(name, RTLD_NOW)) return reinterpret_cast<Bz2Fn>(dlsym(h, "BZ2_bzBuffToBuffDecompress"));
    }
    return nullptr;
  }();
  if (!fn) unsupported("CRAM block compressed with bzip2 and libbz2 is not on this host");
  out->resize(raw_size);
  unsigned n = static_cast<unsigned>(raw_size);
  if (fn(reinterpret_cast<char*>(out->data()), &n, reinterpret_cast<char*>(const_cast<uint8_t*>(in)),
         static_cast<unsigned>(n_in), 0, 0) != 0) {
    bad("bzip2 block failed to inflate");
  }
  out->resize(n);
}
void inflate_lzma(const uint8_t* in, size_t n_in, std::vector<uint8_t>* out, size_t raw_size) {
  static LzmaFn fn = []() -> LzmaFn {
    for (const char* name : {"liblzma.so.5", "liblzma.so"}) {
      if (void* h = dlopen(name, RTLD_NOW)) return reinterpret_cast<LzmaFn>(dlsym(h, "lzma_stream_buffer_decode"));
    }
    return nullptr;
  }();
  if (!fn) unsupported("CRAM block compressed with lzma and liblzma is not on this host");
  out->resize(raw_size);
  uint64_t memlimit = UINT64_MAX;
  size_t in_pos = 0, out_pos = 0;
  if (fn(&memlimit, 0, nullptr, in, &in_pos, n_in, out->data(), &out_pos, raw_size) != 0) {
    bad("lzma block failed to inflate");
  }
  out->resize(out_pos);
}

// rANS 4x8 (CRAM 3.0 section 13 / htslib rANS_static.c): four interleaved states, 12-bit frequencies.
constexpr uint32_t kRansLow = 1u << 23;
constexpr int kTfShift = 12;
constexpr uint32_t kTfMask = (1u << kTfShift) - 1;

struct RansTable {
  uint16_t freq[256];
  uint16_t cum[256];
  uint8_t slot[1 << kTfShift];
  bool present = false;
};

void rans_read_freqs(Cursor& c, RansTable* t) {
  std::memset(t->freq, 0, sizeof(t->freq));
  int sym = c.u8(), rle = 0, last = sym;
  for (;;) {
    uint32_t f = c.u8();
    if (f >= 0x80) f = ((f & 0x7F) << 8) | c.u8();
    t->freq[sym & 255] = static_cast<uint16_t>(f);
    if (rle) {
      --rle;
      ++sym;
    } else {
      sym = c.u8();
      if (sym == last + 1) rle = c.u8();
    }
    last = sym;
    if (sym == 0) break;
    if (sym > 255) bad("bad rANS frequency table");
  }
  uint32_t x = 0;
  for (int s = 0; s < 256; ++s) {
    t->cum[s] = static_cast<uint16_t>(x);
    const uint32_t f = t->freq[s];
    if (f) {
      if (x + f > (1u << kTfShift)) bad("rANS frequencies exceed 4096");
      std::memset(t->slot + x, s, f);
      x += f;
    }
  }
  if (x < (1u << kTfShift)) std::memset(t->slot + x, 0, (1u << kTfShift) - x);
  t->present = true;
}

void rans_decode(const uint8_t* in, size_t n_in, std::vector<uint8_t>* out) {
  Cursor c{in, in + n_in};
  const int order = c.u8();
  c.le32();                       // compressed size
  const uint32_t out_size = c.le32();
  // Refuse sizes that a few bytes of input could not have produced before allocating them (a 9-byte block may not
  // ask for 4 GiB).  The real bound of rANS 4x8: frequencies are normalised to 4095 / 4096 at most, so a dominant
  // symbol costs log2(4096 / 4095) = 3.5e-4 bits -- ~22,700 symbols per payload byte -- and each of the four
  // states absorbs ~23 K symbols before it emits its first byte (1.5 M constant quality values of a 10 k-read slice
  // are a 64-byte payload).  32,768 symbols per input byte + 1 Mi covers both; nothing a slice holds nears 1 GiB.
  if (static_cast<uint64_t>(out_size) > (static_cast<uint64_t>(n_in) + 16) * 32768 + (1u << 20) ||
      out_size > (1u << 30)) {
    bad("rANS block states an impossible size");
  }
  out->assign(out_size, 0);
  if (out_size == 0) return;
  uint8_t* o = out->data();
  if (order == 0) {
    auto t = std::make_unique<RansTable>();
    rans_read_freqs(c, t.get());
    uint32_t r[4];
    for (int j = 0; j < 4; ++j) r[j] = c.le32();
    const uint8_t* p = c.p;
    const uint8_t* const e = c.end;
    for (uint32_t k = 0; k < out_size; ++k) {
      uint32_t x = r[k & 3];
      const uint32_t m = x & kTfMask;
      const uint8_t s = t->slot[m];
      o[k] = s;
      x = t->freq[s] * (x >> kTfShift) + m - t->cum[s];
      while (x < kRansLow && p < e) x = (x << 8) | *p++;
      r[k & 3] = x;
    }
    return;
  }
  if (order != 1) unsupported("rANS order " + std::to_string(order));
  std::vector<std::unique_ptr<RansTable>> tables(256);
  {
    int ctx = c.u8(), rle = 0, last = ctx;
    for (;;) {
      if (ctx > 255) bad("bad rANS context table");
      tables[ctx] = std::make_unique<RansTable>();
      rans_read_freqs(c, tables[ctx].get());
      if (rle) {
        --rle;
        ++ctx;
      } else {
        ctx = c.u8();
        if (ctx == last + 1) rle = c.u8();
      }
      last = ctx;
      if (ctx == 0) break;
    }
  }
  static const RansTable* const empty = []() {
    auto* t = new RansTable();
    std::memset(t, 0, sizeof(*t));
    return t;
  }();
  uint32_t r[4];
  for (int j = 0; j < 4; ++j) r[j] = c.le32();
  const uint8_t* p = c.p;
  const uint8_t* const e = c.end;
  const uint32_t q = out_size >> 2;
  uint32_t pos[4] = {0, q, 2 * q, 3 * q};
  uint8_t prev[4] = {0, 0, 0, 0};
  auto step = [&](int j) {
    const RansTable* t = tables[prev[j]] ? tables[prev[j]].get() : empty;
    uint32_t x = r[j];
    const uint32_t m = x & kTfMask;
    const uint8_t s = t->slot[m];
    o[pos[j]++] = s;
    x = t->freq[s] * (x >> kTfShift) + m - t->cum[s];
    while (x < kRansLow && p < e) x = (x << 8) | *p++;
    r[j] = x;
    prev[j] = s;
  };
  for (uint32_t i = 0; i < q; ++i) {
    step(0);
    step(1);
    step(2);
    step(3);
  }
  while (pos[3] < out_size) step(3);   // the remainder belongs to the fourth stream
}

// ---- blocks --------------------------------------------------------------------------------
struct Block {
  int content_type = 0;
  int32_t content_id = 0;
  const uint8_t* data = nullptr;   // into the mapped file (raw blocks) or into `owned`
  size_t size = 0;
  size_t pos = 0;                  // read position of the EXTERNAL codecs
  std::vector<uint8_t> owned;
};

// Parses the block at c (header + payload + CRC32); `decode` = false only walks past it.
void read_block(Cursor& c, Block* b, bool decode = true) {
  const int method = c.u8();
  b->content_type = c.u8();
  b->content_id = itf8(c);
  const int32_t csize = itf8(c), rsize = itf8(c);
  if (csize < 0 || rsize < 0) bad("negative CRAM block size");
  const uint8_t* payload = c.take(static_cast<size_t>(csize));
  c.take(4);  // CRC32
  b->pos = 0;
  if (!decode) return;
  // A stated raw size that the payload could not have produced is refused before it is allocated.  Deflate expands
  // by at most 1032x; bzip2 (RLE1 + BWT + RLE2) and LZMA reach ~1,000,000x on constant data -- which real CRAM blocks
  // hold (one quality value, one flag for every read) -- so their bound is 2^21 per payload byte; rANS has its own
  // check; and no block of a slice nears 1 GiB whatever the codec.
  if (method != 0) {
    const uint64_t per_byte = method == 1 ? 1100u : (1u << 21);
    if (static_cast<uint64_t>(rsize) > (static_cast<uint64_t>(csize) + 64) * per_byte || rsize > (1 << 30)) {
      bad("CRAM block states an impossible uncompressed size");
    }
  }
  switch (method) {
    case 0:
      b->data = payload;
      b->size = static_cast<size_t>(csize);
      break;
    case 1:
      inflate_gzip(payload, csize, &b->owned, rsize);
      break;
    case 2:
      inflate_bz2(payload, csize, &b->owned, rsize);
      break;
    case 3:
      inflate_lzma(payload, csize, &b->owned, rsize);
      break;
    case 4:
      rans_decode(payload, csize, &b->owned);
      break;
    default:
      unsupported("CRAM block compression method " + std::to_string(method) + " is not supported");
  }
  if (method != 0) {
    b->data = b->owned.data();
    b->size = b->owned.size();
  }
  if (b->size != static_cast<size_t>(rsize)) bad("CRAM block inflates to a size its header does not state");
}

struct Bits {   // the core data block: bits, most significant first
  const uint8_t* data = nullptr;
  size_t size = 0;
  size_t pos = 0;   // in bits
  uint32_t read(int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; ++i) {
      const size_t byte = pos >> 3;
      if (byte >= size) bad("CRAM core block exhausted");
      v = (v << 1) | ((data[byte] >> (7 - (pos & 7))) & 1);
      ++pos;
    }
    return v;
  }
};

// ---- encodings ---------------------------------------------------------------------------
struct Encoding {
  int codec = 0;
  std::vector<uint8_t> params;
};

Encoding parse_encoding(Cursor& c) {
  Encoding e;
  e.codec = itf8(c);
  const int32_t n = itf8(c);
  if (n < 0) bad("negative encoding parameter size");
  const uint8_t* p = c.take(static_cast<size_t>(n));
  e.params.assign(p, p + n);
  return e;
}

struct SliceData {
  std::map<int32_t, Block*> external;
  Bits core;
};

// Reader of one data series of one slice.
struct Decoder {
  int codec = 0;              // 0 = absent / NULL: integers read as 0, arrays as empty
  Block* blk = nullptr;       // EXTERNAL, BYTE_ARRAY_STOP
  Bits* core = nullptr;
  // HUFFMAN (canonical: symbols ordered by (length, value))
  bool constant = false;
  int32_t only = 0;
  std::vector<int32_t> syms;
  int count[34] = {0};
  int32_t offset = 0;
  int nbits = 0;
  uint8_t stop = 0;
  std::unique_ptr<Decoder> len_dec, val_dec;

  void init(const Encoding& e, SliceData* s, bool array) {
    codec = e.codec;
    core = &s->core;
    Cursor c{e.params.data(), e.params.data() + e.params.size()};
    auto block_of = [&](int32_t cid) -> Block* {
      auto it = s->external.find(cid);
      return it == s->external.end() ? nullptr : it->second;
    };
    if (array) {
      if (codec == 5) {            // BYTE_ARRAY_STOP
        stop = c.u8();
        blk = block_of(itf8(c));
      } else if (codec == 4) {     // BYTE_ARRAY_LEN
        const Encoding le = parse_encoding(c), ve = parse_encoding(c);
        len_dec = std::make_unique<Decoder>();
        len_dec->init(le, s, false);
        val_dec = std::make_unique<Decoder>();
        val_dec->init(ve, s, false);
      } else {
        unsupported("CRAM byte-array encoding " + std::to_string(codec) + " is not supported");
      }
      return;
    }
    switch (codec) {
      case 0:
        break;
      case 1:
        blk = block_of(itf8(c));
        break;
      case 3: {
        const std::vector<int32_t> alphabet = itf8_array(c), lengths = itf8_array(c);
        if (alphabet.size() != lengths.size() || alphabet.empty()) bad("bad Huffman encoding");
        if (alphabet.size() == 1 && lengths[0] == 0) {
          constant = true;
          only = alphabet[0];
          break;
        }
        std::vector<size_t> order(alphabet.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = k;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
          return lengths[a] != lengths[b] ? lengths[a] < lengths[b] : alphabet[a] < alphabet[b];
        });
        for (size_t k : order) {
          if (lengths[k] < 1 || lengths[k] > 32) bad("bad Huffman code length");
          ++count[lengths[k]];
          syms.push_back(alphabet[k]);
        }
        break;
      }
      case 6:
        offset = itf8(c);
        nbits = itf8(c);
        if (nbits < 0 || nbits > 32) bad("bad BETA bit count");
        break;
      case 7:
        offset = itf8(c);
        nbits = itf8(c);   // k
        if (nbits < 0 || nbits > 32) bad("bad SUBEXP k");
        break;
      case 9:
        offset = itf8(c);
        break;
      default:
        unsupported("CRAM integer encoding " + std::to_string(codec) + " is not supported");
    }
  }

  int32_t read_int() {
    switch (codec) {
      case 0:
        return 0;
      case 1: {
        if (!blk) bad("CRAM data series refers to a missing external block");
        Cursor c{blk->data + blk->pos, blk->data + blk->size};
        const int32_t v = itf8(c);
        blk->pos = static_cast<size_t>(c.p - blk->data);
        return v;
      }
      case 3: {
        if (constant) return only;
        // canonical decode: codes of one length are consecutive, the first of a longer length is
        // (last of the shorter + 1) shifted
        uint64_t code = 0, first = 0;   // 64-bit unsigned: a 32-bit code shifted once more stays defined
        int64_t index = 0;
        for (int len = 1; len <= 32; ++len) {
          code |= static_cast<uint64_t>(core->read(1));
          const uint64_t n = static_cast<uint64_t>(count[len]);
          if (code >= first && code - first < n) return syms[static_cast<size_t>(index + static_cast<int64_t>(code - first))];
          index += static_cast<int64_t>(n);
          first += n;
          first <<= 1;
          code <<= 1;
        }
        bad("bad Huffman code in CRAM core block");
      }
      case 6:
        return static_cast<int32_t>(core->read(nbits)) - offset;
      case 7: {
        int n = 0;
        while (core->read(1)) ++n;
        if (n == 0) return static_cast<int32_t>(core->read(nbits)) - offset;
        const int bits = n + nbits - 1;
        if (bits < 0 || bits > 31) bad("bad SUBEXP value");
        return static_cast<int32_t>((1u << bits) | core->read(bits)) - offset;
      }
      case 9: {
        int n = 0;
        while (core->read(1) == 0) ++n;
        if (n > 31) bad("bad GAMMA value");
        return static_cast<int32_t>((1u << n) | core->read(n)) - offset;
      }
      default:
        bad("integer read through a byte-array encoding");
    }
  }

  int32_t read_byte() {
    if (codec == 1) {
      if (!blk) bad("CRAM data series refers to a missing external block");
      if (blk->pos >= blk->size) bad("CRAM external block exhausted");
      return blk->data[blk->pos++];
    }
    return read_int();
  }

  // appends the array to `out`, returns its length
  size_t read_bytes(std::vector<uint8_t>* out) {
    if (codec == 0) return 0;
    if (codec == 5) {
      if (!blk) bad("CRAM data series refers to a missing external block");
      const uint8_t* b = blk->data + blk->pos;
      const void* hit = std::memchr(b, stop, blk->size - blk->pos);
      if (!hit) bad("CRAM BYTE_ARRAY_STOP without its stop byte");
      const size_t n = static_cast<size_t>(static_cast<const uint8_t*>(hit) - b);
      out->insert(out->end(), b, b + n);
      blk->pos += n + 1;
      return n;
    }
    const int32_t n = len_dec->read_int();
    if (n < 0) bad("negative CRAM byte-array length");
    if (val_dec->codec == 1) {
      Block* v = val_dec->blk;
      if (!v) bad("CRAM data series refers to a missing external block");
      // (Python slicing semantics of the twin decoder: a short block yields what is left)
      const size_t got = std::min<size_t>(static_cast<size_t>(n), v->size - std::min(v->pos, v->size));
      out->insert(out->end(), v->data + v->pos, v->data + v->pos + got);
      v->pos += static_cast<size_t>(n);
      if (v->pos > v->size) bad("CRAM external block exhausted");
      return got;
    }
    for (int32_t i = 0; i < n; ++i) out->push_back(static_cast<uint8_t>(val_dec->read_byte()));
    return static_cast<size_t>(n);
  }
};

// ---- compression header --------------------------------------------------------------------
inline uint16_t key2(const uint8_t* p) { return static_cast<uint16_t>((p[0] << 8) | p[1]); }
constexpr uint16_t K2(char a, char b) { return static_cast<uint16_t>((static_cast<uint8_t>(a) << 8) | static_cast<uint8_t>(b)); }

struct CompressionHeader {
  bool read_names = true, ap_delta = true, ref_required = true;
  uint8_t subst[5] = {0x1B, 0x1B, 0x1B, 0x1B, 0x1B};
  std::vector<std::vector<std::array<uint8_t, 3>>> tag_lists{{}};
  std::map<uint16_t, Encoding> series;
  std::map<int32_t, Encoding> tags;
  char subst_lookup[5][4];

  explicit CompressionHeader(const Block& b) {
    Cursor c{b.data, b.data + b.size};
    itf8(c);   // map size in bytes
    int32_t n = itf8(c);
    for (int32_t i = 0; i < n; ++i) {
      const uint16_t key = key2(c.take(2));
      if (key == K2('R', 'N')) {
        read_names = c.u8() != 0;
      } else if (key == K2('A', 'P')) {
        ap_delta = c.u8() != 0;
      } else if (key == K2('R', 'R')) {
        ref_required = c.u8() != 0;
      } else if (key == K2('S', 'M')) {
        std::memcpy(subst, c.take(5), 5);
      } else if (key == K2('T', 'D')) {
        const int32_t ln = itf8(c);
        if (ln < 0) bad("bad TD length");
        const uint8_t* td = c.take(static_cast<size_t>(ln));
        tag_lists.clear();
        // entries separated by NUL; a trailing NUL closes the last entry
        size_t s = 0;
        for (size_t k = 0; k <= static_cast<size_t>(ln); ++k) {
          if (k == static_cast<size_t>(ln) || td[k] == 0) {
            if (k == static_cast<size_t>(ln) && ln > 0 && td[ln - 1] == 0) break;
            std::vector<std::array<uint8_t, 3>> entry;
            for (size_t q = s; q + 3 <= k; q += 3) entry.push_back({td[q], td[q + 1], td[q + 2]});
            tag_lists.push_back(std::move(entry));
            s = k + 1;
          }
        }
      } else {
        bad("unknown CRAM preservation key");
      }
    }
    itf8(c);
    n = itf8(c);
    for (int32_t i = 0; i < n; ++i) {
      const uint16_t key = key2(c.take(2));
      series[key] = parse_encoding(c);
    }
    itf8(c);
    n = itf8(c);
    for (int32_t i = 0; i < n; ++i) {
      const int32_t key = itf8(c);
      tags[key] = parse_encoding(c);
    }
    // substitution matrix: for reference base R (A C G T N) the code (0..3) of each other base
    static const char kBases[] = "ACGTN";
    std::memset(subst_lookup, 'N', sizeof(subst_lookup));
    for (int r = 0; r < 5; ++r) {
      int k = 0;
      for (int a = 0; a < 5; ++a) {
        if (a == r) continue;
        subst_lookup[r][(subst[r] >> (6 - 2 * k)) & 3] = kBases[a];
        ++k;
      }
    }
  }
};

// ---- the file ----------------------------------------------------------------------------------
struct ContainerHeader {
  int32_t ref_id = 0, start = 0, span = 0, n_records = 0, n_blocks = 0;
  std::vector<int32_t> landmarks;
  size_t blocks = 0, next = 0;   // file offsets
};

struct CramFile {
  std::string path;
  int fd = -1;
  const uint8_t* buf = nullptr;
  size_t size = 0;
  std::string header_text;
  std::vector<std::string> contig_names;
  std::vector<int64_t> contig_lengths;
  size_t first_data_container = 0;

  explicit CramFile(const char* p) : path(p) {
    fd = ::open(p, O_RDONLY);
    if (fd < 0) bad(std::string("cannot open ") + p);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 26) {
      ::close(fd);
      fd = -1;
      bad(std::string("Failed to parse BAM/CRAM file. ") + p + ": too short");
    }
    size = static_cast<size_t>(st.st_size);
    // mapped, not read: a whole-genome CRAM is tens of gigabytes, a query touches the header container
    // and the containers its index (or a walk over the container headers) selects
    void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      ::close(fd);
      fd = -1;
      bad(std::string("cannot map ") + p);
    }
    buf = static_cast<const uint8_t*>(m);
    try {
      if (std::memcmp(buf, "CRAM", 4) != 0) bad(std::string("Failed to parse BAM/CRAM file. ") + p + ": bad CRAM magic");
      if (buf[4] != 3 || buf[5] != 0) {
        unsupported("CRAM version " + std::to_string(buf[4]) + "." + std::to_string(buf[5]) + " is not supported (3.0 is)");
      }
      const ContainerHeader h = container(26);
      Cursor c{buf + h.blocks, buf + size};
      Block blk;
      read_block(c, &blk);
      if (blk.size < 4) bad("truncated CRAM header block");
      const uint32_t n = blk.data[0] | (blk.data[1] << 8) | (blk.data[2] << 16) | (static_cast<uint32_t>(blk.data[3]) << 24);
      if (4 + static_cast<size_t>(n) > blk.size) bad("truncated CRAM header block");
      header_text.assign(reinterpret_cast<const char*>(blk.data + 4), n);
      parse_sq();
      first_data_container = h.next;
    } catch (...) {
      munmap(const_cast<uint8_t*>(buf), size);
      ::close(fd);
      throw;
    }
  }
  ~CramFile() {
    if (buf) munmap(const_cast<uint8_t*>(buf), size);
    if (fd >= 0) ::close(fd);
  }
  CramFile(const CramFile&) = delete;
  CramFile& operator=(const CramFile&) = delete;

  void parse_sq() {
    size_t i = 0;
    while (i < header_text.size()) {
      size_t e = header_text.find('\n', i);
      if (e == std::string::npos) e = header_text.size();
      if (e - i >= 3 && header_text.compare(i, 3, "@SQ") == 0) {
        std::string name;
        int64_t length = 0;
        size_t f = i;
        while (f < e) {
          size_t t = header_text.find('\t', f);
          if (t == std::string::npos || t > e) t = e;
          if (t - f > 3 && header_text[f + 2] == ':') {
            if (header_text.compare(f, 2, "SN") == 0) name = header_text.substr(f + 3, t - f - 3);
            if (header_text.compare(f, 2, "LN") == 0) length = std::atoll(header_text.substr(f + 3, t - f - 3).c_str());
          }
          f = t + 1;
        }
        contig_names.push_back(name);
        contig_lengths.push_back(length);
      }
      i = e + 1;
    }
  }

  ContainerHeader container(size_t i) const {
    if (i > size || size - i < 4) bad("truncated CRAM container");   // (i + 4 would wrap for an offset near 2^64)
    Cursor c{buf + i, buf + size};
    const int32_t length = static_cast<int32_t>(c.le32());
    if (length < 0) bad("negative CRAM container length");
    ContainerHeader h;
    h.ref_id = itf8(c);
    h.start = itf8(c);
    h.span = itf8(c);
    h.n_records = itf8(c);
    ltf8(c);   // record counter
    ltf8(c);   // bases
    h.n_blocks = itf8(c);
    h.landmarks = itf8_array(c);
    c.take(4);   // CRC32
    h.blocks = static_cast<size_t>(c.p - buf);
    h.next = h.blocks + static_cast<size_t>(length);
    return h;
  }

  bool is_eof_marker(const ContainerHeader& h) const {
    return h.ref_id == -1 && h.n_records == 0 && h.start == 4542278;
  }

  // <path>.crai: gzip text, one line per slice (and per reference of a multi-reference slice):
  // reference id, start, span, container offset, slice offset, slice size (CRAM 3.0 section 12).
  bool index(std::vector<std::array<int64_t, 4>>* rows) const {
    std::vector<std::string> cands{path + ".crai"};
    if (path.size() > 5 && path.compare(path.size() - 5, 5, ".cram") == 0) cands.push_back(path.substr(0, path.size() - 5) + ".crai");
    for (const std::string& cand : cands) {
      FILE* f = std::fopen(cand.c_str(), "rb");
      if (!f) continue;
      std::vector<uint8_t> raw;
      uint8_t tmp[65536];
      size_t got;
      while ((got = std::fread(tmp, 1, sizeof(tmp), f)) > 0) raw.insert(raw.end(), tmp, tmp + got);
      std::fclose(f);
      std::vector<uint8_t> text;
      inflate_gzip(raw.data(), raw.size(), &text, 0);
      size_t i = 0;
      while (i < text.size()) {
        size_t e = i;
        while (e < text.size() && text[e] != '\n') ++e;
        std::array<int64_t, 4> row{};
        int field = 0;
        size_t p = i;
        while (p < e && field < 4) {
          size_t t = p;
          while (t < e && text[t] != '\t') ++t;
          row[static_cast<size_t>(field++)] = std::atoll(std::string(reinterpret_cast<const char*>(&text[p]), t - p).c_str());
          p = t + 1;
        }
        if (field >= 4) rows->push_back(row);
        i = e + 1;
      }
      return true;
    }
    return false;
  }

  // The containers that can hold reads of reference `want` overlapping [start, end) (all data
  // containers when want < 0): by the .crai when there is one, else by walking the container headers.
  std::vector<ContainerHeader> containers_for(int32_t want, int64_t start, int64_t end) const {
    std::vector<ContainerHeader> out;
    std::vector<std::array<int64_t, 4>> rows;
    if (want >= 0 && index(&rows)) {
      std::vector<int64_t> seen;
      for (const auto& r : rows) {
        if (r[0] != want || r[1] - 1 >= end || r[1] - 1 + r[2] <= start) continue;
        if (std::find(seen.begin(), seen.end(), r[3]) != seen.end()) continue;
        seen.push_back(r[3]);
        // the offsets come from a text file: never trust them further than the bytes that are there
        if (r[1] < 0 || r[2] < 0 || r[3] < 0 || static_cast<uint64_t>(r[3]) >= size) bad("CRAM index (.crai) row outside the file");
        ContainerHeader h = container(static_cast<size_t>(r[3]));
        if (h.n_blocks > 0) out.push_back(std::move(h));
      }
      return out;
    }
    size_t i = first_data_container;
    while (i < size) {
      ContainerHeader h = container(i);
      i = h.next;
      if (h.n_blocks <= 0 || is_eof_marker(h)) continue;
      if (want >= 0) {
        // a single-reference container of another contig or interval, or the unmapped tail (-1), cannot
        // hold a read of the query; multi-reference containers (-2) are looked into
        if (h.ref_id == -1) continue;
        if (h.ref_id >= 0 && (h.ref_id != want || static_cast<int64_t>(h.start) - 1 >= end ||
                              static_cast<int64_t>(h.start) - 1 + h.span <= start)) {
          continue;
        }
      }
      out.push_back(std::move(h));
    }
    return out;
  }
};

// ---- one slice ------------------------------------------------------------------------------
struct Rec {
  int32_t flag = 0, cram_flags = 0, ref_id = -1, read_length = 0, pos = 0, mate_flags = 0, mate_ref_id = -1,
          mate_pos = 0, tlen = 0, next_fragment = -1, mapq = 0, ref_len = 0, mate_line = -1, hp = 0;
  bool tlen_known = false, has_hp = false, has_oq = false;
  uint32_t name_off = 0, name_len = 0, seq_off = 0, cig_off = 0, cig_n = 0, oq_off = 0, oq_len = 0;
  // the BAM-encoded values of MM, ML, MN, tp, t0 in SliceOut::aux (kept only when the requirements ask for planes)
  uint32_t aux_off[5] = {0, 0, 0, 0, 0}, aux_len[5] = {0, 0, 0, 0, 0};
  char aux_type[5] = {0, 0, 0, 0, 0};
};

struct SliceOut {
  std::vector<Rec> recs;
  std::vector<uint8_t> bases, quals, names, oq, aux;
  std::vector<uint32_t> cigar;   // BAM words (len << 4 | op), ops M I D N S H P = 0..6
  bool keep_aux = false;
};

using FetchFn = dv_ref_fetch_fn;

struct RefSource {
  const CramFile* file;
  FetchFn fetch;
  void* ctx;
  std::mutex* mu;   // the callback is entered by one thread at a time
  // bases of [start, end) of contig `ref_id` as the callback yields them (may be short at a contig's end)
  std::string get(int32_t ref_id, int64_t start, int64_t end) const {
    if (!fetch) bad("Failed to parse BAM/CRAM file. " + file->path + " needs a reference (--ref) to be decoded");
    if (ref_id < 0 || static_cast<size_t>(ref_id) >= file->contig_names.size()) bad("CRAM record on an unknown reference");
    if (end <= start) return std::string();
    std::string out(static_cast<size_t>(end - start), '\0');
    int64_t n = 0;
    int rc;
    {
      std::lock_guard<std::mutex> lock(*mu);
      rc = fetch(ctx, file->contig_names[static_cast<size_t>(ref_id)].c_str(), start, end, &out[0], &n);
    }
    if (rc != 0) bad("the reference callback failed for " + file->contig_names[static_cast<size_t>(ref_id)]);
    if (n < 0) n = 0;
    out.resize(static_cast<size_t>(std::min<int64_t>(n, end - start)));
    return out;
  }
};

inline uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? static_cast<uint8_t>(c - 32) : c; }

// Decodes the slice whose header block starts at file offset `at`.  Returns false when its header
// says it cannot overlap the query (nothing was inflated).
bool decode_slice(const CramFile& f, const CompressionHeader& ch, size_t at, int32_t want, int64_t lo, int64_t hi,
                  const RefSource& refs, SliceOut* out) {
  Cursor c{f.buf + at, f.buf + f.size};
  Block hdr;
  read_block(c, &hdr);
  Cursor h{hdr.data, hdr.data + hdr.size};
  const int32_t s_ref = itf8(h), s_start = itf8(h), s_span = itf8(h), n_records = itf8(h);
  const int64_t record_counter = ltf8(h);   // index of the slice's first record in the file
  const int32_t n_blocks = itf8(h);
  itf8_array(h);
  const int32_t embedded_id = itf8(h);
  uint8_t ref_md5[16] = {0};
  const bool has_md5 = static_cast<size_t>(h.end - h.p) >= 16;
  if (has_md5) std::memcpy(ref_md5, h.p, 16);
  if (want >= 0 && s_ref >= 0 &&
      (s_ref != want || static_cast<int64_t>(s_start) - 1 >= hi || static_cast<int64_t>(s_start) - 1 + s_span <= lo)) {
    return false;
  }
  if (n_records < 0 || n_blocks < 0) bad("bad CRAM slice header");
  std::vector<std::unique_ptr<Block>> blocks;
  SliceData sd;
  for (int32_t i = 0; i < n_blocks; ++i) {
    auto b = std::make_unique<Block>();
    read_block(c, b.get());
    if (b->content_type == 5) {
      sd.core.data = b->data;
      sd.core.size = b->size;
      sd.core.pos = 0;
    } else if (b->content_type == 4) {
      sd.external[b->content_id] = b.get();
    }
    blocks.push_back(std::move(b));
  }
  auto int_dec = [&](char a, char b2) {
    Decoder d;
    auto it = ch.series.find(K2(a, b2));
    if (it != ch.series.end()) d.init(it->second, &sd, false);
    return d;
  };
  auto arr_dec = [&](char a, char b2) {
    Decoder d;
    auto it = ch.series.find(K2(a, b2));
    if (it != ch.series.end()) d.init(it->second, &sd, true);
    return d;
  };
  Decoder BF = int_dec('B', 'F'), CF = int_dec('C', 'F'), RI = int_dec('R', 'I'), RL = int_dec('R', 'L'),
          AP = int_dec('A', 'P'), RG = int_dec('R', 'G'), MF = int_dec('M', 'F'), NS = int_dec('N', 'S'),
          NP = int_dec('N', 'P'), TS = int_dec('T', 'S'), NF = int_dec('N', 'F'), TL = int_dec('T', 'L'),
          FN = int_dec('F', 'N'), FP = int_dec('F', 'P'), DL = int_dec('D', 'L'), RS = int_dec('R', 'S'),
          PD = int_dec('P', 'D'), HC = int_dec('H', 'C'), MQ = int_dec('M', 'Q');
  Decoder FC = int_dec('F', 'C'), BS = int_dec('B', 'S'), BA = int_dec('B', 'A'), QS = int_dec('Q', 'S');
  Decoder RN = arr_dec('R', 'N'), IN = arr_dec('I', 'N'), SC = arr_dec('S', 'C'), BB = arr_dec('B', 'B'),
          QQ = arr_dec('Q', 'Q');
  std::map<int32_t, std::unique_ptr<Decoder>> tag_dec;

  // reference of the slice: the embedded one, or the callback's (the slice's whole span, fetched on
  // first use and MD5-checked; a multi-reference slice asks per stretch)
  struct Cached {
    int64_t o = 0;
    std::string text;
  };
  std::map<int32_t, Cached> ref_cache;
  if (embedded_id >= 0) {
    auto it = sd.external.find(embedded_id);
    if (it != sd.external.end()) {
      Cached e;
      e.o = static_cast<int64_t>(s_start) - 1;
      e.text.assign(reinterpret_cast<const char*>(it->second->data), it->second->size);
      ref_cache[s_ref] = std::move(e);
    }
  }
  std::string scratch;
  // -> pointer to n reference bases from `start` (fewer when the contig ends: *got), valid until the next call
  auto ref_bases = [&](int32_t ref_id, int64_t start, int64_t n, int64_t* got) -> const char* {
    *got = 0;
    if (n <= 0) return "";
    auto it = ref_cache.find(ref_id);
    if (it != ref_cache.end()) {
      const Cached& e = it->second;
      if (start >= e.o && start + n <= e.o + static_cast<int64_t>(e.text.size())) {
        *got = n;
        return e.text.data() + (start - e.o);
      }
    }
    if (!refs.fetch) bad("Failed to parse BAM/CRAM file. " + f.path + " needs a reference (--ref) to be decoded");
    if (s_ref >= 0 && ref_id == s_ref) {
      Cached e;
      e.o = std::max<int64_t>(0, static_cast<int64_t>(s_start) - 1);
      e.text = refs.get(ref_id, e.o, e.o + s_span + 1);
      // the slice header carries the MD5 of the reference stretch it was encoded against (CRAM 3.0
      // section 8.5; all zero = not recorded): htslib refuses a mismatching FASTA, and decoding against
      // the wrong one would silently produce wrong read bases
      static const uint8_t zero[16] = {0};
      if (has_md5 && std::memcmp(ref_md5, zero, 16) != 0 && s_span > 0 &&
          e.text.size() >= static_cast<size_t>(s_span)) {
        std::string up(e.text.data(), static_cast<size_t>(s_span));
        for (char& ch2 : up) ch2 = static_cast<char>(upper(static_cast<uint8_t>(ch2)));
        uint8_t digest[16];
        md5(reinterpret_cast<const uint8_t*>(up.data()), up.size(), digest);
        if (std::memcmp(digest, ref_md5, 16) != 0) {
          bad("Failed to parse BAM/CRAM file. " + f.path + ": the reference MD5 of the slice at " +
              f.contig_names[static_cast<size_t>(ref_id)] + ":" + std::to_string(s_start) + " does not match --ref");
        }
      }
      if (start >= e.o && start + n <= e.o + static_cast<int64_t>(e.text.size())) {
        Cached& kept = ref_cache[ref_id] = std::move(e);
        *got = n;
        return kept.text.data() + (start - kept.o);
      }
    }
    scratch = refs.get(ref_id, start, start + n);
    *got = static_cast<int64_t>(scratch.size());
    return scratch.data();
  };

  // (a record count is an itf8 from the file: reserve what a slice can plausibly hold, grow for the rest)
  out->recs.reserve(static_cast<size_t>(std::min<int32_t>(std::max<int32_t>(n_records, 0), 1 << 20)));
  int32_t prev_pos = s_start;
  std::vector<uint8_t> tmp;
  for (int32_t rec_i = 0; rec_i < n_records; ++rec_i) {
    Rec r;
    r.flag = BF.read_int();
    r.cram_flags = CF.read_int();
    r.ref_id = s_ref == -2 ? RI.read_int() : s_ref;
    r.read_length = RL.read_int();
    if (r.read_length < 0) bad("negative CRAM read length");
    const int32_t ap = AP.read_int();
    if (ch.ap_delta) {
      prev_pos += ap;
      r.pos = prev_pos;
    } else {
      r.pos = ap;
    }
    RG.read_int();
    auto read_name = [&]() {
      r.name_off = static_cast<uint32_t>(out->names.size());
      r.name_len = static_cast<uint32_t>(RN.read_bytes(&out->names));
    };
    if (ch.read_names) read_name();
    if (r.cram_flags & 0x2) {          // detached: mate information stored verbatim
      r.mate_flags = MF.read_int();
      if (!ch.read_names) read_name();
      r.mate_ref_id = NS.read_int();
      r.mate_pos = NP.read_int();
      r.tlen = TS.read_int();
      r.tlen_known = true;
      if (r.mate_flags & 0x1) r.flag |= 0x20;
      if (r.mate_flags & 0x2) r.flag |= 0x8;
    } else if (r.cram_flags & 0x4) {   // mate is a later record of this slice
      r.next_fragment = NF.read_int();
    }
    const int32_t tl = TL.read_int();
    if (tl < 0 || static_cast<size_t>(tl) >= ch.tag_lists.size()) bad("CRAM tag list index out of range");
    for (const auto& key3 : ch.tag_lists[static_cast<size_t>(tl)]) {
      const int32_t tid = (key3[0] << 16) | (key3[1] << 8) | key3[2];
      auto it = tag_dec.find(tid);
      if (it == tag_dec.end()) {
        auto enc = ch.tags.find(tid);
        if (enc == ch.tags.end()) bad("CRAM tag without an encoding");
        auto d = std::make_unique<Decoder>();
        d->init(enc->second, &sd, true);
        it = tag_dec.emplace(tid, std::move(d)).first;
      }
      tmp.clear();
      it->second->read_bytes(&tmp);
      if (key3[0] == 'H' && key3[1] == 'P') {
        // the integer HP tag (c C s S i I), as bam_reader.cpp reads it
        const uint8_t* v = tmp.data();
        const size_t n = tmp.size();
        switch (key3[2]) {
          case 'c': if (n >= 1) { r.hp = static_cast<int8_t>(v[0]); r.has_hp = true; } break;
          case 'C': if (n >= 1) { r.hp = v[0]; r.has_hp = true; } break;
          case 's': if (n >= 2) { r.hp = static_cast<int16_t>(v[0] | (v[1] << 8)); r.has_hp = true; } break;
          case 'S': if (n >= 2) { r.hp = v[0] | (v[1] << 8); r.has_hp = true; } break;
          case 'i': case 'I':
            if (n >= 4) { r.hp = static_cast<int32_t>(v[0] | (v[1] << 8) | (v[2] << 16) | (static_cast<uint32_t>(v[3]) << 24)); r.has_hp = true; }
            break;
          default: break;
        }
      } else if (key3[0] == 'O' && key3[1] == 'Q' && key3[2] == 'Z') {
        size_t n = tmp.size();
        while (n > 0 && tmp[n - 1] == 0) --n;
        r.oq_off = static_cast<uint32_t>(out->oq.size());
        r.oq_len = static_cast<uint32_t>(n);
        r.has_oq = true;
        out->oq.insert(out->oq.end(), tmp.begin(), tmp.begin() + static_cast<long>(n));
      } else if (out->keep_aux) {
        static const char kKeep[5][3] = {"MM", "ML", "MN", "tp", "t0"};
        for (int k = 0; k < 5; ++k) {
          if (key3[0] == static_cast<uint8_t>(kKeep[k][0]) && key3[1] == static_cast<uint8_t>(kKeep[k][1]) &&
              r.aux_type[k] == 0) {
            r.aux_type[k] = static_cast<char>(key3[2]);
            r.aux_off[k] = static_cast<uint32_t>(out->aux.size());
            r.aux_len[k] = static_cast<uint32_t>(tmp.size());
            out->aux.insert(out->aux.end(), tmp.begin(), tmp.end());
          }
        }
      }
    }
    const size_t L = static_cast<size_t>(r.read_length);
    r.seq_off = static_cast<uint32_t>(out->bases.size());
    out->bases.resize(out->bases.size() + L, 0);
    out->quals.resize(out->quals.size() + L, 0xff);
    uint8_t* const seq = out->bases.data() + r.seq_off;
    uint8_t* const qual = out->quals.data() + r.seq_off;
    r.cig_off = static_cast<uint32_t>(out->cigar.size());
    if (!(r.flag & 0x4)) {
      auto push = [&](uint32_t op, int64_t n) {
        if (n <= 0) return;
        if (out->cigar.size() > r.cig_off && (out->cigar.back() & 0xF) == op) {
          out->cigar.back() += static_cast<uint32_t>(n) << 4;
        } else {
          out->cigar.push_back((static_cast<uint32_t>(n) << 4) | op);
        }
      };
      auto span_ok = [&](int64_t at0, int64_t n) {   // [at0, at0 + n) inside the read (0-based)
        if (at0 < 0 || n < 0 || at0 + n > static_cast<int64_t>(L)) bad("CRAM read feature runs off its read");
      };
      const int32_t n_feat = FN.read_int();
      int64_t rp = 1;                                      // next read base (1-based) not yet filled
      int64_t refp = static_cast<int64_t>(r.pos) - 1;      // 0-based reference position of that base
      int64_t fpos = 0;
      auto fill_matches = [&](int64_t upto) {              // read bases rp .. upto-1 equal the reference
        const int64_t n = upto - rp;
        if (n > 0) {
          span_ok(rp - 1, n);
          int64_t got = 0;
          const char* text = ref_bases(r.ref_id, refp, n, &got);
          for (int64_t k = 0; k < n; ++k) seq[rp - 1 + k] = k < got ? upper(static_cast<uint8_t>(text[k])) : 'N';
          push(0, n);
          rp += n;
          refp += n;
        }
      };
      for (int32_t fi = 0; fi < n_feat; ++fi) {
        const int code = FC.read_byte();
        fpos += FP.read_int();
        if (code == 'Q') {
          span_ok(fpos - 1, 1);
          qual[fpos - 1] = static_cast<uint8_t>(QS.read_byte());
          continue;
        }
        if (code == 'q') {
          tmp.clear();
          QQ.read_bytes(&tmp);
          span_ok(fpos - 1, static_cast<int64_t>(tmp.size()));
          std::memcpy(qual + fpos - 1, tmp.data(), tmp.size());
          continue;
        }
        fill_matches(fpos);
        switch (code) {
          case 'B':
            span_ok(fpos - 1, 1);
            seq[fpos - 1] = static_cast<uint8_t>(BA.read_byte());
            qual[fpos - 1] = static_cast<uint8_t>(QS.read_byte());
            push(0, 1);
            rp += 1;
            refp += 1;
            break;
          case 'X': {
            const int32_t sub = BS.read_byte();
            int64_t got = 0;
            const char* text = ref_bases(r.ref_id, refp, 1, &got);
            const uint8_t rb = got ? upper(static_cast<uint8_t>(text[0])) : 'N';
            const int ri = rb == 'A' ? 0 : rb == 'C' ? 1 : rb == 'G' ? 2 : rb == 'T' ? 3 : 4;
            span_ok(fpos - 1, 1);
            seq[fpos - 1] = static_cast<uint8_t>(ch.subst_lookup[ri][sub & 3]);
            push(0, 1);
            rp += 1;
            refp += 1;
            break;
          }
          case 'I': {
            tmp.clear();
            IN.read_bytes(&tmp);
            span_ok(fpos - 1, static_cast<int64_t>(tmp.size()));
            std::memcpy(seq + fpos - 1, tmp.data(), tmp.size());
            push(1, static_cast<int64_t>(tmp.size()));
            rp += static_cast<int64_t>(tmp.size());
            break;
          }
          case 'i':
            span_ok(fpos - 1, 1);
            seq[fpos - 1] = static_cast<uint8_t>(BA.read_byte());
            push(1, 1);
            rp += 1;
            break;
          case 'S': {
            tmp.clear();
            SC.read_bytes(&tmp);
            span_ok(fpos - 1, static_cast<int64_t>(tmp.size()));
            std::memcpy(seq + fpos - 1, tmp.data(), tmp.size());
            push(4, static_cast<int64_t>(tmp.size()));
            rp += static_cast<int64_t>(tmp.size());
            break;
          }
          case 'D': {
            const int32_t n = DL.read_int();
            push(2, n);
            refp += n;
            break;
          }
          case 'N': {
            const int32_t n = RS.read_int();
            push(3, n);
            refp += n;
            break;
          }
          case 'H':
            push(5, HC.read_int());
            break;
          case 'P':
            push(6, PD.read_int());
            break;
          case 'b': {
            tmp.clear();
            BB.read_bytes(&tmp);
            span_ok(fpos - 1, static_cast<int64_t>(tmp.size()));
            std::memcpy(seq + fpos - 1, tmp.data(), tmp.size());
            push(0, static_cast<int64_t>(tmp.size()));
            rp += static_cast<int64_t>(tmp.size());
            refp += static_cast<int64_t>(tmp.size());
            break;
          }
          default:
            bad("unknown CRAM read feature");
        }
      }
      fill_matches(static_cast<int64_t>(L) + 1);
      r.mapq = MQ.read_int();
      if (r.cram_flags & 0x1) {
        for (size_t q = 0; q < L; ++q) qual[q] = static_cast<uint8_t>(QS.read_byte());
      }
      r.ref_len = static_cast<int32_t>(refp - (static_cast<int64_t>(r.pos) - 1));
    } else {
      for (size_t q = 0; q < L; ++q) seq[q] = static_cast<uint8_t>(BA.read_byte());
      if (r.cram_flags & 0x1) {
        for (size_t q = 0; q < L; ++q) qual[q] = static_cast<uint8_t>(QS.read_byte());
      }
    }
    r.cig_n = static_cast<uint32_t>(out->cigar.size()) - r.cig_off;
    out->recs.push_back(r);
  }

  // mates inside the slice: names, mate fields and htslib's template length
  std::vector<Rec>& recs = out->recs;
  const int32_t n = static_cast<int32_t>(recs.size());
  for (int32_t i = 0; i < n; ++i) {
    if (recs[i].next_fragment >= 0) {
      const int64_t j = static_cast<int64_t>(i) + recs[i].next_fragment + 1;
      if (j < n) recs[i].mate_line = static_cast<int32_t>(j);
    }
  }
  std::vector<int32_t> chain;
  for (int32_t i = 0; i < n; ++i) {
    Rec& r = recs[i];
    if (r.mate_line < 0 || r.tlen_known) continue;
    chain.assign(1, i);
    int32_t j = r.mate_line;
    while (j >= 0 && std::find(chain.begin(), chain.end(), j) == chain.end()) {
      chain.push_back(j);
      j = recs[j].mate_line;
    }
    bool same_ref = true;
    int64_t aleft = INT64_MAX, aright = INT64_MIN;
    for (int32_t m : chain) {
      same_ref = same_ref && recs[m].ref_id == r.ref_id;
      aleft = std::min<int64_t>(aleft, recs[m].pos);
      aright = std::max<int64_t>(aright, static_cast<int64_t>(recs[m].pos) + std::max(recs[m].ref_len, 1) - 1);
    }
    int left_cnt = 0;
    for (int32_t m : chain) left_cnt += recs[m].pos == aleft;
    const int64_t tlen = same_ref ? aright - aleft + 1 : 0;
    for (int32_t m : chain) {
      Rec& rc = recs[m];
      if (!same_ref) {
        rc.tlen = 0;
      } else if (rc.pos == aleft && (left_cnt == 1 || (rc.flag & 0x40))) {
        rc.tlen = static_cast<int32_t>(tlen);
      } else {
        rc.tlen = static_cast<int32_t>(-tlen);
      }
      rc.tlen_known = true;
    }
    // one generated name for the whole template when names were not stored (htslib gives both mates
    // of a pair the same generated name)
    if (recs[chain[0]].name_len == 0) {
      const std::string gen = std::to_string(record_counter + chain[0]);
      recs[chain[0]].name_off = static_cast<uint32_t>(out->names.size());
      recs[chain[0]].name_len = static_cast<uint32_t>(gen.size());
      out->names.insert(out->names.end(), gen.begin(), gen.end());
    }
    for (size_t a = 0; a < chain.size(); ++a) {   // mate of chain[a] is the next in the chain, the last one's is the first
      const Rec& m = recs[chain[(a + 1) % chain.size()]];
      Rec& rc = recs[chain[a]];
      rc.mate_ref_id = m.ref_id;
      rc.mate_pos = m.pos;
      if (m.flag & 0x10) rc.flag |= 0x20;
      if (m.flag & 0x4) rc.flag |= 0x8;
      if (rc.name_len == 0) {
        rc.name_off = recs[chain[0]].name_off;
        rc.name_len = recs[chain[0]].name_len;
      }
    }
  }
  for (int32_t i = 0; i < n; ++i) {
    Rec& r = recs[i];
    if (!r.tlen_known) r.tlen = 0;
    if (r.name_len == 0) {
      const std::string gen = std::to_string(record_counter + i);   // unique across slices
      r.name_off = static_cast<uint32_t>(out->names.size());
      r.name_len = static_cast<uint32_t>(gen.size());
      out->names.insert(out->names.end(), gen.begin(), gen.end());
    }
  }
  return true;
}

// One slice's records -> rows of the table: the region test, nucleus' read requirements and the
// field semantics of bam_reader.cpp's decode_record (ConvertToPb, sam_reader.cc:734-840).
void append_rows(const SliceOut& s, int32_t want, int64_t start, int64_t end, const dv_read_requirements& rq,
                 dv_read_table* t) {
  for (const Rec& r : s.recs) {
    const int32_t flag = r.flag;
    if ((flag & 0x4) || r.ref_id < 0) continue;
    if (want >= 0 && r.ref_id != want) continue;
    const int64_t p0 = static_cast<int64_t>(r.pos) - 1;
    if (!(end > p0 && start < p0 + std::max<int64_t>(r.ref_len, 1))) continue;
    if ((!rq.keep_duplicates && (flag & 0x400)) || (!rq.keep_failed_vendor_quality_checks && (flag & 0x200)) ||
        (!rq.keep_secondary_alignments && (flag & 0x100)) || (!rq.keep_supplementary_alignments && (flag & 0x800))) {
      continue;
    }
    const bool paired = flag & 0x1;
    const bool has_mate_pos = paired && !(flag & 0x8) && r.mate_ref_id >= 0;
    const bool properly_placed = !paired || (flag & 0x2) || !has_mate_pos || r.mate_ref_id == r.ref_id;
    if (!rq.keep_improperly_placed && !properly_placed) continue;
    if (r.mapq < rq.min_mapping_quality) continue;
    const size_t L = static_cast<size_t>(r.read_length);
    const uint8_t* qsrc = s.quals.data() + r.seq_off;
    int qsub = 0;
    if (rq.use_original_base_quality_scores) {
      if (!r.has_oq) throw CramError(DV_ERR_BAD_INPUT, "use_original_base_quality_scores: a read has no OQ tag");
      if (r.oq_len != L) throw CramError(DV_ERR_BAD_INPUT, "OQ tag and sequence are of different length");
      qsrc = s.oq.data() + r.oq_off;
      for (size_t i = 0; i < L; ++i) {
        if (qsrc[i] < 33) throw CramError(DV_ERR_BAD_INPUT, "OQ tag holds a character below '!'");
      }
      qsub = 33;
    } else if (L && qsrc[0] == 0xff) {
      throw CramError(DV_ERR_BAD_INPUT, "Could not read base quality scores");   // sam_reader.cc:752
    }
    t->pos.push_back(static_cast<int32_t>(p0));
    t->end.push_back(p0 + r.ref_len);
    t->mapq.push_back(static_cast<uint8_t>(r.mapq));
    t->flags.push_back(static_cast<uint8_t>(((flag & 0x10) ? DV_READ_REVERSE : 0) | ((flag & 0x800) ? DV_READ_SUPPLEMENTARY : 0)));
    t->read_number.push_back((flag & 0x40) || !paired ? 0 : 1);
    t->frag_len.push_back(r.tlen);
    t->hp.push_back(r.has_hp ? r.hp : DV_HP_NONE);
    for (uint32_t k = 0; k < r.cig_n; ++k) {
      const uint32_t v = s.cigar[r.cig_off + k];
      t->cigar.push_back(((v >> 4) << 4) | ((v & 0xF) + 1));   // kHtslibCigarToProto
    }
    t->cigar_off.push_back(static_cast<uint32_t>(t->cigar.size()));
    t->bases.insert(t->bases.end(), s.bases.begin() + r.seq_off, s.bases.begin() + r.seq_off + static_cast<long>(L));
    const size_t q0 = t->quals.size();
    t->quals.insert(t->quals.end(), qsrc, qsrc + L);
    if (qsub) {
      for (size_t i = q0; i < t->quals.size(); ++i) t->quals[i] = static_cast<uint8_t>(t->quals[i] - qsub);
    }
    t->seq_off.push_back(static_cast<uint32_t>(t->bases.size()));
    if (t->with_mods || t->with_flow) {   // per-base planes from MM / ML / MN and tp / t0 (aux_planes.h)
      dv::AuxFields aux;
      dv::AuxField* slots[5] = {&aux.mm, &aux.ml, &aux.mn, &aux.tp, &aux.t0};
      bool ok = true;
      for (int k = 0; k < 5; ++k) {
        if (!r.aux_type[k]) continue;
        slots[k]->type = r.aux_type[k];
        slots[k]->data = s.aux.data() + r.aux_off[k];
        slots[k]->size = r.aux_len[k];
        // a B value must hold its header and the elements it announces
        if (r.aux_type[k] == 'B') {
          uint32_t n = 0;
          const int sub = r.aux_len[k] >= 5 ? dv::aux_scalar_size(slots[k]->data[0]) : 0;
          if (sub) std::memcpy(&n, slots[k]->data + 1, 4);
          ok = ok && sub && static_cast<uint64_t>(r.aux_len[k]) >= 5 + static_cast<uint64_t>(n) * sub;
        } else if (dv::aux_scalar_size(static_cast<uint8_t>(r.aux_type[k]))) {
          ok = ok && r.aux_len[k] >= static_cast<uint32_t>(dv::aux_scalar_size(static_cast<uint8_t>(r.aux_type[k])));
        }
      }
      if (!ok) throw CramError(DV_ERR_BAD_INPUT, "malformed aux tag value in a CRAM record");
      if (!dv::append_aux_planes(t, t->bases.data() + t->bases.size() - L, L, (flag & 0x10) != 0, aux)) {
        throw CramError(DV_ERR_BAD_INPUT, "MM tag: a position that is not a number");
      }
    }
    t->name_off.push_back(static_cast<uint32_t>(t->names.size()));
    t->names.insert(t->names.end(), s.names.begin() + r.name_off, s.names.begin() + r.name_off + r.name_len);
    t->names.push_back('\0');
  }
}

}  // namespace

extern "C" {

int dv_cram_read_region(const char* path, const char* contig, int64_t start, int64_t end,
                        const dv_read_requirements* req, dv_ref_fetch_fn fetch, void* fetch_ctx, int n_threads,
                        dv_read_table** out) {
  if (!path || !out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_cram_read_region: null");
  try {
    CramFile f(path);
    int32_t want = -1;
    if (contig) {
      for (size_t i = 0; i < f.contig_names.size(); ++i) {
        if (f.contig_names[i] == contig) {
          want = static_cast<int32_t>(i);
          break;
        }
      }
      if (want < 0) return dv::fail(DV_ERR_BAD_INPUT, std::string("contig not in the CRAM header: ") + contig);
    }
    dv_read_requirements rq{};
    if (req) rq = *req;
    // the slices to decode, in file order
    struct Job {
      std::shared_ptr<CompressionHeader> ch;
      size_t at;
      SliceOut out;
      bool used = false;
      std::string error;
      int status = DV_OK;
    };
    std::vector<Job> jobs;
    for (const ContainerHeader& h : f.containers_for(want, start, end)) {
      Cursor c{f.buf + h.blocks, f.buf + f.size};
      Block first;
      read_block(c, &first);
      if (first.content_type != 1) continue;
      auto ch = std::make_shared<CompressionHeader>(first);
      for (int32_t lm : h.landmarks) {
        if (lm < 0) bad("negative CRAM landmark");
        Job j;
        j.ch = ch;
        j.at = h.blocks + static_cast<size_t>(lm);
        j.out.keep_aux = rq.parse_base_modifications != 0 || rq.parse_flow_tags != 0;
        jobs.push_back(std::move(j));
      }
    }
    std::mutex fetch_mu;
    const RefSource refs{&f, fetch, fetch_ctx, &fetch_mu};
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= jobs.size()) return;
        Job& j = jobs[i];
        try {
          j.used = decode_slice(f, *j.ch, j.at, want, start, end, refs, &j.out);
        } catch (const CramError& e) {
          j.status = e.status;
          j.error = e.what();
        } catch (const std::exception& e) {
          j.status = DV_ERR_BAD_INPUT;
          j.error = e.what();
        }
      }
    };
    const int nt = std::max(1, std::min<int>(n_threads, static_cast<int>(jobs.size())));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    std::unique_ptr<dv_read_table> t(new dv_read_table());
    t->with_mods = rq.parse_base_modifications != 0;
    t->with_flow = rq.parse_flow_tags != 0;
    t->seq_off.push_back(0);
    t->cigar_off.push_back(0);
    for (Job& j : jobs) {
      if (j.status != DV_OK) return dv::fail(j.status, j.error);
      if (j.used) append_rows(j.out, want, start, end, rq, t.get());
      j.out = SliceOut();
    }
    if (t->bases.size() >= (1ull << 32) || t->cigar.size() >= (1ull << 32)) {
      return dv::fail(DV_ERR_UNSUPPORTED, "region too large: offsets are 32 bit");
    }
    dv::rank_read_names(t.get());
    *out = t.release();
    return DV_OK;
  } catch (const CramError& e) {
    return dv::fail(e.status, e.what());
  } catch (const std::exception& e) {
    return dv::fail(DV_ERR_BAD_INPUT, e.what());
  }
}

int dv_cram_header(const char* path, char* text, uint64_t capacity, uint64_t* needed) {
  if (!path || !needed) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_cram_header: null");
  try {
    CramFile f(path);
    *needed = f.header_text.size();
    if (text && capacity) {
      const size_t n = std::min<size_t>(f.header_text.size(), static_cast<size_t>(capacity));
      std::memcpy(text, f.header_text.data(), n);
    }
    return DV_OK;
  } catch (const CramError& e) {
    return dv::fail(e.status, e.what());
  } catch (const std::exception& e) {
    return dv::fail(DV_ERR_BAD_INPUT, e.what());
  }
}

}  // extern "C"
