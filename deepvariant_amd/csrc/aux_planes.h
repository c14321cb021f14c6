// aux_planes.h -- the per-base planes the alignment-file decoders derive from a record's aux tags (host only):
//   * base modifications (5mC, 6mA) from MM / ML / MN: nucleus' ParseBaseModifications,
//     third_party/nucleus/io/sam_reader.cc:521-719 (run on every read once its aux fields are parsed, :855-862;
//     make_examples asks for the three tags when a base-modification channel is listed,
//     deepvariant/make_examples_core.py:355-371) -- restated with its quirks, which the reference's own tests and a
//     second restatement in Python (genomics_io.parse_base_modifications) pin: unsupported specifications still
//     consume their ML values; a specification whose positions run past the read leaves no entry AND does not
//     advance the ML offset; strands are merged with a SIGNED char maximum; MN != sequence length or ML too short
//     drop everything;
//   * the Ultima flow-space tags tp (B array, one value per base) and t0 (Z, phred + 33 per base), as
//     channels/homopolymer_indel_quality_channel.cc:68-84 and
//     channels/inter_homopolymer_insertion_quality_channel.cc:76-112 read them (shorter tags leave zeros, longer ones
//     are cut).
// Values arrive BAM-encoded (what follows the type byte), from a BAM record's aux block or from a CRAM tag series.
#ifndef DV_AUX_PLANES_H_
#define DV_AUX_PLANES_H_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "read_table.h"

namespace dv {

struct AuxField {
  char type = 0;                 // 0 = absent; A c C s S i I f Z H B
  const uint8_t* data = nullptr; // the value: after the type byte (B: subtype, int32 count, elements)
  size_t size = 0;
};

struct AuxFields {
  AuxField mm, ml, mn, tp, t0;
};

inline int aux_scalar_size(uint8_t ty) {
  switch (ty) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    default: return 0;
  }
}

inline AuxField* aux_slot(AuxFields* f, uint8_t a, uint8_t b) {
  if (a == 'M' && b == 'M') return &f->mm;
  if (a == 'M' && b == 'L') return &f->ml;
  if (a == 'M' && b == 'N') return &f->mn;
  if (a == 't' && b == 'p') return &f->tp;
  if (a == 't' && b == '0') return &f->t0;
  return nullptr;
}

// Walks a BAM aux block; a malformed block ends the walk (what was found so far stays, as in ParseAuxFields).
inline void scan_bam_aux(const uint8_t* p, const uint8_t* end, AuxFields* out) {
  while (end - p >= 4) {
    AuxField* slot = aux_slot(out, p[0], p[1]);
    const uint8_t ty = p[2];
    p += 3;
    const uint8_t* v = p;
    const int sz = aux_scalar_size(ty);
    if (sz) {
      if (end - p < sz) return;
      p += sz;
    } else if (ty == 'Z' || ty == 'H') {
      while (p < end && *p) ++p;
      if (p >= end) return;
      ++p;
    } else if (ty == 'B') {
      if (end - p < 5) return;
      const int sub = aux_scalar_size(p[0]);
      if (!sub || p[0] == 'A') return;
      uint32_t n;
      std::memcpy(&n, p + 1, 4);
      if (static_cast<uint64_t>(end - (p + 5)) < static_cast<uint64_t>(n) * sub) return;
      p += 5 + static_cast<size_t>(n) * sub;
    } else {
      return;
    }
    if (slot && slot->type == 0) {   // (the first occurrence, as a proto map insert would keep... SetInfoField replaces;
      slot->type = static_cast<char>(ty);  //  duplicate tags are invalid SAM and htslib-made files do not have them)
      slot->data = v;
      slot->size = static_cast<size_t>(p - v);
    }
  }
}

// The int_value()s nucleus would hold for this field: a scalar integer -> one value; a B array of integers -> its
// elements (float arrays hold number_values: int_value() is 0); anything else -> none.  `present` mirrors
// info().contains(): a B:C array without elements is never stored (sam_reader.cc:394-397).
inline void aux_int_values(const AuxField& f, std::vector<int32_t>* out, bool* present) {
  out->clear();
  *present = f.type != 0;
  auto one = [](uint8_t ty, const uint8_t* p) -> int32_t {
    switch (ty) {
      case 'c': return static_cast<int8_t>(p[0]);
      case 'C': return p[0];
      case 's': { int16_t v; std::memcpy(&v, p, 2); return v; }
      case 'S': { uint16_t v; std::memcpy(&v, p, 2); return v; }
      case 'i': { int32_t v; std::memcpy(&v, p, 4); return v; }
      case 'I': { uint32_t v; std::memcpy(&v, p, 4); return static_cast<int32_t>(v); }
      default: return 0;
    }
  };
  if (f.type == 'B') {
    const uint8_t sub = f.data[0];
    uint32_t n;
    std::memcpy(&n, f.data + 1, 4);
    if (sub == 'C' && n == 0) {
      *present = false;
      return;
    }
    const int sz = aux_scalar_size(sub);
    out->reserve(n);
    for (uint32_t i = 0; i < n; ++i) out->push_back(sub == 'f' ? 0 : one(sub, f.data + 5 + static_cast<size_t>(i) * sz));
  } else if (f.type == 'c' || f.type == 'C' || f.type == 's' || f.type == 'S' || f.type == 'i' || f.type == 'I') {
    out->push_back(one(static_cast<uint8_t>(f.type), f.data));
  } else if (f.type == 'f' || f.type == 'A' || f.type == 'Z') {
    out->push_back(0);            // one value of another kind
  } else if (f.type == 'H') {
    *present = false;             // hex strings are skipped (sam_reader.cc:357-360)
  }
}

inline std::string aux_string_value(const AuxField& f) {
  if (f.type != 'Z') return std::string();
  size_t n = f.size;
  while (n > 0 && f.data[n - 1] == 0) --n;
  return std::string(reinterpret_cast<const char*>(f.data), n);
}

inline char complement_base(char c) {
  switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;   // (IUPAC codes do not matter here: only C, A and T are ever looked for)
  }
}

// ([ACGTUN])([-+])([a-z]+|[0-9]+)([.?]?), anchored at both ends (sam_reader.cc:518-519).
inline bool match_modification_spec(const std::string& s, char* base, char* strand, std::string* code) {
  if (s.size() < 3 || !std::strchr("ACGTUN", s[0]) || (s[1] != '+' && s[1] != '-')) return false;
  size_t i = 2;
  if (s[i] >= 'a' && s[i] <= 'z') {
    while (i < s.size() && s[i] >= 'a' && s[i] <= 'z') ++i;
  } else if (s[i] >= '0' && s[i] <= '9') {
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') ++i;
  } else {
    return false;
  }
  *code = s.substr(2, i - 2);
  if (i < s.size() && (s[i] == '.' || s[i] == '?')) ++i;
  if (i != s.size()) return false;
  *base = s[0];
  *strand = s[1];
  return true;
}

// ParseBaseModifications.  -> {"5mC": plane, "6mA": plane} (each of the read's length), possibly empty.  `bad` is set
// for what makes the reference abort (a position that is not a number: std::stoi throws).
inline std::map<std::string, std::string> parse_base_modifications(const uint8_t* seq_in, size_t L, bool reverse,
                                                                   const AuxFields& f, bool* bad) {
  std::map<std::string, std::string> result;
  *bad = false;
  std::vector<int32_t> ml, mn;
  bool has_ml = false, has_mn = false;
  aux_int_values(f.ml, &ml, &has_ml);
  aux_int_values(f.mn, &mn, &has_mn);
  if (f.mm.type == 0 || f.mm.type == 'H' || !has_ml) return result;    // (an MM of any kind holds one value)
  const int64_t mn_size = has_mn && !mn.empty() ? mn[0] : static_cast<int64_t>(L);
  if (mn_size != static_cast<int64_t>(L)) return result;
  std::string seq(reinterpret_cast<const char*>(seq_in), L);
  if (reverse) {
    std::reverse(seq.begin(), seq.end());
    for (char& c : seq) c = complement_base(c);
  }
  std::string mm = aux_string_value(f.mm);
  if (!mm.empty() && mm.back() == ';') mm.pop_back();
  std::vector<std::string> mods;
  {
    size_t at = 0;
    for (;;) {
      const size_t semi = mm.find(';', at);
      mods.push_back(mm.substr(at, semi == std::string::npos ? std::string::npos : semi - at));
      if (semi == std::string::npos) break;
      at = semi + 1;
    }
  }
  auto to_int = [&](const std::string& s, int* v) -> bool {   // std::stoi: leading blanks, sign, digits; rest ignored
    const char* p = s.c_str();
    char* e = nullptr;
    const long x = std::strtol(p, &e, 10);
    if (e == p) return false;
    *v = static_cast<int>(x);
    return true;
  };
  int ml_offset = 0;
  for (const std::string& mod : mods) {
    std::vector<std::string> parts;
    {
      size_t at = 0;
      for (;;) {
        const size_t comma = mod.find(',', at);
        parts.push_back(mod.substr(at, comma == std::string::npos ? std::string::npos : comma - at));
        if (comma == std::string::npos) break;
        at = comma + 1;
      }
    }
    if (parts.size() <= 1) continue;
    char base = 0, strand = 0;
    std::string code;
    const char* spec = nullptr;
    if (match_modification_spec(parts[0], &base, &strand, &code)) {
      if (base == 'C' && strand == '+' && code == "m") spec = "5mC";
      else if (base == 'A' && strand == '+' && code == "a") spec = "6mA";
      else if (base == 'T' && strand == '-' && code == "a") spec = "6mA";
    }
    if (!spec) {
      ml_offset += static_cast<int>(parts.size()) - 1;
      continue;
    }
    std::vector<uint8_t> plane(L, 0);
    parts.erase(parts.begin());
    int idx = 0, base_count = 0, delta = 0;
    if (!to_int(parts[0], &delta)) { *bad = true; return {}; }
    for (size_t pos = 0; pos <= seq.size(); ++pos) {
      const char here = pos < seq.size() ? seq[pos] : '\0';
      if (here != base) continue;
      if (base_count != delta) {
        ++base_count;
        continue;
      }
      if (static_cast<size_t>(ml_offset + idx) >= ml.size()) return {};
      plane[pos] = static_cast<uint8_t>(ml[static_cast<size_t>(idx + ml_offset)]);
      base_count = 0;
      ++idx;
      if (idx >= static_cast<int>(parts.size())) {
        ml_offset += idx;
        std::string str(reinterpret_cast<const char*>(plane.data()), plane.size());
        if (reverse) std::reverse(str.begin(), str.end());
        auto it = result.find(spec);
        if (it != result.end()) {
          for (size_t i = 0; i < str.size(); ++i) {
            it->second[i] = static_cast<char>(std::max<signed char>(static_cast<signed char>(it->second[i]),
                                                                    static_cast<signed char>(str[i])));
          }
        } else {
          result.emplace(spec, std::move(str));
        }
        break;
      }
      if (!to_int(parts[static_cast<size_t>(idx)], &delta)) { *bad = true; return {}; }
    }
  }
  return result;
}

// Appends one read's planes to the table (the table's flags entry of the read must already exist).
inline bool append_aux_planes(dv_read_table* t, const uint8_t* seq, size_t L, bool reverse, const AuxFields& f) {
  if (t->with_mods) {
    bool bad = false;
    const auto mods = parse_base_modifications(seq, L, reverse, f, &bad);
    if (bad) return false;
    const size_t at = t->mod_5mc.size();
    t->mod_5mc.resize(at + L, 0);
    t->mod_6ma.resize(at + L, 0);
    auto it = mods.find("5mC");
    if (it != mods.end()) {
      std::memcpy(t->mod_5mc.data() + at, it->second.data(), L);
      t->flags.back() |= 4;   // DV_READ_HAS_5MC
    }
    it = mods.find("6mA");
    if (it != mods.end()) {
      std::memcpy(t->mod_6ma.data() + at, it->second.data(), L);
      t->flags.back() |= 8;   // DV_READ_HAS_6MA
    }
  }
  if (t->with_flow) {
    const size_t at = t->tp.size();
    t->tp.resize(at + L, 0);
    t->t0.resize(at + L, 0);
    std::vector<int32_t> tp;
    bool has_tp = false;
    aux_int_values(f.tp, &tp, &has_tp);
    if (has_tp) {
      for (size_t i = 0; i < tp.size() && i < L; ++i) t->tp[at + i] = static_cast<int8_t>(tp[i]);
    }
    const std::string t0 = aux_string_value(f.t0);
    for (size_t i = 0; i < t0.size() && i < L; ++i) t->t0[at + i] = static_cast<uint8_t>(t0[i] - 33);
    t->flow_present.push_back(static_cast<uint8_t>((f.tp.type != 0 ? 1 : 0) | (f.t0.type == 'Z' ? 2 : 0)));
  }
  return true;
}

}  // namespace dv

#endif  // DV_AUX_PLANES_H_
