// See direct_phasing.h.
#include "direct_phasing.h"

#include <algorithm>
#include <sstream>

namespace dv {
namespace {

constexpr int kMinRefAlleleDepth = 3;     // direct_phasing.cc:64
constexpr int kNumOfPhases = 2;

// CandidateFilter (direct_phasing.cc:728-752): a site enters the graph only if it has two
// called alleles (or one plus >= 3 reference reads), every called allele is as long as the
// reference span (no indels) and the site does not sit inside an earlier indel's span.
bool candidate_filter(const PhasingCandidate& c, int64_t* indel_end) {
  int called = 0, ref_support = 0;
  for (const PhasingAllele& a : c.alleles) {
    if (a.is_ref) {
      ref_support = static_cast<int>(a.support.size());
    } else {
      ++called;
    }
  }
  if (called <= 1 && ref_support < kMinRefAlleleDepth) return false;
  for (const PhasingAllele& a : c.alleles) {
    if (a.is_ref) continue;
    if (c.end <= *indel_end || static_cast<int64_t>(a.bases.size()) != c.end - c.start) {
      if (*indel_end < c.end) *indel_end = c.end;
      return false;
    }
  }
  return true;
}

}  // namespace

int DirectPhasing::add_vertex(int64_t position, const std::string& bases,
                              const std::vector<PhasingReadSupport>& support, int candidate, int allele) {
  Vertex v;
  v.position = position;
  v.bases = bases;
  v.candidate = candidate;
  v.allele = allele;
  for (const PhasingReadSupport& rs : support) {       // ReadSupportFromProto, :506-521
    if (rs.read >= 0 && rs.read < static_cast<int>(read_to_alleles_.size()) && !rs.is_low_quality) {
      v.reads.push_back(rs.read);
    }
  }
  const int id = static_cast<int>(vertices_.size());
  // UpdateReadToAllelesMap, :558-575
  for (int r : v.reads) {
    v.is_first_allele.push_back(read_to_alleles_[r].empty());
    read_to_alleles_[r].push_back(id);
  }
  vertices_by_position_[position].push_back(id);
  vertices_.push_back(std::move(v));
  in_edges_.emplace_back();
  return id;
}

void DirectPhasing::add_edge(int from, int to, float weight) {
  auto it = edges_.find({from, to});
  if (it == edges_.end()) {
    edges_[{from, to}] = weight;
    in_edges_[to].push_back(from);
  } else {
    it->second += weight;
  }
}

void DirectPhasing::build(const std::vector<PhasingCandidate>& candidates) {   // Build, :765-812
  int64_t indel_end = 0;
  for (size_t i = 0; i < candidates.size(); ++i) {
    const PhasingCandidate& c = candidates[i];
    allele_phases_[i].assign(c.alleles.size(), PhasedAllele());
    if (!candidate_filter(c, &indel_end)) continue;
    // AddCandidate, :685-726: the reference vertex first (if >= 3 reads), then alleles by bases
    std::vector<int> order;
    for (size_t a = 0; a < c.alleles.size(); ++a) {
      if (c.alleles[a].is_ref) {
        if (static_cast<int>(c.alleles[a].support.size()) >= kMinRefAlleleDepth) {
          add_vertex(c.start, "REF", c.alleles[a].support, static_cast<int>(i), static_cast<int>(a));
          allele_phases_[i][a].in_graph = true;
        }
      } else {
        order.push_back(static_cast<int>(a));
      }
    }
    std::sort(order.begin(), order.end(), [&](int x, int y) { return c.alleles[x].bases < c.alleles[y].bases; });
    for (int a : order) {
      add_vertex(c.start, c.alleles[a].bases, c.alleles[a].support, static_cast<int>(i), a);
      allele_phases_[i][a].in_graph = true;
    }
    positions_.push_back(c.start);
  }
  // a read links two alleles when they sit at consecutive graph positions
  for (size_t r = 0; r < read_to_alleles_.size(); ++r) {
    const std::vector<int>& alleles = read_to_alleles_[r];
    for (size_t k = 1; k < alleles.size(); ++k) {
      const Vertex& cur = vertices_[alleles[k]];
      const Vertex& prev = vertices_[alleles[k - 1]];
      const auto it = std::find(positions_.begin(), positions_.end(), cur.position);
      const size_t index = static_cast<size_t>(it - positions_.begin());
      if (index == 0) continue;                     // cannot happen: prev sits at an earlier position
      if (index - 1 == 0 || positions_[index - 1] == prev.position) {
        add_edge(alleles[k - 1], alleles[k], 1.0f);  // both supports are high quality here: 0.5 + 0.5
      }
    }
  }
}

void DirectPhasing::update_starting_score(const std::vector<int>& verts) {   // :474-504
  for (int a : verts) {
    for (int b : verts) scores_.erase({a, b});
  }
  for (size_t i = 0; i < verts.size(); ++i) {
    for (size_t j = i; j < verts.size(); ++j) {
      Score s;
      s.read_support[0].insert(vertices_[verts[i]].reads.begin(), vertices_[verts[i]].reads.end());
      s.read_support[1].insert(vertices_[verts[j]].reads.begin(), vertices_[verts[j]].reads.end());
      s.score = s.read_support[0] == s.read_support[1]
                    ? static_cast<int>(s.read_support[0].size())
                    : static_cast<int>(s.read_support[0].size() + s.read_support[1].size());
      scores_[{verts[i], verts[j]}] = std::move(s);
    }
  }
}

// CalculateScore (:419-472) with FindSupportingReads (:399-417): what the pair of edges adds
// to the score of the pair of paths it extends.
DirectPhasing::Score DirectPhasing::calculate_score(const Pair& edge1, const Pair& edge2) const {
  const int from[2] = {edge1.first, edge2.first}, to[2] = {edge1.second, edge2.second};
  const auto prev_it = scores_.find({from[0], from[1]});
  if (prev_it == scores_.end()) return Score();
  const Score& prev = prev_it->second;
  std::set<int> continuing[kNumOfPhases], first_allele[kNumOfPhases];
  for (int phase = 0; phase < kNumOfPhases; ++phase) {
    const Vertex& v = vertices_[to[phase]];
    for (size_t k = 0; k < v.reads.size(); ++k) {
      if (v.is_first_allele[k]) first_allele[phase].insert(v.reads[k]);
      if (prev.read_support[phase].count(v.reads[k])) continuing[phase].insert(v.reads[k]);
    }
  }
  std::set<int> all_continuing(continuing[0]), all_first(first_allele[0]);
  all_continuing.insert(continuing[1].begin(), continuing[1].end());
  all_first.insert(first_allele[1].begin(), first_allele[1].end());
  Score s;
  s.score = prev.score + static_cast<int>(all_continuing.size()) + static_cast<int>(all_first.size()) / 2;
  if (continuing[0].size() < 2 && continuing[1].size() < 2) s.score = prev.score;
  for (int phase = 0; phase < kNumOfPhases; ++phase) {
    s.from[phase] = from[phase];
    s.read_support[phase] = continuing[phase];
    s.read_support[phase].insert(first_allele[phase].begin(), first_allele[phase].end());
  }
  return s;
}

bool DirectPhasing::compare_vertex_pair_by_bases(int a1, int a2, int b1, int b2) const {   // :213-232
  if (a1 < 0 || a2 < 0) return false;
  if (b1 < 0 || b2 < 0) return true;
  if (vertices_[a1].bases > vertices_[b1].bases) return true;
  if (vertices_[a1].bases < vertices_[b1].bases) return false;
  return vertices_[a2].bases > vertices_[b2].bases;
}

bool DirectPhasing::phase_reads(const std::vector<PhasingCandidate>& candidates, int n_reads,
                                std::vector<int>* phases, std::string* error) {
  vertices_.clear();
  positions_.clear();
  vertices_by_position_.clear();
  edges_.clear();
  in_edges_.clear();
  scores_.clear();
  read_to_alleles_.assign(std::max(n_reads, 0), {});
  allele_phases_.assign(candidates.size(), {});
  for (size_t i = 1; i < candidates.size(); ++i) {
    if (!(candidates[i - 1].start < candidates[i].start)) {
      *error = "Check failed: candidates[i - 1].variant().start() < candidate.variant().start()";
      return false;
    }
  }
  build(candidates);

  for (size_t i = 0; i < positions_.size(); ++i) {
    const std::vector<int>& here = vertices_by_position_[positions_[i]];
    bool has_incoming = false;
    for (int v : here) has_incoming = has_incoming || !in_edges_[v].empty();
    if (i == 0 || !has_incoming) {
      update_starting_score(here);
      continue;
    }
    // all incoming edges; a vertex no read reaches is connected to every vertex before it
    std::map<std::pair<std::string, std::string>, Pair> keyed_edges;
    for (int v : here) {
      if (in_edges_[v].empty()) {
        for (int prev_v : vertices_by_position_[positions_[i - 1]]) add_edge(prev_v, v, 0.0f);
      }
      for (int u : in_edges_[v]) keyed_edges[{vertices_[u].bases, vertices_[v].bases}] = {u, v};
    }
    bool found_advancing_score = false;
    for (const auto& e1 : keyed_edges) {
      for (const auto& e2 : keyed_edges) {
        const Pair &edge1 = e1.second, &edge2 = e2.second;
        const auto prev_it = scores_.find({edge1.first, edge2.first});
        if (prev_it == scores_.end()) continue;
        const int prev_score = prev_it->second.score;
        Score score = calculate_score(edge1, edge2);
        if (prev_score < score.score) found_advancing_score = true;
        const Pair to{edge1.second, edge2.second};
        auto existing = scores_.find(to);
        if (existing == scores_.end()) {
          scores_[to] = std::move(score);
        } else if (existing->second.score < score.score) {
          existing->second = std::move(score);
        } else if (existing->second.score == score.score &&
                   compare_vertex_pair_by_bases(score.from[0], score.from[1], existing->second.from[0],
                                                existing->second.from[1])) {
          existing->second = std::move(score);
        }
      }
    }
    if (i + 1 < positions_.size()) {
      // AllScoresAreTheSame (:183-211): the scores reached here differ by at most one
      int min_score = INT32_MAX, max_score = 0;
      for (const auto& e1 : keyed_edges) {
        for (const auto& e2 : keyed_edges) {
          const auto it = scores_.find({e1.second.second, e2.second.second});
          if (it == scores_.end()) continue;
          min_score = std::min(min_score, it->second.score);
          max_score = std::max(max_score, it->second.score);
        }
      }
      const bool all_the_same = !(max_score - min_score > 1);
      if (!found_advancing_score || all_the_same) update_starting_score(here);   // a new phase block starts
    }
  }
  assign_phases_to_vertices();

  // AssignPhasesToReads, :362-389
  phases->assign(std::max(n_reads, 0), 0);
  for (int r = 0; r < n_reads; ++r) {
    int count[3] = {0, 0, 0};
    for (int v : read_to_alleles_[r]) ++count[vertices_[v].phase];
    if (count[1] > count[2] && count[1] >= min_alleles_to_phase_) {
      (*phases)[r] = 1;
    } else if (count[2] > count[1] && count[2] >= min_alleles_to_phase_) {
      (*phases)[r] = 2;
    }
  }
  for (const Vertex& v : vertices_) {
    allele_phases_[v.candidate][v.allele].phase = v.phase;
    allele_phases_[v.candidate][v.allele].is_first_in_block = v.is_first_in_block;
  }
  return true;
}

// MaxScore (:234-286): the best-scoring vertex pair at a position (ties: larger bases), or
// nothing if all pairs there score the same.
bool DirectPhasing::max_score(int position_index, Pair* best) const {
  const std::vector<int>& verts = vertices_by_position_.at(positions_[position_index]);
  bool have = false;
  int best_score = 0;
  for (int v1 : verts) {
    for (int v2 : verts) {
      const auto it = scores_.find({v1, v2});
      if (it == scores_.end()) continue;
      if (it->second.score > best_score) {
        *best = {v1, v2};
        best_score = it->second.score;
        have = true;
      } else if (it->second.score == best_score) {
        if (!have || compare_vertex_pair_by_bases(v1, v2, best->first, best->second)) {
          *best = {v1, v2};
          best_score = it->second.score;
          have = true;
        }
      }
    }
  }
  for (int v1 : verts) {
    for (int v2 : verts) {
      const auto it = scores_.find({v1, v2});
      if (it != scores_.end() && it->second.score != best_score) return have;
    }
  }
  return false;     // all scores equal
}

void DirectPhasing::assign_phases_to_vertices() {   // :288-360: back-track block by block
  if (scores_.empty()) return;
  int i = static_cast<int>(positions_.size()) - 1;
  bool have_cur = false, have_prev = false;
  Pair cur, prev;
  while (i >= 0) {
    have_cur = false;
    while (i >= 0) {
      have_cur = max_score(i, &cur);
      if (have_cur) break;
      --i;
    }
    if (!have_prev) {
      prev = cur;
      have_prev = have_cur;
    } else {
      vertices_[prev.first].is_first_in_block = true;
      vertices_[prev.second].is_first_in_block = true;
    }
    int verts_in_block = 0;
    while (have_cur) {
      ++verts_in_block;
      const bool het = cur.first != cur.second;
      vertices_[cur.first].phase = het ? 1 : 0;
      vertices_[cur.second].phase = het ? 2 : 0;
      const Score& score = scores_.at(cur);
      if (cur != prev && verts_in_block > 1 && have_prev && score.score == scores_.at(prev).score) {
        vertices_[cur.first].phase = 0;             // "unexpected phasing score"
        vertices_[cur.second].phase = 0;
        --i;
        break;
      }
      const Pair next{score.from[0], score.from[1]};
      if (scores_.find(next) == scores_.end()) {    // the block's first position
        if (verts_in_block == 1) {
          vertices_[cur.first].phase = 0;
          vertices_[cur.second].phase = 0;
        }
        --i;
        prev = cur;
        have_prev = true;
        break;
      }
      if (cur == next) {                            // "loop detected"
        --i;
        break;
      }
      prev = cur;
      have_prev = true;
      cur = next;
      --i;
    }
  }
  if (have_prev && have_cur) {
    vertices_[cur.first].is_first_in_block = true;
    vertices_[cur.second].is_first_in_block = true;
  }
}

std::string DirectPhasing::graphviz() const {
  std::ostringstream out;
  out << "digraph G {\n";
  for (size_t v = 0; v < vertices_.size(); ++v) {
    out << v << "[label=\"" << vertices_[v].position << " " << vertices_[v].bases << "\"];\n";
  }
  for (const auto& e : edges_) out << e.first.first << "->" << e.first.second << " [label=" << e.second << "];\n";
  out << "}\n";
  return out.str();
}

}  // namespace dv
