// See debruijn_graph.h.
#include "debruijn_graph.h"

#include <algorithm>
#include <queue>
#include <set>

namespace dv {
namespace {

inline bool canonical(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// first k of the schedule whose reference k-mers are all distinct; -1 if none
// (KMinMaxFromReference, debruijn_graph.cc:185-212)
int first_k_without_reference_repeat(std::string_view ref, const DeBruijnOptions& o, int max_k) {
  if (o.step_k <= 0) return -1;
  for (int k = o.min_k; k <= max_k; k += o.step_k) {
    if (k <= 0) continue;
    std::set<std::string_view> seen;
    bool repeat = false;
    for (size_t i = 0; i + k <= ref.size(); ++i) {
      if (!seen.insert(ref.substr(i, k)).second) {
        repeat = true;
        break;
      }
    }
    if (!repeat) return k;
  }
  return -1;
}

}  // namespace

std::unique_ptr<DeBruijnGraph> DeBruijnGraph::build(std::string_view ref, const std::vector<AssemblyRead>& reads,
                                                    const DeBruijnOptions& options) {
  const int max_k = std::min(options.max_k, static_cast<int>(ref.size()) - 1);
  const int first_k = first_k_without_reference_repeat(ref, options, max_k);
  if (first_k < 0) return nullptr;
  for (int k = first_k; k <= max_k; k += options.step_k) {
    std::unique_ptr<DeBruijnGraph> g(new DeBruijnGraph(ref, reads, options, k));
    if (g->has_cycle()) continue;
    if (options.disable_graph_pruning) {
      g->prune_lite();
    } else {
      g->prune();
    }
    return g;
  }
  return nullptr;
}

DeBruijnGraph::DeBruijnGraph(std::string_view ref, const std::vector<AssemblyRead>& reads,
                             const DeBruijnOptions& options, int k)
    : options_(options), k_(k) {
  add_kmers_and_edges(ref, 0, static_cast<int>(ref.size()) - k_, true);
  source_ = vertex_of_.at(ref.substr(0, k_));
  sink_ = vertex_of_.at(ref.substr(ref.size() - k_, k_));
  for (const AssemblyRead& read : reads) {
    if (read.mapq >= options_.min_mapq) add_edges_for_read(read);
  }
}

int DeBruijnGraph::ensure_vertex(std::string_view kmer) {
  auto it = vertex_of_.find(kmer);
  if (it != vertex_of_.end()) return it->second;
  const int v = static_cast<int>(kmers_.size());
  kmers_.emplace_back(kmer);
  vertex_alive_.push_back(1);
  out_.emplace_back();
  in_.emplace_back();
  vertex_of_.emplace(std::string_view(kmers_.back()), v);
  return v;
}

void DeBruijnGraph::add_edge(int from, int to, bool is_ref) {
  for (int e : out_[from]) {
    if (edges_[e].to == to) {
      ++edges_[e].weight;
      edges_[e].is_ref |= is_ref;
      return;
    }
  }
  const int e = static_cast<int>(edges_.size());
  edges_.push_back(Edge{from, to, 1, is_ref, true});
  out_[from].push_back(e);
  in_[to].push_back(e);
}

// k-mers starting at start .. end (inclusive) and the edges between consecutive ones.  The
// guard is on `end` alone (debruijn_graph.cc:257-266): a segment too short for an edge still
// leaves its first k-mer behind as a vertex.
void DeBruijnGraph::add_kmers_and_edges(std::string_view bases, int start, int end, bool is_ref) {
  if (end <= 0) return;
  int prev = ensure_vertex(bases.substr(start, k_));
  for (int i = start + 1; i <= end; ++i) {
    const int cur = ensure_vertex(bases.substr(i, k_));
    add_edge(prev, cur, is_ref);
    prev = cur;
  }
}

void DeBruijnGraph::add_edges_for_read(const AssemblyRead& read) {
  std::string bases(read.bases);
  for (char& c : bases) {
    if (c >= 'a' && c <= 'z') c = static_cast<char>(c - 'a' + 'A');
  }
  const int n = static_cast<int>(bases.size());
  auto next_bad_position = [&](int from) {
    for (int i = from; i < n; ++i) {
      if (!canonical(bases[i]) || read.quals[i] < options_.min_base_quality) return i;
    }
    return n;
  };
  const int stop = n - k_;
  int i = 0;
  while (i < stop) {
    const int bad = next_bad_position(i);
    add_kmers_and_edges(bases, i, bad - k_, false);
    i = bad + 1;
  }
}

int DeBruijnGraph::out_degree(int v) const {
  int n = 0;
  for (int e : out_[v]) n += edges_[e].alive;
  return n;
}

// any directed cycle among live edges (three-colour depth-first search, explicit stack)
bool DeBruijnGraph::has_cycle() const {
  const int n = static_cast<int>(kmers_.size());
  std::vector<char> colour(n, 0);
  std::vector<std::pair<int, size_t>> stack;
  for (int root = 0; root < n; ++root) {
    if (colour[root] || !vertex_alive_[root]) continue;
    colour[root] = 1;
    stack.emplace_back(root, 0);
    while (!stack.empty()) {
      auto& [v, next] = stack.back();
      if (next == out_[v].size()) {
        colour[v] = 2;
        stack.pop_back();
        continue;
      }
      const Edge& e = edges_[out_[v][next++]];
      if (!e.alive) continue;
      if (colour[e.to] == 1) return true;
      if (colour[e.to] == 0) {
        colour[e.to] = 1;
        stack.emplace_back(e.to, 0);
      }
    }
  }
  return false;
}

std::vector<char> DeBruijnGraph::reachable(int from, bool reverse) const {
  std::vector<char> seen(kmers_.size(), 0);
  std::vector<int> todo{from};
  seen[from] = 1;
  while (!todo.empty()) {
    const int v = todo.back();
    todo.pop_back();
    for (int id : (reverse ? in_ : out_)[v]) {
      const Edge& e = edges_[id];
      if (!e.alive) continue;
      const int w = reverse ? e.from : e.to;
      if (!seen[w]) {
        seen[w] = 1;
        todo.push_back(w);
      }
    }
  }
  return seen;
}

void DeBruijnGraph::drop_vertices(const std::vector<char>& keep) {
  for (size_t v = 0; v < kmers_.size(); ++v) {
    if (!vertex_alive_[v] || keep[v]) continue;
    vertex_alive_[v] = 0;
    vertex_of_.erase(std::string_view(kmers_[v]));
    for (int e : out_[v]) edges_[e].alive = false;
    for (int e : in_[v]) edges_[e].alive = false;
  }
}

void DeBruijnGraph::prune() {
  for (Edge& e : edges_) {
    if (!e.is_ref && e.weight < options_.min_edge_weight) e.alive = false;
  }
  const std::vector<char> fwd = reachable(source_, false), rev = reachable(sink_, true);
  std::vector<char> keep(kmers_.size());
  for (size_t v = 0; v < keep.size(); ++v) keep[v] = fwd[v] && rev[v];
  drop_vertices(keep);
}

void DeBruijnGraph::prune_lite() {
  std::vector<char> keep(kmers_.size());
  for (size_t v = 0; v < keep.size(); ++v) keep[v] = !out_[v].empty() || !in_[v].empty();
  drop_vertices(keep);
}

std::vector<std::string> DeBruijnGraph::candidate_haplotypes() const {
  using Path = std::vector<int>;
  std::vector<Path> done;
  std::queue<Path> open;
  open.push({source_});
  while (!open.empty()) {
    if (static_cast<int>(done.size() + open.size()) > options_.max_num_paths) return {};
    const Path path = std::move(open.front());
    open.pop();
    for (int id : out_[path.back()]) {
      const Edge& e = edges_[id];
      if (!e.alive) continue;
      Path longer(path);
      longer.push_back(e.to);
      if (e.to == sink_ || out_degree(e.to) == 0) {
        done.push_back(std::move(longer));
      } else {
        open.push(std::move(longer));
      }
    }
  }
  std::vector<std::string> haplotypes;
  haplotypes.reserve(done.size());
  for (const Path& path : done) {
    std::string h;
    h.reserve(path.size() + k_);
    for (int v : path) h.push_back(kmers_[v][0]);
    h.append(kmers_[path.back()], 1, k_ - 1);
    haplotypes.push_back(std::move(h));
  }
  std::sort(haplotypes.begin(), haplotypes.end());
  return haplotypes;
}

std::string DeBruijnGraph::graphviz() const {
  std::vector<int> number(kmers_.size(), -1);
  std::string out = "digraph G {\n";
  int next = 0;
  for (size_t v = 0; v < kmers_.size(); ++v) {
    if (!vertex_alive_[v]) continue;
    number[v] = next++;
    out += std::to_string(number[v]) + "[label=" + kmers_[v] + "];\n";
  }
  for (const Edge& e : edges_) {
    if (!e.alive) continue;
    out += std::to_string(number[e.from]) + "->" + std::to_string(number[e.to]) + " [label=" +
           std::to_string(e.weight) + (e.is_ref ? " color=red" : "") + "];\n";
  }
  out += "}\n";
  return out;
}

}  // namespace dv
