// Local assembly for the window realigner: a de Bruijn graph over the reference window and
// the reads that overlap it, pruned and enumerated into candidate haplotypes.  Behaviour of
// deepvariant/realigner/debruijn_graph.{h,cc} (DeBruijnGraph::Build / CandidateHaplotypes /
// GraphViz); tests mirror deepvariant/realigner/python/debruijn_graph_wrap_test.py.
//
//   1. the smallest k in [min_k, max_k] (step step_k, max_k clipped to |ref| - 1) for which
//      the reference window has no repeated k-mer starts the search (debruijn_graph.cc:185-212);
//   2. per k: reference k-mers, then the k-mers of every read with mapq >= min_mapq, cut at
//      bases that are not A/C/G/T or below min_base_quality; an edge's weight counts the
//      times it was walked; the first k whose graph is acyclic wins (:214-236);
//   3. pruning drops non-reference edges lighter than min_edge_weight and every vertex that
//      is not on a source -> sink walk (:408-441); with disable_graph_pruning only
//      isolated vertices go (:385-406);
//   4. haplotypes are all source -> sink (or dead-end) walks, breadth first, none at all
//      once more than max_num_paths are alive; reported in lexicographic order (:304-356).
//
// Vertices and edges are plain arrays in insertion order (the order the reference's graphviz
// dump numbers them in); there is no graph library underneath.
#ifndef DV_DEBRUIJN_GRAPH_H_
#define DV_DEBRUIJN_GRAPH_H_

#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace dv {

struct DeBruijnOptions {      // DeBruijnGraphOptions, deepvariant/protos/realigner.proto
  int min_k = 10, max_k = 101, step_k = 1;
  int min_mapq = 14, min_base_quality = 15, min_edge_weight = 2, max_num_paths = 256;
  bool disable_graph_pruning = false;
};

struct AssemblyRead {         // what the graph reads of a nucleus Read
  std::string_view bases;     // aligned_sequence
  const uint8_t* quals;       // aligned_quality, one per base
  int mapq;
};

class DeBruijnGraph {
 public:
  // nullptr when no k in range gives an acyclic graph (the caller then keeps the reference
  // as the only haplotype)
  static std::unique_ptr<DeBruijnGraph> build(std::string_view ref, const std::vector<AssemblyRead>& reads,
                                              const DeBruijnOptions& options);

  int kmer_size() const { return k_; }
  std::vector<std::string> candidate_haplotypes() const;
  std::string graphviz() const;

 private:
  struct Edge {
    int from, to, weight;
    bool is_ref, alive;
  };

  DeBruijnGraph(std::string_view ref, const std::vector<AssemblyRead>& reads, const DeBruijnOptions& options,
                int k);
  int ensure_vertex(std::string_view kmer);
  void add_edge(int from, int to, bool is_ref);
  void add_kmers_and_edges(std::string_view bases, int start, int end, bool is_ref);
  void add_edges_for_read(const AssemblyRead& read);
  bool has_cycle() const;
  void prune();
  void prune_lite();
  std::vector<char> reachable(int from, bool reverse) const;
  int out_degree(int v) const;
  void drop_vertices(const std::vector<char>& keep);

  DeBruijnOptions options_;
  int k_;
  int source_ = -1, sink_ = -1;
  std::deque<std::string> kmers_;                      // vertex -> k-mer (stable addresses)
  std::vector<char> vertex_alive_;
  std::unordered_map<std::string_view, int> vertex_of_;
  std::vector<Edge> edges_;                            // insertion order
  std::vector<std::vector<int>> out_, in_;             // vertex -> edge ids
};

}  // namespace dv

#endif  // DV_DEBRUIJN_GRAPH_H_
