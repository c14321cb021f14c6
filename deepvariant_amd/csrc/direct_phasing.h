// Read phasing for the long-read path: candidates (heterozygous sites with the reads that
// support each allele) are chained into a graph, a dynamic programme over consecutive sites
// picks the pair of allele paths that explains the most reads, and every read gets the
// phase (1 / 2, or 0) of the alleles it carries.  The HP tags it produces drive the encoder's
// haplotype channel and row order (sort_by_haplotypes).  Behaviour of
// deepvariant/direct_phasing.{h,cc} (DirectPhasing::PhaseReads, GetPhasedVariants); tests
// mirror deepvariant/direct_phasing_test.cc.
//
// Inputs are plain arrays: reads are indices, an allele is its bases plus the reads that
// support it.  Vertices live in one insertion-ordered vector (no graph library); scores are
// keyed by (vertex, vertex) index pairs.
#ifndef DV_DIRECT_PHASING_H_
#define DV_DIRECT_PHASING_H_

#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace dv {

struct PhasingReadSupport {
  int read = -1;              // index into the reads handed to phase_reads; -1 = not among them
  bool is_low_quality = false;
};

struct PhasingAllele {
  std::string bases;          // alt allele bases as in allele_support_ext's key
  bool is_ref = false;        // the entry built from ref_support_ext
  std::vector<PhasingReadSupport> support;
};

struct PhasingCandidate {    // DeepVariantCall: variant.start / end, allele_support_ext, ref_support_ext
  int64_t start = 0, end = 0;
  std::vector<PhasingAllele> alleles;   // UNCALLED_ALLELE left out; at most one is_ref entry
};

struct PhasedAllele {        // what became of PhasingCandidate.alleles[k]
  bool in_graph = false;
  int phase = 0;
  bool is_first_in_block = false;
};

class DirectPhasing {
 public:
  explicit DirectPhasing(int min_alleles_to_phase) : min_alleles_to_phase_(min_alleles_to_phase) {}

  // PhaseReads (direct_phasing.cc:75-166): phase per read.  false + *error when candidates are
  // not strictly ordered by start (CHECK_LT in Build).
  bool phase_reads(const std::vector<PhasingCandidate>& candidates, int n_reads, std::vector<int>* phases,
                   std::string* error);
  // per input candidate, per allele: phase and block flag after phase_reads
  const std::vector<std::vector<PhasedAllele>>& allele_phases() const { return allele_phases_; }
  std::string graphviz() const;

 private:
  struct Vertex {
    int64_t position;
    std::string bases;
    std::vector<int> reads;            // usable supporting reads (known, not low quality)
    std::vector<char> is_first_allele; // per supporting read: its first allele in the graph
    int phase = 0;
    bool is_first_in_block = false;
    int candidate, allele;             // where it came from
  };
  struct Score {
    int score = 0;
    int from[2] = {-1, -1};
    std::set<int> read_support[2];
  };
  using Pair = std::pair<int, int>;

  void build(const std::vector<PhasingCandidate>& candidates);
  int add_vertex(int64_t position, const std::string& bases, const std::vector<PhasingReadSupport>& support,
                 int candidate, int allele);
  void add_edge(int from, int to, float weight);
  void update_starting_score(const std::vector<int>& verts);
  Score calculate_score(const Pair& edge1, const Pair& edge2) const;
  bool compare_vertex_pair_by_bases(int a1, int a2, int b1, int b2) const;
  bool max_score(int position_index, Pair* best) const;
  void assign_phases_to_vertices();

  int min_alleles_to_phase_;
  std::vector<Vertex> vertices_;
  std::vector<int64_t> positions_;
  std::map<int64_t, std::vector<int>> vertices_by_position_;
  std::map<Pair, float> edges_;
  std::vector<std::vector<int>> in_edges_;           // vertex -> source vertices
  std::map<Pair, Score> scores_;
  std::vector<std::vector<int>> read_to_alleles_;    // read -> vertices, in the order they were added
  std::vector<std::vector<char>> read_allele_low_quality_;
  std::vector<std::vector<PhasedAllele>> allele_phases_;
};

}  // namespace dv

#endif  // DV_DIRECT_PHASING_H_
