// TEST INFRASTRUCTURE (oracle/ref_build): boost::make_reverse_graph -- the same vertices, every edge turned round.
#ifndef DVREF_BOOST_REVERSE_GRAPH_SHIM_HPP_
#define DVREF_BOOST_REVERSE_GRAPH_SHIM_HPP_
#include "boost/graph/adjacency_list.hpp"
namespace boost {
template <class G>
class reverse_graph {
 public:
  explicit reverse_graph(const G& g) : m_g(g) {}
  using vertex_descriptor = void*;
  using edge_descriptor = mini_bgl::edge_descriptor;
  using vertex_iterator = mini_bgl::VertexIter;
  using out_edge_iterator = mini_bgl::EdgeMapIter<true>;
  using in_edge_iterator = mini_bgl::EdgeMapIter<true>;
  using adjacency_iterator = mini_bgl::AdjacencyIter<true>;
  using edge_iterator = mini_bgl::AllEdgesIter;
  using vertices_size_type = size_t;
  using edges_size_type = size_t;
  using degree_size_type = size_t;
  const G& m_g;
};
template <class G>
reverse_graph<G> make_reverse_graph(const G& g) { return reverse_graph<G>(g); }
template <class G>
std::pair<mini_bgl::VertexIter, mini_bgl::VertexIter> vertices(const reverse_graph<G>& g) { return vertices(g.m_g); }
template <class G>
std::pair<mini_bgl::EdgeMapIter<true>, mini_bgl::EdgeMapIter<true>> out_edges(void* v, const reverse_graph<G>&) {
  auto* n = static_cast<mini_bgl::NodeBase*>(v);
  return {mini_bgl::EdgeMapIter<true>(n->in.begin()), mini_bgl::EdgeMapIter<true>(n->in.end())};
}
template <class G>
void* source(const mini_bgl::edge_descriptor& e, const reverse_graph<G>&) { return e.m_source; }
template <class G>
void* target(const mini_bgl::edge_descriptor& e, const reverse_graph<G>&) { return e.m_target; }
template <class G>
size_t num_vertices(const reverse_graph<G>& g) { return num_vertices(g.m_g); }
}  // namespace boost
#endif
