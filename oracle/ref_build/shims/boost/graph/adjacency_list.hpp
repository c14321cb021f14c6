// TEST INFRASTRUCTURE (oracle/ref_build): the slice of the Boost Graph Library the reference's de Bruijn graph
// (realigner/debruijn_graph.cc) and read phasing (direct_phasing.cc) use -- adjacency_list<setS, listS,
// bidirectionalS, VertexProperty, EdgeProperty> with bundled properties, the free-function interface, graph_traits,
// reverse_graph, depth_first_search with a visitor, write_graphviz -- written from BGL's documented semantics; Boost
// is not in the image.  One deliberate difference: BGL's setS edge containers order the edges of a vertex by the
// other end's DESCRIPTOR, which for listS vertices is a heap address; here they are ordered by that vertex's creation
// number (what address order amounts to when vertices are allocated one after another, and the only order a test can
// depend on).
#ifndef DVREF_BOOST_ADJACENCY_LIST_SHIM_HPP_
#define DVREF_BOOST_ADJACENCY_LIST_SHIM_HPP_
#include <cstddef>
#include <iterator>
#include <list>
#include <map>
#include <memory>
#include <utility>
#include <vector>

namespace boost {

struct setS {};
struct listS {};
struct vecS {};
struct bidirectionalS {};
struct directedS {};
struct undirectedS {};
struct no_property {};

namespace mini_bgl {

struct EdgeBase;
struct NodeBase {
  size_t id = 0;
  std::map<size_t, EdgeBase*> out, in;   // keyed by the other end's creation number
  virtual ~NodeBase() {}
};
struct EdgeBase {
  NodeBase* src = nullptr;
  NodeBase* dst = nullptr;
  std::list<EdgeBase*>::iterator at;   // its place in the graph's edge list
  virtual ~EdgeBase() {}
};

struct edge_descriptor {
  void* m_source = nullptr;
  void* m_target = nullptr;
  EdgeBase* m_edge = nullptr;
  friend bool operator==(const edge_descriptor& a, const edge_descriptor& b) { return a.m_edge == b.m_edge; }
  friend bool operator!=(const edge_descriptor& a, const edge_descriptor& b) { return a.m_edge != b.m_edge; }
  friend bool operator<(const edge_descriptor& a, const edge_descriptor& b) { return a.m_edge < b.m_edge; }
};

// iterates a std::map<size_t, EdgeBase*>, yielding edge descriptors (optionally with the ends swapped: reverse_graph)
template <bool Swap>
class EdgeMapIter {
 public:
  using iterator_category = std::forward_iterator_tag;
  using value_type = edge_descriptor;
  using difference_type = std::ptrdiff_t;
  using pointer = const edge_descriptor*;
  using reference = edge_descriptor;
  EdgeMapIter() = default;
  explicit EdgeMapIter(std::map<size_t, EdgeBase*>::const_iterator it) : it_(it) {}
  edge_descriptor operator*() const {
    EdgeBase* e = it_->second;
    return Swap ? edge_descriptor{e->dst, e->src, e} : edge_descriptor{e->src, e->dst, e};
  }
  EdgeMapIter& operator++() { ++it_; return *this; }
  EdgeMapIter operator++(int) { EdgeMapIter t = *this; ++it_; return t; }
  friend bool operator==(const EdgeMapIter& a, const EdgeMapIter& b) { return a.it_ == b.it_; }
  friend bool operator!=(const EdgeMapIter& a, const EdgeMapIter& b) { return a.it_ != b.it_; }
 private:
  std::map<size_t, EdgeBase*>::const_iterator it_;
};

// ... yielding the vertex at the far end
template <bool In>
class AdjacencyIter {
 public:
  using iterator_category = std::forward_iterator_tag;
  using value_type = void*;
  using difference_type = std::ptrdiff_t;
  using pointer = void* const*;
  using reference = void*;
  AdjacencyIter() = default;
  explicit AdjacencyIter(std::map<size_t, EdgeBase*>::const_iterator it) : it_(it) {}
  void* operator*() const { return In ? it_->second->src : it_->second->dst; }
  AdjacencyIter& operator++() { ++it_; return *this; }
  AdjacencyIter operator++(int) { AdjacencyIter t = *this; ++it_; return t; }
  friend bool operator==(const AdjacencyIter& a, const AdjacencyIter& b) { return a.it_ == b.it_; }
  friend bool operator!=(const AdjacencyIter& a, const AdjacencyIter& b) { return a.it_ != b.it_; }
 private:
  std::map<size_t, EdgeBase*>::const_iterator it_;
};

class VertexIter {
 public:
  using iterator_category = std::forward_iterator_tag;
  using value_type = void*;
  using difference_type = std::ptrdiff_t;
  using pointer = void* const*;
  using reference = void*;
  VertexIter() = default;
  explicit VertexIter(std::list<NodeBase*>::const_iterator it) : it_(it) {}
  void* operator*() const { return *it_; }
  VertexIter& operator++() { ++it_; return *this; }
  VertexIter operator++(int) { VertexIter t = *this; ++it_; return t; }
  friend bool operator==(const VertexIter& a, const VertexIter& b) { return a.it_ == b.it_; }
  friend bool operator!=(const VertexIter& a, const VertexIter& b) { return a.it_ != b.it_; }
 private:
  std::list<NodeBase*>::const_iterator it_;
};

// every edge of the graph, in the order the edges were ADDED: a bidirectional adjacency_list keeps one global edge
// list (write_graphviz dumps follow it -- the reference's tests compare such dumps)
class AllEdgesIter {
 public:
  using iterator_category = std::forward_iterator_tag;
  using value_type = edge_descriptor;
  using difference_type = std::ptrdiff_t;
  using pointer = const edge_descriptor*;
  using reference = edge_descriptor;
  AllEdgesIter() = default;
  explicit AllEdgesIter(std::list<EdgeBase*>::const_iterator it) : it_(it) {}
  edge_descriptor operator*() const { return edge_descriptor{(*it_)->src, (*it_)->dst, *it_}; }
  AllEdgesIter& operator++() { ++it_; return *this; }
  AllEdgesIter operator++(int) { AllEdgesIter t = *this; ++it_; return t; }
  friend bool operator==(const AllEdgesIter& a, const AllEdgesIter& b) { return a.it_ == b.it_; }
  friend bool operator!=(const AllEdgesIter& a, const AllEdgesIter& b) { return a.it_ != b.it_; }
 private:
  std::list<EdgeBase*>::const_iterator it_;
};

}  // namespace mini_bgl

template <class OutEdgeListS, class VertexListS, class DirectedS, class VertexProperty = no_property,
          class EdgeProperty = no_property>
class adjacency_list {
 public:
  struct Node : mini_bgl::NodeBase {
    VertexProperty prop;
  };
  struct Edge : mini_bgl::EdgeBase {
    EdgeProperty prop;
  };
  using vertex_descriptor = void*;
  using edge_descriptor = mini_bgl::edge_descriptor;
  using vertex_iterator = mini_bgl::VertexIter;
  using edge_iterator = mini_bgl::AllEdgesIter;
  using out_edge_iterator = mini_bgl::EdgeMapIter<false>;
  using in_edge_iterator = mini_bgl::EdgeMapIter<false>;
  using adjacency_iterator = mini_bgl::AdjacencyIter<false>;
  using inv_adjacency_iterator = mini_bgl::AdjacencyIter<true>;
  using vertices_size_type = size_t;
  using edges_size_type = size_t;
  using degree_size_type = size_t;
  using vertex_bundled = VertexProperty;
  using edge_bundled = EdgeProperty;
  using graph_tag = void;

  adjacency_list() = default;
  adjacency_list(const adjacency_list&) = delete;
  adjacency_list& operator=(const adjacency_list&) = delete;
  ~adjacency_list() { clear(); }

  VertexProperty& operator[](vertex_descriptor v) { return static_cast<Node*>(static_cast<mini_bgl::NodeBase*>(v))->prop; }
  const VertexProperty& operator[](vertex_descriptor v) const {
    return static_cast<const Node*>(static_cast<const mini_bgl::NodeBase*>(v))->prop;
  }
  EdgeProperty& operator[](const edge_descriptor& e) { return static_cast<Edge*>(e.m_edge)->prop; }
  const EdgeProperty& operator[](const edge_descriptor& e) const { return static_cast<const Edge*>(e.m_edge)->prop; }

  void clear() {
    for (mini_bgl::EdgeBase* e : edges_) delete e;
    edges_.clear();
    for (mini_bgl::NodeBase* n : nodes_) delete n;
    nodes_.clear();
  }

  static vertex_descriptor null_vertex() { return nullptr; }

  // (the free functions below are the interface; these are their bodies)
  std::list<mini_bgl::NodeBase*> nodes_;
  std::list<mini_bgl::EdgeBase*> edges_;
  size_t next_id_ = 0;
};

template <class G>
struct graph_traits {
  using vertex_descriptor = typename G::vertex_descriptor;
  using edge_descriptor = typename G::edge_descriptor;
  using vertex_iterator = typename G::vertex_iterator;
  using edge_iterator = typename G::edge_iterator;
  using out_edge_iterator = typename G::out_edge_iterator;
  using in_edge_iterator = typename G::in_edge_iterator;
  using adjacency_iterator = typename G::adjacency_iterator;
  using vertices_size_type = typename G::vertices_size_type;
  using edges_size_type = typename G::edges_size_type;
  using degree_size_type = typename G::degree_size_type;
  static vertex_descriptor null_vertex() { return nullptr; }
};

#define DVREF_BGL_TEMPLATE template <class O, class V, class D, class VP, class EP>
#define DVREF_BGL_GRAPH adjacency_list<O, V, D, VP, EP>

DVREF_BGL_TEMPLATE void* add_vertex(const VP& prop, DVREF_BGL_GRAPH& g) {
  auto* n = new typename DVREF_BGL_GRAPH::Node();
  n->id = g.next_id_++;
  n->prop = prop;
  g.nodes_.push_back(n);
  return static_cast<mini_bgl::NodeBase*>(n);
}
DVREF_BGL_TEMPLATE void* add_vertex(DVREF_BGL_GRAPH& g) { return add_vertex(VP(), g); }

DVREF_BGL_TEMPLATE std::pair<mini_bgl::edge_descriptor, bool> edge(void* u, void* v, const DVREF_BGL_GRAPH&) {
  auto* a = static_cast<mini_bgl::NodeBase*>(u);
  auto* b = static_cast<mini_bgl::NodeBase*>(v);
  auto it = a->out.find(b->id);
  if (it == a->out.end()) return {mini_bgl::edge_descriptor{u, v, nullptr}, false};
  return {mini_bgl::edge_descriptor{u, v, it->second}, true};
}

// setS: no parallel edges -- an existing edge comes back with `false`
DVREF_BGL_TEMPLATE std::pair<mini_bgl::edge_descriptor, bool> add_edge(void* u, void* v, const EP& prop, DVREF_BGL_GRAPH& g) {
  auto* a = static_cast<mini_bgl::NodeBase*>(u);
  auto* b = static_cast<mini_bgl::NodeBase*>(v);
  auto it = a->out.find(b->id);
  if (it != a->out.end()) return {mini_bgl::edge_descriptor{u, v, it->second}, false};
  auto* e = new typename DVREF_BGL_GRAPH::Edge();
  e->src = a;
  e->dst = b;
  e->prop = prop;
  a->out[b->id] = e;
  b->in[a->id] = e;
  e->at = g.edges_.insert(g.edges_.end(), e);
  return {mini_bgl::edge_descriptor{u, v, e}, true};
}
DVREF_BGL_TEMPLATE std::pair<mini_bgl::edge_descriptor, bool> add_edge(void* u, void* v, DVREF_BGL_GRAPH& g) {
  return add_edge(u, v, EP(), g);
}

DVREF_BGL_TEMPLATE std::pair<mini_bgl::VertexIter, mini_bgl::VertexIter> vertices(const DVREF_BGL_GRAPH& g) {
  return {mini_bgl::VertexIter(g.nodes_.begin()), mini_bgl::VertexIter(g.nodes_.end())};
}
DVREF_BGL_TEMPLATE std::pair<mini_bgl::AllEdgesIter, mini_bgl::AllEdgesIter> edges(const DVREF_BGL_GRAPH& g) {
  return {mini_bgl::AllEdgesIter(g.edges_.begin()), mini_bgl::AllEdgesIter(g.edges_.end())};
}
DVREF_BGL_TEMPLATE std::pair<mini_bgl::EdgeMapIter<false>, mini_bgl::EdgeMapIter<false>> out_edges(void* v, const DVREF_BGL_GRAPH&) {
  auto* n = static_cast<mini_bgl::NodeBase*>(v);
  return {mini_bgl::EdgeMapIter<false>(n->out.begin()), mini_bgl::EdgeMapIter<false>(n->out.end())};
}
DVREF_BGL_TEMPLATE std::pair<mini_bgl::EdgeMapIter<false>, mini_bgl::EdgeMapIter<false>> in_edges(void* v, const DVREF_BGL_GRAPH&) {
  auto* n = static_cast<mini_bgl::NodeBase*>(v);
  return {mini_bgl::EdgeMapIter<false>(n->in.begin()), mini_bgl::EdgeMapIter<false>(n->in.end())};
}
DVREF_BGL_TEMPLATE std::pair<mini_bgl::AdjacencyIter<false>, mini_bgl::AdjacencyIter<false>> adjacent_vertices(void* v, const DVREF_BGL_GRAPH&) {
  auto* n = static_cast<mini_bgl::NodeBase*>(v);
  return {mini_bgl::AdjacencyIter<false>(n->out.begin()), mini_bgl::AdjacencyIter<false>(n->out.end())};
}
DVREF_BGL_TEMPLATE size_t out_degree(void* v, const DVREF_BGL_GRAPH&) { return static_cast<mini_bgl::NodeBase*>(v)->out.size(); }
DVREF_BGL_TEMPLATE size_t in_degree(void* v, const DVREF_BGL_GRAPH&) { return static_cast<mini_bgl::NodeBase*>(v)->in.size(); }
DVREF_BGL_TEMPLATE size_t degree(void* v, const DVREF_BGL_GRAPH& g) { return out_degree(v, g) + in_degree(v, g); }
DVREF_BGL_TEMPLATE void* source(const mini_bgl::edge_descriptor& e, const DVREF_BGL_GRAPH&) { return e.m_source; }
DVREF_BGL_TEMPLATE void* target(const mini_bgl::edge_descriptor& e, const DVREF_BGL_GRAPH&) { return e.m_target; }
DVREF_BGL_TEMPLATE size_t num_vertices(const DVREF_BGL_GRAPH& g) { return g.nodes_.size(); }
DVREF_BGL_TEMPLATE size_t num_edges(const DVREF_BGL_GRAPH& g) { return g.edges_.size(); }

DVREF_BGL_TEMPLATE void remove_edge(const mini_bgl::edge_descriptor& e, DVREF_BGL_GRAPH& g) {
  if (!e.m_edge) return;
  e.m_edge->src->out.erase(e.m_edge->dst->id);
  e.m_edge->dst->in.erase(e.m_edge->src->id);
  g.edges_.erase(e.m_edge->at);
  delete e.m_edge;
}
DVREF_BGL_TEMPLATE void remove_edge(void* u, void* v, DVREF_BGL_GRAPH& g) {
  auto found = edge(u, v, g);
  if (found.second) remove_edge(found.first, g);
}
template <class Pred, class O, class V, class D, class VP, class EP>
void remove_edge_if(Pred pred, DVREF_BGL_GRAPH& g) {
  std::vector<mini_bgl::edge_descriptor> doomed;
  for (auto range = edges(g); range.first != range.second; ++range.first) {
    if (pred(*range.first)) doomed.push_back(*range.first);
  }
  for (const auto& e : doomed) remove_edge(e, g);
}
// removes every edge to and from v
DVREF_BGL_TEMPLATE void clear_vertex(void* v, DVREF_BGL_GRAPH& g) {
  auto* n = static_cast<mini_bgl::NodeBase*>(v);
  std::vector<mini_bgl::edge_descriptor> doomed;
  for (auto& kv : n->out) doomed.push_back({kv.second->src, kv.second->dst, kv.second});
  for (auto& kv : n->in) {
    if (kv.second->src != n) doomed.push_back({kv.second->src, kv.second->dst, kv.second});   // (a self loop is in `out` already)
  }
  for (const auto& e : doomed) remove_edge(e, g);
}
// the vertex must have no edges left (BGL's precondition)
DVREF_BGL_TEMPLATE void remove_vertex(void* v, DVREF_BGL_GRAPH& g) {
  auto* n = static_cast<mini_bgl::NodeBase*>(v);
  g.nodes_.remove(n);
  delete n;
}

// get(&Bundle::member, g): a property map from vertex descriptors to that member
template <class G, class Bundle, class T>
struct bundle_member_map {
  const G* g;
  T Bundle::*member;
  const T& operator[](void* v) const { return ((*g)[v]).*member; }
};
template <class Bundle, class T, class O, class V, class D, class VP, class EP>
bundle_member_map<DVREF_BGL_GRAPH, Bundle, T> get(T Bundle::*member, const DVREF_BGL_GRAPH& g) {
  return bundle_member_map<DVREF_BGL_GRAPH, Bundle, T>{&g, member};
}
template <class G, class Bundle, class T>
const T& get(const bundle_member_map<G, Bundle, T>& m, void* v) { return m[v]; }

// an associative container (std::map / flat_hash_map) as a read-only property map
template <class C>
class const_associative_property_map {
 public:
  using key_type = typename C::key_type;
  using value_type = typename C::mapped_type;
  const_associative_property_map() : c_(nullptr) {}
  const_associative_property_map(const C& c) : c_(&c) {}
  const value_type& operator[](const key_type& k) const { return c_->find(k)->second; }
 private:
  const C* c_;
};
template <class C>
const typename C::mapped_type& get(const const_associative_property_map<C>& m, const typename C::key_type& k) { return m[k]; }

}  // namespace boost
#endif
