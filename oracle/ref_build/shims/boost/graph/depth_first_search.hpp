// TEST INFRASTRUCTURE (oracle/ref_build): boost::depth_first_search with a visitor and BGL's named parameters
// (visitor(v).root_vertex(r).vertex_index_map(m)).  As in BGL: the visitor is COPIED, every vertex starts white, the
// root's tree comes first, then a tree from every vertex still white, in vertices(g) order; out-edges in out_edges(v, g)
// order; events discover_vertex / examine_edge / tree_edge / back_edge / forward_or_cross_edge / finish_vertex.
#ifndef DVREF_BOOST_DFS_SHIM_HPP_
#define DVREF_BOOST_DFS_SHIM_HPP_
#include <map>
#include <utility>
#include <vector>
#include "boost/graph/adjacency_list.hpp"
namespace boost {
struct null_visitor {};
template <class Visitors = null_visitor>
class dfs_visitor {
 public:
  template <class V, class G> void initialize_vertex(V, const G&) {}
  template <class V, class G> void start_vertex(V, const G&) {}
  template <class V, class G> void discover_vertex(V, const G&) {}
  template <class E, class G> void examine_edge(E, const G&) {}
  template <class E, class G> void tree_edge(E, const G&) {}
  template <class E, class G> void back_edge(E, const G&) {}
  template <class E, class G> void forward_or_cross_edge(E, const G&) {}
  template <class E, class G> void finish_edge(E, const G&) {}
  template <class V, class G> void finish_vertex(V, const G&) {}
};
template <class Vis>
struct dfs_params {
  Vis vis;
  void* root = nullptr;
  dfs_params root_vertex(void* r) const {
    dfs_params p = *this;
    p.root = r;
    return p;
  }
  template <class M> dfs_params vertex_index_map(const M&) const { return *this; }
  template <class M> dfs_params color_map(const M&) const { return *this; }
};
template <class Vis>
dfs_params<Vis> visitor(const Vis& vis) { return dfs_params<Vis>{vis, nullptr}; }

template <class G, class Vis>
void depth_first_search(const G& g, const dfs_params<Vis>& params) {
  Vis vis = params.vis;
  std::map<void*, int> color;   // 0 white, 1 gray, 2 black
  for (auto r = vertices(g); r.first != r.second; ++r.first) {
    color[*r.first] = 0;
    vis.initialize_vertex(*r.first, g);
  }
  auto visit = [&](void* start) {
    using OutIter = decltype(out_edges(start, g).first);
    struct Frame {
      void* v;
      OutIter it, end;
    };
    std::vector<Frame> stack;
    color[start] = 1;
    vis.discover_vertex(start, g);
    auto r0 = out_edges(start, g);
    stack.push_back({start, r0.first, r0.second});
    while (!stack.empty()) {
      Frame& f = stack.back();
      if (f.it == f.end) {
        color[f.v] = 2;
        vis.finish_vertex(f.v, g);
        stack.pop_back();
        continue;
      }
      const auto e = *f.it;
      ++f.it;
      vis.examine_edge(e, g);
      void* t = target(e, g);
      const int c = color[t];
      if (c == 0) {
        vis.tree_edge(e, g);
        color[t] = 1;
        vis.discover_vertex(t, g);
        auto r = out_edges(t, g);
        stack.push_back({t, r.first, r.second});
      } else if (c == 1) {
        vis.back_edge(e, g);
      } else {
        vis.forward_or_cross_edge(e, g);
      }
    }
  };
  if (params.root) {
    vis.start_vertex(params.root, g);
    visit(params.root);
  }
  for (auto r = vertices(g); r.first != r.second; ++r.first) {
    if (color[*r.first] == 0) {
      vis.start_vertex(*r.first, g);
      visit(*r.first);
    }
  }
}
}  // namespace boost
#endif
