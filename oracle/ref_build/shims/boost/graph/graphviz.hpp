// TEST INFRASTRUCTURE (oracle/ref_build): boost::write_graphviz in BGL's output format -- "digraph G {", one line per
// vertex "<index><vertex writer output>;", one per edge "<i>-><j> <edge writer output>;", "}" -- with label writers.
#ifndef DVREF_BOOST_GRAPHVIZ_SHIM_HPP_
#define DVREF_BOOST_GRAPHVIZ_SHIM_HPP_
#include <ostream>
#include <string>
#include "boost/graph/adjacency_list.hpp"
namespace boost {
struct default_writer {
  void operator()(std::ostream&) const {}
  template <class T> void operator()(std::ostream&, const T&) const {}
};
inline std::string escape_dot_string(const std::string& s) {
  bool plain = !s.empty();
  for (char c : s) plain = plain && (std::isalnum(static_cast<unsigned char>(c)) || c == '_');
  if (plain && !std::isdigit(static_cast<unsigned char>(s[0]))) return s;
  bool number = !s.empty();
  for (char c : s) number = number && (std::isdigit(static_cast<unsigned char>(c)) || c == '.' || c == '-');
  if (number) return s;
  std::string out = "\"";
  for (char c : s) {
    if (c == '"') out += '\\';
    out += c;
  }
  return out + "\"";
}
template <class Map>
class label_writer {
 public:
  explicit label_writer(Map m) : m_(m) {}
  template <class K>
  void operator()(std::ostream& out, const K& k) const { out << "[label=" << escape_dot_string(get(m_, k)) << "]"; }
 private:
  Map m_;
};
template <class Map>
label_writer<Map> make_label_writer(Map m) { return label_writer<Map>(m); }

template <class G, class VW, class EW, class GW, class IndexMap>
void write_graphviz(std::ostream& out, const G& g, VW vw, EW ew, GW gw, IndexMap index) {
  out << "digraph G {" << std::endl;
  gw(out);
  for (auto r = vertices(g); r.first != r.second; ++r.first) {
    out << escape_dot_string(std::to_string(get(index, *r.first)));
    vw(out, *r.first);
    out << ";" << std::endl;
  }
  for (auto r = edges(g); r.first != r.second; ++r.first) {
    const auto e = *r.first;
    out << escape_dot_string(std::to_string(get(index, source(e, g)))) << "->"
        << escape_dot_string(std::to_string(get(index, target(e, g)))) << " ";
    ew(out, e);
    out << ";" << std::endl;
  }
  out << "}" << std::endl;
}
}  // namespace boost
#endif
