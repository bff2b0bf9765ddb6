// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_PTREE
#define RFS_STUB_BOOST_PTREE
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
namespace boost { namespace property_tree {
class ptree {
 public:
  typedef std::pair<std::string, ptree> value_type;
  typedef std::vector<value_type>::const_iterator const_iterator;
  typedef std::vector<value_type>::iterator iterator;
 private:
  std::string data_; std::vector<value_type> kids_;
  const ptree *walk(const std::string &path) const {
    const ptree *t = this; std::size_t a = 0;
    while (t && a <= path.size()) {
      const std::size_t b = path.find('.', a); const std::string key = path.substr(a, b == std::string::npos ? b : b - a);
      const ptree *n = 0;
      for (const value_type &k : t->kids_) if (k.first == key) { n = &k.second; break; }
      t = n; if (b == std::string::npos) break; a = b + 1;
    }
    return t;
  }
  template <class T> static bool parse(const std::string &s, T &v) { std::istringstream is(s); return bool(is >> v); }
  static bool parse(const std::string &s, std::string &v) { v = s; return true; }
 public:
  const std::string &data() const { return data_; }
  std::string &data() { return data_; }
  ptree &add_child_node(const std::string &k) { kids_.push_back(value_type(k, ptree())); return kids_.back().second; }
  const_iterator begin() const { return kids_.begin(); }
  const_iterator end() const { return kids_.end(); }
  const ptree &get_child(const std::string &p) const { const ptree *t = walk(p); if (!t) throw std::runtime_error("ptree: no such node: " + p); return *t; }
  template <class T> T get(const std::string &p) const { T v; if (!parse(get_child(p).data_, v)) throw std::runtime_error("ptree: bad data at " + p); return v; }
  template <class T> T get(const std::string &p, const T &dflt) const { const ptree *t = walk(p); T v; return (t && parse(t->data_, v)) ? v : dflt; }
  std::string get(const std::string &p, const char *dflt) const { const ptree *t = walk(p); return t ? t->data_ : std::string(dflt); }
};
namespace xml_parser {
// elements, text, comments, <?..?>; no attributes / entities (the reference's cfg/*.xml use none)
inline void read_xml(const std::string &file, ptree &root) {
  std::ifstream f(file.c_str()); if (!f) throw std::runtime_error("read_xml: cannot open " + file);
  std::stringstream ss; ss << f.rdbuf(); const std::string s = ss.str();
  std::vector<ptree *> st(1, &root); std::size_t i = 0;
  while (i < s.size()) {
    if (s[i] != '<') { const std::size_t j = s.find('<', i); std::string t = s.substr(i, j == std::string::npos ? j : j - i);
      const std::size_t a = t.find_first_not_of(" \t\r\n"), b = t.find_last_not_of(" \t\r\n");
      if (a != std::string::npos) st.back()->data() += t.substr(a, b - a + 1);
      i = j; continue; }
    if (s.compare(i, 4, "<!--") == 0) { i = s.find("-->", i); i = (i == std::string::npos) ? s.size() : i + 3; continue; }
    const std::size_t j = s.find('>', i); if (j == std::string::npos) break;
    std::string tag = s.substr(i + 1, j - i - 1); i = j + 1;
    if (tag.empty() || tag[0] == '?' || tag[0] == '!') continue;
    if (tag[0] == '/') { if (st.size() > 1) st.pop_back(); continue; }
    const bool selfClose = tag[tag.size() - 1] == '/'; if (selfClose) tag.erase(tag.size() - 1);
    const std::size_t sp = tag.find_first_of(" \t\r\n"); if (sp != std::string::npos) tag.erase(sp);
    ptree &c = st.back()->add_child_node(tag); if (!selfClose) st.push_back(&c);
  }
}
}
using xml_parser::read_xml;
} }
#endif
