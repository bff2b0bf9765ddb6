// xml_cfg.hpp -- tiny XML reader for the reference's cfg/*.xml files: flattens <a><b>v</b></a> into {"a.b": "v"}; repeated
// leaves (<Pd><value>..</value><value>..</value></Pd>) are kept in order in `lists`; comments are skipped.  Stands in for the
// boost::property_tree calls of the reference drivers (pt.get<T>("config.x.y", default), pt.get_child).
#pragma once
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <map>
#include <string>
#include <vector>

namespace rfs_amd {

struct Cfg {
  std::map<std::string, std::string> kv;
  std::map<std::string, std::vector<std::string>> lists;
  bool has(const std::string &k) const { return kv.count(k) != 0; }
  double d(const std::string &k, double def) const { auto it = kv.find(k); return it == kv.end() ? def : std::atof(it->second.c_str()); }
  int i(const std::string &k, int def) const { auto it = kv.find(k); return it == kv.end() ? def : std::atoi(it->second.c_str()); }
  std::string s(const std::string &k, const std::string &def) const { auto it = kv.find(k); return it == kv.end() ? def : it->second; }
  std::vector<double> dl(const std::string &k) const {
    std::vector<double> out;
    auto it = lists.find(k);
    if (it != lists.end())
      for (auto &v : it->second) out.push_back(std::atof(v.c_str()));
    return out;
  }
};

inline Cfg read_xml_cfg(const std::string &fn) {
  Cfg c;
  std::ifstream in(fn);
  if (!in) return c;
  std::string s((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<std::string> stack;
  size_t p = 0;
  std::string text;
  while (p < s.size()) {
    if (s.compare(p, 4, "<!--") == 0) { size_t e = s.find("-->", p); p = (e == std::string::npos) ? s.size() : e + 3; continue; }
    if (s[p] == '<') {
      size_t e = s.find('>', p);
      if (e == std::string::npos) break;
      std::string tag = s.substr(p + 1, e - p - 1);
      if (!tag.empty() && tag[0] == '/') {
        std::string path;
        for (auto &t : stack) path += (path.empty() ? "" : ".") + t;
        size_t a = text.find_first_not_of(" \t\r\n"), b = text.find_last_not_of(" \t\r\n");
        if (a != std::string::npos) {
          const std::string v = text.substr(a, b - a + 1);
          c.kv[path] = v;
          c.lists[path].push_back(v);
        }
        if (!stack.empty()) stack.pop_back();
      } else if (!tag.empty() && tag[0] != '?' && tag.back() != '/') {
        stack.push_back(tag.substr(0, tag.find_first_of(" \t")));
      }
      text.clear();
      p = e + 1;
    } else {
      text += s[p++];
    }
  }
  return c;
}

}  // namespace rfs_amd
