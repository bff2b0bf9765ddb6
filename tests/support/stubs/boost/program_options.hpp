// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_PROGRAM_OPTIONS
#define RFS_STUB_BOOST_PROGRAM_OPTIONS
#include <map>
#include <memory>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace boost { namespace program_options {
struct value_base { virtual ~value_base() {} virtual void assign(const std::string &s) = 0; virtual bool has_default() const = 0; virtual void apply_default() = 0; };
template <class T> struct typed_value : value_base {
  T *dst; T dflt; bool hasDflt; T cur;
  explicit typed_value(T *d) : dst(d), dflt(), hasDflt(false), cur() {}
  typed_value *default_value(const T &v) { dflt = v; hasDflt = true; return this; }
  void set(const T &v) { cur = v; if (dst) *dst = v; }
  void assign(const std::string &s) { T v; std::istringstream is(s); if (!(is >> v)) throw std::runtime_error("bad option value: " + s); set(v); }
  bool has_default() const { return hasDflt; }
  void apply_default() { set(dflt); }
};
template <> inline void typed_value<std::string>::assign(const std::string &s) { set(s); }
template <class T> typed_value<T> *value(T *dst = 0) { return new typed_value<T>(dst); }
struct option { std::string longName, shortName, help; std::shared_ptr<value_base> val; };
class options_description;
struct options_adder {
  options_description *d;
  options_adder &operator()(const char *name, const char *help);
  options_adder &operator()(const char *name, value_base *v, const char *help);
};
class options_description {
 public:
  std::string caption; std::vector<option> opts;
  explicit options_description(const std::string &c = "") : caption(c) {}
  options_adder add_options() { return options_adder{this}; }
  void add(const char *name, value_base *v, const char *help) {
    option o; std::string n(name); const std::size_t c = n.find(',');
    o.longName = n.substr(0, c); if (c != std::string::npos) o.shortName = n.substr(c + 1);
    o.help = help; o.val.reset(v); opts.push_back(o);
  }
};
inline options_adder &options_adder::operator()(const char *n, const char *h) { d->add(n, 0, h); return *this; }
inline options_adder &options_adder::operator()(const char *n, value_base *v, const char *h) { d->add(n, v, h); return *this; }
inline std::ostream &operator<<(std::ostream &os, const options_description &d) {
  os << d.caption << ":\n";
  for (const option &o : d.opts) os << "  " << (o.shortName.empty() ? "" : "-" + o.shortName + " [ ") << "--" << o.longName << (o.shortName.empty() ? "" : " ]") << "  " << o.help << "\n";
  return os;
}
struct variable_value {
  std::shared_ptr<value_base> v;
  template <class T> const T &as() const { const typed_value<T> *t = dynamic_cast<const typed_value<T> *>(v.get()); if (!t) throw std::runtime_error("bad any_cast"); return t->cur; }
};
class variables_map : public std::map<std::string, variable_value> {
 public:
  std::size_t count(const std::string &k) const { return std::map<std::string, variable_value>::count(k); }
  const variable_value &operator[](const std::string &k) const { return find(k)->second; }
  variable_value &slot(const std::string &k) { return std::map<std::string, variable_value>::operator[](k); }
};
struct parsed_options { const options_description *d; std::vector<std::pair<const option *, std::string> > hits; };
inline parsed_options parse_command_line(int argc, char **argv, const options_description &d) {
  parsed_options p; p.d = &d;
  for (int i = 1; i < argc; i++) {
    std::string a(argv[i]), val; bool inl = false; const option *hit = 0;
    if (a.size() > 2 && a.compare(0, 2, "--") == 0) { const std::size_t e = a.find('='); if (e != std::string::npos) { val = a.substr(e + 1); inl = true; a.erase(e); }
      for (const option &o : d.opts) if (o.longName == a.substr(2)) hit = &o; }
    else if (a.size() >= 2 && a[0] == '-') { if (a.size() > 2) { val = a.substr(2); inl = true; }
      for (const option &o : d.opts) if (o.shortName == a.substr(1, 1)) hit = &o; }
    if (!hit) throw std::runtime_error("unrecognised option " + a);
    if (hit->val && !inl) { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); val = argv[++i]; }
    p.hits.push_back(std::make_pair(hit, val));
  }
  return p;
}
inline void store(const parsed_options &p, variables_map &vm) {
  for (const auto &h : p.hits) { if (h.first->val) h.first->val->assign(h.second); vm.slot(h.first->longName).v = h.first->val; }
  for (const option &o : p.d->opts) if (o.val && o.val->has_default() && !vm.count(o.longName)) { o.val->apply_default(); vm.slot(o.longName).v = o.val; }
}
inline void notify(variables_map &) {}
} }
#endif
