// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_LEXICAL_CAST
#define RFS_STUB_BOOST_LEXICAL_CAST
#include <sstream>
#include <stdexcept>
#include <string>
namespace boost {
struct bad_lexical_cast : std::runtime_error { bad_lexical_cast() : std::runtime_error("bad lexical cast") {} };
template <class To, class From> To lexical_cast(const From &v) {
  std::stringstream ss; ss.precision(17); To out;
  if (!(ss << v) || !(ss >> out)) throw bad_lexical_cast();
  return out;
}
template <> inline std::string lexical_cast<std::string, std::string>(const std::string &v) { return v; }
}
#endif
