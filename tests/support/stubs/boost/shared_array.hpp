// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_SHARED_ARRAY
#define RFS_STUB_BOOST_SHARED_ARRAY
#include <cstddef>
#include <memory>
namespace boost {
template <class T> class shared_array {
  std::shared_ptr<T> p_;
 public:
  shared_array() {}
  explicit shared_array(T *p) : p_(p, std::default_delete<T[]>()) {}
  void reset(T *p = 0) { p_.reset(p, std::default_delete<T[]>()); }
  T &operator[](std::ptrdiff_t i) const { return p_.get()[i]; }
  T *get() const { return p_.get(); }
};
}
#endif
