// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_MULTI_ARRAY
#define RFS_STUB_BOOST_MULTI_ARRAY
#include <cstddef>
#include <vector>
namespace boost {
struct extent2 { std::size_t a, b; };
struct extent1 { std::size_t a; extent2 operator[](std::size_t b) const { return extent2{a, b}; } };
struct extent0 { extent1 operator[](std::size_t a) const { return extent1{a}; } };
static const extent0 extents = extent0();
template <class T, int N> class multi_array;
template <class T> class multi_array<T, 2> {
  std::vector<T> d_; std::size_t s_[2];
 public:
  multi_array() { s_[0] = s_[1] = 0; }
  explicit multi_array(const extent2 &e) : d_(e.a * e.b), s_{e.a, e.b} {}
  const std::size_t *shape() const { return s_; }
  T *operator[](std::size_t r) { return d_.data() + r * s_[1]; }
  const T *operator[](std::size_t r) const { return d_.data() + r * s_[1]; }
  void resize(const extent2 &e) {           // keeps the overlapping contents, as boost does
    std::vector<T> n(e.a * e.b);
    for (std::size_t r = 0; r < e.a && r < s_[0]; r++) for (std::size_t c = 0; c < e.b && c < s_[1]; c++) n[r * e.b + c] = d_[r * s_[1] + c];
    d_.swap(n); s_[0] = e.a; s_[1] = e.b;
  }
};
}
#endif
