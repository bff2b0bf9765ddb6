// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_RANDOM
#define RFS_STUB_BOOST_RANDOM
#include <random>
namespace boost {
typedef std::mt19937 mt19937;
template <class T = double> using normal_distribution = std::normal_distribution<T>;    // NOT boost's sample stream
template <class T = double> struct uniform_01 { template <class G> T operator()(G &g) { return std::generate_canonical<T, 53>(g); } };
template <class Engine, class Dist> class variate_generator {
  Engine e_; Dist d_;
 public:
  variate_generator(Engine e, Dist d) : e_(e), d_(d) {}
  typename Dist::result_type operator()() { return d_(e_); }
  Engine &engine() { return e_; }
  Dist &distribution() { return d_; }
};
namespace random { using boost::mt19937; }
}
#endif
