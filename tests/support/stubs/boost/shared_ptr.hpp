// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_SHARED_PTR
#define RFS_STUB_BOOST_SHARED_PTR
#include <memory>
namespace boost {
using std::shared_ptr; using std::weak_ptr; using std::enable_shared_from_this; using std::make_shared;
using std::dynamic_pointer_cast; using std::static_pointer_cast;
}
#endif
