// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_FILESYSTEM
#define RFS_STUB_BOOST_FILESYSTEM
#include <filesystem>
namespace boost { namespace filesystem {
using std::filesystem::path; using std::filesystem::create_directories; using std::filesystem::exists;
struct copy_option { enum enum_type { none, fail_if_exists = none, overwrite_if_exists }; };
inline void copy_file(const path &a, const path &b, copy_option::enum_type o = copy_option::none) {
  std::filesystem::copy_file(a, b, o == copy_option::overwrite_if_exists ? std::filesystem::copy_options::overwrite_existing : std::filesystem::copy_options::none);
}
} }
#endif
