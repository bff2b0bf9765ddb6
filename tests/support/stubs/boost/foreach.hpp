// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef BOOST_FOREACH
#define BOOST_FOREACH(decl, range) for (decl : range)
#endif
