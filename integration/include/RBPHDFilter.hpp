/* Put THIS directory ahead of the reference's `include/` on the compiler's include path (`-I<repo>/integration/include
 * -I<repo>/include -I<reference>/include`): `#include "RBPHDFilter.hpp"` in src/rbphdslam2dSim.cpp:38 and
 * src/rbphdslam_VictoriaPark.cpp:40 then finds the GPU-backed class template instead of include/RBPHDFilter.hpp.
 * That, plus `-lrfsgpu`, is the whole integration. */
#include "../RBPHDFilter_rfsgpu.hpp"
