// Stand-in for the Boost header of this name -- TEST SUPPORT ONLY (tests/test_reference_binding.py): the build image has no
// Boost; this maps what the unmodified reference drivers use onto the C++17 standard library so that the drop-in binding
// (integration/RBPHDFilter_rfsgpu.hpp) can be compiled and linked under them.  Not a parity oracle, not shipped.
#ifndef RFS_STUB_BOOST_TIMER
#define RFS_STUB_BOOST_TIMER
#include <chrono>
#include <ctime>
#include <string>
namespace boost { namespace timer {
typedef long long nanosecond_type;
struct cpu_times { nanosecond_type wall, user, system; };
class cpu_timer {
  bool run_; cpu_times acc_; std::chrono::steady_clock::time_point w0_; std::clock_t c0_;
  cpu_times lap() const {
    cpu_times t = acc_;
    if (run_) {
      t.wall += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0_).count();
      t.user += (nanosecond_type)((std::clock() - c0_) * (1e9 / CLOCKS_PER_SEC));
    }
    return t;
  }
 public:
  cpu_timer() { start(); }
  void start() { acc_ = cpu_times{0, 0, 0}; run_ = true; w0_ = std::chrono::steady_clock::now(); c0_ = std::clock(); }
  void stop() { if (run_) { acc_ = lap(); run_ = false; } }
  void resume() { if (!run_) { run_ = true; w0_ = std::chrono::steady_clock::now(); c0_ = std::clock(); } }
  bool is_stopped() const { return !run_; }
  cpu_times elapsed() const { return lap(); }
};
} }
#endif
