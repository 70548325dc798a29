// Stand-in for the two boost::posix_time calls the reference's updaters use for their debug timers.  TEST INFRASTRUCTURE ONLY.
#ifndef OV_REF_STANDIN_BOOST_POSIX_TIME_HPP
#define OV_REF_STANDIN_BOOST_POSIX_TIME_HPP
#include <chrono>
namespace boost {
namespace posix_time {
struct time_duration {
  long long us;
  long long total_microseconds() const { return us; }
};
struct ptime {
  std::chrono::steady_clock::time_point t;
};
inline time_duration operator-(const ptime &a, const ptime &b) {
  return time_duration{std::chrono::duration_cast<std::chrono::microseconds>(a.t - b.t).count()};
}
struct microsec_clock {
  static ptime local_time() { return ptime{std::chrono::steady_clock::now()}; }
};
}  // namespace posix_time
}  // namespace boost
#endif
