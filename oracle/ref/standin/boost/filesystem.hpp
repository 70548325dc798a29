// Stand-in for the one boost::filesystem call VioManagerOptions.h makes (mask images: never enabled in the oracle/_ref build).
// TEST INFRASTRUCTURE ONLY.
#ifndef OV_REF_STANDIN_BOOST_FILESYSTEM_HPP
#define OV_REF_STANDIN_BOOST_FILESYSTEM_HPP
#include <string>
#include <sys/stat.h>
namespace boost {
namespace filesystem {
inline bool exists(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0;
}
} // namespace filesystem
} // namespace boost
#endif
