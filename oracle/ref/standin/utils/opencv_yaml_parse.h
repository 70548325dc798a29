// Shadows the reference's ov_core/src/utils/opencv_yaml_parse.h (cv::FileStorage + boost::filesystem + optional ROS) in the
// oracle/_ref build.  TEST INFRASTRUCTURE ONLY.  The option structs (`UpdaterOptions`, `FeatureInitializerOptions`,
// `StateOptions`) only touch the parser when one is handed to their print(); oracle/ref/ref_driver.cpp fills the option
// structs field by field and never constructs a parser, so every method here is a no-op that leaves the caller's default.
#ifndef OPENCV_YAML_PARSER_H
#define OPENCV_YAML_PARSER_H
#include <Eigen/Eigen>
#include <boost/filesystem.hpp>
#include <opencv2/opencv.hpp>
#include <memory>
#include <string>
#include <vector>
#include "utils/colors.h"
#include "utils/print.h"
#include "utils/quat_ops.h"
namespace ov_core {
class YamlParser {
public:
  explicit YamlParser(const std::string &config_path, bool = true) : config_path_(config_path) {}
  std::string get_config_folder() { return config_path_.substr(0, config_path_.find_last_of('/')) + "/"; }
  bool successful() const { return true; }
  template <class T> void parse_config(const std::string &, T &, bool = true) {}
  template <class T> void parse_external(const std::string &, const std::string &, const std::string &, T &, bool = true) {}

private:
  std::string config_path_;
};
}  // namespace ov_core
#endif
