// stand-in for ov_core/src/types/Type.h:35-135 (TEST INFRASTRUCTURE)
#pragma once
#include <Eigen/Eigen>
#include <memory>
namespace ov_type {
class Type {
public:
  Type(int size_) : _size(size_) {}
  virtual ~Type() {}
  virtual void set_local_id(int new_id) { _id = new_id; }
  int id() { return _id; }
  int size() { return _size; }
  virtual void update(const Eigen::VectorXd &dx) = 0;
  virtual const Eigen::MatrixXd &value() const { return _value; }
  virtual const Eigen::MatrixXd &fej() const { return _fej; }
  virtual void set_value(const Eigen::MatrixXd &new_value) { _value = new_value; }
  virtual void set_fej(const Eigen::MatrixXd &new_value) { _fej = new_value; }
protected:
  Eigen::MatrixXd _fej, _value;
  int _id = -1, _size = -1;
};
} // namespace ov_type
