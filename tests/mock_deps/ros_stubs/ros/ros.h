#pragma once
// declaration-only stand-in (tests/mock_deps/ros_stubs/README.md)
#include <map>
#include <string>
#include <vector>
#include <boost_shared_ptr_stub.h>
#include "std_msgs/Header.h"
namespace XmlRpc {
class XmlRpcValue {
 public:
  enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString, TypeDateTime, TypeBase64, TypeArray, TypeStruct };
  Type getType() const;
  int size() const;
  XmlRpcValue& operator[](int i);
  XmlRpcValue& operator[](const char* key);
  operator double&();
  operator int&();
};
}  // namespace XmlRpc
namespace ros {
class Publisher {
 public:
  template <class M> void publish(const M& m) const;
  unsigned getNumSubscribers() const;
};
class Subscriber {};
class NodeHandle {
 public:
  NodeHandle(const std::string& ns = std::string());
  template <class M> Publisher advertise(const std::string& topic, unsigned queue);
  template <class M, class T>
  Subscriber subscribe(const std::string& topic, unsigned queue, void (T::*fp)(const boost::shared_ptr<M const>&), T* obj);
  bool getParam(const std::string& key, XmlRpc::XmlRpcValue& v) const;
};
namespace this_node { const std::string& getName(); }
void init(int& argc, char** argv, const std::string& name);
void spin();
void shutdown();
}  // namespace ros
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
