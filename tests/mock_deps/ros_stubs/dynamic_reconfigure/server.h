#pragma once
#include <cstdint>
#include <functional>
namespace dynamic_reconfigure {
template <class ConfigType>
class Server {
 public:
  typedef std::function<void(ConfigType&, uint32_t level)> CallbackType;
  Server();
  void setCallback(const CallbackType& callback);
};
}  // namespace dynamic_reconfigure
