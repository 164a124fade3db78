#pragma once
#include <memory>
namespace boost { template <class T> using shared_ptr = std::shared_ptr<T>; }
