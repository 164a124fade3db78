#pragma once
#include <string>
namespace ros { struct Time { double toSec() const; }; }
namespace std_msgs { struct Header { ros::Time stamp; std::string frame_id; unsigned seq; }; }
