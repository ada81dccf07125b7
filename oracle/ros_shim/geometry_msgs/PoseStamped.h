#pragma once
#include <ros/ros.h>
namespace std_msgs { struct Header { ros::Time stamp; std::string frame_id; }; }
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct Vector3 { double x = 0, y = 0, z = 0; };
}
