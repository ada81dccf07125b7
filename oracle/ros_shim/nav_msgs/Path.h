#pragma once
#include <geometry_msgs/PoseStamped.h>
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }
namespace std_msgs { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }
namespace visualization_msgs {
struct Marker {
    enum { ADD = 0, CUBE_LIST = 6 };
    std_msgs::Header header; std::string ns; int id = 0, type = 0, action = 0;
    geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color; std::vector<geometry_msgs::Point> points;
};
}
namespace sensor_msgs { struct PointCloud2 { std_msgs::Header header; }; }
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0; };
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };
template <class P> struct PointCloud {
    std::vector<P> points; unsigned width = 0, height = 0; bool is_dense = true; std_msgs::Header header;
    void push_back(const P &p) { points.push_back(p); }
};
template <class P> void fromROSMsg(const sensor_msgs::PointCloud2 &, PointCloud<P> &) {}
template <class P> void toROSMsg(const PointCloud<P> &, sensor_msgs::PointCloud2 &) {}
}
