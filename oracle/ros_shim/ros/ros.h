// TEST INFRASTRUCTURE.  Minimal stand-ins for the ROS / PCL / octomap types that the reference's path_searching headers name
// (jps_planner.h, map_util.h), so that JPSPlanner<3> and MapUtil<3> compile unmodified, where they lie, into oracle/_ref/libref_jpsplanner.so.
// Nothing here does anything: publishers swallow messages, parameters keep their defaults, time is zero.  Not ROS.
#pragma once
#include <cstdio>
#include <string>
#include <vector>
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
namespace ros {
struct Time { double t = 0.0; static Time now() { return Time(); } double toSec() const { return t; } };
struct Duration { explicit Duration(double = 0.0) {} };
struct TimerEvent {};
struct Timer {};
struct Publisher { template <class M> void publish(const M &) const {} };
struct Subscriber {};
struct NodeHandle {
    template <class T, class D> void param(const std::string &, T &v, const D &d) const { v = T(d); }      // every parameter keeps its default
    template <class M> Publisher advertise(const std::string &, int) { return Publisher(); }
    template <class C, class Ev> Timer createTimer(Duration, void (C::*)(const Ev &), C *) { return Timer(); }
    template <class C, class M> Subscriber subscribe(const std::string &, int, void (C::*)(const M &), C *) { return Subscriber(); }
};
} // namespace ros
