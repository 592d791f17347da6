// oracle/_ref: stands in for <ros/ros.h> - the hot-path sources use nothing of ROS but its assertion macro
#pragma once
#include <cassert>
#include <cstdio>
#define ROS_ASSERT(cond) assert(cond)
#define ROS_INFO(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_ERROR(...) do { } while (0)

// the threads' run() loops (Track, LocalMapper, GlobalMapper, Localizer): never entered in oracle/_ref
namespace ros {
inline bool ok() { return false; }
inline void shutdown() {}
class NodeHandle {};      // members of MapPublish (the publisher itself is cut off)
class Publisher {};
class Rate {
public:
    explicit Rate(double) {}
    void sleep() {}
};
}  // namespace ros
