// oracle/_ref: stands in for <ros/ros.h> - the hot-path sources use nothing of ROS but its assertion macro
#pragma once
#include <cassert>
#include <cstdio>
#define ROS_ASSERT(cond) assert(cond)
#define ROS_INFO(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_ERROR(...) do { } while (0)
