// oracle/_ref: stands in for <visualization_msgs/Marker.h> - MapPublish.h holds ten of them as members; nothing is published
#pragma once
namespace visualization_msgs { struct Marker {}; }
