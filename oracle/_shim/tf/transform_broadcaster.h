// oracle/_ref: stands in for <tf/transform_broadcaster.h> (a member of MapPublish)
#pragma once
namespace tf { struct TransformBroadcaster {}; }
