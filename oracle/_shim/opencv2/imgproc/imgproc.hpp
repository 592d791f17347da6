// oracle/_ref: stands in for the OpenCV header of this name (see ../../cv_shim.hpp) - test infrastructure only
#pragma once
#include "../../cv_shim.hpp"
