// oracle/_ref: stands in for the OpenCV header of this name (see ../cv_shim.hpp) - test infrastructure only.  Its presence is what
// include/se2lam_amd/conversions.h tests with __has_include, so that the cv:: overloads of the mirrors compile in the drop-in build.
#pragma once
#include "../cv_shim.hpp"
