// oracle/_ref: stands in for the g2o header of this name (see g2o_shim.hpp) - test infrastructure only
#pragma once
#include <g2o_shim.hpp>   // found through -I oracle/_shim
