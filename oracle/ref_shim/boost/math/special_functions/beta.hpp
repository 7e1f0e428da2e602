// TEST INFRASTRUCTURE ONLY: see ../stubs.hpp
#pragma once
#include "boost/math/stubs.hpp"
