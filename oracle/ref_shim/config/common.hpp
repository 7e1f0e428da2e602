// TEST INFRASTRUCTURE ONLY: the model header includes config/common.hpp but uses nothing of it.
#pragma once
