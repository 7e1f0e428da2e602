// TEST INFRASTRUCTURE ONLY: stand-in for basics/aligned_template.hpp: a template is a sequence of reads (see aligned_read.hpp shim).
#pragma once
#include <vector>
#include "basics/aligned_read.hpp"
namespace octopus { using AlignedTemplate = std::vector<AlignedRead>; }
