// TEST INFRASTRUCTURE ONLY: stand-in for config/common.hpp. The model header includes it but uses nothing of it; the array header
// (core/models/haplotype_likelihood_array.hpp) needs the sample name and the per-sample read containers. The real containers
// (MappableMap = unordered_map, MappableFlatMultiSet) need the genomic-region machinery; populate() only iterates them.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "basics/aligned_read.hpp"
#include "basics/aligned_template.hpp"
namespace octopus {
using SampleName = std::string;
template <typename K, typename V> using MappableMap = std::map<K, V>;
using ReadMap = MappableMap<SampleName, std::vector<AlignedRead>>;
using TemplateMap = MappableMap<SampleName, std::vector<AlignedTemplate>>;
} // namespace octopus
