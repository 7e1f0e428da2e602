// TEST INFRASTRUCTURE ONLY: the factory names core/models/haplotype_likelihood_model.cpp mentions (the real factory needs Boost.Filesystem
// and every model class); defined in oracle/ref_model_bridge.cpp, which always passes explicit models instead.
#pragma once
#include <memory>
#include <string>
#include "core/models/error/snv_error_model.hpp"
#include "core/models/error/indel_error_model.hpp"
namespace octopus {
struct ErrorModel { std::unique_ptr<IndelErrorModel> indel; std::unique_ptr<SnvErrorModel> snv; };
std::unique_ptr<SnvErrorModel> make_snv_error_model();
std::unique_ptr<IndelErrorModel> make_indel_error_model();
ErrorModel make_error_model(const std::string& label);
}
