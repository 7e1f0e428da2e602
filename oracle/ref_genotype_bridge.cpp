// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own core/models/genotype/constant_mixture_genotype_likelihood_model.cpp (the per-ploidy /
// per-zygosity case analysis and maths::log_sum_exp from the real utils/maths.hpp), compiled in place on the stand-in array / genotype
// types of oracle/ref_shim. Pins oracle_genotype_likelihoods and, through it, the device read-out.
#include <cstdint>
#include <string>
#include <vector>
#include "core/models/genotype/constant_mixture_genotype_likelihood_model.hpp"

using namespace octopus;
using namespace octopus::model;

// columns: n_haps x n_rows (haplotype-major); genotypes: n_genotypes x ploidy sorted haplotype indices; indexed != 0 uses the
// Genotype<IndexedHaplotype<>> overloads (what the callers use), 0 the Genotype<Haplotype> ones
extern "C" void ref_genotype_likelihoods(const double* columns, uint32_t n_haps, uint32_t n_rows, const uint32_t* genotypes, uint32_t n_genotypes,
                                         uint32_t ploidy, int indexed, double* out)
{
    HaplotypeLikelihoodArray arr;
    for (uint32_t h = 0; h < n_haps; ++h) {
        arr.columns.emplace_back(columns + (size_t)h * n_rows, columns + (size_t)(h + 1) * n_rows);
        Haplotype hp; hp.begin_ = h; hp.sequence_ = std::to_string(h); arr.haplotypes.push_back(hp);
    }
    const ConstantMixtureGenotypeLikelihoodModel model {arr};
    for (uint32_t g = 0; g < n_genotypes; ++g) {
        if (indexed) {
            Genotype<IndexedHaplotype<>> gt;
            for (uint32_t j = 0; j < ploidy; ++j) gt.haplotypes_.emplace_back(genotypes[(size_t)g * ploidy + j]);
            out[g] = model.evaluate(gt);
        } else {
            Genotype<Haplotype> gt;
            for (uint32_t j = 0; j < ploidy; ++j) gt.haplotypes_.push_back(arr.haplotypes[genotypes[(size_t)g * ploidy + j]]);
            out[g] = model.evaluate(gt);
        }
    }
}
