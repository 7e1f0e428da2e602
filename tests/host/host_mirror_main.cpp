// Test driver for octopus_amd/host/haplotype_likelihood_array.hpp: reads a scenario on stdin, runs populate() through the C++ mirror,
// prints the matrix via the reference-style accessors. Linked against liboct_phmm.so (GPU) or tests/sim/libphmm_sim.so (CPU simulator).
//   scenario: "cfg band use_mapq flank lhs rhs templates" / "H n" then n x "begin seq" / "S n_samples" then per sample "name n_rows",
//   per row "k" then k x "begin reverse mapq seq quals(comma separated)"
#include <cstdio>
#include <iostream>
#include <sstream>
#include "../../octopus_amd/host/haplotype_likelihood_array.hpp"

using namespace octopus_amd;

int main()
{
    int band, use_mapq, has_flank, templates; unsigned lhs, rhs;
    std::string tok;
    std::cin >> tok >> band >> use_mapq >> has_flank >> lhs >> rhs >> templates;
    std::size_t nh; std::cin >> tok >> nh;
    std::vector<Haplotype> haps(nh);
    for (auto& h : haps) std::cin >> h.begin_ >> h.sequence_;
    std::size_t ns; std::cin >> tok >> ns;
    TemplateMap tm; ReadMap rm; std::vector<SampleName> names;
    for (std::size_t s = 0; s < ns; ++s) {
        std::string name; std::size_t nrows; std::cin >> name >> nrows; names.push_back(name);
        tm.emplace_back(name, std::vector<AlignedTemplate> {}); rm.emplace_back(name, std::vector<AlignedRead> {});
        for (std::size_t r = 0; r < nrows; ++r) {
            std::size_t k; std::cin >> k; AlignedTemplate t;
            for (std::size_t j = 0; j < k; ++j) {
                AlignedRead rd; int rev, mq; std::string q; std::cin >> rd.begin_ >> rev >> mq >> rd.sequence_ >> q;
                rd.reverse_ = rev; rd.mapping_quality_ = static_cast<std::uint8_t>(mq);
                std::stringstream ss {q}; std::string item; while (std::getline(ss, item, ',')) rd.base_qualities_.push_back(static_cast<std::uint8_t>(std::stoi(item)));
                t.push_back(rd); rm.back().second.push_back(rd);
            }
            tm.back().second.push_back(t);
        }
    }
    HaplotypeLikelihoodModel::Config cfg; cfg.max_indel_error = band; cfg.use_mapping_quality = use_mapq;
    try {
        HaplotypeLikelihoodModel model {cfg};
        std::printf("pad_requirement %u\n", model.pad_requirement());
        HaplotypeLikelihoodArray arr {model, names};
        HaplotypeLikelihoodArray::FlankState fs {lhs, rhs};
        if (templates) arr.populate(tm, haps, has_flank ? &fs : nullptr); else arr.populate(rm, haps, has_flank ? &fs : nullptr);
        for (std::size_t h = 0; h < haps.size(); ++h)
            for (const auto& n : names) { std::printf("L %zu %s", h, n.c_str()); for (double v : arr(n, haps[h])) std::printf(" %.17g", v); std::printf("\n"); }
        arr.prime(names.front());
        std::printf("primed %zu contains %d\n", arr.num_likelihoods(), arr.contains(haps.front()) ? 1 : 0);
        {   // genotype read-out on the resident matrix (primed sample) and realignment of the first sample's reads against haplotype 0
            ConstantMixtureGenotypeLikelihoodModel gm {arr};
            std::vector<ConstantMixtureGenotypeLikelihoodModel::Genotype> gts;
            for (std::size_t a = 0; a < haps.size(); ++a) for (std::size_t b = a; b < haps.size(); ++b) gts.push_back({a, b});
            const auto gl = gm.evaluate(gts);
            for (std::size_t i = 0; i < gts.size(); ++i) std::printf("G %zu %zu %.17g\n", gts[i][0], gts[i][1], gl[i]);
            if (haps.size() >= 3) std::printf("G3 %.17g\n", gm.evaluate({0, 1, 2}));
            const auto alns = model.align(rm.front().second, haps.front(), has_flank ? &fs : nullptr);
            for (std::size_t i = 0; i < alns.size(); ++i) std::printf("A %zu %zu %s %.17g\n", i, alns[i].mapping_position, alns[i].cigar.c_str(), alns[i].likelihood);
        }
        const auto merged = arr.merge_samples(names);
        std::printf("merged %zu\n", merged[0].size());
        {   // extract_sample (ref hpp:85) and copy semantics: a copy has its own handle and no device matrix, and can populate on its own
            const auto ex = arr.extract_sample(names.back());
            bool same = ex.size() == haps.size();
            for (std::size_t h = 0; h < haps.size() && same; ++h) same = ex.at(haps[h]) == arr(names.back(), haps[h]);
            HaplotypeLikelihoodArray copy {arr};
            const bool distinct = copy.handle() != arr.handle() && copy.resident_batch() == nullptr && arr.resident_batch() != nullptr;
            bool equal = true;
            if (templates) copy.populate(tm, haps, has_flank ? &fs : nullptr); else copy.populate(rm, haps, has_flank ? &fs : nullptr);
            for (std::size_t h = 0; h < haps.size(); ++h) for (const auto& n : names) equal = equal && copy(n, haps[h]) == arr(n, haps[h]);
            std::printf("extract %d copy %d %d\n", same ? 1 : 0, distinct ? 1 : 0, equal ? 1 : 0);
        }
        if (haps.size() > 1) {
            arr.reset({haps.back()}); std::printf("reset %zu %d\n", arr.haplotypes().size(), arr.contains(haps.front()) ? 1 : 0);
            arr.prime(names.front());                           // no device matrix any more: the mirror says so instead of computing on the host
            try { ConstantMixtureGenotypeLikelihoodModel {arr}.evaluate({0, 0}); std::printf("Ghost computed\n"); }
            catch (const HaplotypeLikelihoodModel::DeviceError&) { std::printf("Ghost refused\n"); }
        }
    } catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) {
        std::printf("ShortHaplotypeError %zu %zu\n", e.haplotype_index(), e.required_extension());
    } catch (const HaplotypeLikelihoodModel::TooLargeBandSizeError& e) {
        std::printf("TooLargeBandSizeError %d %d\n", e.requested(), e.max());
    }
    return 0;
}
