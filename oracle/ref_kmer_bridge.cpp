// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own k-mer mapper header, compiled in place from /root/reference/src/utils/kmer_mapper.hpp
// (self-contained: standard headers only). Pins the oracle's restatement of
//   compute_kmer_hashes<6>, make_kmer_hash_table<6>, map_query_to_target        utils/kmer_mapper.hpp:57-159
// as HaplotypeLikelihoodArray::populate drives them (core/models/haplotype_likelihood_array.cpp:118-158, mapperKmerSize = 6).
#include <cstdint>
#include <string>
#include <vector>
#include "utils/kmer_mapper.hpp"

extern "C" int ref_kmer_map(const char* query, int query_len, const char* target, int target_len, int max_positions, uint32_t* out_positions)
{
    using namespace octopus;
    const std::string q(query, query + query_len), t(target, target + target_len);
    const auto query_hashes = compute_kmer_hashes<6>(q);
    const auto table = make_kmer_hash_table<6>(t);
    auto counts = init_mapping_counts(table);
    std::vector<std::size_t> result(static_cast<std::size_t>(max_positions) + 1);
    const auto last = map_query_to_target(query_hashes, table, counts, result.begin(), static_cast<std::size_t>(max_positions));
    const int n = static_cast<int>(last - result.begin());
    for (int i = 0; i < n; ++i) out_positions[i] = static_cast<uint32_t>(result[i]);
    return n;
}
