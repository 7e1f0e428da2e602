// TEST INFRASTRUCTURE ONLY: stand-in for the reference's basics/aligned_read.hpp (which pulls in HTSlib-facing types) exposing exactly the
// members core/models/haplotype_likelihood_model.cpp touches; shadows the real header when that file is compiled in place for oracle/_ref.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
namespace octopus {
class AlignedRead
{
public:
    using NucleotideSequence = std::string;
    using MappingQuality = std::uint8_t;
    using BaseQuality = std::uint8_t;
    using BaseQualityVector = std::vector<BaseQuality>;
    NucleotideSequence sequence_; BaseQualityVector base_qualities_; MappingQuality mapping_quality_ = 0; bool reverse_ = false; std::int64_t begin_ = 0;
    const NucleotideSequence& sequence() const noexcept { return sequence_; }
    const BaseQualityVector& base_qualities() const noexcept { return base_qualities_; }
    MappingQuality mapping_quality() const noexcept { return mapping_quality_; }
    bool is_marked_reverse_mapped() const noexcept { return reverse_; }
    // the realignment seam (core/tools/read_realigner.cpp:97-104): where the read lies now and how (one word per operation: length << 8 | the flag's character)
    std::int64_t realigned_end_ = 0; std::vector<std::uint32_t> realigned_cigar_;
    template <typename Region, typename Cigar> void realign(const Region& region, Cigar cigar)
    {
        begin_ = static_cast<std::int64_t>(region.begin); realigned_end_ = static_cast<std::int64_t>(region.end);
        realigned_cigar_.clear();
        for (const auto& op : cigar) realigned_cigar_.push_back(static_cast<std::uint32_t>(op.size()) << 8 | static_cast<unsigned char>(op.flag()));
    }
    std::string name() const { return {}; }                                        // debug printers of haplotype_likelihood_array.hpp only
    std::string cigar() const { return {}; }
};
inline std::int64_t mapped_region(const AlignedRead& r) noexcept { return r.begin_; }
inline std::size_t sequence_size(const AlignedRead& r) noexcept { return r.sequence_.size(); }
} // namespace octopus
