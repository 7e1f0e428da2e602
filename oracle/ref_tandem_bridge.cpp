// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's vendored tandem-repeat library, compiled in place from
// /root/reference/lib/tandem (tandem.cpp + libdivsufsort; nothing is copied into this repo). It pins the oracle's
// restatement of the repeat extraction the error models depend on:
//   tandem::extract_exact_tandem_repeats(str, min_period, max_period)      lib/tandem/tandem.hpp:497-514
// used by RepeatBasedIndelErrorModel (periods 1-5, core/models/error/repeat_based_indel_error_model.cpp:15-18) and
// BasicRepeatBasedSNVErrorModel (periods 1-3, core/models/error/repeat_based_snv_error_model.cpp:43-46).
#include <cstdint>
#include <string>
#include <vector>
#include "tandem/tandem.hpp"

extern "C" int ref_tandem_repeats(const char* str, uint32_t n, uint32_t min_period, uint32_t max_period,
                                  uint32_t* out_pos_len_period, uint32_t capacity)
{
    const std::string s(str, str + n);
    const auto repeats = tandem::extract_exact_tandem_repeats(s, min_period, max_period);
    uint32_t k = 0;
    for (const auto& r : repeats) {
        if (k < capacity) { out_pos_len_period[3 * k] = r.pos; out_pos_len_period[3 * k + 1] = r.length; out_pos_len_period[3 * k + 2] = r.period; }
        ++k;
    }
    return (int)k;
}
