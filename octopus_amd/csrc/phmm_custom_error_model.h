// The reference's indel error model READ FROM A FILE (`--sequence-error-model <path>`): CustomRepeatBasedIndelErrorModel
// (core/models/error/custom_repeat_based_indel_error_model.cpp) built by make_error_model(path) (error_model_factory.cpp:572-590) from
// make_penalty_map's reading of the file (custom_repeat_based_indel_error_model.cpp:105-159). HOST code only (string-keyed look-ups; no kernel
// includes this file): the repeats come from the same lz_tandem_repeats / sort_by_length as the built-in models (phmm_error_model.hpp), the
// penalties from the file's rows instead of the period tables.
//
// A model file, as the reference reads it (restated; every habit kept because a drop-in must accept and refuse the same files):
//   * a line that starts with '#' is a comment; empty lines are skipped;
//   * any other line is  <motif>:<p0>,<p1>,...  - gap-OPEN penalties of a repeat of that motif by its number of periods (index length / period,
//     the last entry for longer repeats); a motif that ends in '+' (the '+' is dropped) gives gap-EXTENSION penalties instead;
//   * the motif is whatever stands between the line's start and the NEXT ':' of the file (not of the line); a line that starts with ':' , a motif
//     that is only "+", a missing ':' before the end of the file, a row without numbers, and any entry that is not [+-]?digits (a blank, a '\r',
//     an empty entry - a trailing comma FOLLOWED BY A NEWLINE, or two commas; a row whose last character, at the very end of the file, is a comma parses like the reference's
//     make_penalty_map, which stops at the end of input before it asks for another entry: "A:1," is read, "A:1,\n" is refused) or does not fit int8 refuse the whole file;
//   * the first row of a motif wins (unordered_map::emplace);
//   * a file without any open row is malformed (MalformedErrorModelFile, error_model_factory.cpp:577-579).
// Look-ups (:68-103): the repeat's own motif as it stands at the repeat's start; else the row of min(period, 10) letters 'N'; else the default:
// entry 0 of the FIRST row in the open map's iteration order (:38-40, :53-58) - an artefact of libstdc++'s unordered_map, reproduced here by
// building the same container with the same sequence of emplace calls (this library and the reference are compiled against the same libstdc++);
// extension penalty without any '+' row: the constructor's default 3 (custom_repeat_based_indel_error_model.hpp:28).
#pragma once
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "phmm_error_model.hpp"

namespace octphmm { namespace em {

struct CustomIndelModel {
    using Map = std::unordered_map<std::string, std::vector<int8_t>>;
    Map open, extend;
    bool has_extend = false;                                  // boost::optional<MotifPenaltyMap> gap_extend_penalties_
    int8_t default_open = 0, default_extend = 3;
    std::string n_motif[11];                                  // ns_: "", "N", "NN", ... ten 'N'
    CustomIndelModel() { for (size_t i = 0; i <= 10; ++i) n_motif[i].assign(i, 'N'); }

    static int8_t at(const std::vector<int8_t>& row, uint32_t i) { return i < row.size() ? row[i] : row.back(); }          // get_min_penalty :24-28
    // the two constructors' defaults :30-62 (call once the maps are complete)
    void take_defaults_from_first_rows()
    {
        if (!open.empty()) default_open = at(open.cbegin()->second, 0);
        if (has_extend) { default_extend = 0; if (!extend.empty()) default_extend = at(extend.cbegin()->second, 0); }
    }
    int8_t look_up(const Map& rows, const uint8_t* motif, uint32_t period, uint32_t length, int8_t otherwise, std::string& key) const   // get_open_penalty :68-80, get_extension_penalty :88-101
    {
        key.assign((const char*)motif, period);
        auto it = rows.find(key);
        if (it == rows.end()) {
            it = rows.find(n_motif[period < 10 ? period : 10]);
            if (it == rows.end()) return otherwise;
        }
        return at(it->second, length / period);
    }
    int8_t open_penalty(const uint8_t* motif, uint32_t period, uint32_t length, std::string& key) const { return look_up(open, motif, period, length, default_open, key); }
    int8_t extend_penalty(const uint8_t* motif, uint32_t period, uint32_t length, std::string& key) const
    {
        return has_extend ? look_up(extend, motif, period, length, default_extend, key) : default_extend;
    }
};

// boost::lexical_cast<int> of a token followed by boost::numeric_cast<int8_t> (:138): an optional sign, then digits and nothing else; the value must fit int (else
// bad_lexical_cast) and then int8 (else bad_numeric_cast) - either way the reference leaves make_penalty_map by exception
inline bool parse_penalty(const char* p, const char* end, int8_t* out)
{
    if (p == end) return false;
    const bool neg = *p == '-';
    if (*p == '-' || *p == '+') ++p;
    if (p == end) return false;
    long long v = 0;
    for (; p != end; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
        if (v > 0x80000000ll) return false;                   // beyond int either way
    }
    if (neg) v = -v;
    if (v < -128 || v > 127) return false;
    *out = (int8_t)v;
    return true;
}

// make_penalty_map :105-159 + the factory's check :577-579 + the constructors' defaults. false: the reference throws on this text.
inline bool parse_custom_indel_model(const char* text, size_t len, CustomIndelModel& m)
{
    const char* const end = text + len;
    auto find = [&](const char* from, char c) { while (from != end && *from != c) ++from; return from; };
    bool any_open = false;
    for (const char* p = text; p != end;) {
        if (*p == '#') { p = find(p, '\n'); if (p != end) ++p; continue; }
        if (*p == '\n') { ++p; continue; }
        const char* colon = find(p, ':');
        if (colon == end || colon == p) return false;
        std::string motif(p, colon);
        const bool extension_row = motif.back() == '+';
        if (extension_row) { motif.pop_back(); if (motif.empty()) return false; m.has_extend = true; }
        else any_open = true;
        std::vector<int8_t> row;
        p = colon + 1;
        while (p != end) {                                    // entries up to and including the line's '\n' (or the end of the text)
            const char* q = p;
            while (q != end && *q != ',' && *q != '\n') ++q;
            int8_t v;
            if (!parse_penalty(p, q, &v)) return false;
            row.push_back(v);
            if (q == end) { p = end; break; }
            p = q + 1;
            if (*q == '\n') break;
        }
        if (row.empty()) return false;
        (extension_row ? m.extend : m.open).emplace(std::move(motif), std::move(row));
    }
    if (!any_open) return false;
    m.take_defaults_from_first_rows();
    return true;
}

// RepeatBasedIndelErrorModel::do_set_penalties, vector overload (repeat_based_indel_error_model.cpp:67-83), with the custom model's look-ups
inline int custom_indel_penalties(const CustomIndelModel& m, const uint8_t* s, uint32_t n, uint32_t* w, uint32_t grow, int8_t* gap_open, int8_t* gap_extend)
{
    for (uint32_t i = 0; i < n; ++i) { gap_open[i] = m.default_open; gap_extend[i] = m.default_extend; }
    if (n == 0) return kOk;
    const Layout lay = layout(n, grow);
    Repeat* runs = (Repeat*)(w + lay.runs);
    uint32_t nr = 0;
    const int rc = lz_tandem_repeats(Seq {}, s, n, 1, 5, w, grow, runs, &nr);
    if (rc != kOk) return rc;
    sort_by_length(runs, nr);
    std::string key;
    for (uint32_t i = 0; i < nr; ++i) {
        const Repeat& q = runs[i];
        const int8_t open = m.open_penalty(s + q.pos, q.period, q.length, key), extend = m.extend_penalty(s + q.pos, q.period, q.length, key);
        for (uint32_t k = 0; k < q.length; ++k) {
            if (open < gap_open[q.pos + k]) gap_open[q.pos + k] = open;
            gap_extend[q.pos + k] = extend;
        }
    }
    return kOk;
}

}} // namespace octphmm::em
