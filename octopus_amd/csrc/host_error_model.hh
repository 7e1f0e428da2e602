// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - per-haplotype penalty vectors: kernels, host threads, named and file-read models (C ABI entries).
// ---------------------------------------------------------------------------------------------------------------
// per-haplotype penalty vectors (phmm_error_model.hpp): host threads, or one device lane per haplotype
// ---------------------------------------------------------------------------------------------------------------
struct PenaltyOut { int8_t* go; int8_t* ge; uint8_t* mf; int8_t* pf; uint8_t* mr; int8_t* pr; };
constexpr size_t kPenaltyLdsBytes = 64 * 1024;     // LDS a wave of k_penalty_vectors_wave may take: the 160 KB of a CU then hold two haplotypes

OCT_KERNEL(k_penalty_vectors)(const oct_phmm_error_model* model, const uint8_t* hbases, const uint32_t* hoff, uint32_t hap0, uint32_t hap1,
                              const uint8_t* sub_mask, uint32_t* workspace, size_t words_per_hap, PenaltyOut out, uint32_t* overflow)
{
    const uint32_t h = hap0 + hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (h >= hap1) return;
    const uint32_t o = hoff[h], n = hoff[h + 1] - o;
    uint32_t* w = workspace + (size_t)(h - hap0) * words_per_hap;
    const int rc = em::penalty_vectors(*model, hbases + o, n, sub_mask ? sub_mask + o : nullptr, w, 1, out.go + o, out.ge + o, out.mf + o, out.pf + o, out.mr + o, out.pr + o);
    if (rc != em::kOk) overflow[h] = 1;
}

// One wave per haplotype, everything but the six output vectors in LDS: the haplotype's bases and the flat workspace of
// phmm_error_model.hpp at its tight sizing. The wave's lanes share the parallel phases; lane 0 runs the sequential ones at LDS latency.
OCT_KERNEL(k_penalty_vectors_wave)(const oct_phmm_error_model* model, const uint8_t* hbases, const uint32_t* hoff, uint32_t n_haps,
                                   const uint8_t* sub_mask, uint32_t lds_words, PenaltyOut out, uint32_t* overflow, unsigned long long* prof)
{
    OCT_DYN_SMEM(lds_raw);
    uint32_t* w = (uint32_t*)lds_raw;
    const uint32_t h = hw::block_idx();
    if (h >= n_haps) return;
    em::Wave x;
    x.prof = prof;
    x.tick(0);
    const uint32_t o = hoff[h], n = hoff[h + 1] - o;
    const size_t need = em::workspace_words(n, 0);
    uint8_t* s = (uint8_t*)(w + need);
    if (need + (n + 3) / 4 > lds_words) { if (x.lane() == 0) overflow[h] = 1; return; }
    for (uint32_t i = x.lane(); i < n; i += 64) s[i] = hbases[o + i];
    x.sync();
    const int rc = em::penalty_vectors(x, *model, s, n, sub_mask ? sub_mask + o : nullptr, w, 0, out.go + o, out.ge + o, out.mf + o, out.pf + o, out.mr + o, out.pr + o);
    if (rc != em::kOk && x.lane() == 0) overflow[h] = 1;
}

namespace {

// one haplotype on the calling thread; the workspace grows until the run lists fit (pathological repeat structure only)
void host_penalty_vectors_one(const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* sub, std::vector<uint32_t>& w, PenaltyOut out, size_t o,
                              const em::CustomIndelModel* custom = nullptr)
{
    for (uint32_t grow = 1; ; grow *= 4) {
        const size_t need = em::workspace_words(n, grow);
        if (w.size() < need) w.resize(need);
        if (!custom) {
            if (em::penalty_vectors(m, s, n, sub, w.data(), grow, out.go + o, out.ge + o, out.mf + o, out.pf + o, out.mr + o, out.pr + o) == em::kOk) return;
        } else if (em::custom_indel_penalties(*custom, s, n, w.data(), grow, out.go + o, out.ge + o) == em::kOk) {     // the file's rows for the gaps, `m` for the SNV vectors only
            em::snv_priors(em::Seq {}, m, s, n, sub, w.data(), out.mf + o, out.pf + o, out.mr + o, out.pr + o);
            return;
        }
    }
}

void host_penalty_vectors(const oct_phmm_error_model& m, uint32_t n_haps, const uint8_t* bases, const uint32_t* off, const uint8_t* sub, PenaltyOut out,
                          const em::CustomIndelModel* custom = nullptr)
{
    static const unsigned kCores = std::thread::hardware_concurrency();     // (asked once, see host_parallel)
    unsigned T = kCores ? std::min(kCores, 16u) : 1;
    const uint32_t n_bases = n_haps ? off[n_haps] : 0;
    if (n_bases < 2000 || n_haps < 4) T = 1;                           // a thread start costs more than a few short haplotypes
    T = std::min<unsigned>(T, std::max<uint32_t>(1, n_haps / 2));
    std::atomic<uint32_t> next {0};
    auto work = [&] {
        std::vector<uint32_t> w;
        for (uint32_t h = next.fetch_add(1); h < n_haps; h = next.fetch_add(1))
            host_penalty_vectors_one(m, bases + off[h], off[h + 1] - off[h], sub ? sub + off[h] : nullptr, w, out, off[h], custom);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}

bool model_is_valid(const oct_phmm_error_model* m)       // every table entry a penalty in [0, 127]; table by table (the struct's padding bytes are the caller's)
{
    bool ok = true;
    auto table = [&](const int8_t* t, size_t n) { for (size_t i = 0; i < n; ++i) ok = ok && t[i] >= 0; };
    table(m->at_homopolymer_open, OCT_PHMM_INDEL_TABLE); table(m->cg_homopolymer_open, OCT_PHMM_INDEL_TABLE);
    table(m->dinucleotide_open, OCT_PHMM_INDEL_TABLE); table(m->trinucleotide_open, OCT_PHMM_INDEL_TABLE);
    table(m->homopolymer_extend, OCT_PHMM_INDEL_TABLE); table(m->dinucleotide_extend, OCT_PHMM_INDEL_TABLE); table(m->trinucleotide_extend, OCT_PHMM_INDEL_TABLE);
    for (int k = 0; k < 3; ++k) table(m->snv_caps[k], OCT_PHMM_SNV_TABLE);
    return ok;
}

} // namespace

extern "C" void oct_phmm_error_model_expand(int8_t* dst, uint32_t capacity, const int8_t* src, uint32_t n)
{
    if (!dst || !src || !n) return;
    for (uint32_t i = 0; i < capacity; ++i) dst[i] = src[i < n ? i : n - 1];
}

// The reference's built-in parameter sets (error_model_factory.cpp:220-517) as data: phmm_error_model_tables.hpp, generated from the reference's source by
// tools/make_error_model_tables.py. Names are matched the way the reference's operator>> does (:88-104, :158-182): capitalised, "PCR-FREE" also "PCRF".
namespace {
int library_by_name(const char* name)
{
    if (!name || !*name) return emt::kDefaultLibrary;
    std::string t(name); for (char& c : t) c = (char)toupper((unsigned char)c);          // utils::capitalise
    if (t == "PCR") return emt::pcr;
    if (t == "PCR-FREE" || t == "PCRF") return emt::pcr_free;
    if (t == "10X") return emt::tenx;
    if (t == "MDA") return emt::mda;
    return -1;                                                                            // UnknownLibraryPreparation
}
int sequencer_by_name(const char* name)
{
    if (!name || !*name) return emt::kDefaultSequencer;
    std::string t(name); for (char& c : t) c = (char)toupper((unsigned char)c);
    static const char* const names[emt::kSequencers] = {"HISEQ-2000", "HISEQ-2500", "HISEQ-4000", "X10", "NOVASEQ", "BGISEQ-500", "PACBIO", "PACBIOCCS"};
    for (int i = 0; i < emt::kSequencers; ++i) if (t == names[i]) return i;
    return -1;                                                                            // UnknownSequencer
}
int builtin_model(int lib, int seq, oct_phmm_error_model* m)
{
    if (!m || lib < 0 || seq < 0) return OCT_PHMM_EINVAL;
    if (emt::indel_open[lib][seq][0] < 0) return OCT_PHMM_EINVAL;                          // builtin_indel_models.at() throws: 10X / MDA have no PacBio entries (:366-473)
    memset(m, 0, sizeof(*m));
    auto put = [&](int8_t* dst, uint32_t cap, int row) { const emt::Row& r = emt::rows[row]; oct_phmm_error_model_expand(dst, cap, r.v, r.n); };
    put(m->at_homopolymer_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][0]); put(m->cg_homopolymer_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][1]);
    put(m->dinucleotide_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][2]); put(m->trinucleotide_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][3]);
    put(m->homopolymer_extend, OCT_PHMM_INDEL_TABLE, emt::extend[0]); put(m->dinucleotide_extend, OCT_PHMM_INDEL_TABLE, emt::extend[1]); put(m->trinucleotide_extend, OCT_PHMM_INDEL_TABLE, emt::extend[2]);
    for (int k = 0; k < 3; ++k) put(m->snv_caps[k], OCT_PHMM_SNV_TABLE, emt::snv_caps[lib][k]);
    m->use_snv_model = (seq == emt::pacbio || seq == emt::pacbio_ccs) ? 0 : 1;             // use_snv_error_model :480-483
    return OCT_PHMM_OK;
}
} // namespace

extern "C" void oct_phmm_error_model_default(oct_phmm_error_model* m)
{
    if (m) builtin_model(emt::kDefaultLibrary, emt::kDefaultSequencer, m);                 // default_model_config = {PCR-free, HiSeq-2500} (error_model_factory.hpp:26-28)
}

extern "C" int oct_phmm_error_model_by_name(const char* library_preparation, const char* sequencer, oct_phmm_error_model* m)
{
    return builtin_model(library_by_name(library_preparation), sequencer_by_name(sequencer), m);
}

extern "C" int oct_phmm_error_model_by_label(const char* label, oct_phmm_error_model* m)   // parse_model_config :628-644: "<library>[.<sequencer>]", either part may be empty
{
    if (!label) return OCT_PHMM_EINVAL;
    const std::string l(label);
    const size_t dot = l.find('.');
    const std::string lib = l.substr(0, dot), seq = dot == std::string::npos ? std::string() : l.substr(dot + 1);
    return builtin_model(library_by_name(lib.c_str()), sequencer_by_name(seq.c_str()), m);
}

extern "C" int oct_phmm_penalty_vectors(const oct_phmm_error_model* model, uint32_t n_haps, const char* bases, const uint32_t* offsets,
                                        const uint8_t* substitution_mask, int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd,
                                        int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev, oct_phmm_status* status)
{
    if (!model || (n_haps && (!bases || !offsets || !gap_open || !gap_extend || !snv_mask_fwd || !snv_prior_fwd || !snv_mask_rev || !snv_prior_rev)))
        return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (n_haps && !monotone(offsets, n_haps)) return fail(status, OCT_PHMM_EINVAL, "offsets not monotone");
    if (!model_is_valid(model)) return fail(status, OCT_PHMM_EINVAL, "negative penalty in the error model's tables");
    try {
        host_penalty_vectors(*model, n_haps, (const uint8_t*)bases, offsets, substitution_mask,
                             PenaltyOut {gap_open, gap_extend, (uint8_t*)snv_mask_fwd, snv_prior_fwd, (uint8_t*)snv_mask_rev, snv_prior_rev});
    } catch (const std::exception&) { return fail(status, OCT_PHMM_EHIP, "host allocation"); }
    return ok(status);
}

extern "C" int oct_phmm_set_error_model(oct_phmm_handle* h, const oct_phmm_error_model* model)
{
    if (!h) return OCT_PHMM_EINVAL;
    if (model && !model_is_valid(model)) return OCT_PHMM_EINVAL;
    h->has_model = model != nullptr;
    h->custom.reset();
    if (model) h->model = *model;
    if (h->d_model) { rt::set_device(h->cfg.device_id); rt::stream_sync(h->stream); h->pool.release(h->d_model); h->d_model = nullptr; }
    return OCT_PHMM_OK;
}

// ---- CustomRepeatBasedIndelErrorModel: the indel model of `--sequence-error-model <file>` (phmm_custom_error_model.h) -----------------------------------------
struct oct_phmm_custom_indel_model { std::shared_ptr<const em::CustomIndelModel> m; };

extern "C" int oct_phmm_custom_indel_model_parse(const char* text, size_t len, oct_phmm_custom_indel_model** out)
{
    if (!out || (len && !text)) return OCT_PHMM_EINVAL;
    *out = nullptr;
    try {
        auto m = std::make_shared<em::CustomIndelModel>();
        if (!em::parse_custom_indel_model(text, len, *m)) return OCT_PHMM_EINVAL;     // "Bad model" / MalformedErrorModelFile
        *out = new oct_phmm_custom_indel_model {std::move(m)};
    } catch (const std::exception&) { return OCT_PHMM_EHIP; }
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_custom_indel_model_create(const oct_phmm_motif_penalties* open, uint32_t n_open, int8_t default_open,
                                                  const oct_phmm_motif_penalties* extend, uint32_t n_extend, int32_t has_extend, int8_t default_extend,
                                                  oct_phmm_custom_indel_model** out)
{
    if (!out || (n_open && !open) || (n_extend && !extend) || (n_extend && !has_extend)) return OCT_PHMM_EINVAL;
    *out = nullptr;
    try {
        auto m = std::make_shared<em::CustomIndelModel>();
        auto put = [](em::CustomIndelModel::Map& rows, const oct_phmm_motif_penalties* r, uint32_t n) {
            for (uint32_t i = 0; i < n; ++i) {
                if (!r[i].motif || !r[i].motif_len || !r[i].penalties || !r[i].n_penalties) return false;
                rows.emplace(std::string(r[i].motif, r[i].motif_len), std::vector<int8_t>(r[i].penalties, r[i].penalties + r[i].n_penalties));
            }
            return true;
        };
        if (!put(m->open, open, n_open) || !put(m->extend, extend, n_extend)) return OCT_PHMM_EINVAL;
        m->has_extend = has_extend != 0; m->default_open = default_open; m->default_extend = default_extend;
        *out = new oct_phmm_custom_indel_model {std::move(m)};
    } catch (const std::exception&) { return OCT_PHMM_EHIP; }
    return OCT_PHMM_OK;
}

extern "C" void oct_phmm_custom_indel_model_destroy(oct_phmm_custom_indel_model* m) { delete m; }

extern "C" int oct_phmm_custom_indel_model_info(const oct_phmm_custom_indel_model* m, int8_t* default_open, int8_t* default_extend, uint32_t* n_open_rows, uint32_t* n_extend_rows, int32_t* has_extend)
{
    if (!m) return OCT_PHMM_EINVAL;
    if (default_open) *default_open = m->m->default_open;
    if (default_extend) *default_extend = m->m->default_extend;
    if (n_open_rows) *n_open_rows = (uint32_t)m->m->open.size();
    if (n_extend_rows) *n_extend_rows = (uint32_t)m->m->extend.size();
    if (has_extend) *has_extend = m->m->has_extend ? 1 : 0;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_custom_penalty_vectors(const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv, uint32_t n_haps, const char* bases, const uint32_t* offsets,
                                               const uint8_t* substitution_mask, int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd,
                                               int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev, oct_phmm_status* status)
{
    if (!indel || (n_haps && (!bases || !offsets || !gap_open || !gap_extend || !snv_mask_fwd || !snv_prior_fwd || !snv_mask_rev || !snv_prior_rev)))
        return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (n_haps && !monotone(offsets, n_haps)) return fail(status, OCT_PHMM_EINVAL, "offsets not monotone");
    oct_phmm_error_model dflt;
    if (!snv) { oct_phmm_error_model_default(&dflt); snv = &dflt; }                    // make_snv_error_model(default_model_config), error_model_factory.cpp:587
    if (!model_is_valid(snv)) return fail(status, OCT_PHMM_EINVAL, "negative penalty in the error model's tables");
    try {
        host_penalty_vectors(*snv, n_haps, (const uint8_t*)bases, offsets, substitution_mask,
                             PenaltyOut {gap_open, gap_extend, (uint8_t*)snv_mask_fwd, snv_prior_fwd, (uint8_t*)snv_mask_rev, snv_prior_rev}, indel->m.get());
    } catch (const std::exception&) { return fail(status, OCT_PHMM_EHIP, "host allocation"); }
    return ok(status);
}

extern "C" int oct_phmm_set_custom_error_model(oct_phmm_handle* h, const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv)
{
    if (!h || !indel) return OCT_PHMM_EINVAL;
    oct_phmm_error_model dflt;
    if (!snv) { oct_phmm_error_model_default(&dflt); snv = &dflt; }
    const int rc = oct_phmm_set_error_model(h, snv);
    if (rc != OCT_PHMM_OK) return rc;
    h->custom = indel->m;                                  // shared: the caller may destroy its model object
    return OCT_PHMM_OK;
}

