// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - status helpers, the packed upload image, input facts, kernel launchers by band / flavour, traceback scratch, run_dp_kind.
namespace {

int fail(oct_phmm_status* st, int code, const char* msg)
{
    if (st) {
        memset(st, 0, sizeof(*st));
        st->code = code;
        if (code == OCT_PHMM_EHIP) st->hip_error = rt::last_error_code;
        if (msg) snprintf(st->message, sizeof(st->message), "%s", msg);
    }
    return code;
}
int ok(oct_phmm_status* st) { if (st) { memset(st, 0, sizeof(*st)); } return OCT_PHMM_OK; }

int band_for(int max_indel_error)   // simd_pair_hmm_wrapper.hpp:219-241
{
    for (int b = 8; b <= 256; b *= 2) if (max_indel_error <= b) return b;
    return -1;
}

#define RT(expr) do { if (!(expr)) return fail(status, OCT_PHMM_EHIP, #expr); } while (0)

// All device memory of a batch is ONE pool block: `upload` / `dalloc` only record what is needed, `commit` allocates, fills in the
// pointers and sends every input array up in a single copy out of the handle's pinned staging buffer (each array keeps a zeroed
// 16-byte tail pad, as the kernels' vector loads expect).
// Run f(lo, hi) over [0, n) on a few host threads (memory-bound passes over a big batch's arrays); small n stays on the caller's thread.
template <class F> void host_parallel(size_t n, size_t grain, F&& f)
{
    static const unsigned kCores = std::thread::hardware_concurrency();     // (asked once: glibc reads /sys for it, ~15 us per call - five calls were a third of a region call's host time)
    unsigned T = kCores > 4 ? 4 : (kCores ? kCores : 1);
    if (n / grain < T) T = (unsigned)(n / grain);
    if (T <= 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th; th.reserve(T - 1);
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&f, n, t, T] { f(n * t / T, n * (t + 1) / T); });
    f((size_t)0, n / T);
    for (auto& x : th) x.join();
}

struct Packer {
    struct Item { const void* src; size_t bytes; size_t off; void** dst; };
    std::vector<Item> items;
    size_t in_bytes = 0, total = 0;
    static size_t aligned(size_t n) { return (n + 16 + 255) & ~(size_t)255; }
    template <class T> void upload(const T* host, size_t n, const T** dev) { items.push_back({host, n * sizeof(T), 0, (void**)dev}); }
    template <class T> void dalloc(T** dev, size_t n) { items.push_back({nullptr, n * sizeof(T), 0, (void**)dev}); }
    bool commit(oct_phmm_handle* h, oct_phmm_batch* b, rt::Stream s);
};
// Inputs up to this size are packed into the pinned staging buffer and copied in one piece; larger ones stream through its two halves.
// OCT_PHMM_STAGE_MAX_KB: test hook (small batches through the streaming path).
static size_t stage_max()
{
    long long kb; if (tune::number("OCT_PHMM_STAGE_MAX_KB", &kb) && kb >= 2) return (size_t)kb << 10;
    return (size_t)64 << 20;
}

constexpr size_t kHostMappedCopyMax = (size_t)1 << 20;     // inputs up to here go up through k_copy_from_host, results of up to kHostMappedOutMax values (one region's) come back through the epilogue's own stores
constexpr uint64_t kHostMappedOutMax = 12288;             // (the epilogue's stores over the host link: 4 us for one region's 58 KB, 30 us for four regions', 82 for eight - a DMA copy wins from two regions on)
bool Packer::commit(oct_phmm_handle* h, oct_phmm_batch* b, rt::Stream s)
{
    std::stable_partition(items.begin(), items.end(), [](const Item& it) { return it.src != nullptr; });   // inputs first, contiguous
    total = 0;
    for (auto& it : items) { it.off = total; total += aligned(it.bytes); if (it.src) in_bytes = total; }
    void* base = nullptr;
    if (!h->pool.alloc(&base, total)) return false;
    b->allocs.push_back(base);
    for (auto& it : items) *it.dst = (char*)base + it.off;
    if (!in_bytes) return true;
    const size_t kStageMax = stage_max();
    if (in_bytes <= kStageMax) {
        if (h->stage_bytes < in_bytes) {
            rt::host_pinned_free(h->stage); h->stage = nullptr; h->stage_bytes = 0;
            size_t want = (size_t)1 << 20; while (want < in_bytes) want <<= 1;
            if (!rt::host_pinned_malloc(&h->stage, want)) return false;
            h->stage_bytes = want;
        }
        // the image of the input arrays in the pinned buffer, copied by a few host threads once it is worth their start-up (one thread moves
        // ~10 GB/s: the 30 MB of a 100k x 128 batch took 3 ms of the call on one thread)
        size_t n_in = 0; while (n_in < items.size() && items[n_in].src) ++n_in;
        host_parallel(in_bytes, (size_t)2 << 20, [&](size_t lo, size_t hi) {
            for (size_t i = 0; i < n_in; ++i) {
                const Item& it = items[i];
                const size_t slot_end = it.off + aligned(it.bytes), a = std::max(lo, it.off), z = std::min(hi, slot_end);
                if (a >= z) continue;
                const size_t data_end = it.off + it.bytes;
                if (a < data_end) memcpy((char*)h->stage + a, (const char*)it.src + (a - it.off), std::min(z, data_end) - a);
                if (z > data_end) { const size_t p0 = std::max(a, data_end); memset((char*)h->stage + p0, 0, z - p0); }
            }
        });
        if (in_bytes <= kHostMappedCopyMax && tune::host_mapped()) {                                // region-sized: a copy kernel reads the pinned image itself
            const uint32_t n16 = (uint32_t)((in_bytes + 15) / 16);
            OCT_LAUNCH(k_copy_from_host, (n16 + 255) / 256, 256, 0, s, (uint4*)base, (const uint4*)h->stage, n16);
            return rt::launch_ok();
        }
        return rt::h2d(base, h->stage, in_bytes, s);
    }
    size_t n_in = 0; while (n_in < items.size() && items[n_in].src) ++n_in;
    {   // Big batch with arrays in page-locked caller memory (oct_phmm_host_alloc, hipHostMalloc, hipHostRegister): the DMA engine reads those arrays themselves; the
        // others (the library's own small tables, pageable caller arrays) go through the staging halves one by one
        std::vector<char> direct(n_in, 0); bool any = false;
        for (size_t i = 0; i < n_in; ++i) if (items[i].bytes >= tune::pinned_min_bytes((size_t)1 << 20) && rt::host_is_pinned(items[i].src, items[i].bytes)) { direct[i] = 1; any = true; }
        if (any) {
            if (h->stage_bytes < kStageMax) {
                rt::host_pinned_free(h->stage); h->stage = nullptr; h->stage_bytes = 0;
                if (!rt::host_pinned_malloc(&h->stage, kStageMax)) return false;
                h->stage_bytes = kStageMax;
            }
            const size_t half = (kStageMax / 2) & ~(size_t)255;
            rt::Event ev[2] {}; bool used[2] = {false, false};
            if (!h->get_event(&ev[0]) || !h->get_event(&ev[1])) return false;
            bool ok = true; int k = 0;
            for (size_t i = 0; i < n_in && ok; ++i) {
                const Item& it = items[i];
                const size_t slot = aligned(it.bytes);
                if (direct[i]) {
                    ok = rt::dev_memset((char*)base + it.off + it.bytes, 0, slot - it.bytes, s) && rt::h2d((char*)base + it.off, it.src, it.bytes, s);   // (kernels read up to 16 bytes past an array)
                    continue;
                }
                for (size_t pos = 0; pos < slot && ok; pos += half, k ^= 1) {
                    const size_t len = slot - pos < half ? slot - pos : half;
                    char* buf = (char*)h->stage + (size_t)k * half;
                    if (used[k]) ok = rt::event_sync(ev[k]);
                    host_parallel(len, (size_t)4 << 20, [&](size_t lo, size_t hi) {
                        const size_t a = pos + lo, z = pos + hi;                  // bytes [a, z) of the slot: payload, then zero padding
                        if (a < it.bytes) memcpy(buf + lo, (const char*)it.src + a, (z < it.bytes ? z : it.bytes) - a);
                        if (z > it.bytes) { const size_t p0 = a > it.bytes ? a : it.bytes; memset(buf + (p0 - pos), 0, z - p0); }
                    });
                    ok = ok && rt::h2d((char*)base + it.off + pos, buf, len, s) && rt::event_record(ev[k], s);
                    used[k] = true;
                }
            }
            for (int i = 0; i < 2; ++i) { if (used[i]) ok = rt::event_sync(ev[i]) && ok; h->put_event(ev[i]); }
            return ok;
        }
    }
    // Big batch: the device image [0, in_bytes) goes through the two halves of the pinned staging buffer. While the DMA drains one half
    // a few host threads fill the other (one thread copies at ~10 GB/s, a pageable hipMemcpy no faster; PCIe takes ~50 GB/s).
    if (h->stage_bytes < kStageMax) {
        rt::host_pinned_free(h->stage); h->stage = nullptr; h->stage_bytes = 0;
        if (!rt::host_pinned_malloc(&h->stage, kStageMax)) return false;
        h->stage_bytes = kStageMax;
    }
    const size_t half = (kStageMax / 2) & ~(size_t)255;
    rt::Event ev[2] {}; bool used[2] = {false, false};
    if (!h->get_event(&ev[0]) || !h->get_event(&ev[1])) return false;
    auto fill = [&](char* dst, size_t lo, size_t hi) {       // image of device bytes [lo, hi): item payloads, zero padding between them
        size_t i = (size_t)(std::upper_bound(items.begin(), items.begin() + n_in, lo, [](size_t v, const Item& it) { return v < it.off; }) - items.begin());
        i = i ? i - 1 : 0;
        for (size_t pos = lo; pos < hi; ) {
            const Item& it = items[i];
            const size_t end = i + 1 < n_in ? items[i + 1].off : in_bytes;      // this item's slot (payload + padding)
            const size_t stop = end < hi ? end : hi;
            if (pos < it.off + it.bytes) {
                const size_t n = (it.off + it.bytes < stop ? it.off + it.bytes : stop) - pos;
                memcpy(dst + (pos - lo), (const char*)it.src + (pos - it.off), n);
                pos += n;
            }
            if (pos < stop) { memset(dst + (pos - lo), 0, stop - pos); pos = stop; }
            if (pos >= end) ++i;
        }
    };
    bool ok = true; int k = 0;
    for (size_t lo = 0; lo < in_bytes && ok; lo += half, k ^= 1) {
        const size_t len = in_bytes - lo < half ? in_bytes - lo : half;
        char* buf = (char*)h->stage + (size_t)k * half;
        if (used[k]) ok = rt::event_sync(ev[k]);
        host_parallel(len, kStageMax >= ((size_t)32 << 20) ? (size_t)4 << 20 : 256, [&](size_t a, size_t z) { fill(buf + a, lo + a, lo + z); });
        ok = ok && rt::h2d((char*)base + lo, buf, len, s) && rt::event_record(ev[k], s);
        used[k] = true;
    }
    for (int i = 0; i < 2; ++i) { if (used[i]) ok = rt::event_sync(ev[i]) && ok; h->put_event(ev[i]); }   // the staging buffer is the handle's: drained before anyone reuses it
    return ok;
}

// Byte-set questions over the input arrays, eight bytes per step (the compiler left the byte loops scalar: 0.17 ms of a 16-region upload, the only thing that made a
// device-sized batch of 150 k pairs slower than a host-sized one). high bit of every byte of the result: clear where the byte of x equals c.
inline uint64_t swar_ne(uint64_t x, uint8_t c) { const uint64_t y = x ^ (0x0101010101010101ull * c); return ((y & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | y; }
bool any_byte_outside_acgt(const uint8_t* p, size_t n)
{
    uint64_t bad = 0; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t x; memcpy(&x, p + i, 8); bad |= swar_ne(x, 'A') & swar_ne(x, 'C') & swar_ne(x, 'G') & swar_ne(x, 'T'); }
    uint32_t tail = 0;
    for (; i < n; ++i) tail |= ((p[i] == 'A') | (p[i] == 'C') | (p[i] == 'G') | (p[i] == 'T')) ? 0u : 1u;
    return (bad & 0x8080808080808080ull) != 0 || tail != 0;
}
bool any_byte_equals(const uint8_t* p, size_t n, uint8_t c)
{
    uint64_t all_ne = ~0ull; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t x; memcpy(&x, p + i, 8); all_ne &= swar_ne(x, c); }
    uint32_t tail = 0;
    for (; i < n; ++i) tail |= p[i] == c ? 1u : 0u;
    return (~all_ne & 0x8080808080808080ull) != 0 || tail != 0;
}

// What an upload must know about EVERY byte of its input before it packs it: the contract's range checks (quality <= 127, penalties >= 0, no empty read), the bounds the
// FASTADD decision needs (largest per-read quality sum, largest gap penalties) and - for device-sized batches - whether any base is outside ACGT / any SNV mask byte '0'.
// upload_impl makes them itself (one pass per array, threaded from ~2 MB on); the region server's CALLERS make them for their own region before they queue - 64 threads that would
// otherwise sleep - and a device batch inherits the merge (facts_of_reads / facts_of_haps are what both run).
struct InputFacts {
    uint32_t q_or = 0, pen_or = 0, gomax = 0, gemax = 0, t_min = 0xffffffffu; uint64_t sum_q_max = 0;
    int dirty = -1;                                       // -1 not looked at, 0 every base ACGT and every SNV mask byte set, 1 not so
    bool have_haps = false;                               // the penalty vectors were looked at (false: the library makes them)
    void merge(const InputFacts& o)
    {
        q_or |= o.q_or; pen_or |= o.pen_or; gomax = std::max(gomax, o.gomax); gemax = std::max(gemax, o.gemax); t_min = std::min(t_min, o.t_min); sum_q_max = std::max(sum_q_max, o.sum_q_max);
        dirty = (dirty < 0 || o.dirty < 0) ? -1 : (dirty | o.dirty);
    }
};
void facts_of_reads(const oct_phmm_reads* R, size_t r0, size_t r1, bool want_dirty, InputFacts* f)
{
    uint32_t v = 0, shortest = 0xffffffffu; uint64_t best = 0;
    for (size_t r = r0; r < r1; ++r) {
        const uint8_t* q = R->qualities + R->offsets[r]; const uint32_t n = R->offsets[r + 1] - R->offsets[r];
        uint32_t sq = 0, o = 0;                           // reads are < 32,768 bases of quality <= 127
        for (uint32_t i = 0; i < n; ++i) { sq += q[i]; o |= q[i]; }
        v |= o; best = std::max<uint64_t>(best, sq); shortest = std::min(shortest, n);
    }
    f->q_or |= v; f->sum_q_max = std::max(f->sum_q_max, best); f->t_min = std::min(f->t_min, shortest);
    if (want_dirty && r1 > r0 && any_byte_outside_acgt((const uint8_t*)R->bases + R->offsets[r0], (size_t)R->offsets[r1] - R->offsets[r0])) f->dirty = 1;
}
void facts_of_haps(const oct_phmm_haplotypes* H, size_t lo, size_t hi, bool want_dirty, InputFacts* f)      // bases [lo, hi) of the concatenated haplotypes, vectors given
{
    uint32_t v = 0, a = 0, e = 0;
    for (size_t i = lo; i < hi; ++i) {
        const uint32_t go = (uint8_t)H->gap_open[i], ge = (uint8_t)H->gap_extend[i];
        v |= go | ge | (uint8_t)H->snv_prior_fwd[i] | (uint8_t)H->snv_prior_rev[i];
        a = std::max(a, go); e = std::max(e, ge);           // (as bytes: with no sign bit anywhere - checked by the caller - these are the values)
    }
    f->pen_or |= v; f->gomax = std::max(f->gomax, a); f->gemax = std::max(f->gemax, e);
    if (want_dirty && (any_byte_outside_acgt((const uint8_t*)H->bases + lo, hi - lo) || any_byte_equals((const uint8_t*)H->snv_mask_fwd + lo, hi - lo, '0')
                       || any_byte_equals((const uint8_t*)H->snv_mask_rev + lo, hi - lo, '0'))) f->dirty = 1;
}

bool monotone(const uint32_t* off, uint32_t n) { for (uint32_t i = 0; i < n; ++i) if (off[i + 1] < off[i]) return false; return true; }

// kernel dispatch over (band, traceback, generic bytes, 32-bit adds)
template <int B, bool TR, bool GEN, bool FA>
bool launch_dp_inst(const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    if (lds > 64 * 1024 && !rt::allow_lds((k_dp<B, TR, GEN, FA>), lds)) return false;    // up to 64 KB needs no opt-in (and the call is a driver round trip)
    OCT_LAUNCH((k_dp<B, TR, GEN, FA>), n_blocks, kBlockWaves * 64, lds, s, p);
    return rt::launch_ok();
}
template <int B, bool FA>
bool launch_dp_band(bool tr, bool gen, const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    if (tr) return gen ? launch_dp_inst<B, true, true, FA>(p, n_blocks, lds, s) : launch_dp_inst<B, true, false, FA>(p, n_blocks, lds, s);
    return gen ? launch_dp_inst<B, false, true, FA>(p, n_blocks, lds, s) : launch_dp_inst<B, false, false, FA>(p, n_blocks, lds, s);
}
bool launch_dp(int band, bool tr, bool gen, bool fa, const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    switch (band) {
        case 8:  return fa ? launch_dp_band<8, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<8, false>(tr, gen, p, n_blocks, lds, s);
        case 16: return fa ? launch_dp_band<16, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<16, false>(tr, gen, p, n_blocks, lds, s);
        case 32: return fa ? launch_dp_band<32, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<32, false>(tr, gen, p, n_blocks, lds, s);
        case 64: return fa ? launch_dp_band<64, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<64, false>(tr, gen, p, n_blocks, lds, s);
        default: return false;
    }
}
// the traceback list and the score-only list of one flavour in one launch (device-sized steps)
template <int B, bool GEN, bool FA>
bool launch_dp_pair_inst(const DpParams& pt, const DpParams& ps, uint32_t n_blocks_t, uint32_t n_blocks_s, size_t lds, rt::Stream s)
{
    if (lds > 64 * 1024 && !rt::allow_lds((k_dp_pair<B, GEN, FA>), lds)) return false;
    OCT_LAUNCH((k_dp_pair<B, GEN, FA>), n_blocks_t + n_blocks_s, kBlockWaves * 64, lds, s, pt, ps, n_blocks_t);
    return rt::launch_ok();
}
template <int B>
bool launch_dp_pair_band(bool gen, bool fa, const DpParams& pt, const DpParams& ps, uint32_t nt, uint32_t ns, size_t lds, rt::Stream s)
{
    if (gen) return fa ? launch_dp_pair_inst<B, true, true>(pt, ps, nt, ns, lds, s) : launch_dp_pair_inst<B, true, false>(pt, ps, nt, ns, lds, s);
    return fa ? launch_dp_pair_inst<B, false, true>(pt, ps, nt, ns, lds, s) : launch_dp_pair_inst<B, false, false>(pt, ps, nt, ns, lds, s);
}
bool launch_dp_pair(int band, bool gen, bool fa, const DpParams& pt, const DpParams& ps, uint32_t nt, uint32_t ns, size_t lds, rt::Stream s)
{
    switch (band) {
        case 8:  return launch_dp_pair_band<8>(gen, fa, pt, ps, nt, ns, lds, s);
        case 16: return launch_dp_pair_band<16>(gen, fa, pt, ps, nt, ns, lds, s);
        case 32: return launch_dp_pair_band<32>(gen, fa, pt, ps, nt, ns, lds, s);
        case 64: return launch_dp_pair_band<64>(gen, fa, pt, ps, nt, ns, lds, s);
        default: return false;
    }
}
template <int B, bool TR>
bool launch_dp32_inst(const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    if (lds > 64 * 1024 && !rt::allow_lds((k_dp32<B, TR>), lds)) return false;
    OCT_LAUNCH((k_dp32<B, TR>), n_blocks, kBlockWaves * 64, lds, s, p);
    return rt::launch_ok();
}
bool launch_dp32(int band, bool tr, const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    switch (band) {
        case 8:  return tr ? launch_dp32_inst<8, true>(p, n_blocks, lds, s) : launch_dp32_inst<8, false>(p, n_blocks, lds, s);
        case 16: return tr ? launch_dp32_inst<16, true>(p, n_blocks, lds, s) : launch_dp32_inst<16, false>(p, n_blocks, lds, s);
        case 32: return tr ? launch_dp32_inst<32, true>(p, n_blocks, lds, s) : launch_dp32_inst<32, false>(p, n_blocks, lds, s);
        case 64: return tr ? launch_dp32_inst<64, true>(p, n_blocks, lds, s) : launch_dp32_inst<64, false>(p, n_blocks, lds, s);
        default: return false;
    }
}
template <int B, int TPR, int C>
bool launch_walk_inst(const WalkParams& w, rt::Stream s, int stage)     // stage: 0 lockstep walker out of registers, 1 lockstep out of LDS-staged tiles, 2 one walk per 16-lane row
{
    const uint32_t blocks = (w.n_tasks + 255) / 256;
    const size_t lds = 256 * kWalkEvents * sizeof(uint32_t);
    if constexpr (C == 1) {
        const size_t stage_lds = walk_stage_lds_bytes(B, TPR);
        if (stage == 2) { const uint32_t th = 64; OCT_LAUNCH((k_walk_rows<B, TPR>), (w.n_tasks + th / 16 - 1) / (th / 16), th, walk_rows_lds_bytes(B, th), s, w); }   // region-sized launch: four walks per wave, runs of matches in one move
        else if (stage && stage_lds <= rt::kMaxLdsBytes) {          // one wave per workgroup, the tiles staged in LDS
            if (stage_lds > 64 * 1024 && !rt::allow_lds((k_walk<B, TPR, C, true>), stage_lds)) return false;
            OCT_LAUNCH((k_walk<B, TPR, C, true>), (w.n_tasks + 63) / 64, 64, stage_lds, s, w);
        } else OCT_LAUNCH((k_walk<B, TPR, C, false>), blocks, 256, lds, s, w);
    } else {
        // bands 128 / 256: one wave walks one task out of LDS-staged lines (k_walk_long) unless the launch is big enough for the lockstep walker to fill its waves
        if (stage) OCT_LAUNCH((k_walk_long<B, C>), w.n_tasks, 64, walk_long_lds_bytes(), s, w);
        else OCT_LAUNCH((k_walk<B, TPR, C>), blocks, 256, lds, s, w);
    }
    if (w.out_align1 != nullptr) OCT_LAUNCH((k_walk_strings<B, TPR, C>), blocks, 256, 0, s, w);   // test seam: gapped strings from the simple per-step walker
    if (w.pair_key != nullptr) OCT_LAUNCH((k_walk_cigar<B, TPR, C>), blocks, 256, 0, s, w);       // align mode: the pairs' winning tasks write their CIGARs
    return rt::launch_ok();
}
bool launch_walk(int band, bool one_per_row, const WalkParams& w, rt::Stream s, int stage)
{
    switch (band) {
        case 8:   return one_per_row ? launch_walk_inst<8, 1, 1>(w, s, stage) : launch_walk_inst<8, 2, 1>(w, s, stage);
        case 16:  return one_per_row ? launch_walk_inst<16, 1, 1>(w, s, stage) : launch_walk_inst<16, 2, 1>(w, s, stage);
        case 32:  return one_per_row ? launch_walk_inst<32, 1, 1>(w, s, stage) : launch_walk_inst<32, 2, 1>(w, s, stage);
        case 64:  return one_per_row ? launch_walk_inst<64, 1, 1>(w, s, stage) : launch_walk_inst<64, 2, 1>(w, s, stage);
        case 128: return launch_walk_inst<128, 1, 2>(w, s, stage);
        case 256: return launch_walk_inst<256, 1, 4>(w, s, stage);
        default: return false;
    }
}
template <int B, bool TR>
bool launch_dp_wide_inst(bool w16, const DpParams& p, rt::Stream s)
{
    constexpr uint32_t ROWS = B < 64 ? 64 / B : 1;                       // tasks per wave
    const uint32_t waves = (p.n_tasks + ROWS - 1) / ROWS, blocks = (waves + kBlockWaves - 1) / kBlockWaves;
    if (w16) OCT_LAUNCH((k_dp_wide<B, TR, true>), blocks, kBlockWaves * 64, 0, s, p);
    else     OCT_LAUNCH((k_dp_wide<B, TR, false>), blocks, kBlockWaves * 64, 0, s, p);
    return rt::launch_ok();
}
bool launch_dp_rows(bool tr, bool gen, const DpParams& p, rt::Stream s)      // long reads at band 16, int32 lanes: one task per row of 16 lanes (k_dp_rows)
{
    // a launch of about a wave per SIMD (ccs256x12: 1,340 waves of ~13,000 dependent iterations each): one wave per workgroup, so that the dispatcher spreads single waves, not fours
    // (4.47-4.59 against 4.62-4.70 ms per step; a device-sized launch counts its bound, ~11 tasks per pair). Measured and not kept (profiles/EXPERIMENTS.md): unused LDS as a cap on the
    // workgroups per CU, and a persistent grid of one workgroup per CU whose waves stride over the groups.
    const uint32_t waves = (p.n_tasks + 3) / 4, wpb = waves <= 16384 ? 1u : (uint32_t)kBlockWaves, blocks = (waves + wpb - 1) / wpb;
    if (tr) { if (gen) OCT_LAUNCH((k_dp_rows<true, true>), blocks, wpb * 64, 0, s, p); else OCT_LAUNCH((k_dp_rows<true, false>), blocks, wpb * 64, 0, s, p); }
    else    { if (gen) OCT_LAUNCH((k_dp_rows<false, true>), blocks, wpb * 64, 0, s, p); else OCT_LAUNCH((k_dp_rows<false, false>), blocks, wpb * 64, 0, s, p); }
    return rt::launch_ok();
}
bool launch_dp_wide(int band, bool tr, bool w16, const DpParams& p, rt::Stream s)
{
    switch (band) {
        case 8:   return tr ? launch_dp_wide_inst<8, true>(w16, p, s) : launch_dp_wide_inst<8, false>(w16, p, s);
        case 16:  return tr ? launch_dp_wide_inst<16, true>(w16, p, s) : launch_dp_wide_inst<16, false>(w16, p, s);
        case 32:  return tr ? launch_dp_wide_inst<32, true>(w16, p, s) : launch_dp_wide_inst<32, false>(w16, p, s);
        case 64:  return tr ? launch_dp_wide_inst<64, true>(w16, p, s) : launch_dp_wide_inst<64, false>(w16, p, s);
        case 128: return tr ? launch_dp_wide_inst<128, true>(w16, p, s) : launch_dp_wide_inst<128, false>(w16, p, s);
        case 256: return tr ? launch_dp_wide_inst<256, true>(w16, p, s) : launch_dp_wide_inst<256, false>(w16, p, s);
        default: return false;
    }
}

template <int B, int PL>
bool launch_dp_mw_band(bool tr, bool gen, const DpParams& p, rt::Stream s)
{
    const uint32_t blocks = p.n_tasks, threads = B / PL;                 // one task per workgroup of B / (64 PL) waves
    if (tr) { if (gen) OCT_LAUNCH((k_dp_mw<B, PL, true, true>), blocks, threads, 0, s, p); else OCT_LAUNCH((k_dp_mw<B, PL, true, false>), blocks, threads, 0, s, p); }
    else    { if (gen) OCT_LAUNCH((k_dp_mw<B, PL, false, true>), blocks, threads, 0, s, p); else OCT_LAUNCH((k_dp_mw<B, PL, false, false>), blocks, threads, 0, s, p); }
    return rt::launch_ok();
}
// one_wave: all planes of a task in one wave (a launch with a task for every SIMD of the chip), else one plane per wave (few tasks: spread them out)
bool launch_dp_mw(int band, bool one_wave, bool tr, bool gen, const DpParams& p, rt::Stream s)
{
    if (band == 128) return one_wave ? launch_dp_mw_band<128, 2>(tr, gen, p, s) : launch_dp_mw_band<128, 1>(tr, gen, p, s);
    if (band == 256) return one_wave ? launch_dp_mw_band<256, 4>(tr, gen, p, s) : launch_dp_mw_band<256, 1>(tr, gen, p, s);
    return false;
}

// Read records of the packed int16 kernels: whole reads in LDS while three workgroups then fit on a CU (150-base reads: 47 KB with traceback tiles, 30 KB without), else
// the largest chunk of iterations (a multiple of 32, at least 64) with which three do, else two - 500-base chunks of long reads against 1.8 kb haplotypes: 115 KB -> 75 KB
// with traceback (one wave per SIMD -> two), 98 -> 51 KB without (-> three). OCT_PHMM_REC_CHUNK=n forces a chunk (test hook: restaging on small reads; 0 = never).
// dense (round 5): a mid-size batch (a region server's device batch: a few rounds of workgroups, both DP forms on the chip at once) takes 64-iteration chunks where that lets a
// fourth traceback workgroup (46 -> 39 KB) and a sixth or seventh score-only one (30 -> 22 KB) onto a CU - the launch is bound by rounds x wave latency, not by issue slots:
// 16 regions of the configs[3] stream 1.054 -> 0.983 ms per populate; the 12.8 M-pair step does not care (traceback 8.60 -> 8.58 ms per launch, score-only 6.25 -> 6.48).
uint32_t dp_rec_chunk(uint32_t t_cap, uint32_t lh_cap, uint32_t B, bool trace, bool dense = false)
{
    long long v;
    if (tune::number("OCT_PHMM_REC_CHUNK", &v)) return v > 0 ? (uint32_t)((v + 3) & ~3ll) : 0u;
    if (dense && t_cap > 100 && dp_lds_bytes(t_cap, lh_cap, B, trace, 64) <= rt::kMaxLdsBytes / (trace ? 4 : 6) - 512
        && dp_lds_bytes(t_cap, lh_cap, B, trace) > rt::kMaxLdsBytes / (trace ? 4 : 6) - 512) return 64u;
    if (t_cap <= 128) return 0;
    for (size_t per_cu : {(size_t)3, (size_t)2}) {
        const size_t budget = rt::kMaxLdsBytes / per_cu - 1024;
        if (dp_lds_bytes(t_cap, lh_cap, B, trace) <= budget) return 0;
        for (uint32_t c = ((t_cap - 1) / 32) * 32; c >= 64; c -= 32) if (dp_lds_bytes(t_cap, lh_cap, B, trace, c) <= budget) return c;
    }
    return dp_lds_bytes(t_cap, lh_cap, B, trace) <= rt::kMaxLdsBytes ? 0u : 64u;      // (no chunk gives two per CU: whole reads if they fit at all)
}

bool ensure_bp(oct_phmm_handle* h, int slice, size_t bytes)
{
    if (h->bp_bytes[slice] >= bytes) return true;
    // grow by half at least (below 4 GB): a region thread's calls differ in size, and every regrowth is a hipFree + hipMalloc that stalls the device
    // (doubling: three workers of a region server each met their biggest batch late in a run of 8,000 calls, and every regrowth of a multi-gigabyte block took
    // 0.1 - 1 s of hipMalloc: profiles/r04_step3_server_api_trace.txt); never beyond the handle's budget
    const size_t old = h->bp_bytes[slice];
    size_t roomy = old < ((size_t)4 << 30) ? std::max(bytes, 2 * old) : bytes;
    if (roomy > h->bp_budget) roomy = std::max(bytes, h->bp_budget);
    rt::dev_free(h->bp[slice]); h->bp[slice] = nullptr; h->bp_bytes[slice] = 0;
    void* p = nullptr; size_t got = roomy;
    if (h->fail_bp_allocs > 0) { --h->fail_bp_allocs; return false; }      // test hook (OCT_PHMM_TEST_FAIL_BP_ALLOCS): the device "has no room": the caller halves its chunk
    if (!rt::dev_malloc(&p, roomy)) {
        rt::clear_error();
        h->pool.trim();                                 // cached blocks of earlier batches may be in the way
        got = bytes;
        if (!rt::dev_malloc(&p, bytes)) {
            rt::clear_error();
            DevPool::trim_device(h->pool.device);       // ... or the sibling handles' (a region server's, a caller's other threads')
            if (!rt::dev_malloc(&p, bytes)) { rt::clear_error(); return false; }
        }
    }
    h->bp[slice] = (uint32_t*)p; h->bp_bytes[slice] = got;
    return true;
}

// Run one kind's task list through the DP kernel (+ walk for traceback kinds), chunked so the traceback scratch fits.
// `ref` (device-sized launch): `tasks` is the array that holds all six lists, `n_tasks` the host's bound for one list; the kernels take the list itself
// from the totals in device memory.
// One workgroup scans ~10 us per tile of 8,192 pairs (41 us at four regions, 82 at eight: profiles/r03_step7_multi_region_timelines.txt); the four launches of the tiled scan
// cost ~20 us whatever the size
constexpr size_t   kPinnedOutMinBytes = (size_t)8 << 20;  // results from here on: is the caller's `out` page-locked? (the question costs microseconds: not asked for region-sized calls)
constexpr uint64_t kLaneMapMinPairs = 200000;          // k-mer mapper: one lane per pair from here on (k_kmer_map_lanes), one wave per pair below
constexpr uint64_t kDslMaxPairs = 400000;              // device-sized launches (no read-back inside the step, grids sized by the host's bound) up to here. Round 4 stopped at 100 k: first 6 / 8 / 12 / 16 / 64
                                                       // regions of the configs[3] stream (50 k / 70 k / 110 k / 150 k / 660 k pairs), one populate from host buffers: 0.54 / 0.60 / 1.04 / 1.31 / 4.14 ms
                                                       // device-sized, 0.57 / 0.57 / 0.97 / 1.21 / 3.56 host-sized. Round 5 (gpurun_out/r05_s01): on the DEVICE the two forms take the same time (16 regions:
                                                       // run + wait 0.643 against 0.650 ms, 64 regions 2.21 against 2.24) - the difference was the upload's byte-by-byte scan for the cost flavours
                                                       // (0.17 / 0.29 ms), now eight bytes per step (any_byte_outside_acgt). What the device-sized form buys a caller who has other work - the region
                                                       // server's workers - is that oct_phmm_batch_run never waits.
constexpr uint64_t kWalkRowsMaxPairs = 49152;          // traceback walks of batches up to here: one walk per 16-lane row (k_walk_rows)
constexpr uint64_t kDslMergeMaxPairs = 12000;          // device-sized step: up to here the traceback and the score-only list of a flavour share one launch (k_dp_pair)
constexpr uint32_t kDslMaxBlocks = 1024;               // grid of a device-sized DP launch: the bound, at most this (workgroups stride over the groups; 4,096 / 16,384 one-unit workgroups: the same populate times, profiles/r06_s14_dsl_grid.txt)
int run_dp_kind(oct_phmm_handle* h, oct_phmm_batch* b, int slice, int kind, const DevTask* tasks, uint32_t n_tasks, TraceEnd* ends,
                int nuc_prior, const WalkParams* seam_walk, oct_phmm_status* status, const rt::Stream* on_stream = nullptr, bool late = false,
                TaskListRef ref = TaskListRef {nullptr, nullptr, 0, nullptr}, const rt::Event* after_first_dp = nullptr,
                int paired_score_list = -1, uint32_t paired_score_bound = 0,     // device-sized traceback launch: the score-only list of the same flavour rides along (k_dp_pair)
                uint32_t joined_late_from = 0xffffffffu,                         // host-sized traceback launch: the flavour's late-start list lies behind the list proper (this many tasks) and is part of `n_tasks`
                const uint32_t* paired_end = nullptr)                            // `tasks` went through k_pair_sort: per haplotype, the list index up to which tasks 2i and 2i + 1 share a window
{
    if (!n_tasks) return OCT_PHMM_OK;
    const bool dsl = ref.totals != nullptr;
    rt::Stream st = on_stream ? *on_stream : h->slice_stream(slice);
    const int B = h->band;
    const uint32_t C = (uint32_t)h->lanes_c;
    const uint32_t G = b->stream ? (B < 64 ? 64u / (uint32_t)B : 1u) : (h->wide ? 1 : 2) * (64 / B);
    const bool tr = kind == kTraceFast || kind == kTraceGen, gen = kind == kScoreGen || kind == kTraceGen;
    const bool dense = b->n_pairs > 20000 && b->n_pairs <= kDslMaxPairs;        // (a region-sized call is one round of workgroups: nothing to gain from a restage every 64 iterations)
    const uint32_t rec_chunk = (b->stream || h->wide) ? 0u : dp_rec_chunk(b->t_cap, b->lh_cap, (uint32_t)B, tr, dense);
    const size_t lds = b->stream ? 0 : dp_lds_bytes(b->t_cap, b->lh_cap, (uint32_t)B, tr, rec_chunk, paired_end != nullptr);
    if (lds > rt::kMaxLdsBytes) return fail(status, OCT_PHMM_EUNSUPPORTED, "read/haplotype too long for the LDS-resident DP kernel");
    DpParams p {};
    p.rec_chunk = rec_chunk;
    p.ref = ref;
    p.rbases = b->d.rbases; p.rquals = b->d.rquals; p.roff = b->d.roff; p.rrev = b->d.rrev; p.hoff = b->d.hoff;
    p.rrec = b->d.rrec; p.rrec_stride = b->d.rrec_stride; p.rrecW = b->d.rrecW;
    p.tabF = gen ? b->d.tabGenF : b->d.tabFastF; p.tabR = gen ? b->d.tabGenR : b->d.tabFastR;
    p.pair_best = b->d.pair_best;
    p.k_cap = bp_tiles(b->t_cap, (uint32_t)B); p.t_cap = b->t_cap; p.lh_cap = b->lh_cap;
    const uint32_t n4 = ((uint32_t)(int8_t)nuc_prior << 2) & 0xffffu;       // vectorise_left_shift_bits(int8_t), simd_pair_hmm.hpp:74-78,257
    p.nuc4 = n4 | n4 << 16;
    const uint32_t n_groups = n_tasks / G;
    // workgroups walk kGroupsPerWave groups per wave to amortise the haplotype-table staging; a small launch (one active region) instead
    // spreads over the chip: one group per wave until there are enough workgroups for every CU
    p.groups_per_block = kBlockWaves * std::max<uint32_t>(1, std::min<uint32_t>(kGroupsPerWave, n_groups / (kBlockWaves * 2048)));
    if (dsl) p.groups_per_block = kBlockWaves;             // (region-sized by construction)
    // late traceback start is PERMITTED wherever the walk may stop early (below); which task groups take it is geometry (dp_groups). `late`: the launch is a late-start list.
    // (a launch of a traceback list proper, beside late-start lists of its own, has no such group by construction: p.late stays 0 and its groups skip the question)
    const bool may_start_late = tr && !seam_walk && !b->align_mode && b->fast_adds && !b->stream && !h->wide && tune::late_start();
    p.late = (may_start_late && (late || ref.join_late || joined_late_from != 0xffffffffu)) ? 1 : 0;
    p.late_from = late ? 0u : joined_late_from;
    p.hap_region = b->d.hap_region; p.reg_rhs = b->d.reg_rhs; p.reg_lhs = b->d.reg_lhs;
    p.paired_end = paired_end; p.task0 = 0;
    uint32_t chunk_groups = n_groups;
    if (tr) {
        const size_t per_group = (size_t)p.k_cap * 4096 * (b->stream ? C : 1);
        const size_t fit = dsl ? n_groups : std::max<size_t>(1, h->bp_budget / std::max<size_t>(1, b->slices.size()) / per_group);   // (device-sized: the bound was checked at upload)
        chunk_groups = (uint32_t)std::min<size_t>(n_groups, fit);
        if (chunk_groups < n_groups) chunk_groups = std::max<uint32_t>(p.groups_per_block, chunk_groups / p.groups_per_block * p.groups_per_block);   // several launches: whole workgroups each
        // the device may not have the budget free (other handles, other processes): fall back to smaller chunks of whole workgroups
        while (!ensure_bp(h, slice, (size_t)std::min(chunk_groups, n_groups) * per_group)) {
            if (chunk_groups <= p.groups_per_block || b->align_mode || dsl) return fail(status, OCT_PHMM_EHIP, "traceback scratch allocation");
            chunk_groups = std::max<uint32_t>(p.groups_per_block, chunk_groups / 2 / p.groups_per_block * p.groups_per_block);
        }
    }
    for (uint32_t g0 = 0; g0 < n_groups; g0 += chunk_groups) {
        const uint32_t ng = std::min(chunk_groups, n_groups - g0);
        p.tasks = tasks + (size_t)g0 * G; p.n_tasks = ng * G; p.task0 = g0 * G;
        if (!late && joined_late_from != 0xffffffffu) p.late_from = joined_late_from > g0 * G ? joined_late_from - g0 * G : 0u;      // (relative to this chunk's first task)
        p.bp = h->bp[slice]; p.ends = tr ? ends + (size_t)g0 * G : nullptr;
        uint32_t n_blocks = (ng + p.groups_per_block - 1) / p.groups_per_block;
        const long long dsl_blocks = kDslMaxBlocks;
        if (dsl) n_blocks = std::min<uint32_t>(n_blocks, (uint32_t)std::max<long long>(64, dsl_blocks));
        rt::Event e0 {}, e1 {};
        if (h->timing) { RT(h->get_event(&e0)); RT(h->get_event(&e1)); RT(rt::event_record(e0, st)); }
        // (a device-sized launch does not know its task count: region-sized, so the spread-out form)
        const bool one_wave = tune::mw_planes() >= 0 ? tune::mw_planes() != 0 : (!dsl && p.n_tasks >= 640);
        if (paired_score_list >= 0 && g0 == 0) {
            DpParams ps = p;                                   // same tables (same flavour), the score-only list of the same task array, no traceback scratch
            ps.ref.list = (uint32_t)paired_score_list; ps.tasks = tasks; ps.n_tasks = paired_score_bound / G * G; ps.bp = nullptr; ps.ends = nullptr; ps.late = 0;
            const uint32_t n_blocks_s = std::min<uint32_t>((ps.n_tasks / G + ps.groups_per_block - 1) / ps.groups_per_block, (uint32_t)std::max<long long>(64, dsl_blocks));
            if (!launch_dp_pair(B, gen, b->fast_adds, p, ps, n_blocks, n_blocks_s, lds, st)) return fail(status, OCT_PHMM_EHIP, "DP kernel launch");
        } else
        if (!(b->multi_wave ? launch_dp_mw(B, one_wave, tr, gen, p, st) : b->rows32 ? launch_dp_rows(tr, gen, p, st) : b->stream ? launch_dp_wide(B, tr, !h->wide, p, st)
                        : h->wide ? launch_dp32(B, tr, p, n_blocks, lds, st) : launch_dp(B, tr, gen, b->fast_adds, p, n_blocks, lds, st)))
            return fail(status, OCT_PHMM_EHIP, "DP kernel launch");
        if (h->timing) { RT(rt::event_record(e1, st)); b->timers.emplace_back(e0, e1); b->timer_kind.push_back(kind); }
        if (after_first_dp && g0 == 0) RT(rt::event_record(*after_first_dp, st));      // whoever waits for it runs beside this launch's walk, not beside its DP
        if (tr) {
            WalkParams w {};
            if (seam_walk) w = *seam_walk;
            w.ref = ref;
            w.tasks = p.tasks; w.n_tasks = p.n_tasks; w.ends = p.ends; w.bp = h->bp[slice]; w.k_cap = p.k_cap; w.band = B;
            w.rbases = b->d.rbases; w.rquals = b->d.rquals; w.roff = b->d.roff; w.rrev = b->d.rrev;
            w.hbases = b->d.hbases; w.hoff = b->d.hoff; w.go = b->d.go; w.ge = b->d.ge;
            w.maskF = b->d.maskF; w.priorF = b->d.priorF; w.maskR = b->d.maskR; w.priorR = b->d.priorR;
            w.hap_region = b->d.hap_region; w.reg_lhs = b->d.reg_lhs; w.reg_rhs = b->d.reg_rhs;
            w.nuc_prior = nuc_prior; w.pair_best = b->d.pair_best;
            w.early_stop = (!seam_walk && !b->align_mode && b->fast_adds && !b->stream && !h->wide) ? 1 : 0;
            if (seam_walk) {   // seam outputs are indexed by task: advance to this chunk
                const size_t o = (size_t)g0 * G;
                w.out_first_pos += o; w.out_align_off += o;
                if (w.seam_lhs) { w.seam_lhs += o; w.seam_rhs += o; w.out_flank += o; w.out_mask_size += o; }
            }
            if (b->align_mode) {
                if (ng != n_groups) return fail(status, OCT_PHMM_EUNSUPPORTED, "alignment batch too large for the traceback scratch (raise OCT_PHMM_BP_BUDGET_GB or split the batch)");
                w.pair_key = b->d.pair_key; w.task_key = b->slices[slice].d_keys; w.pos = b->d.pos; w.npos = b->d.npos; w.max_pos = b->d.max_pos;
                w.err_flags = b->d_err_flags; w.cig_ops = b->d_aln_ops; w.cig_n = b->d_aln_n; w.cig_mpos = b->d_aln_mpos; w.cig_cap = b->cig_cap;
            }
            // region-sized launches (a few hundred waves at most) stage their tiles in LDS; big ones hide the line fetches behind other waves
            // ... and a few regions' worth of walks (up to kWalkRowsMaxPairs pairs) get a 16-lane row each: measured per 300 x 24 call 20 against 60 us, and a 1k x 64 batch
            // 0.52 against 0.46 ms (profiles/r03_step7_small_batch_walkers_ab.log) - from there on the lockstep walker's 64 walks per wave win again
            const bool small = dsl || (size_t)p.n_tasks <= 64 * 1024;
            const int stage = tune::walk_stage() >= 0 ? tune::walk_stage() : (small ? (b->n_pairs <= kWalkRowsMaxPairs ? 2 : 1) : 0);
            if (!launch_walk(B, h->wide || b->stream, w, st, stage)) return fail(status, OCT_PHMM_EHIP, "walk kernel launch");
        }
    }
    return OCT_PHMM_OK;
}

} // namespace

