// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - clock probe, kernel times, genotype read-out on the resident matrix, page-locked host memory.
// ---------------------------------------------------------------------------------------------------------------
// diagnostic: the shader clock while other work runs (bench.py prices its VALU roofline at the clock the DP kernels actually get)
// ---------------------------------------------------------------------------------------------------------------
#if !defined(OCTPHMM_SIM)
__global__ void k_clock_probe(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = clock64(), r0 = wall_clock64();          // s_memtime: shader cycles; s_memrealtime: the constant reference clock
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
#endif
extern "C" int oct_phmm_probe_clock(oct_phmm_handle* h, double window_ms, double* shader_ghz)
{
    if (!h || !shader_ghz || !(window_ms > 0) || window_ms > 1000) return OCT_PHMM_EINVAL;
#if defined(OCTPHMM_SIM)
    return OCT_PHMM_EUNSUPPORTED;
#else
    if (!rt::set_device(h->cfg.device_id)) return OCT_PHMM_EHIP;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id) != hipSuccess || khz <= 0) return OCT_PHMM_EHIP;
    // its own stream (made once per handle: creating one, like hipMalloc, synchronises the device): the probe wave runs beside whatever the handle's streams are doing
    if (!h->probe_ready) {
        if (!rt::stream_create(&h->probe_stream) || !rt::dev_malloc((void**)&h->d_probe, 16) || !rt::host_pinned_malloc((void**)&h->h_probe, 16)) return OCT_PHMM_EHIP;
        h->probe_ready = true;
    }
    rt::Stream s = h->probe_stream; unsigned long long* v = h->h_probe;
    v[0] = v[1] = 0;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, h->d_probe, (unsigned long long)(window_ms * khz));
    const bool ok = rt::launch_ok() && rt::d2h(v, h->d_probe, 16, s) && rt::stream_sync(s);
    if (!ok || !v[1]) return OCT_PHMM_EHIP;
    *shader_ghz = (double)v[0] / (double)v[1] * khz * 1e-6;
    return OCT_PHMM_OK;
#endif
}

extern "C" int oct_phmm_batch_kernel_time(const oct_phmm_batch* b, double* ms, uint32_t* launches)
{
    if (!b) return OCT_PHMM_EINVAL;
    if (ms) *ms = b->dp_ms;
    if (launches) *launches = b->dp_launches;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_batch_kernel_time_by_kind(const oct_phmm_batch* b, double ms[4], uint32_t launches[4])
{
    if (!b) return OCT_PHMM_EINVAL;
    for (int k = 0; k < kNumKinds; ++k) { if (ms) ms[k] = b->kind_ms[k]; if (launches) launches[k] = b->kind_launches[k]; }
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// genotype read-out (phmm_readout.hpp)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int oct_phmm_batch_genotype_likelihoods(oct_phmm_handle* h, oct_phmm_batch* b, const oct_phmm_genotype_sets* gs,
                                                   double* out, oct_phmm_status* status)
{
    if (!h || !b || b->owner != h || !gs) return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (!b->ran) return fail(status, OCT_PHMM_EINVAL, "batch was not run");
    if (gs->n_sets == 0) return ok(status);
    if (!gs->ploidy || !gs->gt_offsets || gs->gt_offsets[0] != 0) return fail(status, OCT_PHMM_EINVAL, "genotype set tables");
    const uint32_t n_gt = gs->gt_offsets[gs->n_sets];
    if (n_gt == 0) return ok(status);
    if (!gs->hap_indices || !out) return fail(status, OCT_PHMM_EINVAL, "null genotype indices or output");
    const int rc = oct_phmm_batch_wait(h, b, status);
    if (rc != OCT_PHMM_OK) return rc;

    constexpr uint32_t kTargetBlocks = 2048;                     // >= 8 workgroups per CU before rows are split
    constexpr size_t kTileBytes = 96 * 1024;
    std::vector<ReadoutSet> sets(gs->n_sets);
    std::vector<uint4> blocks, sum_blocks;
    uint64_t idx_off = 0, partial_off = 0; size_t lds = 0;
    uint64_t total_blocks_unsplit = 0;
    for (uint32_t s = 0; s < gs->n_sets; ++s) total_blocks_unsplit += (gs->gt_offsets[s + 1] - gs->gt_offsets[s] + kReadoutThreads - 1) / kReadoutThreads;
    for (uint32_t s = 0; s < gs->n_sets; ++s) {
        ReadoutSet& q = sets[s];
        if (gs->gt_offsets[s + 1] < gs->gt_offsets[s]) return fail(status, OCT_PHMM_EINVAL, "genotype offsets must not decrease");
        q.gt0 = gs->gt_offsets[s]; q.n_genotypes = gs->gt_offsets[s + 1] - q.gt0; q.ploidy = gs->ploidy[s];
        q.gt_idx_off = idx_off; q.partial_off = partial_off;
        if (q.ploidy < 1 || q.ploidy > OCT_PHMM_MAX_PLOIDY) return fail(status, OCT_PHMM_EUNSUPPORTED, "ploidy outside 1..16");
        const uint32_t* gi = gs->hap_indices + idx_off;
        idx_off += (uint64_t)q.n_genotypes * q.ploidy;
        if (q.n_genotypes == 0) { q.n_splits = 1; continue; }
        if (gi[0] >= b->n_haps) return fail(status, OCT_PHMM_EINVAL, "haplotype index out of range");
        const uint32_t reg = b->h_hap_region[gi[0]];
        q.hap0 = b->h_reg_hap0[reg]; q.n_haps = b->h_reg_hap0[reg + 1] - q.hap0;
        for (uint64_t i = 0; i < (uint64_t)q.n_genotypes * q.ploidy; ++i) {
            if (gi[i] < q.hap0 || gi[i] >= q.hap0 + q.n_haps) return fail(status, OCT_PHMM_EINVAL, "genotypes of one set must use haplotypes of one region");
            if (i % q.ploidy && gi[i] < gi[i - 1]) return fail(status, OCT_PHMM_EINVAL, "genotype haplotype indices must be sorted");
        }
        const uint32_t rows = (uint32_t)(b->h_hap_out_off[q.hap0 + 1] - b->h_hap_out_off[q.hap0]);
        q.row_begin = gs->row_begin ? gs->row_begin[s] : 0; q.row_end = gs->row_end ? gs->row_end[s] : rows;
        if (q.row_begin > q.row_end || q.row_end > rows) return fail(status, OCT_PHMM_EINVAL, "row range outside the region");
        const uint32_t nrows = q.row_end - q.row_begin;
        const size_t fit = kTileBytes / (8 * (size_t)q.n_haps);
        if (fit < 3) return fail(status, OCT_PHMM_EUNSUPPORTED, "too many haplotypes in one region for the read-out tile");
        q.tile_rows = (uint32_t)std::min<size_t>(32, fit - 1);
        lds = std::max(lds, (size_t)q.n_haps * (q.tile_rows + 1) * 8);
        const uint32_t n_chunks = (q.n_genotypes + kReadoutThreads - 1) / kReadoutThreads;
        const uint32_t n_tiles = std::max(1u, (nrows + q.tile_rows - 1) / q.tile_rows);
        uint32_t want = total_blocks_unsplit >= kTargetBlocks ? 1 : (uint32_t)((kTargetBlocks + total_blocks_unsplit - 1) / total_blocks_unsplit);
        want = std::min(want, n_tiles);
        const uint32_t tiles_per_split = (n_tiles + want - 1) / want;
        q.rows_per_split = tiles_per_split * q.tile_rows;
        q.n_splits = (n_tiles + tiles_per_split - 1) / tiles_per_split;
        if (q.n_splits > 1) partial_off += (uint64_t)q.n_splits * q.n_genotypes;
        for (uint32_t sp = 0; sp < q.n_splits; ++sp)
            for (uint32_t c = 0; c < n_chunks; ++c) blocks.push_back(uint4{s, c, sp, 0});
        if (q.n_splits > 1) for (uint32_t c = 0; c < n_chunks; ++c) sum_blocks.push_back(uint4{s, c, 0, 0});
    }
    if (blocks.empty()) return ok(status);
    if (blocks.size() > 0x7fffffffull) return fail(status, OCT_PHMM_EUNSUPPORTED, "too many genotypes in one call");

    RT(rt::set_device(h->cfg.device_id));
    rt::Stream st = h->stream;
    struct Tmp { oct_phmm_handle* h; std::vector<void*> v; ~Tmp() { for (void* p : v) h->pool.release(p); } } tmp {h, {}};
    auto put = [&](const void* host, size_t bytes, void** dev) {
        if (!h->pool.alloc(dev, bytes)) return false;
        tmp.v.push_back(*dev);
        return host ? rt::h2d(*dev, host, bytes, st) : true;
    };
    void *d_gt = nullptr, *d_sets = nullptr, *d_blocks = nullptr, *d_sum = nullptr, *d_partial = nullptr, *d_res = nullptr;
    RT(put(gs->hap_indices, idx_off * sizeof(uint32_t), &d_gt));
    RT(put(sets.data(), sets.size() * sizeof(ReadoutSet), &d_sets));
    RT(put(blocks.data(), blocks.size() * sizeof(uint4), &d_blocks));
    if (!sum_blocks.empty()) { RT(put(sum_blocks.data(), sum_blocks.size() * sizeof(uint4), &d_sum)); RT(put(nullptr, partial_off * sizeof(double), &d_partial)); }
    RT(put(nullptr, (size_t)n_gt * sizeof(double), &d_res));
    ReadoutParams p {};
    p.lik = b->d_out; p.hap_out_off = b->d.hap_out_off; p.gt = (const uint32_t*)d_gt; p.sets = (const ReadoutSet*)d_sets;
    p.blocks = (const uint4*)d_blocks; p.partial = (double*)d_partial; p.out = (double*)d_res;
    if (lds > 64 * 1024) RT(rt::allow_lds(k_genotype_lik, lds));
    OCT_LAUNCH(k_genotype_lik, (uint32_t)blocks.size(), kReadoutThreads, lds, st, p);
    RT(rt::launch_ok());
    if (!sum_blocks.empty()) {
        OCT_LAUNCH(k_genotype_sum, (uint32_t)sum_blocks.size(), kReadoutThreads, 0, st, p, (const uint4*)d_sum);
        RT(rt::launch_ok());
    }
    RT(rt::d2h(out, d_res, (size_t)n_gt * sizeof(double), st));
    RT(rt::stream_sync(st));
    return ok(status);
}

extern "C" void* oct_phmm_host_alloc(size_t bytes) { void* p = nullptr; return rt::host_pinned_malloc(&p, bytes) ? p : nullptr; }
extern "C" void oct_phmm_host_free(void* p) { rt::host_pinned_free(p); }

