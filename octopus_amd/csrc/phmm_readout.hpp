// Genotype read-out over the likelihood matrix populate() left in HBM:
//   ln p(reads | genotype) = sum_read [ ln sum_{haplotype in genotype} p(read | haplotype) - ln ploidy ]
// ConstantMixtureGenotypeLikelihoodModel::evaluate, core/models/genotype/constant_mixture_genotype_likelihood_model.cpp:29-330
// (the IndexedHaplotype overloads, :76-330: one case per ploidy <= 4 and zygosity pattern, generic log-sum-exp above),
// maths::log_sum_exp (utils/maths.hpp:292-330).
//
// One thread per genotype; a workgroup stages a tile of rows of ALL the region's haplotype columns in LDS ([haplotype][rows + 1]
// doubles, so lanes holding consecutive haplotypes hit distinct banks and lanes sharing a haplotype broadcast) and every thread
// walks the tile's rows in order, accumulating in fp64. Rows are split over `n_splits` workgroups per genotype chunk; the
// partial sums are combined in split order by k_genotype_sum, so the result does not depend on scheduling.
#pragma once
#include "phmm_device.hpp"

namespace octphmm {

constexpr int kMaxPloidy = 16;
constexpr int kReadoutThreads = 256;

struct ReadoutSet {                 // one evaluate(genotypes, model) call: one region, one primed row range, one ploidy
    uint64_t gt_idx_off;            // first entry of the set's genotypes in ReadoutParams::gt
    uint64_t partial_off;           // first partial sum of the set in ReadoutParams::partial ([n_splits][n_genotypes])
    uint32_t gt0, n_genotypes, ploidy;
    uint32_t hap0, n_haps;          // the region's haplotypes [hap0, hap0 + n_haps)
    uint32_t row_begin, row_end;    // primed rows of the region
    uint32_t tile_rows;             // rows per LDS tile
    uint32_t n_splits, rows_per_split;
};

struct ReadoutParams {
    const double*   lik;            // populate output: lik[hap_out_off[h] + region_row]
    const uint64_t* hap_out_off;
    const uint32_t* gt;             // batch haplotype indices, ploidy per genotype, Genotype order
    const ReadoutSet* sets;
    const uint4*    blocks;         // per workgroup {set, chunk, split, 0}
    double*  partial;
    double*  out;                   // [sum of n_genotypes]
};

// constant_mixture_genotype_likelihood_model.cpp:46-63 (the ln lookup table)
OCT_DEVICE double ln_small(uint32_t n)
{
    switch (n) {
        case 1: return 0.0;
        case 2: return 0.693147180559945309417232121458176568075500134360255254120;
        case 3: return 1.098612288668109691395245236922525704647490557822749451734;
        case 4: return 1.386294361119890618834464242916353136151000268720510508241;
        default: return log((double)n);
    }
}

OCT_DEVICE double lse2(double a, double b)      // utils/maths.hpp:294-298
{
    const double lo = b < a ? b : a, hi = b < a ? a : b;     // std::minmax
    return hi + log1p(exp(lo - hi));
}

// Ploidy 1..4: duplicates collapsed into weighted terms exactly as the reference's per-case lambdas do.
template <int P>
struct SmallGenotype {
    uint32_t idx[P];        // LDS column offsets of the distinct haplotypes
    double   w[P];          // ln multiplicity added to the term (0 when 1)
    int      n;             // distinct haplotypes
    double   c;             // constant subtracted per read

    OCT_DEVICE void init(const uint32_t* g, uint32_t hap0, uint32_t stride)
    {
        uint32_t mult[P];
        n = 0;
        for (int j = 0; j < P; ++j) {
            const uint32_t col = (g[j] - hap0) * stride;
            if (n > 0 && idx[n - 1] == col) ++mult[n - 1];
            else { idx[n] = col; mult[n] = 1; ++n; }
        }
        for (int j = n; j < P; ++j) { idx[j] = idx[0]; mult[j] = 1; }
        c = ln_small(P);
        for (int j = 0; j < P; ++j) w[j] = ln_small(mult[j]);
        if (P == 4 && n == 2 && mult[0] == 2) { w[0] = 0; w[1] = 0; c = ln_small(2); }   // :258-266: {a,a,b,b} is evaluated as a diploid
    }
    OCT_DEVICE double eval(const double* tile, uint32_t r) const
    {
        if (n == 1) return tile[idx[0] + r];                                            // homozygous: plain accumulate
        if (P >= 2 && n == 2) return lse2(w[0] + tile[idx[0] + r], w[P >= 2 ? 1 : 0] + tile[idx[P >= 2 ? 1 : 0] + r]) - c;
        double x[P];
        double mx = -1.7976931348623157e308;
        for (int j = 0; j < P; ++j) {
            x[j] = w[j] + tile[idx[j] + r];
            if (j < n && x[j] > mx) mx = x[j];
        }
        double s = exp(x[0] - mx);
        for (int j = 1; j < P; ++j) if (j < n) s = s + exp(x[j] - mx);
        return mx + log(s) - c;
    }
};

template <int P>
OCT_DEVICE void genotype_lik_block(const ReadoutParams& p, const ReadoutSet& q, uint32_t chunk, uint32_t split, double* tile)
{
    const uint32_t tid = hw::thread_idx();
    const uint32_t g = chunk * kReadoutThreads + tid;
    const uint32_t stride = q.tile_rows + 1;
    const bool live = g < q.n_genotypes;
    const uint32_t* gi = p.gt + q.gt_idx_off + (size_t)g * q.ploidy;
    SmallGenotype<(P > 0 ? P : 1)> sg;
    if (P > 0 && live) sg.init(gi, q.hap0, stride);
    const uint32_t r_first = q.row_begin + split * q.rows_per_split;
    const uint32_t r_last = r_first + q.rows_per_split < q.row_end ? r_first + q.rows_per_split : q.row_end;
    const double ln_ploidy = log((double)q.ploidy);                 // :141, :321
    double acc = 0;
    for (uint32_t r0 = r_first; r0 < r_last; r0 += q.tile_rows) {
        const uint32_t nr = r_last - r0 < q.tile_rows ? r_last - r0 : q.tile_rows;
        for (uint32_t i = tid; i < q.n_haps * q.tile_rows; i += kReadoutThreads) {
            const uint32_t h = i / q.tile_rows, r = i - h * q.tile_rows;
            if (r < nr) tile[h * stride + r] = p.lik[p.hap_out_off[q.hap0 + h] + r0 + r];
        }
        hw::block_sync();
        if (live) {
            if (P > 0) {
                for (uint32_t r = 0; r < nr; ++r) acc = acc + sg.eval(tile, r);
            } else {                                                // evaluate_polyploid, :316-329
                for (uint32_t r = 0; r < nr; ++r) {
                    double mx = tile[(gi[0] - q.hap0) * stride + r];
                    for (uint32_t j = 1; j < q.ploidy; ++j) { const double x = tile[(gi[j] - q.hap0) * stride + r]; if (x > mx) mx = x; }
                    double s = 0;
                    for (uint32_t j = 0; j < q.ploidy; ++j) s = s + exp(tile[(gi[j] - q.hap0) * stride + r] - mx);
                    acc = acc + ((mx + log(s)) - ln_ploidy);
                }
            }
        }
        hw::block_sync();
    }
    if (live) {
        if (q.n_splits == 1) p.out[q.gt0 + g] = acc;
        else p.partial[q.partial_off + (size_t)split * q.n_genotypes + g] = acc;
    }
}

OCT_KERNEL(k_genotype_lik)(ReadoutParams p)
{
    OCT_DYN_SMEM(smem);
    double* tile = (double*)smem;                                   // [n_haps][tile_rows + 1]
    const uint4 blk = p.blocks[hw::block_idx()];
    const ReadoutSet q = p.sets[blk.x];
    switch (q.ploidy) {                                             // :65-74
        case 1: genotype_lik_block<1>(p, q, blk.y, blk.z, tile); break;
        case 2: genotype_lik_block<2>(p, q, blk.y, blk.z, tile); break;
        case 3: genotype_lik_block<3>(p, q, blk.y, blk.z, tile); break;
        case 4: genotype_lik_block<4>(p, q, blk.y, blk.z, tile); break;
        default: genotype_lik_block<0>(p, q, blk.y, blk.z, tile); break;
    }
}

// combine the row splits in split order; one workgroup per {set, chunk} (the blocks with split == 0 of sets that were split)
OCT_KERNEL(k_genotype_sum)(ReadoutParams p, const uint4* sum_blocks)
{
    const uint4 blk = sum_blocks[hw::block_idx()];
    const ReadoutSet q = p.sets[blk.x];
    const uint32_t g = blk.y * kReadoutThreads + hw::thread_idx();
    if (g >= q.n_genotypes) return;
    double acc = 0;
    for (uint32_t s = 0; s < q.n_splits; ++s) acc = acc + p.partial[q.partial_off + (size_t)s * q.n_genotypes + g];
    p.out[q.gt0 + g] = acc;
}

} // namespace octphmm
