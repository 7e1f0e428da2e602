// Drop-in calling pattern without an interpreter in the way: T host threads, one oct_phmm handle each, every thread calls
// oct_phmm_populate once per active region (the reference's region tasks, caller.cpp:475 / octopus.cpp:867). Regions are synthetic
// (R reads x H haplotypes, 150 bp x 300 bp, default-model constants); the point is calls/s and how it scales with threads.
//   g++ -O2 -std=c++17 tools/region_calls_bench.cpp -Iinclude -Loctopus_amd -loct_phmm -lpthread -o tools/region_calls_bench
//   LD_LIBRARY_PATH=octopus_amd [OCT_BENCH_DEVICES=0,1,...] tools/region_calls_bench [regions=400] [reads=300] [haps=24] [threads...]
// or, on regions somebody else made (bench.py: the SURVEY 8d config-4 stream of octopus_amd/synth.py, written by synth.write_regions_file):
//   tools/region_calls_bench --file regions.bin --out results.bin [threads...]
// which issues the same calls (one handle from one thread, then the region server from each thread count) and writes the server's answers of the LAST thread
// count - every region's H x R matrix, in file order - for the caller to compare with the reference.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <sys/resource.h>
#include <vector>
#include "oct_phmm.h"

struct Region {
    std::string rb, hb, mf, mr; std::vector<uint8_t> q, mq, rev; std::vector<uint32_t> ro {0}, ho {0};
    std::vector<int64_t> rbeg, hbeg; std::vector<int8_t> go, ge, pf, pr; std::vector<double> out;
};

static Region make_region(std::mt19937_64& rng, int R, int H, int T = 150, int Lh = 300, int B = 16)
{
    static const char A[] = "ACGT";
    Region g;
    std::string base(Lh, 'A');
    for (auto& c : base) c = A[rng() & 3];
    for (int h = 0; h < H; ++h) {
        std::string s = base;
        for (int e = 0; e < (h ? 2 : 0); ++e) s[60 + rng() % 180] = A[rng() & 3];
        g.hb += s; g.ho.push_back((uint32_t)g.hb.size()); g.hbeg.push_back(0);
        for (int i = 0; i < Lh; ++i) { g.mf += s[(i + Lh - 1) % Lh]; g.mr += s[(i + 1) % Lh]; }
    }
    g.go.assign(g.hb.size(), 45); g.ge.assign(g.hb.size(), 3); g.pf.assign(g.hb.size(), 125); g.pr.assign(g.hb.size(), 125);
    for (int r = 0; r < R; ++r) {
        const int h = (int)(rng() % H), start = B + (int)(rng() % (Lh - T - 2 * B));
        std::string s = g.hb.substr((size_t)h * Lh + start, T);
        for (int i = 0; i < T; ++i) {
            const uint8_t qual = (rng() % 10) < 7 ? 37 : ((rng() % 3) ? 25 : 12);
            g.q.push_back(qual);
            if ((rng() % 1000) < (qual == 37 ? 1u : qual == 25 ? 4u : 60u)) s[i] = A[rng() & 3];
        }
        g.rb += s; g.ro.push_back((uint32_t)g.rb.size()); g.mq.push_back(60); g.rev.push_back(rng() & 1); g.rbeg.push_back(start);
    }
    g.out.assign((size_t)R * H, 0.0);
    return g;
}


// octopus_amd/synth.py::write_regions_file: "OCTR", u32 n; per region u32 {R, H, has_flank, lhs, rhs}, u64 {read bases, haplotype bases}, then the arrays below
static bool read_regions(const char* path, std::vector<Region>& out, std::vector<int>& has_flank, std::vector<oct_phmm_flank_state>& flank)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    auto get = [&](void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; };
    char magic[4]; uint32_t n = 0;
    if (!get(magic, 4) || memcmp(magic, "OCTR", 4) != 0 || !get(&n, 4)) { fclose(f); return false; }
    out.resize(n); has_flank.resize(n); flank.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t h[5]; uint64_t nb[2];
        if (!get(h, sizeof h) || !get(nb, sizeof nb)) { fclose(f); return false; }
        Region& g = out[i];
        const uint32_t R = h[0], H = h[1]; has_flank[i] = (int)h[2]; flank[i] = oct_phmm_flank_state {h[3], h[4]};
        g.ro.resize(R + 1); g.rb.resize(nb[0]); g.q.resize(nb[0]); g.mq.resize(R); g.rev.resize(R); g.rbeg.resize(R);
        g.ho.resize(H + 1); g.hb.resize(nb[1]); g.hbeg.resize(H); g.go.resize(nb[1]); g.ge.resize(nb[1]); g.mf.resize(nb[1]); g.pf.resize(nb[1]); g.mr.resize(nb[1]); g.pr.resize(nb[1]);
        if (!get(g.ro.data(), 4 * (R + 1)) || !get(&g.rb[0], nb[0]) || !get(g.q.data(), nb[0]) || !get(g.mq.data(), R) || !get(g.rev.data(), R) || !get(g.rbeg.data(), 8 * R)
            || !get(g.ho.data(), 4 * (H + 1)) || !get(&g.hb[0], nb[1]) || !get(g.hbeg.data(), 8 * H) || !get(g.go.data(), nb[1]) || !get(g.ge.data(), nb[1])
            || !get(&g.mf[0], nb[1]) || !get(g.pf.data(), nb[1]) || !get(&g.mr[0], nb[1]) || !get(g.pr.data(), nb[1])) { fclose(f); return false; }
        g.out.assign((size_t)R * H, 0.0);
    }
    fclose(f);
    return true;
}

int main(int argc, char** argv)
{
    std::vector<Region> regions; std::vector<int> has_flank; std::vector<oct_phmm_flank_state> flanks;
    std::vector<int> threads;
    const char* out_path = nullptr;
    int n_regions, R = 0, H = 0;
    double n_loglik = 0;
    const bool from_file = argc > 2 && !strcmp(argv[1], "--file");
    if (from_file) {
        if (!read_regions(argv[2], regions, has_flank, flanks)) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        int i = 3;
        if (argc > 4 && !strcmp(argv[3], "--out")) { out_path = argv[4]; i = 5; }
        for (; i < argc; ++i) threads.push_back(atoi(argv[i]));
        n_regions = (int)regions.size();
    } else {
        n_regions = argc > 1 ? atoi(argv[1]) : 400; R = argc > 2 ? atoi(argv[2]) : 300; H = argc > 3 ? atoi(argv[3]) : 24;
        for (int i = 4; i < argc; ++i) threads.push_back(atoi(argv[i]));
        std::mt19937_64 rng(42);
        for (int i = 0; i < n_regions; ++i) regions.push_back(make_region(rng, R, H));
        has_flank.assign((size_t)n_regions, 1); flanks.assign((size_t)n_regions, oct_phmm_flank_state {40, 40});
    }
    if (threads.empty()) threads = {1, 2, 4, 8, 16};
    for (const Region& g : regions) n_loglik += (double)g.out.size();
    const int reps = from_file ? (getenv("OCT_BENCH_REPS") ? atoi(getenv("OCT_BENCH_REPS")) : 4) : 1;      // timed rounds over the file's regions per server configuration
    for (int T : threads) {                                     // ---- one region server shared by T calling threads (oct_phmm_server) ----
        oct_phmm_config c; oct_phmm_config_default(&c); c.max_indel_error = 16;
        oct_phmm_server* srv = nullptr;
        // OCT_BENCH_DEVICES="0,1,..." (default: every visible MI355X): the server's workers and handles are spread over these devices
        std::vector<int32_t> devs;
        if (const char* e = getenv("OCT_BENCH_DEVICES")) { for (const char* p = e; *p; ) { devs.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; } }
        else for (int d = 0; d < oct_phmm_device_count(); ++d) devs.push_back(d);
        if (devs.empty() || oct_phmm_server_create_multi(&c, devs.data(), (uint32_t)devs.size(), 0, &srv) != OCT_PHMM_OK) { fprintf(stderr, "no device\n"); return 1; }
        int failures = 0;
        auto work = [&](int t, int count) {
            for (int k = t; k < count; k += T) {
                const int i = k % n_regions;
                Region& g = regions[i];
                oct_phmm_reads rd {(uint32_t)g.mq.size(), g.rb.data(), g.q.data(), g.ro.data(), g.mq.data(), g.rev.data(), g.rbeg.data(), 0, nullptr};
                oct_phmm_haplotypes hp {(uint32_t)g.hbeg.size(), g.hb.data(), g.ho.data(), g.hbeg.data(), g.go.data(), g.ge.data(), g.mf.data(), g.pf.data(), g.mr.data(), g.pr.data()};
                oct_phmm_status st;
                if (oct_phmm_server_populate(srv, &rd, &hp, has_flank[i] ? &flanks[i] : nullptr, nullptr, g.out.data(), &st) != OCT_PHMM_OK) ++failures;
            }
        };
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0 warms the handles' pools (file mode: one whole round - the regions differ in size by orders of magnitude); pass 1 is timed, `reps` rounds
            const int count = pass ? n_regions * reps : (from_file ? n_regions : std::min(n_regions, 4 * T));
            uint64_t c0 = 0, b0 = 0, c1 = 0, b1 = 0; oct_phmm_server_stats(srv, &c0, &b0);
            auto cpu_s = [] { rusage u; getrusage(RUSAGE_SELF, &u); return (double)u.ru_utime.tv_sec + u.ru_utime.tv_usec * 1e-6 + (double)u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6; };
            const double cpu0 = cpu_s();
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t, count);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double cpu = cpu_s() - cpu0;                       // host cores the process kept busy (callers + workers + the runtime's threads)
            oct_phmm_server_stats(srv, &c1, &b1);
            if (pass) printf("{\"mode\": \"server\", \"devices\": %d, \"threads\": %d, \"regions_per_s\": %.1f, \"M_loglik_per_s\": %.2f, \"regions_per_device_batch\": %.2f, \"host_cores_busy\": %.2f, \"failures\": %d}\n",
                             (int)devs.size(), T, (double)count / dt, n_loglik * reps / dt / 1e6, (double)(c1 - c0) / (double)std::max<uint64_t>(1, b1 - b0), cpu / dt, failures);
        }
        if (T == threads.back()) {                                  // the server's answers against plain populate calls on a handle of our own, bit for bit
            oct_phmm_handle* hv = nullptr;
            if (oct_phmm_create(&c, &hv) != OCT_PHMM_OK) { fprintf(stderr, "no device\n"); return 1; }
            const int n_check = std::min(n_regions, 64); int bad = 0;
            for (int i = 0; i < n_check; ++i) {
                Region& g = regions[i];
                std::vector<double> want(g.out.size());
                oct_phmm_reads rd {(uint32_t)g.mq.size(), g.rb.data(), g.q.data(), g.ro.data(), g.mq.data(), g.rev.data(), g.rbeg.data(), 0, nullptr};
                oct_phmm_haplotypes hp {(uint32_t)g.hbeg.size(), g.hb.data(), g.ho.data(), g.hbeg.data(), g.go.data(), g.ge.data(), g.mf.data(), g.pf.data(), g.mr.data(), g.pr.data()};
                oct_phmm_status st;
                if (oct_phmm_populate(hv, &rd, &hp, nullptr, has_flank[i] ? &flanks[i] : nullptr, nullptr, want.data(), &st) != OCT_PHMM_OK || want != g.out) ++bad;
            }
            oct_phmm_destroy(hv);
            printf("{\"mode\": \"server vs plain calls\", \"regions_compared\": %d, \"regions_that_differ\": %d}\n", n_check, bad);
            if (bad) failures += bad;
            if (out_path) {                                           // the server's answers at this caller count, region after region
                FILE* f = fopen(out_path, "wb");
                if (!f) { fprintf(stderr, "cannot write %s\n", out_path); return 1; }
                for (const Region& g : regions) fwrite(g.out.data(), sizeof(double), g.out.size(), f);
                fclose(f);
            }
        }
        oct_phmm_server_destroy(srv);
    }
    for (int T : threads) {                                     // ---- one handle per calling thread ----
        if (from_file && T != 1) continue;                      // (the file mode asks for the one-thread, one-handle figure only)
        std::vector<oct_phmm_handle*> hs(T, nullptr);
        oct_phmm_config c; oct_phmm_config_default(&c); c.max_indel_error = 16;
        for (auto& h : hs) if (oct_phmm_create(&c, &h) != OCT_PHMM_OK) { fprintf(stderr, "no device\n"); return 1; }
        int failures = 0;
        auto work = [&](int t, int count) {
            for (int i = t; i < count; i += T) {
                Region& g = regions[i];
                oct_phmm_reads rd {(uint32_t)g.mq.size(), g.rb.data(), g.q.data(), g.ro.data(), g.mq.data(), g.rev.data(), g.rbeg.data(), 0, nullptr};
                oct_phmm_haplotypes hp {(uint32_t)g.hbeg.size(), g.hb.data(), g.ho.data(), g.hbeg.data(), g.go.data(), g.ge.data(), g.mf.data(), g.pf.data(), g.mr.data(), g.pr.data()};
                oct_phmm_status st;
                if (oct_phmm_populate(hs[t], &rd, &hp, nullptr, has_flank[i] ? &flanks[i] : nullptr, nullptr, g.out.data(), &st) != OCT_PHMM_OK) ++failures;
            }
        };
        for (int pass = 0; pass < 2; ++pass) {                  // pass 0 warms the handles' pools
            const int count = pass ? n_regions : std::min(n_regions, 4 * T);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t, count);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (pass) printf("{\"mode\": \"handle per thread\", \"threads\": %d, \"regions_per_s\": %.1f, \"ms_per_call\": %.3f, \"M_loglik_per_s\": %.2f, \"failures\": %d}\n",
                             T, n_regions / dt, dt / n_regions * T * 1e3, n_loglik / dt / 1e6, failures);
        }
        for (auto h : hs) oct_phmm_destroy(h);
    }
    return 0;
}
