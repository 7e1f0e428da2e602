// Drop-in calling pattern without an interpreter in the way: T host threads, one oct_phmm handle each, every thread calls
// oct_phmm_populate once per active region (the reference's region tasks, caller.cpp:475 / octopus.cpp:867). Regions are synthetic
// (R reads x H haplotypes, 150 bp x 300 bp, default-model constants); the point is calls/s and how it scales with threads.
//   g++ -O2 -std=c++17 tools/region_calls_bench.cpp -Iinclude -Loctopus_amd -loct_phmm -lpthread -o tools/region_calls_bench
//   LD_LIBRARY_PATH=octopus_amd [OCT_BENCH_DEVICES=0,1,...] tools/region_calls_bench [regions=400] [reads=300] [haps=24] [threads...]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
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

int main(int argc, char** argv)
{
    const int n_regions = argc > 1 ? atoi(argv[1]) : 400, R = argc > 2 ? atoi(argv[2]) : 300, H = argc > 3 ? atoi(argv[3]) : 24;
    std::vector<int> threads; for (int i = 4; i < argc; ++i) threads.push_back(atoi(argv[i]));
    if (threads.empty()) threads = {1, 2, 4, 8, 16};
    std::mt19937_64 rng(42);
    std::vector<Region> regions; for (int i = 0; i < n_regions; ++i) regions.push_back(make_region(rng, R, H));
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
            for (int i = t; i < count; i += T) {
                Region& g = regions[i];
                oct_phmm_reads rd {(uint32_t)g.mq.size(), g.rb.data(), g.q.data(), g.ro.data(), g.mq.data(), g.rev.data(), g.rbeg.data(), 0, nullptr};
                oct_phmm_haplotypes hp {(uint32_t)g.hbeg.size(), g.hb.data(), g.ho.data(), g.hbeg.data(), g.go.data(), g.ge.data(), g.mf.data(), g.pf.data(), g.mr.data(), g.pr.data()};
                oct_phmm_flank_state fl {40, 40}; oct_phmm_status st;
                if (oct_phmm_server_populate(srv, &rd, &hp, &fl, nullptr, g.out.data(), &st) != OCT_PHMM_OK) ++failures;
            }
        };
        for (int pass = 0; pass < 2; ++pass) {
            const int count = pass ? n_regions : std::min(n_regions, 4 * T);
            uint64_t c0 = 0, b0 = 0, c1 = 0, b1 = 0; oct_phmm_server_stats(srv, &c0, &b0);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t, count);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            oct_phmm_server_stats(srv, &c1, &b1);
            if (pass) printf("{\"mode\": \"server\", \"devices\": %d, \"threads\": %d, \"regions_per_s\": %.1f, \"M_loglik_per_s\": %.2f, \"regions_per_device_batch\": %.2f, \"failures\": %d}\n",
                             (int)devs.size(), T, n_regions / dt, (double)n_regions * R * H / dt / 1e6, (double)(c1 - c0) / (double)std::max<uint64_t>(1, b1 - b0), failures);
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
                oct_phmm_flank_state fl {40, 40}; oct_phmm_status st;
                if (oct_phmm_populate(hv, &rd, &hp, nullptr, &fl, nullptr, want.data(), &st) != OCT_PHMM_OK || want != g.out) ++bad;
            }
            oct_phmm_destroy(hv);
            printf("{\"mode\": \"server vs plain calls\", \"regions_compared\": %d, \"regions_that_differ\": %d}\n", n_check, bad);
            if (bad) failures += bad;
        }
        oct_phmm_server_destroy(srv);
    }
    for (int T : threads) {                                     // ---- one handle per calling thread ----
        std::vector<oct_phmm_handle*> hs(T, nullptr);
        oct_phmm_config c; oct_phmm_config_default(&c); c.max_indel_error = 16;
        for (auto& h : hs) if (oct_phmm_create(&c, &h) != OCT_PHMM_OK) { fprintf(stderr, "no device\n"); return 1; }
        int failures = 0;
        auto work = [&](int t, int count) {
            for (int i = t; i < count; i += T) {
                Region& g = regions[i];
                oct_phmm_reads rd {(uint32_t)g.mq.size(), g.rb.data(), g.q.data(), g.ro.data(), g.mq.data(), g.rev.data(), g.rbeg.data(), 0, nullptr};
                oct_phmm_haplotypes hp {(uint32_t)g.hbeg.size(), g.hb.data(), g.ho.data(), g.hbeg.data(), g.go.data(), g.ge.data(), g.mf.data(), g.pf.data(), g.mr.data(), g.pr.data()};
                oct_phmm_flank_state fl {40, 40}; oct_phmm_status st;
                if (oct_phmm_populate(hs[t], &rd, &hp, nullptr, &fl, nullptr, g.out.data(), &st) != OCT_PHMM_OK) ++failures;
            }
        };
        for (int pass = 0; pass < 2; ++pass) {                  // pass 0 warms the handles' pools
            const int count = pass ? n_regions : std::min(n_regions, 4 * T);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t, count);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (pass) printf("{\"mode\": \"handle per thread\", \"threads\": %d, \"regions_per_s\": %.1f, \"ms_per_call\": %.3f, \"M_loglik_per_s\": %.2f, \"failures\": %d}\n",
                             T, n_regions / dt, dt / n_regions * T * 1e3, (double)n_regions * R * H / dt / 1e6, failures);
        }
        for (auto h : hs) oct_phmm_destroy(h);
    }
    return 0;
}
