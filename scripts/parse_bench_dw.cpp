// Host-only timing of the front end's DEVICE-WALK path (lines -> records -> best mapping -> plans + (tile, barcode) groups: what a
// lone update's staging does on the host before the GPU can start its CIGAR walk), chr20+21 geometry:
//   ./parse_bench_dw paf.txt reads.txt [threads] [reps]      (reads.txt: "name<TAB>length" per line; BOSSX_STAGE_TIMING=1 prints the phases)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <sstream>
#include "engine.hpp"
using namespace bossx;
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::ifstream pf(argv[1]); std::stringstream ss; ss << pf.rdbuf(); std::string paf = ss.str();
    std::ifstream rf(argv[2]);
    std::string names; std::vector<int64_t> name_off{0}, seq_off{0}, seq_len;
    std::string nm; int64_t len;
    while (rf >> nm >> len) { names += nm; name_off.push_back(int64_t(names.size())); seq_len.push_back(len); seq_off.push_back(seq_off.back() + ((len + 1) & ~int64_t(1))); }
    const int64_t lens[2] = {64444167, 46709983};
    const char *cn[2] = {"chr20", "chr21"};
    std::vector<ContigInfo> contigs(2);
    std::unordered_map<std::string, int32_t> index;
    int64_t tile = 0, site = 0;
    for (int i = 0; i < 2; ++i) {
        ContigInfo &c = contigs[size_t(i)];
        c.name = cn[i]; c.length = lens[i]; c.filt_index = i;
        c.n_tiles = (c.length + kTileSites - 1) / kTileSites; c.T = c.length / kWindow; c.n_buckets = c.length / kBucket + 1;
        c.tile_off = tile; c.site_off = site; tile += c.n_tiles; site += c.n_tiles * kTileSites;
        index[c.name] = i;
    }
    const int32_t n = int32_t(name_off.size() - 1);
    std::vector<int32_t> ri(n), ci(n); std::vector<uint8_t> rv(n); std::vector<int64_t> ts(n), te(n), ql(n);
    bossx_batch_summary sm{ri.data(), ci.data(), rv.data(), ts.data(), te.data(), ql.data()};
    std::vector<uint8_t> dirty(size_t(n), 0);
    const int reps = argc > 4 ? atoi(argv[4]) : 12;
    std::vector<double> ms;
    ParsedBatch pb;                  // (kept between calls, as the engine keeps a slot's)
    for (int rep = 0; rep < reps; ++rep) {
        ParseInput in{paf.data(), paf.size(), names.data(), name_off.data(), seq_off.data(), nullptr, n, 200, 1};
        in.seq_len = seq_len.data();
        in.device_walk = true; in.read_dirty = dirty.data(); in.n_tiles = tile;
        if (argc > 3) in.n_threads = atoi(argv[3]);
        double t_early = 0;
        auto t0 = std::chrono::steady_clock::now();
        in.early_walk = [&](ParsedBatch &) { t_early = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        std::string err;
        int rc = parse_paf_batch(in, contigs, index, &sm, pb, err);
        auto t1 = std::chrono::steady_clock::now();
        const double t = std::chrono::duration<double, std::milli>(t1 - t0).count();
        ms.push_back(t_early);
        if (rep == reps - 1 || rc) printf("rc=%d %s plans=%zu groups=%zu touched=%zu emit=%llu | plans ready after %.3f ms, whole call %.3f ms\n", rc, err.c_str(), pb.plans.size(), pb.tiles.size(),
               pb.n_touched_tiles, (unsigned long long)pb.total_emit, t_early, t);
    }
    std::sort(ms.begin(), ms.end());
    printf("plans ready: median %.3f ms, min %.3f ms over %d calls\n", ms[ms.size() / 2], ms[0], reps);
}
