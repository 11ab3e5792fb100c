// ThreadSanitizer run of the host half of staging (no GPU): the two-phase WorkPool job of
// parse_paf_batch in its device-walk planning form, with caller tasks that write per-read flags
// and bump a counter while the calling thread groups and plans — what bossx_stage_batch_ptrs does
// with its gather / upload tasks.  Build + run: scripts/tsan_two_phase.sh
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include "engine.hpp"
using namespace bossx;
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    std::ifstream pf(argv[1]); std::stringstream ss; ss << pf.rdbuf(); std::string paf = ss.str();
    std::ifstream rf(argv[2]);
    std::string names; std::vector<int64_t> name_off{0}, seq_off{0};
    std::string nm; int64_t len;
    while (rf >> nm >> len) { names += nm; name_off.push_back(int64_t(names.size())); seq_off.push_back(seq_off.back() + len); }
    std::vector<ContigInfo> contigs(1);
    ContigInfo &c = contigs[0];
    c.name = "ecoli"; c.length = atoll(argv[3]); c.filt_index = 0;
    c.n_tiles = (c.length + kTileSites - 1) / kTileSites; c.T = c.length / kWindow; c.n_buckets = c.length / kBucket + 1;
    std::unordered_map<std::string, int32_t> index{{"ecoli", 0}};
    const int32_t n = int32_t(name_off.size() - 1);
    int bad = 0;
    for (int rep = 0; rep < 40; ++rep) {
        ParseInput in{paf.data(), paf.size(), names.data(), name_off.data(), seq_off.data(), nullptr, n, 200, 1};
        in.device_walk = true; in.n_tiles = c.n_tiles;
        in.n_threads = argc > 4 ? atoi(argv[4]) : 8;
        std::vector<uint8_t> dirty(size_t(n), 0);
        std::atomic<int> done{0};
        int at_collection = -1;
        const int n_extra = 11;
        in.extra_n = n_extra;
        in.extra_fn = [&](int t) {
            for (int32_t i = int32_t(int64_t(n) * t / n_extra); i < int32_t(int64_t(n) * (t + 1) / n_extra); ++i) dirty[size_t(i)] = uint8_t((i + rep) % 3 == 0);
            done.fetch_add(1);
        };
        in.after_pass1 = [&]() { at_collection = done.load(); };
        in.read_dirty = dirty.data();
        ParsedBatch pb; std::string err;
        const int rc = parse_paf_batch(in, contigs, index, nullptr, pb, err);
        size_t wrong = 0;
        for (size_t i = 0; i < pb.plans.size(); ++i)
            wrong += ((pb.plans[i].flags & kPlanCheckBases) != 0) != (dirty[size_t(pb.plan_read[i])] != 0);
        if (rc || at_collection != n_extra || wrong) { ++bad; printf("rep %d: rc=%d %s collected %d/%d wrong flags %zu\n", rep, rc, err.c_str(), at_collection, n_extra, wrong); }
    }
    printf("%d of 40 repetitions bad\n", bad);
    return bad ? 1 : 0;
}
