// Host-only timing of the PAF/CIGAR front end (no GPU): ./parse_bench paf.txt reads.txt L [threads]
// reads.txt: one "name<TAB>length" per line.  Build: see scripts/README.md.
#include <chrono>
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
    std::vector<int32_t> ri(n), ci(n); std::vector<uint8_t> rv(n); std::vector<int64_t> ts(n), te(n), ql(n);
    bossx_batch_summary sm{ri.data(), ci.data(), rv.data(), ts.data(), te.data(), ql.data()};
    std::vector<EmitOp> buf(ops_capacity_for(paf.size()));
    for (int rep = 0; rep < 8; ++rep) {
        ParseInput in{paf.data(), paf.size(), names.data(), name_off.data(), seq_off.data(), nullptr, n, 200, 1};
        in.ops_buf = buf.data(); in.ops_cap = buf.size();
        if (argc > 4) in.n_threads = atoi(argv[4]);
        ParsedBatch pb; std::string err;
        auto t0 = std::chrono::steady_clock::now();
        int rc = parse_paf_batch(in, contigs, index, &sm, pb, err);
        auto t1 = std::chrono::steady_clock::now();
        printf("rc=%d %s ops=%zu segs=%zu tiles=%zu emit=%llu  %.2f ms\n", rc, err.c_str(), pb.n_ops, pb.segs.size(),
               pb.tiles.size(), (unsigned long long)pb.total_emit, std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
}
