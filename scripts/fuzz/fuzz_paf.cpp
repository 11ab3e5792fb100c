// Mutation fuzzer of the native PAF / CIGAR front end (bossx_host_parse, include/bossx.h), built with
// AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (scripts/fuzz_paf.sh).  No device, no engine.
//
// Every iteration builds a small, consistent batch (random reads, random CIGARs, PAF columns that agree
// with them), applies 0-4 random mutations — truncated lines, deleted / duplicated / swapped fields and
// lines, huge and negative integers, non-integers, missing / malformed / unknown-type tags, CIGARs that
// are too long or too short for the read, garbage and NUL bytes, empty reads, reads cut short, other
// letters in reads, duplicate read names, barcodes out of range — and calls the parser with 1 or 3 threads.
// Pass = no sanitizer report, and the return code is one the reference's behaviour maps to:
//   0 ok | -3 ValueError | -4 KeyError | -5 IndexError | -8 TypeError | -9 AssertionError | -10 OverflowError
// (paf.py:18-75, 631-672; sequences.py:678-794; reference.py:138).  BOSSX_E_INVALID (-1) is what the
// entry point returns when its own consistency checks of the emit runs / tile segments / device-walk plans
// fail: a bug, reported with the input.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "bossx.h"

namespace {
struct Batch {
    std::string paf;
    std::vector<std::string> names, seqs;
    std::vector<int32_t> barcodes;
};

const char kOps[] = "MIDNSHP=XB";

std::string rand_seq(std::mt19937_64 &g, size_t n) {
    std::string s(n, 'A');
    for (auto &c : s) c = "ACGT"[g() & 3];
    return s;
}

Batch make_batch(std::mt19937_64 &g, int nb) {
    Batch b;
    const int n_reads = 1 + int(g() % 10);
    const int64_t clen[3] = {3000, 2500, 4};
    const char *cname[3] = {"ctgA", "ctgB", "ctgR"};
    for (int i = 0; i < n_reads; ++i) {
        std::string name = "r" + std::to_string(g() % 1000) + "_" + std::to_string(i);
        if (g() % 23 == 0) name = std::to_string(g() % 100);          // numeric names are normalised ("007" is "7")
        const int n_lines = g() % 9 == 0 ? 2 : 1;
        // CIGAR
        std::string cg;
        int64_t q = 0, r = 0;
        const int n_ops = 1 + int(g() % 12);
        for (int k = 0; k < n_ops; ++k) {
            const int64_t len = g() % 17 == 0 ? 0 : 1 + int64_t(g() % 60);
            const int pick = int(g() % 20);
            const char op = pick < 12 ? 'M' : pick < 15 ? 'I' : pick < 18 ? 'D' : kOps[3 + g() % 7];
            cg += std::to_string(len) + op;
            if (op != 'D') q += len;
            if (op != 'I') r += len;
        }
        const int64_t fl = int64_t(g() % 20), fr = int64_t(g() % 20);
        const int64_t qlen = fl + q + fr;
        b.names.push_back(name);
        b.seqs.push_back(rand_seq(g, size_t(qlen)));
        b.barcodes.push_back(int32_t(g() % uint64_t(nb)));
        for (int l = 0; l < n_lines; ++l) {
            const int c = int(g() % 3);
            const bool rev = g() & 1;
            const int64_t room = clen[c] - r;
            const int64_t ts = room > 0 ? int64_t(g() % uint64_t(room)) : 0;
            char buf[256];
            snprintf(buf, sizeof buf, "%s\t%lld\t%lld\t%lld\t%c\t%s\t%lld\t%lld\t%lld\t%lld\t%lld\t%d\ttp:A:%c\tcm:i:3\ts1:i:%lld\tcg:Z:",
                     name.c_str(), (long long)qlen, (long long)(rev ? fr : fl), (long long)(rev ? qlen - fl : qlen - fr), rev ? '-' : '+', cname[c], (long long)clen[c],
                     (long long)ts, (long long)(ts + r), (long long)r, (long long)(q + r), int(g() % 61), g() % 13 == 0 ? 'S' : 'P', (long long)r);
            b.paf += buf;
            b.paf += cg;
            b.paf += "\tAS:i:" + std::to_string(g() % 500);
            b.paf += '\n';
        }
    }
    if (g() & 1 && !b.paf.empty()) b.paf.pop_back();                   // with or without the final newline
    return b;
}

void mutate(std::mt19937_64 &g, Batch &b) {
    auto lines_of = [&]() {
        std::vector<std::string> v;
        size_t s = 0;
        while (s <= b.paf.size()) {
            size_t e = b.paf.find('\n', s);
            if (e == std::string::npos) { if (s < b.paf.size()) v.push_back(b.paf.substr(s)); break; }
            v.push_back(b.paf.substr(s, e - s));
            s = e + 1;
        }
        return v;
    };
    auto join = [&](const std::vector<std::string> &v) {
        std::string o;
        for (size_t i = 0; i < v.size(); ++i) { o += v[i]; if (i + 1 < v.size()) o += '\n'; }
        b.paf = o;
    };
    auto split = [](const std::string &l) {
        std::vector<std::string> f;
        size_t s = 0;
        while (true) { size_t e = l.find('\t', s); if (e == std::string::npos) { f.push_back(l.substr(s)); break; } f.push_back(l.substr(s, e - s)); s = e + 1; }
        return f;
    };
    auto unsplit = [](const std::vector<std::string> &f) { std::string o; for (size_t i = 0; i < f.size(); ++i) { o += f[i]; if (i + 1 < f.size()) o += '\t'; } return o; };
    std::vector<std::string> L = lines_of();
    const int kind = int(g() % 26);
    if (L.empty() && kind < 18) return;
    const size_t li = L.empty() ? 0 : g() % L.size();
    static const char *junk[] = {"", "abc", "-1", "99999999999999999999999999", "1e5", " 12 ", "1_0", "+7", "0x10", "\x00", "12\x00"};
    switch (kind) {
        case 0: { auto f = split(L[li]); f.resize(g() % (f.size() + 1)); L[li] = unsplit(f); join(L); break; }          // truncated line
        case 1: { auto f = split(L[li]); if (!f.empty()) f.erase(f.begin() + long(g() % f.size())); L[li] = unsplit(f); join(L); break; }
        case 2: { auto f = split(L[li]); if (!f.empty()) f[g() % f.size()] = junk[g() % 11]; L[li] = unsplit(f); join(L); break; }
        case 3: { auto f = split(L[li]); if (f.size() > 1) std::swap(f[g() % f.size()], f[g() % f.size()]); L[li] = unsplit(f); join(L); break; }
        case 4: L.insert(L.begin() + long(li), L[li]); join(L); break;                                                   // duplicated line
        case 5: L.erase(L.begin() + long(li)); join(L); break;
        case 6: L.insert(L.begin() + long(li), std::string(g() % 3, ' ')); join(L); break;                              // blank line
        case 7: { auto f = split(L[li]); for (auto &x : f) if (x.rfind("cg:Z:", 0) == 0) { const size_t p = 5 + g() % (x.size() - 4); x.insert(std::min(p, x.size()), std::string(1, char(g() % 256))); } L[li] = unsplit(f); join(L); break; }
        case 8: { auto f = split(L[li]); for (auto &x : f) if (x.rfind("cg:Z:", 0) == 0) x = "cg:Z:" + std::to_string(1 + g() % 50) + kOps[g() % 10] + x.substr(5); L[li] = unsplit(f); join(L); break; }
        case 9: { auto f = split(L[li]); for (auto &x : f) if (x.rfind("cg:Z:", 0) == 0 && x.size() > 7) x.resize(5 + g() % (x.size() - 5)); L[li] = unsplit(f); join(L); break; }
        case 10: { auto f = split(L[li]); for (auto &x : f) if (x.rfind("cg:Z:", 0) == 0) x += std::to_string(g()) + (g() & 1 ? "M" : ""); L[li] = unsplit(f); join(L); break; }
        case 11: { auto f = split(L[li]); f.push_back(g() & 1 ? "zz:q:1" : (g() & 1 ? "zz:Z:a:b" : "zz")); L[li] = unsplit(f); join(L); break; }
        case 12: { auto f = split(L[li]); for (auto &x : f) if (x.rfind("AS:", 0) == 0) x = g() & 1 ? "AS:f:1.5" : "AS:i:x"; L[li] = unsplit(f); join(L); break; }
        case 13: { auto f = split(L[li]); f.erase(std::remove_if(f.begin(), f.end(), [&](const std::string &x) { return x.rfind(g() & 1 ? "cg:" : "tp:", 0) == 0; }), f.end()); L[li] = unsplit(f); join(L); break; }
        case 14: if (!b.paf.empty()) b.paf[g() % b.paf.size()] = char(g() % 256); break;                                 // byte flip (NUL included)
        case 15: if (!b.paf.empty()) b.paf.erase(g() % b.paf.size(), 1 + g() % 8); break;
        case 16: if (!b.paf.empty()) b.paf.insert(g() % b.paf.size(), std::string(1 + g() % 4, char(g() % 256))); break;
        case 17: { auto f = split(L[li]); if (f.size() > 8) { f[7] = std::to_string(g() % 4000); f[8] = std::to_string(g() % 4000); } L[li] = unsplit(f); join(L); break; }
        case 18: if (!b.seqs.empty()) b.seqs[g() % b.seqs.size()].clear(); break;                                       // empty read
        case 19: if (!b.seqs.empty()) { auto &s = b.seqs[g() % b.seqs.size()]; s.resize(s.size() / 2); } break;        // read cut short
        case 20: if (!b.seqs.empty()) { auto &s = b.seqs[g() % b.seqs.size()]; if (!s.empty()) s[g() % s.size()] = "Nacgt\0-"[g() % 7]; } break;
        case 21: if (b.names.size() > 1) b.names[g() % b.names.size()] = b.names[g() % b.names.size()]; break;         // duplicate names
        case 22: if (!b.names.empty()) { const size_t k = g() % b.names.size(); b.names.erase(b.names.begin() + long(k)); b.seqs.erase(b.seqs.begin() + long(k)); b.barcodes.erase(b.barcodes.begin() + long(k)); } break;
        case 23: if (!b.barcodes.empty()) b.barcodes[g() % b.barcodes.size()] = int32_t(g() % 7) - 2; break;
        case 24: if (!b.names.empty()) b.names[g() % b.names.size()] = g() & 1 ? "" : std::string("a\0b", 3); break;
        default: break;
    }
}
}  // namespace

int main(int argc, char **argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 100000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 g(seed);
    long counts[16] = {0};
    const char *cnames[3] = {"ctgA", "ctgB", "ctgR"};
    const int64_t clens[3] = {3000, 2500, 4};
    const int32_t cflags[3] = {0, 0, BOSSX_CONTIG_REJECTED};
    std::vector<int32_t> s_read(64), s_contig(64), o_contig;
    std::vector<uint8_t> s_rev(64), o_code, o_bc;
    std::vector<int64_t> s_ts(64), s_te(64), s_ql(64), o_pos;
    for (long it = 0; it < iters; ++it) {
        const int nb = 1 + int(g() % 3);
        Batch b = make_batch(g, nb);
        const int n_mut = int(g() % 5);
        for (int k = 0; k < n_mut; ++k) mutate(g, b);
        const int32_t n = int32_t(b.names.size());
        std::vector<const char *> np(size_t(n) + 1), sp(size_t(n) + 1);
        std::vector<int64_t> nl(size_t(n) + 1), sl(size_t(n) + 1);
        for (int32_t i = 0; i < n; ++i) { np[size_t(i)] = b.names[size_t(i)].data(); nl[size_t(i)] = int64_t(b.names[size_t(i)].size()); sp[size_t(i)] = b.seqs[size_t(i)].data(); sl[size_t(i)] = int64_t(b.seqs[size_t(i)].size()); }
        if (size_t(n) + 1 > s_read.size()) { s_read.resize(size_t(n) + 1); s_contig.resize(size_t(n) + 1); s_rev.resize(size_t(n) + 1); s_ts.resize(size_t(n) + 1); s_te.resize(size_t(n) + 1); s_ql.resize(size_t(n) + 1); }
        bossx_batch_summary summ{s_read.data(), s_contig.data(), s_rev.data(), s_ts.data(), s_te.data(), s_ql.data()};
        int32_t n_rec = 0;
        int64_t aligned = 0;
        char err[512] = {0};
        const int threads = g() & 1 ? 1 : 3;
        // the text as an exact-size heap block (no terminator: reads past the end are the sanitizer's to find)
        std::vector<char> text(b.paf.begin(), b.paf.end());
        int rc = bossx_host_parse(cnames, clens, cflags, 3, nb, text.data(), text.size(), np.data(), nl.data(), sp.data(), sl.data(),
                                  b.barcodes.data(), n, 40, threads, &summ, &n_rec, &aligned, nullptr, nullptr, nullptr, nullptr, 0, err, sizeof err);
        if (rc == BOSSX_OK && aligned > 0) {       // the expansion the ingest kernels would perform
            o_contig.resize(size_t(aligned)); o_pos.resize(size_t(aligned)); o_code.resize(size_t(aligned)); o_bc.resize(size_t(aligned));
            rc = bossx_host_parse(cnames, clens, cflags, 3, nb, text.data(), text.size(), np.data(), nl.data(), sp.data(), sl.data(),
                                  b.barcodes.data(), n, 40, threads, &summ, &n_rec, &aligned, o_contig.data(), o_pos.data(), o_code.data(), o_bc.data(),
                                  aligned, err, sizeof err);
            for (int64_t e = 0; rc == BOSSX_OK && e < aligned; ++e)
                if (o_contig[size_t(e)] < 0 || o_contig[size_t(e)] > 1 || o_pos[size_t(e)] < 0 || o_pos[size_t(e)] >= clens[o_contig[size_t(e)]] || o_code[size_t(e)] > 4 || o_bc[size_t(e)] >= nb) {
                    fprintf(stderr, "iteration %ld (seed %llu): accepted batch expands outside the reference\n", it, (unsigned long long)seed);
                    return 2;
                }
        }
        const bool known = rc == BOSSX_OK || rc == BOSSX_E_PARSE || rc == BOSSX_E_KEY || rc == BOSSX_E_RANGE || rc == BOSSX_E_TYPE || rc == BOSSX_E_ASSERT || rc == BOSSX_E_OVERFLOW;
        if (!known) {
            fprintf(stderr, "iteration %ld (seed %llu): return code %d: %s\n---- PAF (%zu bytes) ----\n", it, (unsigned long long)seed, rc, err, b.paf.size());
            fwrite(b.paf.data(), 1, b.paf.size(), stderr);
            fprintf(stderr, "\n---- %d reads ----\n", n);
            for (int32_t i = 0; i < n; ++i) fprintf(stderr, "%s\t%zu\n", b.names[size_t(i)].c_str(), b.seqs[size_t(i)].size());
            return 1;
        }
        ++counts[-rc];
    }
    printf("%ld batches: ok %ld, ValueError %ld, KeyError %ld, IndexError %ld, TypeError %ld, AssertionError %ld, OverflowError %ld\n", iters, counts[0], counts[3], counts[4],
           counts[5], counts[8], counts[9], counts[10]);
    return 0;
}
