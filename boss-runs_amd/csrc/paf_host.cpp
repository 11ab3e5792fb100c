// Host-side front end of the ingest path: PAF text -> chosen mapping per read -> emit runs.
//
// Replaces, for the decision-update path only,
//   Paf.parse_PAF / _parse_content   /root/reference/boss/paf.py:631-672
//   PafLine.__init__ (12 columns + tags AS, cg, tp)         paf.py:18-75
//   Paf.choose_best_mapper                                  paf.py:709-722
//   CoverageConverter.convert_records / _parse_cigar        boss/runs/sequences.py:678-794
// The per-base expansion itself happens on the GPU (ingest_scatter_kernel); this file only
// walks the O(#CIGAR runs) structure.  Error behaviour mirrors the reference's exceptions
// (ValueError/AssertionError -> BOSSX_E_PARSE, KeyError -> BOSSX_E_KEY, IndexError ->
// BOSSX_E_RANGE); on error nothing is ingested.
#include "engine.hpp"
#include "bossx_py.h"

#include <sched.h>
#include <immintrin.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <cmath>
#include <cstdint>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string_view>
#include <thread>

#include <chrono>
#include <cstdio>
// phase times of parse_paf_batch, printed when BOSSX_STAGE_TIMING is set
#define PT(x) auto x = std::chrono::steady_clock::now()
#define PTMS(a, b) std::chrono::duration<double, std::milli>(b - a).count()
#define PTREPORT() do { if (getenv("BOSSX_STAGE_TIMING")) fprintf(stderr, "  [parse] names+pass1+groups %.3f  plan %.3f  walk/plans %.3f  merge %.3f ms\n", PTMS(T0, T1), PTMS(T1, T2), PTMS(T2, T3), PTMS(T3, T4)); } while (0)

namespace bossx {
namespace {

struct Rec {
    // (copies, not views into the text: the grouping — ONE thread — compares the names of 4000 records, and their copies lie side by side
    // in the heap of the worker that parsed them, while the names in the text lie a line, 1.4 KB, apart: as views the grouping took
    // 0.25 ms longer, a cache miss per record — tried in round 6)
    std::string qname, tname;
    int32_t cidx = -1;               // index of tname in the contig table (-1: not there), resolved by the line task
    int64_t qlen, qstart, qend, tstart, tend, alnlen, mapq, as;
    bool rev;
    const char *cg; size_t cg_len;   // points into the PAF text
    bool has_cg;
    bool cg_not_str = false;         // cg:i: / cg:f: whose value converts: the reference hands an int / float to re.findall (TypeError)
    // columns that are not integers stay strings in the reference (conv_type, paf.py:103-108) and only
    // matter where it computes with them: bit = column index
    uint32_t bad_cols = 0;
    // what np.array([(mapq, AS), ...], dtype=int) of choose_best_mapper does with this record (paf.py:716-718):
    // 0, ValueError (mapq stayed a str) or OverflowError (mapq or AS beyond int64)
    int key_err = 0;
};

// ---- text the way Python sees it -------------------------------------------------------------
// The PAF text crosses the C-ABI as UTF-8.  str.strip() and int() work on code points: they strip
// every Unicode whitespace character and take every decimal digit (category Nd), so "١٢" IS 12 for the
// reference (tests/golden/g_errors_fuzz.json holds such cases).
int utf8_at(std::string_view s, size_t i, uint32_t &cp) {
    const unsigned char c = static_cast<unsigned char>(s[i]);
    if (c < 0x80) { cp = c; return 1; }
    int n = c >= 0xf0 ? 4 : c >= 0xe0 ? 3 : c >= 0xc0 ? 2 : 0;
    if (!n || i + size_t(n) > s.size()) { cp = 0xfffd; return 1; }
    cp = c & (0xffu >> (n + 1));
    for (int k = 1; k < n; ++k) {
        const unsigned char d = static_cast<unsigned char>(s[i + size_t(k)]);
        if ((d & 0xc0) != 0x80) { cp = 0xfffd; return 1; }
        cp = (cp << 6) | (d & 0x3fu);
    }
    return n;
}

bool py_space(uint32_t cp) {          // str.isspace(), Unicode 13
    if (cp < 0x80) return cp == ' ' || (cp >= 0x09 && cp <= 0x0d) || (cp >= 0x1c && cp <= 0x1f);
    return cp == 0x85 || cp == 0xa0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200a) || cp == 0x2028 || cp == 0x2029 ||
           cp == 0x202f || cp == 0x205f || cp == 0x3000;
}

// value of a decimal digit (category Nd: 65 runs of ten consecutive code points, Unicode 13), or -1
int nd_digit(uint32_t cp) {
    if (cp < 0x80) return cp >= '0' && cp <= '9' ? int(cp - '0') : -1;
    static constexpr uint32_t zero[] = {
        0x660, 0x6F0, 0x7C0, 0x966, 0x9E6, 0xA66, 0xAE6, 0xB66, 0xBE6, 0xC66, 0xCE6, 0xD66, 0xDE6, 0xE50, 0xED0, 0xF20, 0x1040,
        0x1090, 0x17E0, 0x1810, 0x1946, 0x19D0, 0x1A80, 0x1A90, 0x1B50, 0x1BB0, 0x1C40, 0x1C50, 0xA620, 0xA8D0, 0xA900, 0xA9D0,
        0xA9F0, 0xAA50, 0xABF0, 0xFF10, 0x104A0, 0x10D30, 0x11066, 0x110F0, 0x11136, 0x111D0, 0x112F0, 0x11450, 0x114D0, 0x11650,
        0x116C0, 0x11730, 0x118E0, 0x11950, 0x11C50, 0x11D50, 0x11DA0, 0x16A60, 0x16B50, 0x1D7CE, 0x1D7D8, 0x1D7E2, 0x1D7EC,
        0x1D7F6, 0x1E140, 0x1E2F0, 0x1E950, 0x1FBF0};
    for (uint32_t z : zero) {
        if (cp < z) return -1;
        if (cp < z + 10) return int(cp - z);
    }
    return -1;
}

std::string_view strip(std::string_view s) {      // str.strip()
    size_t b = 0, e = s.size();
    while (b < e) {
        uint32_t cp;
        const int n = utf8_at(s, b, cp);
        if (!py_space(cp)) break;
        b += size_t(n);
    }
    while (e > b) {
        size_t k = e - 1;
        while (k > b && e - k < 4 && (static_cast<unsigned char>(s[k]) & 0xc0) == 0x80) --k;
        uint32_t cp;
        const int n = utf8_at(s, k, cp);
        if (k + size_t(n) != e || !py_space(cp)) break;
        e = k;
    }
    return s.substr(b, e - b);
}

// Python `int(s)` for a str: surrounding whitespace, optional sign, decimal digits with single
// underscores between them.  Python's integers do not overflow: `v` is clamped to +-2^62 (nothing on the
// path can hold more, and the checks downstream reject such a value like the reference does), `*big`
// says that the value does not fit an int64 (where numpy is handed it: OverflowError), `*canon` gets
// str(int(s)).
bool parse_int(std::string_view s, int64_t &v, bool *big = nullptr, std::string *canon = nullptr) {
    // (what a mapper writes: up to eighteen ASCII digits and nothing else — below 10^18, inside int64 and below the clamp further
    // down; seven columns and a tag per line go through here, and the general path below — code points, underscores, 128-bit
    // arithmetic, a std::string of digits — was the largest single item of a line task)
    if (!canon && !s.empty() && s.size() <= 18) {
        uint64_t x = 0;
        size_t i = 0;
        for (; i < s.size(); ++i) {
            const unsigned d = unsigned(static_cast<unsigned char>(s[i])) - unsigned('0');
            if (d > 9u) break;
            x = x * 10u + d;
        }
        if (i == s.size()) { v = int64_t(x); if (big) *big = false; return true; }
    }
    s = strip(s);
    size_t i = 0;
    const size_t n = s.size();
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
    if (i >= n) return false;
    unsigned __int128 x = 0;
    const unsigned __int128 cap = static_cast<unsigned __int128>(1) << 100;
    bool prev_us = true, any = false;           // (an underscore may not lead)
    std::string digits;
    while (i < n) {
        uint32_t cp;
        const int len = utf8_at(s, i, cp);
        i += size_t(len);
        if (cp == '_') { if (prev_us) return false; prev_us = true; continue; }
        const int d = nd_digit(cp);
        if (d < 0) return false;
        prev_us = false; any = true;
        x = x >= cap ? cap : x * 10 + unsigned(d);
        if (canon && (d || !digits.empty())) digits.push_back(char('0' + d));
    }
    if (!any || prev_us) return false;
    const unsigned __int128 lim = (static_cast<unsigned __int128>(1) << 63) - (neg ? 0 : 1);
    if (big) *big = x > lim;
    const int64_t kClamp = int64_t(1) << 62;
    const int64_t mag = x > static_cast<unsigned __int128>(kClamp) ? kClamp : int64_t(x);
    v = neg ? -mag : mag;
    if (canon) *canon = digits.empty() ? std::string("0") : (neg ? "-" + digits : digits);
    return true;
}

// Python `float(s)` as far as a tag value can reach it: whitespace, sign, inf / infinity / nan, decimal
// digits with single underscores between them, fraction, exponent.  (No hex floats: strtod takes them,
// float() does not.)
bool parse_float(std::string_view s, double &d) {
    s = strip(s);
    std::string t;
    t.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        uint32_t cp;
        const int len = utf8_at(s, i, cp);
        i += size_t(len);
        const int dg = nd_digit(cp);
        if (dg >= 0) t.push_back(char('0' + dg));
        else if (cp < 0x80) t.push_back(char(cp >= 'A' && cp <= 'Z' ? cp + 32 : cp));
        else return false;
    }
    size_t k = (!t.empty() && (t[0] == '+' || t[0] == '-')) ? 1 : 0;
    const std::string_view body = std::string_view(t).substr(k);
    if (body == "inf" || body == "infinity") { d = t[0] == '-' ? -HUGE_VAL : HUGE_VAL; return true; }
    if (body == "nan") { d = NAN; return true; }
    std::string u;
    u.reserve(t.size());
    bool digit_before = false;
    for (size_t i = 0; i < t.size(); ++i) {
        const char c = t[i];
        if (c == '_') {
            if (!digit_before || i + 1 >= t.size() || t[i + 1] < '0' || t[i + 1] > '9') return false;
            digit_before = false;
            continue;
        }
        if (!((c >= '0' && c <= '9') || c == '.' || c == 'e' || c == '+' || c == '-')) return false;
        digit_before = c >= '0' && c <= '9';
        u.push_back(c);
    }
    if (u.empty()) return false;
    bool has_digit = false;
    for (char c : u) has_digit |= (c >= '0' && c <= '9');
    if (!has_digit) return false;
    char *ep = nullptr;
    d = strtod(u.c_str(), &ep);
    return ep && *ep == '\0';
}

// PafLine stores names as str(conv_type(x, int)): "007" becomes "7" (paf.py:55-56, 103-108).
std::string normalise_name(std::string_view s) {
    if (!s.empty()) {
        // (int() takes nothing that starts with any other ASCII character: whitespace is stripped, then a sign or a digit — of any script)
        const unsigned char c = static_cast<unsigned char>(s[0]);
        if (!(c >= 0x80 || (c >= '0' && c <= '9') || c == '+' || c == '-' || c <= ' ')) return std::string(s);
    }
    int64_t v;
    std::string canon;
    if (parse_int(s, v, nullptr, &canon)) return canon;
    return std::string(s);
}

struct Group {
    Rec *best;               // the record chosen so far: stays where its line task put it (moving 4000 records with their strings was a third of the grouping)
    int64_t key_q, key_dp;
    int32_t read = -1;       // index of the read in the batch (-1: the name is not in the batch)
    int32_t n_recs = 1;      // kept records of this read
    int key_err = 0;         // first of them (line order) that np.array(..., dtype=int) of choose_best_mapper refuses, and how
};

struct CigarTable {
    bool ok[256];
    constexpr CigarTable() : ok() { for (const char *p = "MIDNSHP=XB"; *p; ++p) ok[static_cast<unsigned char>(*p)] = true; }
    constexpr bool operator[](unsigned char c) const { return ok[c]; }
};
constexpr CigarTable kCigarOp{};

// True if every byte of [p, p + n) is one of 'A', 'C', 'G', 'T' (the reference's base2int table,
// sequences.py:666-667; anything else becomes an out-of-range column index in np.add.at).
// Branch-free byte loop: vectorised by the compiler (this file is built with -O3).
bool all_acgt(const char *p, size_t n) {
    unsigned bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = static_cast<unsigned char>(p[i]);
        bad |= unsigned(!((c == 'A') | (c == 'C') | (c == 'G') | (c == 'T')));
    }
    return bad == 0;
}
// ... or one of the digits the reference counts as well: every byte minus '0' is its column index (sequences.py:790), so '0'..'3'
// are A C G T, '4' the deletion column, '7' becomes a deletion with the D operations (:803); anything else is out of range (IndexError)
bool all_counted(const char *p, size_t n) {
    unsigned bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = static_cast<unsigned char>(p[i]);
        const unsigned d = unsigned(c) - unsigned('0');
        bad |= unsigned(!((c == 'A') | (c == 'C') | (c == 'G') | (c == 'T') | (d <= 4u) | (d == 7u)));
    }
    return bad == 0;
}

}  // namespace

namespace {

struct LineOut {
    std::vector<Rec> recs;       // records that pass the min_len / primary filters, line order
    int64_t n_lines = 0;
    int64_t err_line = 0;        // 1-based within the range, 0 = none
    int err_code = BOSSX_OK;     // class of the reference's exception for that line
    std::string err_msg;
};

// Paf.parse_PAF / PafLine.__init__ over the lines of [p, end) (paf.py:18-75, 631-672), with the
// reference's exception classes (tests/golden/g_errors.json and g_errors_fuzz.json hold what it does,
// case by case):
//   fewer than 12 columns, blank lines included     IndexError     (f[i], paf.py:50-51)
//   tag that is not key:type:value                  ValueError     (x.split(":") unpacking, paf.py:95-98)
//   tag type other than i / A / f / Z               KeyError       (c[tag])
//   AS that int() does not take                     ValueError     (paf.py:62; the LAST AS tag of the line counts:
//                                                                   the tags are a dict), OverflowError for AS:f:inf
//   alignment block length that is not an integer   TypeError      (str < int, paf.py:666)
// Other non-integer columns stay strings there and only matter if the path computes with them
// (Rec::bad_cols; the unused ones — tlen, number of matches — never do).
// One pass over a line (round 6): the offsets of its tabs and colons up to the newline.  The fields used to be found by a memchr for the
// newline, a find per tab and — for every tag — finds for its colons, the third of them over the whole 1.2-KB CIGAR string: every byte of
// the text was looked at three times (1.05 of the 2.9 ms a 6-MB batch takes one thread; one pass: 0.43).  Returns the newline (or `end`);
// `n` > kLineSpecials: too many for the table — the caller splits that line the old way.
constexpr int kLineSpecials = 192;
struct LineSpecials { uint32_t off[kLineSpecials]; int n; };
__attribute__((target("avx2"))) const char *scan_line_avx2(const char *p, const char *end, LineSpecials &sp) {
    const __m256i vt = _mm256_set1_epi8('\t'), vc = _mm256_set1_epi8(':'), vn = _mm256_set1_epi8('\n');
    const char *q = p;
    int n = 0;
    for (; q + 32 <= end; q += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q));
        const unsigned mn = unsigned(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vn)));
        unsigned m = unsigned(_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v, vt), _mm256_cmpeq_epi8(v, vc))));
        if (mn) m &= (mn & (0u - mn)) - 1u;                     // only what stands in front of the first newline
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1u;
            if (n < kLineSpecials) sp.off[n] = uint32_t(q - p) + uint32_t(b);
            ++n;
        }
        if (mn) { sp.n = n; return q + __builtin_ctz(mn); }
    }
    for (; q < end; ++q) {
        const char c = *q;
        if (c == '\n') { sp.n = n; return q; }
        if (c == '\t' || c == ':') { if (n < kLineSpecials) sp.off[n] = uint32_t(q - p); ++n; }
    }
    sp.n = n;
    return end;
}
const char *scan_line_scalar(const char *p, const char *end, LineSpecials &sp) {
    int n = 0;
    const char *q = p;
    for (; q < end; ++q) {
        const char c = *q;
        if (c == '\n') break;
        if (c == '\t' || c == ':') { if (n < kLineSpecials) sp.off[n] = uint32_t(q - p); ++n; }
    }
    sp.n = n;
    return q;
}

void parse_lines(const char *p, const char *end, int64_t min_len, LineOut &lo, const std::unordered_map<std::string, int32_t> *contig_index = nullptr) {
    static const bool has_avx2 = __builtin_cpu_supports("avx2");
    const bool avx2 = has_avx2 && !getenv("BOSSX_PARSE_SCALAR");       // (the variable: tests hold the two scanners to each other)
    std::vector<std::string_view> f;
    std::vector<uint32_t> fc;        // per field: its first two colons (offsets inside the field, ~0u: none) and how many it holds
    LineSpecials sp;
    auto fail = [&](int code, const char *msg) { lo.err_line = lo.n_lines; lo.err_code = code; lo.err_msg = msg; };
    while (p < end) {
        const char *le = avx2 ? scan_line_avx2(p, end, sp) : scan_line_scalar(p, end, sp);
        const char *nl = le < end ? le : nullptr;
        std::string_view line = strip(std::string_view(p, size_t(le - p)));
        const char *const line0 = p;
        p = nl ? nl + 1 : end;
        ++lo.n_lines;
        f.clear(); fc.clear();
        if (sp.n <= kLineSpecials) {
            // the fields of the STRIPPED line, from the table: tabs outside it (str.strip() takes them) split nothing
            const uint32_t a = uint32_t(line.data() - line0), b = a + uint32_t(line.size());
            uint32_t fs = a, c1 = ~0u, c2 = ~0u, nc = 0;
            for (int k = 0; k < sp.n; ++k) {
                const uint32_t o = sp.off[k];
                if (o < a || o >= b) continue;
                if (line0[o] == '\t') {
                    f.push_back(std::string_view(line0 + fs, o - fs)); fc.push_back(c1); fc.push_back(c2); fc.push_back(nc);
                    fs = o + 1; c1 = c2 = ~0u; nc = 0;
                } else {
                    if (nc == 0) c1 = o - fs; else if (nc == 1) c2 = o - fs;
                    ++nc;
                }
            }
            f.push_back(std::string_view(line0 + fs, b - fs)); fc.push_back(c1); fc.push_back(c2); fc.push_back(nc);
        } else {
            size_t s = 0;
            while (true) {
                size_t t = line.find('\t', s);
                const std::string_view fld = t == std::string_view::npos ? line.substr(s) : line.substr(s, t - s);
                const size_t k1 = fld.find(':'), k2 = k1 == std::string_view::npos ? k1 : fld.find(':', k1 + 1);
                const bool third = k2 != std::string_view::npos && fld.find(':', k2 + 1) != std::string_view::npos;
                f.push_back(fld);
                fc.push_back(k1 == std::string_view::npos ? ~0u : uint32_t(k1)); fc.push_back(k2 == std::string_view::npos ? ~0u : uint32_t(k2));
                fc.push_back(k1 == std::string_view::npos ? 0u : k2 == std::string_view::npos ? 1u : third ? 3u : 2u);
                if (t == std::string_view::npos) break;
                s = t + 1;
            }
        }
        if (f.size() < 12) return fail(BOSSX_E_RANGE, ": fewer than 12 columns");
        Rec r;
        r.qlen = r.qstart = r.qend = r.tstart = r.tend = r.alnlen = r.mapq = 0;
        bool mapq_big = false;
        auto col = [&](size_t k, int64_t &dst, bool *big = nullptr) { if (!parse_int(f[k], dst, big)) { dst = 0; r.bad_cols |= 1u << k; } };
        col(1, r.qlen); col(2, r.qstart); col(3, r.qend); col(7, r.tstart); col(8, r.tend); col(10, r.alnlen); col(11, r.mapq, &mapq_big);
        r.rev = !(f[4].size() == 1 && f[4][0] == '+');   // paf.py:58
        r.as = 0; r.has_cg = false; r.cg = nullptr; r.cg_len = 0;
        bool primary = false;
        bool has_as = false;
        char as_typ = 'i';
        std::string_view as_val;
        for (size_t k = 12; k < f.size(); ++k) {
            std::string_view tag = f[k];
            const size_t c1 = fc[3 * k], c2 = fc[3 * k + 1];
            if (fc[3 * k + 2] != 2u) return fail(BOSSX_E_PARSE, ": malformed tag");   // x.split(':') unpack: exactly two colons
            std::string_view key = tag.substr(0, c1), typ = tag.substr(c1 + 1, c2 - c1 - 1), val = tag.substr(c2 + 1);
            if (!(typ.size() == 1 && (typ[0] == 'i' || typ[0] == 'A' || typ[0] == 'f' || typ[0] == 'Z')))
                return fail(BOSSX_E_KEY, ": unknown tag type");
            // a repeated key replaces the earlier one (the tags are collected in a dict, paf.py:95-101)
            if (key == "AS") {
                has_as = true; as_typ = typ[0]; as_val = val;
            } else if (key == "cg") {
                r.has_cg = true; r.cg = val.data(); r.cg_len = val.size();
                int64_t iv; double dv;
                r.cg_not_str = (typ[0] == 'i' && parse_int(val, iv)) || (typ[0] == 'f' && parse_float(val, dv));
            } else if (key == "tp") {
                primary = (val == "P");
            }
        }
        bool as_big = false;
        if (has_as) {
            // int(tags_parsed.get("AS", 0)): an int, a float (truncated) or a str that int() has to take
            bool ok = false;
            double d;
            if (as_typ == 'f' && parse_float(as_val, d)) {
                if (d != d) return fail(BOSSX_E_PARSE, ": AS is not a number");            // int(nan)
                if (d == HUGE_VAL || d == -HUGE_VAL) return fail(BOSSX_E_OVERFLOW, ": AS is infinite");   // int(inf)
                as_big = d >= 9223372036854775808.0 || d < -9223372036854775808.0;
                r.as = d >= 4e18 ? (int64_t(1) << 62) : d <= -4e18 ? -(int64_t(1) << 62) : int64_t(d);
                ok = true;
            }
            if (!ok && !parse_int(as_val, r.as, &as_big)) return fail(BOSSX_E_PARSE, ": AS is not an integer");
        }
        if (r.bad_cols & (1u << 10)) return fail(BOSSX_E_TYPE, ": alignment block length is not an integer");   // paf.py:666
        if (r.alnlen < min_len) continue;     // paf.py:666-667
        if (!primary) continue;               // paf.py:668-669
        r.key_err = (r.bad_cols & (1u << 11)) ? BOSSX_E_PARSE : (mapq_big || as_big) ? BOSSX_E_OVERFLOW : 0;
        r.qname = normalise_name(f[0]);
        r.tname = normalise_name(f[5]);
        if (contig_index) {              // (a read-only table: looked up here, in parallel, instead of in the serial pre-pass)
            if (!lo.recs.empty() && lo.recs.back().tname == r.tname) r.cidx = lo.recs.back().cidx;
            else { auto it = contig_index->find(r.tname); r.cidx = it != contig_index->end() ? it->second : -1; }
        }
        lo.recs.push_back(std::move(r));
    }
}

}  // namespace

namespace {

// One chosen mapping, resolved against the batch and the contig table (sequential pre-pass).
struct Plan {
    const Rec *rec;
    int64_t gi;              // index of the chosen mapping in record (first appearance) order
    int32_t read, cidx, bc;
    uint64_t emit0;          // index of its first emitted base in the batch-wide emit order
    size_t ops_at;           // where its emit runs start in the caller's buffer (upper-bound spacing)
    int64_t tlo, thi;        // the sites it emits to, [tlo, thi) (np's slice of the coverage array: both ends below zero wrap)
    int64_t q_first, q_len;  // read[q_first] is the first base the walk reads ('-': walks down from there), q_len how many the slice holds
    uint64_t span_add = 0;   // what it adds to the emit order / to the run capacity (0 for a mapping the walk is not given): summed in
    size_t ops_add = 0;      // record order once every mapping has been looked at (the look itself runs in parallel)
    // what the device walk's MapPlan needs of the record, copied while the pre-pass has it in its cache: the records lie all over the
    // line tasks' vectors, and the MapPlan loops used to take a second round of cache misses over them
    const char *cg = nullptr; size_t cg_len = 0; bool rev = false;
};

// seq[start:end] of a Python sequence of n elements: first index and length.
void py_slice(int64_t start, int64_t end, int64_t n, int64_t &first, int64_t &len) {
    const int64_t s = start < 0 ? std::max<int64_t>(start + n, 0) : std::min(start, n);
    const int64_t e = end < 0 ? std::max<int64_t>(end + n, 0) : std::min(end, n);
    first = s;
    len = e > s ? e - s : 0;
}

// The stretch of the read a mapping's CIGAR is laid over: int_seq[start:end] of sequences.py:707-716, 790 —
// of the read itself on '+', of its reverse complement on '-' (qlen - qend : qlen - qstart).  Python slicing:
// indices past the ends are clipped, negative ones count from the end (a qlen column smaller than the
// read — 0, 1, -7 — still selects bases; the reference goes on with those).
void read_slice(const Rec &r, int64_t seq_len, int64_t &q_first, int64_t &q_len) {
    int64_t first;
    if (r.rev) {
        py_slice(r.qlen - r.qend, r.qlen - r.qstart, seq_len, first, q_len);
        q_first = seq_len - 1 - first;       // element `first` of the reverse complement is base seq_len-1-first of the read
    } else {
        py_slice(r.qstart, r.qend, seq_len, first, q_len);
        q_first = first;
    }
}

struct WalkError {
    int64_t group = INT64_MAX;
    int code = BOSSX_OK;
    std::string msg;
};

struct WalkOut {
    const EmitOp *base = nullptr;
    size_t n_ops = 0;                      // runs written, contiguous from `base`
    std::vector<TileSeg> segs;             // op_lo / op_hi relative to the thread's base
    std::vector<uint32_t> seg_tile;
    std::vector<uint8_t> seg_bc;
    std::vector<uint64_t> emitted_per_contig;
    WalkError err;          // first ValueError / AssertionError / KeyError class failure (stops the walk)
    WalkError range_err;    // first IndexError class failure (raised later in the reference: the walk goes on)
};

}  // namespace

// CoverageConverter._parse_cigar + the span assertion (sequences.py:744-794, 732) as far as the TEXT of a
// CIGAR decides them, in the order the reference gets to them:
//   no (\d+)([MIDNSHP=XB]) token at all               ValueError      zip(*[]), :769
//   a run of 2^32 bases or more                       OverflowError   np.array(lengths, dtype=np.uint32), :770
//   (a run of 10^9 .. 2^32-1 bases: the reference allocates it and fails on the shape; ValueError here)
//   [q_ok false: qstart / qend stayed strings]        TypeError       int_seq[start:end], :790
//   query bases consumed != bases of the read slice   ValueError      cig_rep[notdel] = ..., :790
//   [t_ok false: tstart / tend stayed strings]        TypeError       min / max, :730-731
//   reference bases emitted != |tend - tstart|        AssertionError  :732
// The one authority on the CLASS of a CIGAR failure: the host walk and the device walk detect, this
// classifies.  (A read slice of exactly one base is broadcast by numpy over however many bases the CIGAR
// consumes: the reference goes on with that, and so does this path.)
int check_cigar_text(const char *cg, size_t n, int64_t q_len, int64_t span, bool q_ok, bool t_ok, std::string &msg) {
    size_t n_tok = 0;
    bool overflow = false, huge = false;
    unsigned __int128 consumed = 0, emitted = 0;
    const char *cp = cg, *ce = cg + n;
    while (cp < ce) {
        unsigned __int128 len = 0;
        const char *d0 = cp;
        unsigned d;
        while (cp < ce && (d = unsigned(*cp) - unsigned('0')) < 10u) { len = len >= (static_cast<unsigned __int128>(1) << 80) ? len : len * 10 + d; ++cp; }
        if (cp >= ce) break;                         // digits without a letter at the end
        const char op = *cp++;
        if (cp - 1 == d0 || !kCigarOp[static_cast<unsigned char>(op)]) continue;
        ++n_tok;
        if (len >= (static_cast<unsigned __int128>(1) << 32)) overflow = true;
        else if (len >= 1000000000u) huge = true;
        if (op != 'D') consumed += len;
        if (op != 'I') emitted += len;
    }
    if (n_tok == 0) { msg = "no CIGAR operation"; return BOSSX_E_PARSE; }
    if (overflow) { msg = "CIGAR run of 2^32 bases or more"; return BOSSX_E_OVERFLOW; }
    if (huge) { msg = "CIGAR run of 10^9 bases or more"; return BOSSX_E_PARSE; }
    if (!q_ok) { msg = "query coordinate is not an integer"; return BOSSX_E_TYPE; }
    // (a slice of exactly ONE base is broadcast by numpy over however many bases the CIGAR consumes: no error there, and
    // the walks read that one base under every run — kPlanBroadcast)
    if (consumed != static_cast<unsigned __int128>(q_len) && q_len != 1) {
        msg = "CIGAR consumes " + std::to_string(uint64_t(consumed)) + " query bases, the PAF columns select " + std::to_string(q_len);
        return BOSSX_E_PARSE;
    }
    if (!t_ok) { msg = "coordinate is not an integer"; return BOSSX_E_TYPE; }
    if (emitted != static_cast<unsigned __int128>(span)) {
        msg = "CIGAR spans " + std::to_string(uint64_t(emitted)) + " reference bases, PAF says " + std::to_string(span);
        return BOSSX_E_ASSERT;
    }
    return BOSSX_OK;
}

namespace {

// CIGAR walk of plans [p0, p1): emit runs written densely from `base`, tile segments collected.
void walk_plans(const ParseInput &in, const std::vector<ContigInfo> &contigs, const std::vector<Plan> &plans,
                size_t p0, size_t p1, EmitOp *base, WalkOut &wo) {
    wo.emitted_per_contig.assign(contigs.size(), 0);
    wo.base = base;
    EmitOp *w = base;
    for (size_t pi = p0; pi < p1; ++pi) {
        const Plan &pl = plans[pi];
        if (pl.cidx < 0) continue;           // ignored contig (pre-pass marks it)
        const Rec &r = *pl.rec;
        // a failure of the ValueError / OverflowError / AssertionError class: check_cigar_text says which
        auto fail = [&]() {
            wo.err.group = int64_t(pi);
            wo.err.code = check_cigar_text(r.cg, r.cg_len, pl.q_len, pl.thi - pl.tlo, true, true, wo.err.msg);
            if (!wo.err.code) { wo.err.code = BOSSX_E_INVALID; wo.err.msg = "internal: the CIGAR walk and its check disagree"; }
            wo.err.msg = "read '" + r.qname + "': " + wo.err.msg;
        };
        // IndexError class (np.add.at inside Contig.increment_coverage, reference.py:138): the
        // reference raises it after convert_records has gone through every read, so it never hides
        // a parse error of a later read — noted, and the walk continues
        auto range_fail = [&](std::string msg) {
            if (!wo.range_err.code) { wo.range_err.group = int64_t(pi); wo.range_err.code = BOSSX_E_RANGE; wo.range_err.msg = std::move(msg); }
        };
        const ContigInfo &c = contigs[size_t(pl.cidx)];
        const int64_t seq_b = in.seq_off[pl.read], seq_len = in.seq_len ? in.seq_len[pl.read] : in.seq_off[pl.read + 1] - seq_b;
        const int64_t tlo = pl.tlo, thi = pl.thi;
        // query walk: '+' reads seq[q_first + i]; '-' reads comp(seq[q_first - i]) (sequences.py:707-716)
        int64_t q = pl.q_first;
        const bool bcast = pl.q_len == 1;               // numpy broadcasts a one-base slice over every query-consuming run
        const int64_t qstep = bcast ? 0 : (r.rev ? -1 : 1);
        const int64_t q_need = pl.q_len;
        // quick look at the whole aligned stretch of the read; only a read that holds something
        // other than A/C/G/T there gets the per-run check below (insertions may hold anything)
        bool check_bases = false;
        if (in.seqs && q_need > 0) {
            int64_t lo = r.rev ? q - (q_need - 1) : q, hi = r.rev ? q : q + (q_need - 1);
            if (lo < 0) lo = 0;
            if (hi >= seq_len) hi = seq_len - 1;
            if (hi >= lo) check_bases = !all_acgt(in.seqs + seq_b + lo, size_t(hi - lo + 1));
        }
        int64_t consumed = 0, ref_pos = tlo;
        uint64_t cur_emit = pl.emit0;
        EmitOp *const first = w;
        const uint32_t meta_base = (uint32_t(pl.bc) << 8) | (r.rev ? kOpRev : 0u) | (bcast ? kOpBcast : 0u);
        const char *cp = r.cg, *ce = r.cg + r.cg_len;
        // the reference tokenises with re.findall(r"(\d+)([MIDNSHP=XB])") (sequences.py:672,767): whatever
        // is not digits directly followed by one of these letters is skipped, silently
        size_t n_tok = 0;
        bool failed = false;
        while (cp < ce) {
            int64_t len = 0;
            const char *d0 = cp;
            unsigned d;
            while (cp < ce && (d = unsigned(*cp) - unsigned('0')) < 10u) { len = len >= (int64_t(1) << 40) ? len : len * 10 + int64_t(d); ++cp; }
            if (cp >= ce) break;                         // digits without a letter at the end
            const char op = *cp++;
            if (cp - 1 == d0 || !kCigarOp[static_cast<unsigned char>(op)]) continue;
            ++n_tok;
            if (len >= int64_t(1000000000)) { failed = true; break; }
            if (len == 0) continue;
            if (op == 'I') {                       // consumes query, emits nothing (sequences.py:781)
                consumed += len; q += qstep * len;
                continue;
            }
            const bool del = (op == 'D');          // emits code 4, consumes nothing (sequences.py:782,793)
            if (!del) {
                if (!bcast && consumed + len > q_need) { failed = true; break; }       // more than the slice holds: a shape mismatch in cig_rep[notdel] = int_seq[start:end]
                const int64_t q_last = q + qstep * (len - 1);
                if (check_bases && !all_counted(in.seqs + seq_b + (r.rev ? q_last : q), size_t(bcast ? 1 : len)))
                    range_fail("read '" + r.qname + "': base other than A/C/G/T inside an aligned segment");
            }
            if (ref_pos + len > c.length)
                range_fail("read '" + r.qname + "': mapping extends past the end of " + c.name);
            const uint64_t site = uint64_t(c.site_off + ref_pos);
            w->emit_start = uint32_t(cur_emit);
            w->site_lo = uint32_t(site & 0xffffffffu);
            w->qpos = del ? 0u : uint32_t(seq_b + q);
            w->meta = uint32_t((site >> 32) & 0xffu) | meta_base | (del ? kOpDel : 0u);
            ++w;
            cur_emit += uint64_t(len);
            ref_pos += len;
            if (!del) { consumed += len; q += qstep * len; }
        }
        if (failed || n_tok == 0 || (consumed != q_need && !bcast) || ref_pos - tlo != thi - tlo) return fail();
        wo.emitted_per_contig[size_t(pl.cidx)] += uint64_t(thi - tlo);
        // split the read's emitted stretch at sweep-tile boundaries (padded site space)
        if (w > first) {
            const uint64_t site0 = uint64_t(c.site_off + tlo);
            const uint64_t site1 = uint64_t(c.site_off + thi);
            const uint64_t e0 = pl.emit0;
            const EmitOp *op = first, *const last = w - 1;
            for (uint64_t t = site0 / kTileSites; t * kTileSites < site1; ++t) {
                const uint64_t s_lo = std::max<uint64_t>(t * kTileSites, site0);
                const uint64_t s_hi = std::min<uint64_t>((t + 1) * kTileSites, site1);
                // pieces of at most kSegMax emitted bases, each with its exact emit-run range
                for (uint64_t p_lo = s_lo; p_lo < s_hi; p_lo += kSegMax) {
                    const uint64_t p_hi = std::min<uint64_t>(p_lo + kSegMax, s_hi);
                    TileSeg sg;
                    sg.e_lo = uint32_t(e0 + (p_lo - site0));
                    sg.e_hi = uint32_t(e0 + (p_hi - site0));
                    // emit_start is compared modulo 2^32 only within one read (< 2^32 bases)
                    while (op < last && uint32_t(op[1].emit_start - uint32_t(e0)) <= uint32_t(sg.e_lo - uint32_t(e0))) ++op;
                    sg.op_lo = uint32_t(op - base);
                    const EmitOp *oh = op;
                    while (oh < last && uint32_t(oh[1].emit_start - uint32_t(e0)) < uint32_t(sg.e_hi - uint32_t(e0))) ++oh;
                    sg.op_hi = uint32_t(oh - base);
                    wo.segs.push_back(sg);
                    wo.seg_tile.push_back(uint32_t(t));
                    wo.seg_bc.push_back(uint8_t(pl.bc));
                }
            }
        }
    }
    wo.n_ops = size_t(w - base);
}

}  // namespace

namespace {

class WorkPool {
  public:
    // Tasks [0, n_first) come first; the caller may go on once THEY are done (wait_first) while the
    // workers finish the rest, and collects the whole job later (wait_all).
    struct Job {
        const std::function<void(int)> *fn;
        int n, n_first;
        std::atomic<int> next_first{0}, next_rest{0}, pending_first{0}, pending{0};
    };
    static WorkPool &get() { static WorkPool p; return p; }
    // worker threads a job of n tasks would run on (besides the caller)
    int workers_for(int n) { ensure(n - 1); std::lock_guard<std::mutex> lk(m_); return int(threads_.size()); }
    void run(int n, const std::function<void(int)> &fn) {
        auto job = start(n, n, fn);
        if (job) wait_all(*job);
    }
    // Starts the job and returns once the first n_first tasks are done (the calling thread works on
    // those only); null when everything already ran inline.
    std::shared_ptr<Job> start(int n, int n_first, const std::function<void(int)> &fn) {
        if (n <= 0) return nullptr;
        ensure(n - 1);
        if (n == 1 || threads_.empty()) { for (int i = 0; i < n; ++i) fn(i); return nullptr; }
        auto job = std::make_shared<Job>();
        job->fn = &fn; job->n = n; job->n_first = std::min(n_first, n);
        job->next_rest.store(job->n_first);
        job->pending.store(n); job->pending_first.store(job->n_first);
        { std::lock_guard<std::mutex> lk(m_); job_ = job; ++gen_; }
        gen_a_.fetch_add(1, std::memory_order_release);
        cv_.notify_all();
        work(*job, /*first_only=*/job->n_first < n);
        // (spin briefly, then yield: the task waited for may sit on a worker that is runnable but has no CPU — other processes share the
        // host — and a caller that keeps spinning on its own CPU waits a scheduler time slice for it: 7-9 ms stalls of a 1.6-ms update)
        for (int s = 0; job->pending_first.load(std::memory_order_acquire) > 0; ++s) { if (s < 2000) __builtin_ia32_pause(); else std::this_thread::yield(); }
        return job;
    }
    static void wait_all(Job &job) {
        work(job, false);                                  // lend a hand with what is left
        while (job.pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    }
  private:
    static void work(Job &j, bool first_only = false) {
        int i;
        while ((i = j.next_first.fetch_add(1)) < j.n_first) {
            (*j.fn)(i);
            j.pending_first.fetch_sub(1, std::memory_order_release);
            j.pending.fetch_sub(1, std::memory_order_release);
        }
        if (first_only) return;
        while ((i = j.next_rest.fetch_add(1)) < j.n) { (*j.fn)(i); j.pending.fetch_sub(1, std::memory_order_release); }
    }
    // Workers never outnumber the host's other hardware threads, and an explicit BOSSX_PARSE_THREADS
    // bounds the pool as well as the line parse (tasks are pulled dynamically: fewer workers still
    // finish the job, the caller takes part).
    static int worker_cap() {
        int cap = std::min<int>(cpu_budget() - 1, 2 * parse_threads() - 1);
        if (getenv("BOSSX_PARSE_THREADS")) cap = std::min(cap, parse_threads() - 1);
        if (const char *e = getenv("BOSSX_POOL_THREADS")) cap = std::min(cap, std::max(atoi(e), 1) - 1);
        return std::max(cap, 0);
    }
    void ensure(int want) {
        want = std::min(want, worker_cap());
        std::lock_guard<std::mutex> lk(m_);
        while (int(threads_.size()) < want) threads_.emplace_back([this] { loop(); });
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // a short spin first: the parallel regions of one batch follow each other within microseconds
            for (int s = 0; s < 4000 && gen_a_.load(std::memory_order_acquire) == seen; ++s) __builtin_ia32_pause();
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    ~WorkPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    std::shared_ptr<Job> job_;
    uint64_t gen_ = 0;
    std::atomic<uint64_t> gen_a_{0};
    bool stop_ = false;
};

}  // namespace

void pool_run(int n_tasks, const std::function<void(int)> &fn) { WorkPool::get().run(n_tasks, fn); }

namespace {
// collects a two-phase job on every way out of the scope that started it
struct JobGuard {
    std::shared_ptr<WorkPool::Job> job;
    void finish() { if (job) { WorkPool::wait_all(*job); job.reset(); } }
    ~JobGuard() { finish(); }
};
}  // namespace

size_t ops_capacity_for(size_t paf_len) { return paf_len / 2 + paf_len / 16 + 64; }

// CPUs this process may actually use: the hardware threads, the affinity mask, and the container's CFS quota
// (cgroup v2 cpu.max, v1 cpu.cfs_quota_us) — a pool sized for 256 hardware threads inside a 16-CPU quota is
// throttled by the scheduler for the rest of every 100-ms period it overdraws.
int cpu_budget() {
    static const int budget = [] {
        unsigned hc = std::thread::hardware_concurrency();
        int n = int(hc ? hc : 1u);
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) n = std::min(n, k); }
        long long quota = -1, period = -1;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
        }
        if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, quota / period));
        return std::max(n, 1);
    }();
    return budget;
}

int parse_threads() {
    if (const char *e = getenv("BOSSX_PARSE_THREADS")) { const int v = atoi(e); if (v >= 1) return std::min(v, 64); }
    // (two of the budget are left to the calling thread's own work and the runtime's threads)
    return std::max(1, std::min(cpu_budget() - 2, 16));
}

int parse_paf_batch(const ParseInput &in, const std::vector<ContigInfo> &contigs,
                    const std::unordered_map<std::string, int32_t> &contig_index,
                    bossx_batch_summary *summary, ParsedBatch &out, std::string &err) {
    PT(T0);
    out.reset();                     // (keeps the vectors' memory: see ParsedBatch::reset)
    out.emitted_per_contig.assign(contigs.size(), 0);

    // read id -> index in the batch (built by one task of pass 1's parallel region): open addressing over
    // one flat array — no node allocations (4000 of them under the allocator's lock, next to sixteen
    // gathering threads, took 0.4 ms)
    struct NameIndex {
        std::vector<int32_t> slot;
        uint64_t mask = 0;
        const ParseInput *in = nullptr;
        static uint64_t hash(std::string_view s) {
            uint64_t a = 0, b = 0;
            const size_t n = s.size(), k = n < 8 ? n : 8;
            memcpy(&a, s.data(), k);
            memcpy(&b, s.data() + n - k, k);
            // (names of one batch share their first and last bytes' low bits — "r1003_17", "r1003_18": every input
            // bit has to reach the slot bits, so mix, multiply and take the HIGH half)
            uint64_t h = a ^ (b << 32 | b >> 32) ^ (uint64_t(n) << 56);
            h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
            h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
            return h;
        }
        std::string_view name(int32_t i) const { return std::string_view(in->names + in->name_off[i], size_t(in->name_off[i + 1] - in->name_off[i])); }
        void reserve(const ParseInput &input) {
            in = &input;
            size_t cap = 16;
            while (cap < size_t(input.n_reads) * 2 + 1) cap <<= 1;
            slot.assign(cap, -1);
            mask = cap - 1;
        }
        // dict semantics: a later duplicate replaces the earlier one; returns false if the name was there already
        bool insert_or_assign(int32_t i) {
            const std::string_view s = name(i);
            for (uint64_t p = hash(s) & mask;; p = (p + 1) & mask) {
                if (slot[p] < 0) { slot[p] = i; return true; }
                if (name(slot[p]) == s) { slot[p] = i; return false; }
            }
        }
        int32_t find(std::string_view s) const {
            if (slot.empty()) return -1;
            for (uint64_t p = hash(s) & mask;; p = (p + 1) & mask) {
                if (slot[p] < 0) return -1;
                if (name(slot[p]) == s) return slot[p];
            }
        }
    } read_index;
    bool dup_names = false;
    std::atomic<bool> index_ready{false};
    auto read_name = [&](int32_t i) { return std::string_view(in.names + in.name_off[i], size_t(in.name_off[i + 1] - in.name_off[i])); };
    auto build_read_index = [&]() {
        read_index.reserve(in);
        for (int32_t i = 0; i < in.n_reads; ++i)
            if (!read_index.insert_or_assign(i)) dup_names = true;     // later duplicates win, like a dict
        index_ready.store(true, std::memory_order_release);
    };
    // (the index is built by a worker NEXT TO the grouping below, which needs it only for a record
    // whose read is neither the previous record's nor the next one of the batch)
    auto wait_index = [&]() { for (int s = 0; !index_ready.load(std::memory_order_acquire); ++s) { if (s < 2000) __builtin_ia32_pause(); else std::this_thread::yield(); } };

    // ---- pass 1 (threads over line ranges): lines -> filtered records, in line order; then the
    // best record per query name, groups in first-appearance order --------------------------
    // One two-phase job: the line ranges and the read-name index first — the calling thread goes on
    // with the grouping and the plans as soon as THOSE are done — then whatever independent work the
    // caller brought along (gathering the reads, copying the text, their uploads), which the workers
    // finish meanwhile and which is collected before the device walk is launched.
    // (scratch that keeps its memory between calls, like ParsedBatch: fresh 100-350 KB vectors come from mmap and fault page by page)
    // (bound to plain references: a lambda that runs on a pool thread would otherwise see THAT thread's instance of a thread_local)
    static thread_local std::vector<Group> tl_groups;
    std::vector<Group> &groups = tl_groups;
    groups.clear();
    // Many small line ranges, pulled dynamically: a worker that wakes up late finds nothing left instead of
    // holding a sixteenth of the text back (one straggler used to set the pace: 0.28 ms against a mean of 0.06).
    size_t task_kb = 48;
    if (const char *e = getenv("BOSSX_LINE_TASK_KB")) task_kb = size_t(std::max(atoi(e), 4));
    int nt = in.n_threads > 0 ? in.n_threads : int(std::min<size_t>(256, std::max<size_t>(1, in.paf_len / (task_kb << 10))));
    if (in.paf_len < (size_t(1) << 16)) nt = 1;
    std::vector<const char *> cuts(size_t(nt) + 1, in.paf + in.paf_len);
    cuts[0] = in.paf;
    for (int t = 1; t < nt; ++t) {
        const char *c = in.paf + in.paf_len * size_t(t) / size_t(nt);
        if (c < cuts[size_t(t) - 1]) c = cuts[size_t(t) - 1];
        const char *nl = static_cast<const char *>(memchr(c, '\n', size_t(in.paf + in.paf_len - c)));
        cuts[size_t(t)] = nl ? nl + 1 : in.paf + in.paf_len;
    }
    std::vector<LineOut> los(static_cast<size_t>(nt));
    // (line task t has left its records: the calling thread groups them in line order WHILE the later tasks still run)
    std::unique_ptr<std::atomic<uint8_t>[]> line_done(new std::atomic<uint8_t>[size_t(nt)]);
    for (int t = 0; t < nt; ++t) line_done[size_t(t)].store(0, std::memory_order_relaxed);
    const bool trace = getenv("BOSSX_STAGE_TIMING") != nullptr;
    std::vector<double> tb, te;
    const auto r0 = std::chrono::steady_clock::now();
    const int n_tasks = nt + 1 + in.extra_n;
    if (trace) { tb.assign(size_t(n_tasks), 0.0); te.assign(size_t(n_tasks), 0.0); }
    const std::function<void(int)> pass1_fn = [&](int t) {
        if (trace) tb[size_t(t)] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
        if (t < nt) { parse_lines(cuts[size_t(t)], cuts[size_t(t) + 1], in.min_len, los[size_t(t)], &contig_index); line_done[size_t(t)].store(1, std::memory_order_release); }
        else if (t == nt) build_read_index();
        else in.extra_fn(t - nt - 1);
        if (trace) te[size_t(t)] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
    };
    // Streamed grouping (round 6): with enough workers the calling thread takes no line task — it consumes the tasks' records in
    // line order as they appear (the grouping is inherently serial; it used to START when the last line task had finished).
    // BOSSX_STREAM_GROUPING=0: the caller parses lines too and groups afterwards.
    const bool stream_env = !(getenv("BOSSX_STREAM_GROUPING") && atoi(getenv("BOSSX_STREAM_GROUPING")) == 0);
    const bool streamed = stream_env && in.device_walk && !in.summary_only && nt >= 8 && WorkPool::get().workers_for(n_tasks) >= 4;
    // Order in which the workers pull the tasks when the caller takes none: the name index first (the grouping waits for it at the
    // first record that is out of order — as task nt it started when the lines were done and ended 0.09 ms after them), then the
    // caller's `extra_first` tasks (copies of the PAF text: the device walk needs the text in HBM, and the upload needs nothing of
    // the parse), the lines, the rest of the caller's tasks.
    std::vector<int> task_order(static_cast<size_t>(n_tasks));
    for (int j = 0; j < n_tasks; ++j) task_order[size_t(j)] = j;
    if (streamed) {
        const int ef = std::min(std::max(in.extra_first, 0), in.extra_n);
        int j = 0;
        task_order[size_t(j++)] = nt;
        for (int e = 0; e < ef; ++e) task_order[size_t(j++)] = nt + 1 + e;
        for (int t = 0; t < nt; ++t) task_order[size_t(j++)] = t;
        for (int e = ef; e < in.extra_n; ++e) task_order[size_t(j++)] = nt + 1 + e;
    }
    const std::function<void(int)> pass1_ordered = [&](int j) { pass1_fn(task_order[size_t(j)]); };
    JobGuard pass1;                        // (declared after everything its tasks touch: collected first on every way out)
    auto collect_pass1 = [&]() {
        pass1.finish();
        if (trace) {
            auto stat = [&](int a, int b, const char *what) {
                if (b <= a) return;
                double s0 = 1e9, e1 = 0, dsum = 0, dmax = 0;
                for (int i = a; i < b; ++i) { s0 = std::min(s0, tb[size_t(i)]); e1 = std::max(e1, te[size_t(i)]); dsum += te[size_t(i)] - tb[size_t(i)]; dmax = std::max(dmax, te[size_t(i)] - tb[size_t(i)]); }
                fprintf(stderr, "  [pass1] %-6s %2d tasks: first start %.3f, last end %.3f, mean %.3f, max %.3f ms\n", what, b - a, s0, e1, dsum / (b - a), dmax);
            };
            stat(0, nt, "parse"); stat(nt, nt + 1, "index"); stat(nt + 1, n_tasks, "extra");
            fprintf(stderr, "  [pass1] all tasks collected after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count());
        }
        if (in.after_pass1) in.after_pass1();
    };
    pass1.job = WorkPool::get().start(n_tasks, streamed ? 0 : nt, pass1_ordered);       // (the caller goes on once the LINES are done — or, streamed, at once)
    if (!pass1.job) index_ready.store(true);                         // everything ran inline
    if (trace) fprintf(stderr, "  [pass1] lines + name index done after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count());
    if (!in.device_walk || in.summary_only) collect_pass1();     // nothing to overlap with on these paths
    {
        // best record per query name, groups in first-appearance order.  Names are resolved to batch
        // indices here; only names that are NOT in the batch — a KeyError for the reference — need a
        // map of their own.
        // A mapper reports its hits read by read, in the order the reads were handed over: the read
        // of a record is the previous record's or the next one in the batch far more often than not,
        // and comparing two names is much cheaper than hashing one — so the grouping starts right
        // away, OPTIMISTICALLY, while a worker still builds the name index: it waits for the index only
        // at the first record that is out of order.  Duplicate names in the batch (a dict keeps the last
        // one; impossible from a Python dict, possible through the C-ABI) show when the index is done:
        // the grouping is then repeated on freshly parsed lines with index lookups only.
        static thread_local std::vector<int32_t> tl_group_of_read;
        std::vector<int32_t> &group_of_read = tl_group_of_read;
        std::unordered_map<std::string_view, int32_t> group_of_unknown;
        int32_t cursor = -1;
        auto group_reset = [&]() {
            group_of_read.assign(size_t(in.n_reads), -1);
            group_of_unknown.clear();
            groups.clear();
            groups.reserve(size_t(in.n_reads) + 64);
            cursor = -1;
        };
        auto group_feed = [&](LineOut &lo, bool use_cursor) {
            for (Rec &r : lo.recs) {
                int32_t read = -1;
                const std::string_view qn(r.qname);
                if (use_cursor) {
                    // the previous record's read, or one of the next few of the batch (reads without a
                    // surviving mapping are skipped over): a handful of name compares, no hash
                    if (cursor >= 0 && read_name(cursor) == qn) read = cursor;
                    else
                        for (int32_t k = cursor + 1, ke = std::min<int64_t>(int64_t(cursor) + 25, in.n_reads); k < ke; ++k)
                            if (read_name(k) == qn) { read = k; break; }
                }
                if (read < 0) {
                    wait_index();
                    read = read_index.find(qn);
                }
                if (read >= 0) cursor = read;
                int32_t *slot = nullptr;
                int32_t unknown_slot = -1;
                if (read >= 0) slot = &group_of_read[size_t(read)];
                else {
                    auto it = group_of_unknown.find(qn);
                    if (it != group_of_unknown.end()) unknown_slot = it->second;
                    slot = &unknown_slot;
                }
                const int key_err = r.key_err;
                if (*slot < 0) {
                    groups.push_back(Group{&r, r.mapq, r.as, read, 1, key_err});
                    if (read >= 0) *slot = int32_t(groups.size() - 1);
                    else group_of_unknown.emplace(std::string_view(r.qname), int32_t(groups.size() - 1));     // (the key views the FIRST record's name: it stays in place)
                } else {
                    Group &g = groups[size_t(*slot)];
                    ++g.n_recs;
                    if (!g.key_err) g.key_err = key_err;
                    // argsort by (mapq, AS), last element wins; stable for ties (paf.py:716-721)
                    if (r.mapq > g.key_q || (r.mapq == g.key_q && r.as >= g.key_dp)) {
                        g.key_q = r.mapq; g.key_dp = r.as;
                        g.best = &r;
                    }
                }
            }
        };
        group_reset();
        int64_t line_base = 0;
        for (int t = 0; t < nt; ++t) {          // line order: the first failing line in file order is the one reported
            for (int sp = 0; !line_done[size_t(t)].load(std::memory_order_acquire); ++sp) { if (sp < 4000) __builtin_ia32_pause(); else std::this_thread::yield(); }
            LineOut &lo = los[size_t(t)];
            if (lo.err_line) { err = "PAF line " + std::to_string(line_base + lo.err_line) + lo.err_msg; return lo.err_code; }
            line_base += lo.n_lines;
            group_feed(lo, true);
        }
        wait_index();
        if (dup_names) {
            for (int t = 0; t < nt; ++t) { los[size_t(t)] = LineOut(); parse_lines(cuts[size_t(t)], cuts[size_t(t) + 1], in.min_len, los[size_t(t)], &contig_index); }
            group_reset();
            for (int t = 0; t < nt; ++t) group_feed(los[size_t(t)], false);
        }
    }

    PT(T1);
    // ---- pass 2a: resolve every chosen mapping, fill the summary, lay out the emit order.  A failing
    // record stops the pre-pass; records before it are still walked so that the first failure in record
    // order is the one reported (the reference raises inside its per-record loop, sequences.py:700-735).
    // Round 6: what a record needs of its predecessors is two running sums; everything else about it —
    // the checks, the slice of the read, the range of sites — is looked at for all records IN PARALLEL
    // (ranges of records on the pool's workers, idle by now), each range stopping at its own first failure;
    // the first failure overall, the running sums and the first IndexError-class finding follow in one
    // cheap serial sweep.  Small batches and hosts without workers take the same code as one range.
    static thread_local std::vector<Plan> tl_plans;
    std::vector<Plan> &plans = tl_plans;
    plans.clear();
    plans.resize(groups.size());
    struct RangeState {
        WalkError fail; bool fail_counted = false;      // first failure of the range (fail_counted: the record had reached the summary)
        WalkError range;                                // first IndexError-class finding of the range
    };
    auto plan_one = [&](size_t gi, RangeState &rs) -> bool {       // false: the range stops here
        const Rec &r = *groups[gi].best;
        auto pre_fail = [&](int code, std::string msg, bool counted) { rs.fail.group = int64_t(gi); rs.fail.code = code; rs.fail.msg = std::move(msg); rs.fail_counted = counted; };
        if (groups[gi].n_recs > 1 && groups[gi].key_err) {
            pre_fail(groups[gi].key_err, "read '" + r.qname + "': mapping quality / AS of one of its mappings is not an int64", false);   // choose_best_mapper, paf.py:716-718
            return false;
        }
        if (groups[gi].read < 0) {
            pre_fail(BOSSX_E_KEY, "read '" + r.qname + "' is mapped in the PAF but absent from the batch", false);
            return false;               // seqs[rec.qname], sequences.py:708/713
        }
        const int32_t read = groups[gi].read;
        const int32_t cidx = r.cidx;
        // (record gi is entry gi: nothing is skipped in front of a failure, and a failing batch's summary is void — records BEHIND a
        // read that is not in the batch may have an index beyond the caller's arrays, which hold one entry per read of the batch)
        if (summary && gi < size_t(in.n_reads)) {
            summary->read_idx[gi] = read;
            summary->contig_idx[gi] = cidx;
            summary->rev[gi] = r.rev ? 1 : 0;
            summary->tstart[gi] = r.tstart;
            summary->tend[gi] = r.tend;
            summary->qlen[gi] = r.qlen;
        }
        Plan &pl = plans[gi];
        pl = Plan{&r, int64_t(gi), read, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (in.summary_only) return true;
        // columns that stayed strings: the reference computes qlen - qend / qlen - qstart on '-' mappings
        // (sequences.py:709-710) before it looks at the CIGAR, slices with qstart / qend on '+' ones only after
        // tokenising it (:790), and takes min / max of the target coordinates after that (:730-731) — a
        // TypeError each, at its place in that order (check_cigar_text); a '+' mapping never looks at qlen
        const uint32_t bad_q = r.bad_cols & ((1u << 2) | (1u << 3) | (r.rev ? (1u << 1) : 0u));
        const uint32_t bad_t = r.bad_cols & ((1u << 7) | (1u << 8));
        if (r.rev && bad_q) { pre_fail(BOSSX_E_TYPE, "read '" + r.qname + "': query coordinate is not an integer", true); return false; }
        if (!r.has_cg) {
            pre_fail(BOSSX_E_ASSERT, "read '" + r.qname + "': mapping without cg tag", true);   // assert rec.cigar is not None
            return false;
        }
        if (r.cg_not_str) { pre_fail(BOSSX_E_TYPE, "read '" + r.qname + "': cg tag is not a string", true); return false; }      // re.findall on an int / float
        const bool local = cidx >= 0 && !contigs[size_t(cidx)].rejected && !contigs[size_t(cidx)].remote;
        const bool remote = cidx >= 0 && !contigs[size_t(cidx)].rejected && contigs[size_t(cidx)].remote;
        pl.cidx = local ? cidx : -1;
        const int64_t seq_len = in.seq_len ? in.seq_len[read] : in.seq_off[read + 1] - in.seq_off[read];
        if (!bad_q) read_slice(r, seq_len, pl.q_first, pl.q_len);
        pl.tlo = std::min(r.tstart, r.tend);
        pl.thi = std::max(r.tstart, r.tend);
        // Where the increments go (Contig.increment_coverage: np.add.at(coverage[start:end], (arange(end - start), ...)),
        // reference.py:138): a slice, so an end past the contig clips it (IndexError: the walk reports it) and two
        // NEGATIVE ends count from the contig's end — inside the contig the reference goes on there, anything
        // else leaves a slice shorter than the mapping (IndexError)
        bool range_bad = false;
        if (local && !bad_t && pl.thi - pl.tlo > int64_t(INT32_MAX)) range_bad = true;      // (longer than any contig: checked here, never laid out)
        else if (local && !bad_t && pl.tlo < 0) {
            const int64_t clen = contigs[size_t(cidx)].length;
            if (pl.thi == pl.tlo) range_bad = false;                 // nothing to add
            else if (pl.thi < 0 && pl.tlo + clen >= 0) { pl.tlo += clen; pl.thi += clen; }
            else range_bad = true;
        }
        // A mapping the walk is not given — its contig is not one of ours (the reference converts EVERY
        // chosen mapping and drops those afterwards, sequences.py:700-735 / core.py:83-86), a coordinate stayed
        // a string, or its sites lie before the contig's start — is still held to the CIGAR checks, here.
        // (Mappings on another device's contigs are that device's to check.)
        const bool walk_it = local && !bad_q && !bad_t && !range_bad && !(pl.thi == pl.tlo && pl.tlo < 0);
        if (!walk_it && !remote) {
            std::string msg;
            const int code = check_cigar_text(r.cg, r.cg_len, pl.q_len, pl.thi - pl.tlo, !bad_q, !bad_t, msg);
            if (code) { pre_fail(code, "read '" + r.qname + "': " + msg, true); return false; }
            if (range_bad && !rs.range.code) {
                rs.range.group = int64_t(gi); rs.range.code = BOSSX_E_RANGE;
                rs.range.msg = "read '" + r.qname + "': mapping lies outside " + contigs[size_t(cidx)].name;
            }
        }
        if (!walk_it) {
            pl.cidx = -1;               // core.py:83-86: only (local) contigs_filt receive coverage
        } else {
            pl.bc = in.barcodes ? in.barcodes[read] : 0;
            if (pl.bc < 0 || pl.bc >= in.nbarcodes) {
                if (!rs.range.code) { rs.range.group = int64_t(gi); rs.range.code = BOSSX_E_RANGE; rs.range.msg = "read '" + r.qname + "': barcode index out of range"; }
                pl.bc = 0;
            }
            pl.span_add = uint64_t(pl.thi - pl.tlo);
            pl.ops_add = r.cg_len / 2 + 1;            // every run is at least one digit + one letter
            pl.cg = r.cg; pl.cg_len = r.cg_len; pl.rev = r.rev;
        }
        return true;
    };
    // ranges of records: as many as the pool has hands for, none below ~256 records (BOSSX_PLAN_RANGE: tests force small ranges)
    const char *plan_range_env = getenv("BOSSX_PLAN_RANGE");
    const size_t plan_range_min = plan_range_env ? size_t(std::max(atoi(plan_range_env), 1)) : 256;
    // (One range unless asked for: on the GPU boxes the ranges measured SLOWER than the single one — 0.07-0.13 + 0.19-0.21 ms against
    // 0.09 + 0.07 — because the workers are still gathering reads at this point and a job waits for the slowest hand.  BOSSX_PLAN_PARALLEL=1,
    // or an explicit thread count (tests), takes the ranges.)
    const int pool_hands = in.n_threads > 0 ? in.n_threads : (getenv("BOSSX_PLAN_PARALLEL") ? WorkPool::get().workers_for(64) + 1 : 1);
    const size_t n_ranges = std::max<size_t>(1, std::min<size_t>(size_t(pool_hands), groups.size() / plan_range_min));
    std::vector<RangeState> range_state(n_ranges);
    auto range_lo = [&](size_t k) { return groups.size() * k / n_ranges; };
    {
        const std::function<void(int)> plan_range = [&](int k) {
            RangeState &rs = range_state[size_t(k)];
            for (size_t gi = range_lo(size_t(k)), ge = range_lo(size_t(k) + 1); gi < ge; ++gi) {
                // (the chosen records lie all over the line tasks' vectors: ask for the one eight ahead while this one is looked at)
                if (gi + 8 < ge) { const char *nx = reinterpret_cast<const char *>(groups[gi + 8].best); __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128); }
                if (!plan_one(gi, rs)) break;
            }
        };
        if (n_ranges == 1) plan_range(0); else pool_run(int(n_ranges), plan_range);
    }
    // the serial sweep: first failure in record order, first IndexError-class finding in front of it, the running sums
    WalkError pre_err, pre_range;
    bool pre_err_counted = false;
    for (RangeState &rs : range_state)
        if (rs.fail.code && rs.fail.group < pre_err.group) { pre_err = std::move(rs.fail); pre_err_counted = rs.fail_counted; }
    const size_t n_ok = pre_err.code ? size_t(pre_err.group) : groups.size();      // records in front of the first failure
    for (RangeState &rs : range_state)
        if (rs.range.code && rs.range.group < int64_t(n_ok) && rs.range.group < pre_range.group) pre_range = std::move(rs.range);
    plans.resize(n_ok);
    uint64_t cur_emit = 0;
    size_t ops_at = 0;
    for (Plan &pl : plans) { pl.emit0 = cur_emit; pl.ops_at = ops_at; cur_emit += pl.span_add; ops_at += pl.ops_add; }
    const int32_t n_rec = int32_t(n_ok) + (pre_err_counted ? 1 : 0);
    if (in.summary_only) plans.clear();

    PT(T2);
    if (in.device_walk && !in.summary_only) {
        // The CIGAR walk runs on the GPU (front_end.hip.inc): hand over one MapPlan per mapping that
        // lands on a local contig, the (tile, barcode) groups its emitted stretch touches, and the
        // buffer sizes the device needs.  Errors of the pre-pass are reported by the caller after the
        // device walk, so that an earlier record's CIGAR error still wins (record order).
        std::vector<uint64_t> &marks = out.marks;
        const size_t n_keys = size_t(in.n_tiles) * size_t(in.nbarcodes);
        if (n_keys >= (size_t(1) << 32)) { err = "more than 2^32 (tile, barcode) keys"; return BOSSX_E_RANGE; }
        marks.assign((n_keys + 63) / 64, 0);
        // which plans travel: those on a local contig, in front of the first mapping of 2^32 bases or more (a span no contig holds:
        // the reference fails on it as well; records after it are not walked).  Their slots in the upload are their ranks.
        static thread_local std::vector<uint32_t> tl_slot;
        std::vector<uint32_t> &slot = tl_slot;
        slot.assign(plans.size(), UINT32_MAX);
        uint32_t n_up = 0;
        for (size_t i = 0; i < plans.size(); ++i) {
            const Plan &pl = plans[i];
            if (pl.cidx < 0) continue;
            if (pl.cg_len > size_t(UINT32_MAX) || pl.thi - pl.tlo > int64_t(UINT32_MAX)) {
                const Rec &r = *pl.rec;
                if (!pre_err.code || pl.gi < pre_err.group) {
                    pre_err.group = pl.gi;
                    pre_err.code = check_cigar_text(r.cg, r.cg_len, pl.q_len, pl.thi - pl.tlo, true, true, pre_err.msg);
                    if (!pre_err.code) { pre_err.code = BOSSX_E_RANGE; pre_err.msg = "mapping of 2^32 bases or more"; }
                    pre_err.msg = "read '" + r.qname + "': " + pre_err.msg;
                }
                break;
            }
            slot[i] = n_up++;
        }
        out.plans.resize(n_up);
        out.plan_read.resize(n_up);
        out.plan_gi.resize(n_up);
        // the MapPlans and the bitmap of touched (tile, barcode) keys: ranges of plans in parallel (bits set with atomic ORs)
        const size_t n_cr = std::max<size_t>(1, std::min<size_t>(size_t(pool_hands), plans.size() / plan_range_min));
        std::vector<size_t> seg_part(n_cr, 0);
        std::vector<std::vector<uint64_t>> emit_part(n_cr, std::vector<uint64_t>(contigs.size(), 0));
        const std::function<void(int)> map_range = [&](int k) {
            size_t seg_sum = 0;
            std::vector<uint64_t> &emitted = emit_part[size_t(k)];
            for (size_t i = plans.size() * size_t(k) / n_cr, ie = plans.size() * (size_t(k) + 1) / n_cr; i < ie; ++i) {
                if (slot[i] == UINT32_MAX) continue;
                const Plan &pl = plans[i];
                const ContigInfo &c = contigs[size_t(pl.cidx)];
                const int64_t seq_b = in.seq_off[pl.read], seq_len = in.seq_len ? in.seq_len[pl.read] : in.seq_off[pl.read + 1] - seq_b;
                const int64_t tlo = pl.tlo, thi = pl.thi;
                const int64_t q = pl.q_first, q_need = pl.q_len;
                MapPlan mp{};
                mp.cg_off = uint32_t(size_t(pl.cg - in.paf) + size_t(in.paf_base));
                mp.cg_len = uint32_t(pl.cg_len);
                mp.emit0 = uint32_t(pl.emit0);
                mp.span = uint32_t(thi - tlo);
                mp.site0 = uint64_t(c.site_off + tlo);
                mp.q_rel = int32_t(q);
                mp.q0 = uint32_t(seq_b + (q >= 0 && q < seq_len ? q : 0));
                mp.q_need = uint32_t(q_need);
                mp.seq_b = uint32_t(seq_b);
                mp.seq_len = uint32_t(seq_len);
                const int64_t room = c.length - tlo;
                mp.room = uint32_t(room < 0 ? 0 : (room > int64_t(UINT32_MAX) ? int64_t(UINT32_MAX) : room));
                mp.ops_cap = uint32_t(pl.cg_len / 2 + 1);
                mp.flags = uint32_t(pl.bc & 0xff) | (pl.rev ? kPlanRev : 0u) | (q_need == 1 ? kPlanBroadcast : 0u);      // (kPlanCheckBases: below, once the reads have been looked at)
                // groups: every sweep tile the stretch [site0, site0 + span) touches, for this barcode.
                // A stretch that runs past its contig (an IndexError reported by the device walk) is
                // clipped here so that no key outside the table is marked.
                int64_t s_end = c.site_off + thi;
                const int64_t c_end = c.site_off + c.n_tiles * kTileSites;
                if (s_end > c_end) s_end = c_end;
                uint32_t ntile = 0;
                for (int64_t t = int64_t(mp.site0) / kTileSites; thi > tlo && t * kTileSites < s_end; ++t) {      // (a mapping of insertions only emits nothing)
                    const size_t key = size_t(t) * size_t(in.nbarcodes) + size_t(pl.bc);
                    if (key < n_keys) {
                        const uint64_t bit = 1ull << (key & 63);
                        if (n_cr == 1) marks[key >> 6] |= bit; else __atomic_fetch_or(&marks[key >> 6], bit, __ATOMIC_RELAXED);
                    }
                    ++ntile;
                }
                mp.seg_cap = mp.span / kSegMax + ntile + 1;
                seg_sum += mp.seg_cap;
                emitted[size_t(pl.cidx)] += uint64_t(thi - tlo);
                out.plans[slot[i]] = mp;
                out.plan_read[slot[i]] = pl.read;
                out.plan_gi[slot[i]] = pl.gi;
            }
            seg_part[size_t(k)] = seg_sum;
        };
        if (n_cr == 1) map_range(0); else pool_run(int(n_cr), map_range);
        size_t seg_cap = 0;
        for (size_t k = 0; k < n_cr; ++k) {
            seg_cap += seg_part[k];
            for (size_t c = 0; c < contigs.size(); ++c) out.emitted_per_contig[c] += emit_part[k][c];
        }
        if (cur_emit >= (1ull << 32) - kEmitTile) {
            err = "batch too large: more than 2^32 aligned bases";
            return BOSSX_E_RANGE;
        }
        PT(T2a);
        // The (tile, barcode) groups = the set bits, in key order: the LIST is built on the device (build_groups_kernel) from the
        // bitmap and the per-word ranks — the host only counts (round 5 pushed 13.6 k TileRefs here, 0.1 ms of the serial part,
        // and uploaded 218 KB).  Rank of a key among the marked ones = index of its group.
        {
            std::vector<uint32_t> &rank = out.rank;
            rank.resize(marks.size() + 1);
            uint32_t acc = 0;
            for (size_t w = 0; w < marks.size(); ++w) { rank[w] = acc; acc += uint32_t(__builtin_popcountll(marks[w])); }
            rank[marks.size()] = acc;
            out.n_groups = acc;
            if (in.nbarcodes == 1) out.n_touched_tiles = acc;
            else {
                out.n_touched_tiles = 0;
                uint32_t last_tile = UINT32_MAX;
                for (size_t w = 0; w < marks.size(); ++w) {
                    uint64_t bits = marks[w];
                    while (bits) {
                        const uint32_t key = uint32_t((w << 6) + size_t(__builtin_ctzll(bits)));
                        bits &= bits - 1;
                        const uint32_t t = key / uint32_t(in.nbarcodes);
                        if (t != last_tile) { ++out.n_touched_tiles; last_tile = t; }
                    }
                }
            }
            for (MapPlan &mp : out.plans) {
                const size_t key = size_t(mp.site0 / kTileSites) * size_t(in.nbarcodes) + size_t(mp.flags & 0xffu);
                mp.g_first = key < n_keys ? rank[key >> 6] + uint32_t(__builtin_popcountll(marks[key >> 6] & ((1ull << (key & 63)) - 1ull))) : 0u;
            }
        }
        PT(T2b);
        if (getenv("BOSSX_STAGE_TIMING")) fprintf(stderr, "  [parse] device-walk tail: MapPlans + marks %.3f, ranks %.3f ms\n", PTMS(T2, T2a), PTMS(T2a, T2b));
        out.ops_cap = ops_at + 1;
        out.segs_cap = seg_cap + 1;
        out.total_emit = cur_emit;
        out.n_rec = n_rec;
        // The device walk needs the plans and the text, not the reads: the caller may launch it now, while
        // the workers still gather and upload the reads (early_walk also waits for the text slices).
        if (in.early_walk) in.early_walk(out);
        // the caller's tasks (gather + base check + uploads) have had the grouping and the plans to finish
        collect_pass1();
        out.any_check_bases = false;
        for (size_t i = 0; i < out.plans.size(); ++i)
            if (!in.read_dirty || in.read_dirty[out.plan_read[i]]) { out.plans[i].flags |= kPlanCheckBases; out.any_check_bases = true; }
        // failures of the pre-pass travel with the batch: the caller merges them with the device
        // walk's per-mapping outcome (first ValueError / KeyError class failure in record order; the
        // IndexError class only if nothing else failed)
        out.pre_code = pre_err.code; out.pre_msg = pre_err.msg; out.pre_gi = pre_err.code ? pre_err.group : -1;
        out.pre_range_gi = pre_range.code ? pre_range.group : -1; out.pre_range_msg = pre_range.msg;
        PT(T3); PT(T4);
        PTREPORT();
        return BOSSX_OK;
    }
    // ---- pass 2b (threads over contiguous record ranges): CIGAR walk -> emit runs + segments ----
    std::vector<WalkOut> wos;
    if (!in.summary_only && !plans.empty()) {
        if (!in.ops_buf || in.ops_cap < ops_at + 1) {
            err = "internal: emit-run buffer too small";
            return BOSSX_E_INVALID;
        }
        int nw = in.n_threads > 0 ? in.n_threads : parse_threads();
        if (plans.size() < 256) nw = 1;
        nw = int(std::min<size_t>(size_t(nw), plans.size()));
        // balance by CIGAR bytes (ops_at is their running upper bound)
        std::vector<size_t> cut(size_t(nw) + 1, plans.size());
        cut[0] = 0;
        for (int t = 1; t < nw; ++t) {
            const size_t want = ops_at * size_t(t) / size_t(nw);
            size_t lo = cut[size_t(t) - 1], hi = plans.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (plans[mid].ops_at < want) lo = mid + 1; else hi = mid; }
            cut[size_t(t)] = lo;
        }
        wos.resize(size_t(nw));
        auto work = [&](int t) {
            const size_t p0 = cut[size_t(t)], p1 = cut[size_t(t) + 1];
            EmitOp *base = in.ops_buf + (p0 < plans.size() ? plans[p0].ops_at : ops_at);
            walk_plans(in, contigs, plans, p0, p1, base, wos[size_t(t)]);
        };
        pool_run(nw, work);
    }
    PT(T3);
    // first failure in record order; the IndexError class only if nothing else failed (the
    // reference raises those in _effect_increments, after convert_records has seen every read)
    const WalkError *first_err = pre_err.code ? &pre_err : nullptr;
    for (const WalkOut &wo : wos)
        if (wo.err.code && (!first_err || wo.err.group < first_err->group)) first_err = &wo.err;
    if (!first_err) {
        if (pre_range.code) first_err = &pre_range;
        for (const WalkOut &wo : wos)
            if (wo.range_err.code && (!first_err || wo.range_err.group < first_err->group)) first_err = &wo.range_err;
    }
    if (first_err) { err = first_err->msg; return first_err->code; }
    if (cur_emit >= (1ull << 32) - kEmitTile) {
        err = "batch too large: more than 2^32 aligned bases";
        return BOSSX_E_RANGE;
    }

    // ---- merge: device positions of the chunks, segments grouped by tile (stable) ------------
    size_t n_segs = 0, dev_off = 0;
    for (const WalkOut &wo : wos) n_segs += wo.segs.size();
    std::vector<uint32_t> tile_of, order, tmp;
    std::vector<uint8_t> bc_of;
    tile_of.reserve(n_segs);
    bc_of.reserve(n_segs);
    std::vector<const TileSeg *> seg_ptr;
    seg_ptr.reserve(n_segs);
    std::vector<uint32_t> seg_base;
    seg_base.reserve(n_segs);
    uint32_t max_tile = 0;
    for (WalkOut &wo : wos) {
        if (wo.n_ops) out.chunks.push_back(OpsChunk{wo.base, wo.n_ops, dev_off});
        for (size_t i = 0; i < wo.segs.size(); ++i) {
            tile_of.push_back(wo.seg_tile[i]);
            bc_of.push_back(wo.seg_bc[i]);
            max_tile = std::max(max_tile, wo.seg_tile[i]);
            seg_ptr.push_back(&wo.segs[i]);
            seg_base.push_back(uint32_t(dev_off));
        }
        for (size_t c = 0; c < contigs.size(); ++c) out.emitted_per_contig[c] += wo.emitted_per_contig[c];
        dev_off += wo.n_ops;
    }
    out.n_ops = dev_off;
    // stable LSD radix sort of the segment indices by (tile, barcode): barcode pass first, then the
    // tile id 11 bits per pass
    order.resize(n_segs);
    for (size_t i = 0; i < n_segs; ++i) order[i] = uint32_t(i);
    tmp.resize(n_segs);
    if (in.nbarcodes > 1) {
        uint32_t count[257] = {0};
        for (size_t i = 0; i < n_segs; ++i) ++count[uint32_t(bc_of[order[i]]) + 1];
        for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
        for (size_t i = 0; i < n_segs; ++i) tmp[count[bc_of[order[i]]]++] = order[i];
        order.swap(tmp);
    }
    for (uint32_t shift = 0; shift < 32 && (max_tile >> shift) != 0; shift += 11) {
        uint32_t count[2049] = {0};
        for (size_t i = 0; i < n_segs; ++i) ++count[((tile_of[order[i]] >> shift) & 2047u) + 1];
        for (int b = 0; b < 2048; ++b) count[b + 1] += count[b];
        for (size_t i = 0; i < n_segs; ++i) tmp[count[(tile_of[order[i]] >> shift) & 2047u]++] = order[i];
        order.swap(tmp);
    }
    // one TileRef per (tile, barcode) group; the groups of a tile are consecutive
    out.segs.reserve(n_segs);
    out.n_touched_tiles = 0;
    for (uint32_t idx : order) {
        const uint32_t t = tile_of[idx], bc = bc_of[idx];
        if (out.tiles.empty() || out.tiles.back().tile != t) ++out.n_touched_tiles;
        if (out.tiles.empty() || out.tiles.back().tile != t || out.tiles.back().bc != bc)
            out.tiles.push_back(TileRef{t, uint32_t(out.segs.size()), uint32_t(out.segs.size()), bc});
        TileSeg sg = *seg_ptr[idx];
        sg.op_lo += seg_base[idx]; sg.op_hi += seg_base[idx];
        out.segs.push_back(sg);
        out.tiles.back().seg_hi = uint32_t(out.segs.size());
    }
    PT(T4);
    PTREPORT();
    out.n_groups = out.tiles.size();
    out.total_emit = cur_emit;
    out.n_rec = n_rec;
    return BOSSX_OK;
}

}  // namespace bossx

// ------------------------------------------------------------------------------------------
// Host-only check entry (include/bossx.h): the front end without a device.  Parses like
// bossx_stage_batch_ptrs and expands the emit runs base by base exactly as the ingest kernels
// read them, so CPU tests can compare the parser (and its thread partition) with the oracle.
// ------------------------------------------------------------------------------------------
extern "C" int bossx_host_parse(const char *const *contig_names, const int64_t *contig_lengths,
                                const int32_t *contig_flags, int32_t n_contigs, int32_t nbarcodes,
                                const char *paf, size_t paf_len, const char *const *name_ptrs,
                                const int64_t *name_lens, const char *const *seq_ptrs, const int64_t *seq_lens,
                                const int32_t *barcodes, int32_t n_reads, int32_t min_len, int32_t n_threads,
                                bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases,
                                int32_t *out_contig, int64_t *out_pos, uint8_t *out_code, uint8_t *out_barcode,
                                int64_t out_cap, char *err_buf, size_t err_cap) {
    using namespace bossx;
    auto fail = [&](int code, const std::string &msg) {
        if (err_buf && err_cap) { strncpy(err_buf, msg.c_str(), err_cap - 1); err_buf[err_cap - 1] = '\0'; }
        return code;
    };
    if (n_contigs < 0 || n_reads < 0 || nbarcodes < 1) return fail(BOSSX_E_INVALID, "bad host_parse call");
    std::vector<ContigInfo> contigs(static_cast<size_t>(n_contigs));
    std::unordered_map<std::string, int32_t> index;
    int64_t site = 0;
    for (int32_t i = 0; i < n_contigs; ++i) {
        ContigInfo &c = contigs[size_t(i)];
        c.name = contig_names[i]; c.length = contig_lengths[i];
        c.rejected = (contig_flags[i] & BOSSX_CONTIG_REJECTED) != 0;
        c.remote = (contig_flags[i] & BOSSX_CONTIG_REMOTE) != 0;
        index[c.name] = i;
        if (c.rejected || c.remote) continue;
        c.site_off = site;
        c.n_tiles = (c.length + kTileSites - 1) / kTileSites;
        site += c.n_tiles * kTileSites;
    }
    std::vector<int64_t> name_off(size_t(n_reads) + 1, 0), seq_off(size_t(n_reads) + 1, 0);
    for (int32_t i = 0; i < n_reads; ++i) {
        name_off[size_t(i) + 1] = name_off[size_t(i)] + name_lens[i];
        seq_off[size_t(i) + 1] = seq_off[size_t(i)] + seq_lens[i];
    }
    std::string names(size_t(name_off[size_t(n_reads)]), '\0'), blob(size_t(seq_off[size_t(n_reads)]), '\0');
    for (int32_t i = 0; i < n_reads; ++i) {
        memcpy(&names[size_t(name_off[size_t(i)])], name_ptrs[i], size_t(name_lens[i]));
        memcpy(&blob[size_t(seq_off[size_t(i)])], seq_ptrs[i], size_t(seq_lens[i]));
    }
    std::vector<EmitOp> buf(ops_capacity_for(paf ? paf_len : 0));
    ParseInput in{paf ? paf : "", paf ? paf_len : 0, names.data(), name_off.data(), seq_off.data(), barcodes, n_reads, min_len, nbarcodes};
    in.ops_buf = buf.data(); in.ops_cap = buf.size(); in.n_threads = n_threads;
    in.seqs = blob.data();
    ParsedBatch pb;
    std::string err;
    int rc = parse_paf_batch(in, contigs, index, summary, pb, err);
    if (rc) return fail(rc, err);
    if (n_rec) *n_rec = pb.n_rec;
    if (aligned_bases) *aligned_bases = int64_t(pb.total_emit);
    // the device array: chunks back to back
    std::vector<EmitOp> ops(pb.n_ops);
    for (const OpsChunk &ck : pb.chunks) memcpy(ops.data() + ck.dev_off, ck.host, ck.n * sizeof(EmitOp));
    for (size_t i = 1; i < ops.size(); ++i)
        if (ops[i].emit_start <= ops[i - 1].emit_start) return fail(BOSSX_E_INVALID, "emit runs are not strictly increasing");
    // segments: every emitted base exactly once, inside its tile, inside its run range
    uint64_t seg_total = 0;
    for (const TileRef &tr : pb.tiles) {
        for (uint32_t k = tr.seg_lo; k < tr.seg_hi; ++k) {
            const TileSeg &sg = pb.segs[k];
            if (sg.e_hi <= sg.e_lo || sg.e_hi - sg.e_lo > uint32_t(kSegMax) || sg.op_hi < sg.op_lo || sg.op_hi >= ops.size())
                return fail(BOSSX_E_INVALID, "malformed tile segment");
            if (ops[sg.op_lo].emit_start > sg.e_lo || (sg.op_lo + 1 < ops.size() && ops[sg.op_lo + 1].emit_start <= sg.e_lo))
                return fail(BOSSX_E_INVALID, "segment does not start in its first run");
            if (ops[sg.op_hi].emit_start >= sg.e_hi || (sg.op_hi + 1 < ops.size() && ops[sg.op_hi + 1].emit_start < sg.e_hi))
                return fail(BOSSX_E_INVALID, "segment does not end in its last run");
            const EmitOp &o = ops[sg.op_lo];
            const uint64_t s0 = ((uint64_t(o.meta & 0xffu) << 32) | o.site_lo) + (sg.e_lo - o.emit_start);
            if (s0 / kTileSites != tr.tile || (s0 + (sg.e_hi - sg.e_lo) - 1) / kTileSites != tr.tile)
                return fail(BOSSX_E_INVALID, "segment crosses its tile");
            for (uint32_t o2 = sg.op_lo; o2 <= sg.op_hi; ++o2)
                if (((ops[o2].meta >> 8) & 0xffu) != tr.bc) return fail(BOSSX_E_INVALID, "segment in the wrong barcode group");
            seg_total += sg.e_hi - sg.e_lo;
        }
    }
    if (seg_total != pb.total_emit) return fail(BOSSX_E_INVALID, "segments do not cover the batch");
    for (size_t i = 1; i < pb.tiles.size(); ++i) {
        const TileRef &a = pb.tiles[i - 1], &b = pb.tiles[i];
        if (b.tile < a.tile || (b.tile == a.tile && b.bc <= a.bc)) return fail(BOSSX_E_INVALID, "tile groups out of order");
    }
    // the host half of the device walk (plans + groups + capacities) must describe the same batch
    {
        ParseInput in2 = in;
        in2.device_walk = true; in2.ops_buf = nullptr; in2.ops_cap = 0;
        int64_t n_tiles = 0;
        for (const ContigInfo &c : contigs) if (!c.rejected && !c.remote) n_tiles += c.n_tiles;
        in2.n_tiles = n_tiles;
        // ... with the caller's share of the two-phase job as bossx_stage_batch_ptrs brings it along: slices
        // of reads looked at for bytes other than A/C/G/T while the calling thread groups and plans;
        // collected (after_pass1) before the plans are flagged for the walk's base check
        std::vector<uint8_t> dirty(size_t(n_reads), 0);
        const int n_extra = n_reads > 0 ? std::min(n_reads, 7) : 0;
        std::atomic<int> extras_done{0};
        int done_at_collection = -1;
        const std::function<void(int)> extra_fn = [&](int t) {
            const int32_t i0 = int32_t(int64_t(n_reads) * t / n_extra), i1 = int32_t(int64_t(n_reads) * (t + 1) / n_extra);
            for (int32_t i = i0; i < i1; ++i) {
                const char *q = seq_ptrs ? seq_ptrs[i] : nullptr;
                bool bad = false;
                for (int64_t j = 0; q && j < seq_lens[i]; ++j) bad |= !(q[j] == 'A' || q[j] == 'C' || q[j] == 'G' || q[j] == 'T');
                dirty[size_t(i)] = bad ? 1 : 0;
            }
            extras_done.fetch_add(1);
        };
        in2.extra_n = n_extra; in2.extra_fn = extra_fn;
        in2.after_pass1 = [&]() { done_at_collection = extras_done.load(); };
        in2.read_dirty = dirty.data();
        ParsedBatch pd;
        std::string err2;
        rc = parse_paf_batch(in2, contigs, index, nullptr, pd, err2);
        if (rc || pd.pre_code || pd.pre_range_gi >= 0) return fail(BOSSX_E_INVALID, "device-walk planning fails where the host walk passed: " + err2 + pd.pre_msg);
        if (done_at_collection != n_extra) return fail(BOSSX_E_INVALID, "device-walk planning: the caller's tasks were not all done when the job was collected");
        for (size_t i = 0; i < pd.plans.size(); ++i)
            if (((pd.plans[i].flags & kPlanCheckBases) != 0) != (dirty[size_t(pd.plan_read[i])] != 0))
                return fail(BOSSX_E_INVALID, "device-walk planning: base-check flags do not follow the reads");
        if (pd.total_emit != pb.total_emit || pd.n_rec != pb.n_rec || pd.emitted_per_contig != pb.emitted_per_contig)
            return fail(BOSSX_E_INVALID, "device-walk planning: totals differ");
        if (pd.ops_cap < pb.n_ops || pd.segs_cap < pb.segs.size()) return fail(BOSSX_E_INVALID, "device-walk planning: capacity too small");
        // the group list as build_groups_kernel writes it from the bitmap and the ranks (front_end.hip.inc), here on the CPU
        std::vector<TileRef> dg(pd.n_groups);
        if (pd.rank.size() != pd.marks.size() + 1 || pd.rank.back() != pd.n_groups) return fail(BOSSX_E_INVALID, "device-walk planning: ranks do not add up");
        for (size_t w = 0; w < pd.marks.size(); ++w) {
            uint64_t bits = pd.marks[w];
            uint32_t at = pd.rank[w];
            while (bits) {
                const uint32_t key = uint32_t((w << 6) + size_t(__builtin_ctzll(bits)));
                bits &= bits - 1;
                if (at >= dg.size()) return fail(BOSSX_E_INVALID, "device-walk planning: rank beyond the group count");
                const uint32_t t = key / uint32_t(nbarcodes);
                dg[at++] = TileRef{t, 0u, 0u, key - t * uint32_t(nbarcodes)};
            }
        }
        if (dg.size() != pb.tiles.size() || pd.n_touched_tiles != pb.n_touched_tiles)
            return fail(BOSSX_E_INVALID, "device-walk planning: group count differs");
        for (size_t i = 0; i < dg.size(); ++i)
            if (dg[i].tile != pb.tiles[i].tile || dg[i].bc != pb.tiles[i].bc)
                return fail(BOSSX_E_INVALID, "device-walk planning: groups differ");
        for (const MapPlan &mp : pd.plans) {       // the group index the device walk starts from
            const uint32_t t0 = uint32_t(mp.site0 / kTileSites), bc = mp.flags & 0xffu;
            if (mp.span && (mp.g_first >= dg.size() || dg[mp.g_first].tile != t0 || dg[mp.g_first].bc != bc))
                return fail(BOSSX_E_INVALID, "device-walk planning: first group of a mapping is wrong");
        }
        uint64_t span_sum = 0;
        for (const MapPlan &mp : pd.plans) {
            if (mp.emit0 != uint32_t(span_sum)) return fail(BOSSX_E_INVALID, "device-walk planning: emit order broken");
            span_sum += mp.span;
            if (size_t(mp.cg_off) + mp.cg_len > paf_len) return fail(BOSSX_E_INVALID, "device-walk planning: CIGAR outside the text");
        }
        if (span_sum != pb.total_emit) return fail(BOSSX_E_INVALID, "device-walk planning: spans do not add up");
    }
    if (!out_contig) return BOSSX_OK;
    if (out_cap < int64_t(pb.total_emit)) return fail(BOSSX_E_INVALID, "output arrays too small");
    for (size_t i = 0; i < ops.size(); ++i) {
        const EmitOp &o = ops[i];
        const uint64_t e_end = i + 1 < ops.size() ? ops[i + 1].emit_start : pb.total_emit;
        const uint64_t site0 = (uint64_t(o.meta & 0xffu) << 32) | o.site_lo;
        int32_t cidx = -1;
        for (int32_t c = 0; c < n_contigs; ++c) {
            const ContigInfo &ci = contigs[size_t(c)];
            if (!ci.rejected && !ci.remote && int64_t(site0) >= ci.site_off && int64_t(site0) < ci.site_off + ci.n_tiles * kTileSites) cidx = c;
        }
        for (uint64_t e = o.emit_start; e < e_end; ++e) {
            const uint64_t j = e - o.emit_start;
            uint8_t code = 4;
            if (!(o.meta & kOpDel)) {
                const bool rev = (o.meta & kOpRev) != 0;
                const char ch = blob[size_t((o.meta & kOpBcast) ? o.qpos : (rev ? o.qpos - j : o.qpos + j))];
                code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 255;
                if (rev && code != 255) code = uint8_t(3 - code);
                if (code == 255) { const unsigned d = unsigned(static_cast<unsigned char>(ch)) - unsigned('0'); if (d <= 4u) code = uint8_t(d); else if (d == 7u) code = 4; }      // (digits: their own column, either strand)
            }
            out_contig[e] = cidx;
            out_pos[e] = int64_t(site0 + j) - contigs[size_t(cidx)].site_off;
            out_code[e] = code;
            out_barcode[e] = uint8_t((o.meta >> 8) & 0xffu);
        }
    }
    return BOSSX_OK;
}

// Lines of a PAF text whose query name (column 1, normalised as PafLine does) is a read of the
// batch with keep[read] != 0, copied to `out` in file order, '\n' separated.
extern "C" int bossx_paf_select_lines(const char *paf, size_t paf_len, const char *const *name_ptrs,
                                      const int64_t *name_lens, int32_t n_reads, const uint8_t *keep,
                                      char *out, size_t out_cap, size_t *out_len) {
    using namespace bossx;
    if (!out_len || n_reads < 0 || (n_reads > 0 && (!name_ptrs || !name_lens || !keep)) || (paf_len && (!paf || !out)))
        return BOSSX_E_INVALID;
    std::unordered_map<std::string_view, int32_t> read_index;
    read_index.reserve(size_t(n_reads) * 2 + 1);
    for (int32_t i = 0; i < n_reads; ++i) read_index[std::string_view(name_ptrs[i], size_t(name_lens[i]))] = i;
    size_t w = 0;
    const char *p = paf, *end = paf + paf_len;
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', size_t(end - p)));
        const char *le = nl ? nl : end;
        std::string_view line = strip(std::string_view(p, size_t(le - p)));
        p = nl ? nl + 1 : end;
        if (line.empty()) continue;
        const size_t t = line.find('\t');
        const std::string_view q = line.substr(0, t);
        auto it = read_index.find(q);
        if (it == read_index.end()) {
            int64_t v;
            std::string norm;                                   // "007" is stored as "7" (paf.py:55-56)
            if (!parse_int(q, v, nullptr, &norm)) continue;
            it = read_index.find(std::string_view(norm));
            if (it == read_index.end()) continue;
        }
        if (!keep[it->second]) continue;
        if (w + line.size() + 1 > out_cap) return BOSSX_E_INVALID;
        if (w) out[w++] = '\n';
        memcpy(out + w, line.data(), line.size());
        w += line.size();
    }
    *out_len = w;
    return BOSSX_OK;
}

// Binding helper for CPython callers (bossx_py.h, private): buffer pointer + length of every str in
// the Python list `list` through the interpreter's own PyList_GetItem / PyUnicode_AsUTF8AndSize
// (passed in by address, so this library does not link against libpython).  Must be called
// with the GIL held (ctypes.PyDLL).
extern "C" int bossx_py_str_pointers(void *list, int64_t n, void *list_get_item, void *as_utf8_and_size,
                                     const char **ptrs, int64_t *lens) {
    using get_t = void *(*)(void *, long);             // Py_ssize_t == long on LP64
    using utf_t = const char *(*)(void *, long *);
    static_assert(sizeof(long) == sizeof(void *), "LP64 only");
    get_t get = reinterpret_cast<get_t>(list_get_item);
    utf_t utf = reinterpret_cast<utf_t>(as_utf8_and_size);
    if (!get || !utf || !list || n < 0 || (n > 0 && (!ptrs || !lens))) return BOSSX_E_INVALID;
    for (int64_t i = 0; i < n; ++i) {
        void *item = get(list, long(i));               // borrowed reference
        long sz = 0;
        const char *p = item ? utf(item, &sz) : nullptr;
        if (!p) return BOSSX_E_INVALID;                // the interpreter has set the exception
        ptrs[i] = p; lens[i] = int64_t(sz);
    }
    return BOSSX_OK;
}

extern "C" int64_t bossx_py_dict_pointers(void *dict, int64_t cap, void *dict_next, void *as_utf8_and_size,
                                          const char **key_ptrs, int64_t *key_lens, const char **val_ptrs, int64_t *val_lens) {
    using next_t = int (*)(void *, long *, void **, void **);      // PyDict_Next(dict, &pos, &key, &value): borrowed references
    using utf_t = const char *(*)(void *, long *);
    next_t next = reinterpret_cast<next_t>(dict_next);
    utf_t utf = reinterpret_cast<utf_t>(as_utf8_and_size);
    if (!next || !utf || !dict || cap < 0 || (cap > 0 && (!key_ptrs || !key_lens || !val_ptrs || !val_lens))) return BOSSX_E_INVALID;
    // Two passes: the dict's entries lie side by side, the str objects they point to all over the heap (4000 reads of 6 kb each): the first
    // pass collects the object pointers and asks for their headers, the second reads them — the misses overlap instead of queueing
    // (0.27 -> ~0.1 ms for a 4000-read batch, in front of everything else a staging call does).
    long pos = 0;
    void *k = nullptr, *v = nullptr;
    int64_t n = 0;
    while (next(dict, &pos, &k, &v)) {
        if (n >= cap) return BOSSX_E_INVALID;
        __builtin_prefetch(k); __builtin_prefetch(v);
        key_ptrs[n] = static_cast<const char *>(k); val_ptrs[n] = static_cast<const char *>(v);
        ++n;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (i + 24 < n) {                                            // (ASCII str: the characters start 48 bytes into the object: the line after the header's)
            __builtin_prefetch(key_ptrs[i + 24] + 48); __builtin_prefetch(val_ptrs[i + 24] + 48);
        }
        long kl = 0, vl = 0;
        const char *kp = utf(const_cast<char *>(key_ptrs[i]), &kl), *vp = utf(const_cast<char *>(val_ptrs[i]), &vl);
        if (!kp || !vp) {
            // not a str (the interpreter has set its error indicator).  The outputs held object pointers up to here and character
            // pointers before: never hand a mixture back — everything is cleared, the contents mean nothing on an error
            for (int64_t j = 0; j < n; ++j) { key_ptrs[j] = nullptr; val_ptrs[j] = nullptr; key_lens[j] = 0; val_lens[j] = 0; }
            return BOSSX_E_INVALID;
        }
        key_ptrs[i] = kp; key_lens[i] = kl; val_ptrs[i] = vp; val_lens[i] = vl;
    }
    return n;
}

