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

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string_view>

namespace bossx {
namespace {

struct Rec {
    std::string qname, tname;
    int64_t qlen, qstart, qend, tstart, tend, alnlen, mapq, as;
    bool rev;
    const char *cg; size_t cg_len;   // points into the PAF text
    bool has_cg;
};

// Python `int(s)` for the plain forms PAF uses: optional sign + decimal digits.
bool parse_int(std::string_view s, int64_t &v) {
    size_t i = 0, n = s.size();
    while (i < n && (s[i] == ' ')) ++i;
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
    if (i >= n) return false;
    int64_t x = 0;
    for (; i < n; ++i) {
        char c = s[i];
        if (c < '0' || c > '9') return false;
        x = x * 10 + (c - '0');
    }
    v = neg ? -x : x;
    return true;
}

// PafLine stores names as str(conv_type(x, int)): "007" becomes "7" (paf.py:55-56, 103-108).
std::string normalise_name(std::string_view s) {
    int64_t v;
    if (parse_int(s, v)) return std::to_string(v);
    return std::string(s);
}

std::string_view strip(std::string_view s) {
    size_t b = 0, e = s.size();
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; };
    while (b < e && ws(s[b])) ++b;
    while (e > b && ws(s[e - 1])) --e;
    return s.substr(b, e - b);
}

struct Group {
    Rec best;
    int64_t key_q, key_dp;
};

struct CigarTable {
    bool ok[256];
    constexpr CigarTable() : ok() { for (const char *p = "MIDNSHP=XB"; *p; ++p) ok[static_cast<unsigned char>(*p)] = true; }
    constexpr bool operator[](unsigned char c) const { return ok[c]; }
};
constexpr CigarTable kCigarOp{};

}  // namespace

int parse_paf_batch(const ParseInput &in, const std::vector<ContigInfo> &contigs,
                    const std::unordered_map<std::string, int32_t> &contig_index,
                    bossx_batch_summary *summary, ParsedBatch &out, std::string &err) {
    out = ParsedBatch();
    out.emitted_per_contig.assign(contigs.size(), 0);

    // read id -> index in the batch
    std::unordered_map<std::string_view, int32_t> read_index;
    read_index.reserve(size_t(in.n_reads) * 2 + 1);
    for (int32_t i = 0; i < in.n_reads; ++i) {
        std::string_view nm(in.names + in.name_off[i], size_t(in.name_off[i + 1] - in.name_off[i]));
        read_index[nm] = i;   // later duplicates win, like a dict
    }

    // ---- pass 1: lines -> best record per query name, first-appearance order -------------
    std::vector<Group> groups;
    std::unordered_map<std::string, int32_t> group_of;
    const char *p = in.paf, *end = in.paf + in.paf_len;
    int64_t lineno = 0;
    std::vector<std::string_view> f;
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', size_t(end - p)));
        const char *le = nl ? nl : end;
        std::string_view line = strip(std::string_view(p, size_t(le - p)));
        p = nl ? nl + 1 : end;
        ++lineno;
        if (line.empty()) continue;
        f.clear();
        size_t s = 0;
        while (true) {
            size_t t = line.find('\t', s);
            if (t == std::string_view::npos) { f.push_back(line.substr(s)); break; }
            f.push_back(line.substr(s, t - s));
            s = t + 1;
        }
        if (f.size() < 12) {
            err = "PAF line " + std::to_string(lineno) + ": fewer than 12 columns";
            return BOSSX_E_PARSE;
        }
        Rec r;
        int64_t tlen, nmatch;
        bool ok = parse_int(f[1], r.qlen) && parse_int(f[2], r.qstart) && parse_int(f[3], r.qend) &&
                  parse_int(f[6], tlen) && parse_int(f[7], r.tstart) && parse_int(f[8], r.tend) &&
                  parse_int(f[9], nmatch) && parse_int(f[10], r.alnlen) && parse_int(f[11], r.mapq);
        if (!ok) {
            err = "PAF line " + std::to_string(lineno) + ": non-integer core column";
            return BOSSX_E_PARSE;
        }
        r.rev = !(f[4].size() == 1 && f[4][0] == '+');   // paf.py:58
        r.as = 0; r.has_cg = false; r.cg = nullptr; r.cg_len = 0;
        bool primary = false;
        for (size_t k = 12; k < f.size(); ++k) {
            std::string_view tag = f[k];
            size_t c1 = tag.find(':');
            size_t c2 = c1 == std::string_view::npos ? c1 : tag.find(':', c1 + 1);
            if (c1 == std::string_view::npos || c2 == std::string_view::npos ||
                tag.find(':', c2 + 1) != std::string_view::npos) {
                err = "PAF line " + std::to_string(lineno) + ": malformed tag";   // x.split(':') unpack
                return BOSSX_E_PARSE;
            }
            std::string_view key = tag.substr(0, c1), val = tag.substr(c2 + 1);
            if (key == "AS") {
                if (!parse_int(val, r.as)) {
                    err = "PAF line " + std::to_string(lineno) + ": AS is not an integer";
                    return BOSSX_E_PARSE;
                }
            } else if (key == "cg") {
                r.has_cg = true; r.cg = val.data(); r.cg_len = val.size();
            } else if (key == "tp") {
                primary = (val == "P");
            }
        }
        if (r.alnlen < in.min_len) continue;     // paf.py:666-667
        if (!primary) continue;                   // paf.py:668-669
        r.qname = normalise_name(f[0]);
        r.tname = normalise_name(f[5]);
        auto it = group_of.find(r.qname);
        if (it == group_of.end()) {
            group_of.emplace(r.qname, int32_t(groups.size()));
            groups.push_back(Group{r, r.mapq, r.as});
        } else {
            Group &g = groups[size_t(it->second)];
            // argsort by (mapq, AS), last element wins; stable for ties (paf.py:716-721)
            if (r.mapq > g.key_q || (r.mapq == g.key_q && r.as >= g.key_dp)) {
                g.best = r; g.key_q = r.mapq; g.key_dp = r.as;
            }
        }
    }

    // ---- pass 2: chosen mappings -> emit runs ------------------------------------------------
    uint64_t cur_emit = 0;
    out.ops.reserve(in.paf_len / 3 + 16);
    std::vector<TileSeg> raw_segs;
    std::vector<uint32_t> raw_tile;
    int32_t n_rec = 0;
    for (const Group &g : groups) {
        const Rec &r = g.best;
        auto ri = read_index.find(std::string_view(r.qname));
        if (ri == read_index.end()) {
            err = "read '" + r.qname + "' is mapped in the PAF but absent from the batch";
            return BOSSX_E_KEY;       // seqs[rec.qname], sequences.py:708/713
        }
        const int32_t read = ri->second;
        int32_t cidx = -1;
        auto ci = contig_index.find(r.tname);
        if (ci != contig_index.end()) cidx = ci->second;
        if (summary) {
            summary->read_idx[n_rec] = read;
            summary->contig_idx[n_rec] = cidx;
            summary->rev[n_rec] = r.rev ? 1 : 0;
            summary->tstart[n_rec] = r.tstart;
            summary->tend[n_rec] = r.tend;
            summary->qlen[n_rec] = r.qlen;
        }
        ++n_rec;
        if (in.summary_only) continue;
        if (!r.has_cg) {
            err = "read '" + r.qname + "': mapping without cg tag";   // assert rec.cigar is not None
            return BOSSX_E_PARSE;
        }
        if (cidx < 0 || contigs[size_t(cidx)].rejected || contigs[size_t(cidx)].remote) continue;   // core.py:83-86: only (local) contigs_filt
        const ContigInfo &c = contigs[size_t(cidx)];
        const int64_t seq_b = in.seq_off[read], seq_len = in.seq_off[read + 1] - seq_b;
        const int64_t tlo = r.tstart < r.tend ? r.tstart : r.tend;
        const int64_t thi = r.tstart < r.tend ? r.tend : r.tstart;
        int32_t bc = in.barcodes ? in.barcodes[read] : 0;
        if (bc < 0 || bc >= in.nbarcodes) {
            err = "read '" + r.qname + "': barcode index out of range";
            return BOSSX_E_RANGE;
        }
        // query walk: '+' reads seq[qstart + i]; '-' reads comp(seq[len-1-(qlen-qend) - i])
        // (sequences.py:707-716: slice [qlen-qend, qlen-qstart) of the reverse complement)
        int64_t q = r.rev ? (seq_len - 1 - (r.qlen - r.qend)) : r.qstart;
        const int64_t qstep = r.rev ? -1 : 1;
        const int64_t q_need = r.qend - r.qstart;
        int64_t consumed = 0, ref_pos = tlo;
        const size_t first_op = out.ops.size();
        const char *cp = r.cg, *ce = r.cg + r.cg_len;
        while (cp < ce) {
            int64_t len = 0;
            const char *d0 = cp;
            while (cp < ce && *cp >= '0' && *cp <= '9') { len = len * 10 + (*cp - '0'); ++cp; }
            if (cp == d0 || cp >= ce) {
                err = "read '" + r.qname + "': malformed CIGAR";
                return BOSSX_E_PARSE;
            }
            const char op = *cp++;
            if (!kCigarOp[static_cast<unsigned char>(op)]) {
                err = "read '" + r.qname + "': unknown CIGAR op";
                return BOSSX_E_PARSE;
            }
            if (len == 0) continue;
            if (op == 'I') {                       // consumes query, emits nothing (sequences.py:781)
                consumed += len; q += qstep * len;
                continue;
            }
            const bool del = (op == 'D');          // emits code 4, consumes nothing (sequences.py:782,793)
            if (!del) {
                const int64_t q_last = q + qstep * (len - 1);
                if (q < 0 || q >= seq_len || q_last < 0 || q_last >= seq_len) {
                    err = "read '" + r.qname + "': CIGAR walks outside the read";
                    return BOSSX_E_PARSE;       // shape mismatch in cig_rep[notdel] = int_seq[start:end]
                }
            }
            if (ref_pos + len > c.length) {
                err = "read '" + r.qname + "': mapping extends past the end of " + c.name;
                return BOSSX_E_RANGE;
            }
            const uint64_t site = uint64_t(c.site_off + ref_pos);
            EmitOp e;
            e.emit_start = uint32_t(cur_emit);
            e.site_lo = uint32_t(site & 0xffffffffu);
            e.qpos = del ? 0u : uint32_t(seq_b + q);
            e.meta = uint32_t((site >> 32) & 0xffu) | (uint32_t(bc) << 8) | (r.rev ? kOpRev : 0u) |
                     (del ? kOpDel : 0u);
            out.ops.push_back(e);
            cur_emit += uint64_t(len);
            ref_pos += len;
            if (!del) { consumed += len; q += qstep * len; }
        }
        if (consumed != q_need) {
            err = "read '" + r.qname + "': CIGAR consumes " + std::to_string(consumed) +
                  " query bases, PAF says " + std::to_string(q_need);
            out.ops.resize(first_op);
            return BOSSX_E_PARSE;
        }
        if (ref_pos - tlo != thi - tlo) {
            err = "read '" + r.qname + "': CIGAR spans " + std::to_string(ref_pos - tlo) +
                  " reference bases, PAF says " + std::to_string(thi - tlo);   // sequences.py:732
            return BOSSX_E_PARSE;
        }
        out.emitted_per_contig[size_t(cidx)] += uint64_t(thi - tlo);
        // split the read's emitted stretch at sweep-tile boundaries (padded site space)
        if (out.ops.size() > first_op) {
            const uint64_t site0 = uint64_t(c.site_off + tlo);
            const uint64_t site1 = uint64_t(c.site_off + thi);
            const uint64_t e0 = out.ops[first_op].emit_start;
            size_t op = first_op;
            for (uint64_t t = site0 / kTileSites; t * kTileSites < site1; ++t) {
                const uint64_t s_lo = std::max<uint64_t>(t * kTileSites, site0);
                const uint64_t s_hi = std::min<uint64_t>((t + 1) * kTileSites, site1);
                // pieces of at most kSegMax emitted bases, each with its exact emit-run range
                for (uint64_t p_lo = s_lo; p_lo < s_hi; p_lo += kSegMax) {
                    const uint64_t p_hi = std::min<uint64_t>(p_lo + kSegMax, s_hi);
                    TileSeg sg;
                    sg.e_lo = uint32_t(e0 + (p_lo - site0));
                    sg.e_hi = uint32_t(e0 + (p_hi - site0));
                    while (op + 1 < out.ops.size() && out.ops[op + 1].emit_start <= sg.e_lo) ++op;
                    sg.op_lo = uint32_t(op);
                    size_t oh = op;
                    while (oh + 1 < out.ops.size() && out.ops[oh + 1].emit_start < sg.e_hi) ++oh;
                    sg.op_hi = uint32_t(oh);
                    raw_segs.push_back(sg);
                    raw_tile.push_back(uint32_t(t));
                }
            }
        }
    }
    // group segments by tile (stable sort of the small tile-id list)
    {
        std::vector<uint32_t> order(raw_segs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = uint32_t(i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return raw_tile[a] < raw_tile[b]; });
        out.segs.reserve(order.size());
        for (uint32_t idx : order) {
            const uint32_t t = raw_tile[idx];
            if (out.tiles.empty() || out.tiles.back().tile != t)
                out.tiles.push_back(TileRef{t, uint32_t(out.segs.size()), uint32_t(out.segs.size()), 0});
            out.segs.push_back(raw_segs[idx]);
            out.tiles.back().seg_hi = uint32_t(out.segs.size());
        }
    }
    if (cur_emit >= (1ull << 32) - kEmitTile) {
        err = "batch too large: more than 2^32 aligned bases";
        return BOSSX_E_RANGE;
    }
    out.total_emit = cur_emit;
    out.n_rec = n_rec;
    return BOSSX_OK;
}

// Emit-order tiling for the fallback scatter kernel: tile t -> the run that holds element
// t * kEmitTile (built on demand; the normal path bins by sweep tile instead).
void build_emit_tiles(ParsedBatch &pb) {
    const size_t n_tiles = size_t((pb.total_emit + kEmitTile - 1) / kEmitTile);
    pb.tile_first_op.assign(n_tiles + 1, 0);
    size_t op = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        const uint64_t e = uint64_t(t) * kEmitTile;
        while (op + 1 < pb.ops.size() && pb.ops[op + 1].emit_start <= e) ++op;
        pb.tile_first_op[t] = uint32_t(op);
    }
    pb.tile_first_op[n_tiles] = pb.ops.empty() ? 0u : uint32_t(pb.ops.size() - 1);
}

}  // namespace bossx
