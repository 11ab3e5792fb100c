// Internal definitions shared by the host-side PAF/CIGAR front end (paf_host.cpp) and the
// HIP engine (bossx.hip).  Not part of the C-ABI (that is include/bossx.h).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "bossx.h"

namespace bossx {

constexpr int kWindow = BOSSX_WINDOW;      // 100-bp strategy window
constexpr int kBucket = BOSSX_BUCKET;      // 20-kb activation bucket
constexpr int kTileSites = 2000;           // sites per sweep tile: 20 bins, 1/10 bucket
constexpr int kTileBins = kTileSites / kWindow;
constexpr int kEmitTile = 2048;            // emitted reference bases per ingest tile (fallback scatter)
constexpr int kSegMax = kTileSites;        // emitted bases per tile segment: one piece per (mapping, tile)

// Per-site state in HBM, TILE-MAJOR: what a sweep block needs of one (tile, barcode) is ONE contiguous
// record — five planes of kTileSites uint16 counters, then kTileSites state bytes.  The four base planes are
// REFERENCE-RELATIVE: plane k (k < 4) of a site counts base (ref + k) & 3, ref = the site's reference base
// (state bits 0-1); plane 4 counts deletions.  ~93 % of a batch's increments then land in plane 0, so the
// 16-byte vectors of planes 1-4 mostly stay clean and are not written back, and every site's pattern looks
// like (depth, few, few, few, few) whatever its base — the score table is indexed in that space (see
// device_lut_index) and its hot entries are a few kilobytes.  bossx_export / _import un-rotate.
// One record per (tile, barcode),
// so a tile that receives bases costs one or two pages / DRAM rows instead of six, one in each of six
// arrays of gigabytes (at 3.1 Gb the plane-major layout spent its time in address translation:
// 1.5 ms for 16 k scattered tiles against 0.16 ms for as many tiles of a 110 Mb reference).
constexpr int kTilePlaneBytes = kTileSites * 2;
constexpr int kTileMetaOff = 5 * kTilePlaneBytes;
constexpr int kTileStride = 22016;         // 22,000 bytes rounded up to whole 128-byte lines
// ... and in the 16 spare bytes behind the state bytes: the depth total of the (tile, barcode) at its last sweep (what the
// sweep adds to a 20-kb bucket total is the CHANGE of it), so that it arrives with the tile's other loads
constexpr int kTileTotalOff = kTileMetaOff + kTileSites;
static_assert(kTileStride >= kTileTotalOff + 4 && kTileStride % 128 == 0, "tile record");

// One emitting CIGAR run (M-like or D) of a chosen mapping, 16 bytes, loaded as one uint4.
//   emit_start : index of its first emitted base in the batch-wide emit order
//   site_lo    : low 32 bits of the padded global site index of that base
//   qpos       : byte index into the uploaded read blob of the first query base (M-like);
//                walks +1 per base on '+' mappings, -1 (and complemented) on '-' mappings
//   meta       : bits 0-7 site_hi, 8-15 barcode, 16 reverse strand, 17 deletion run
struct alignas(16) EmitOp {
    uint32_t emit_start;
    uint32_t site_lo;
    uint32_t qpos;
    uint32_t meta;
};
constexpr uint32_t kOpRev = 1u << 16;
constexpr uint32_t kOpDel = 1u << 17;
constexpr uint32_t kOpBcast = 1u << 18;   // every base of the run is read[qpos]: a one-base slice of the read laid under a longer CIGAR (numpy broadcasts it, sequences.py:790)

struct ContigInfo {
    std::string name;
    int64_t length = 0;       // Contig.length (4 for a rejected dummy)
    bool rejected = false;
    bool remote = false;      // non-rejected, but its sites live on another device (multi-GPU)
    int32_t filt_index = -1;  // index among non-rejected contigs (merge order), -1 if rejected
    // geometry of non-rejected contigs
    int64_t site_off = 0;     // first padded global site (multiple of kTileSites)
    int64_t n_tiles = 0;
    int64_t tile_off = 0;
    int64_t T = 0;            // length // 100  = strat rows
    int64_t bin_off = 0;      // first merged bin (block of T+1 bins)
    int64_t row_off = 0;      // first strat row = sum of T of predecessors (core.py:155)
    int64_t strat_off = 0;    // byte offset in the strat buffer (rows*2*nb)
    int64_t n_buckets = 0;    // length // 20000 + 1
    int64_t bucket_off = 0;
    uint64_t cov_total = 0;   // sum of all counters of this contig (all barcodes)
};

// A read's stretch of at most kSegMax emitted bases that falls into one sweep tile.
struct TileSeg {
    uint32_t e_lo, e_hi;      // emit-index range [e_lo, e_hi)
    uint32_t op_lo, op_hi;    // first and last (inclusive) emit run overlapping the range
};
// What the sweep reads of a segment (front_end.hip.inc: expand_codes_kernel): where the piece starts in
// the per-base code array and which sites of its tile it covers.
struct TilePiece {
    uint32_t e_lo;            // emit index of its first base = index into the code array
    uint32_t loc_len;         // bits 0-15 first site within the tile, bits 16-31 number of sites (<= kTileSites)
};
// Base codes of the emitted bases of a batch, one byte each, in emit order: 0..3 = A C G T on the
// reference strand, 4 = deletion, kCodeSkip = nothing to count.  kCodePad bytes precede entry 0 so
// that a thread may read the 8 bytes that START up to 7 bytes before a piece.
// The read blob in HBM holds FOUR BITS per base (round 5: 12 MB instead of 24 over PCIe for a 4000-read batch — the upload was the
// long pole of a lone update's staging): 0..3 = A C G T, 4..8 = the digits '0'..'4' and 9 = '7' (which the reference counts as
// columns 0..4 and as a deletion, sequences.py:790, 803), kNibBad = any other byte (IndexError where it is aligned).  Blob indices
// (MapPlan::seq_b / q0, EmitOp::qpos) are BASE indices: base i sits in byte i >> 1, low nibble first; every read starts on an index that
// is a multiple of four.  Round 6: a batch whose reads hold nothing but A C G T — every batch a basecaller writes — travels with TWO bits
// per base (base i in byte i >> 2, bits 2 (i & 3) up; the same base indices): 6 MB instead of 12 over PCIe, whose reads by the upload
// launches slow every kernel running next to them.  A batch with any other byte is packed again, as nibbles, once the gather has seen it.
constexpr uint32_t kNibBad = 15;
constexpr uint32_t kCodeSkip = 7;
constexpr uint32_t kCodePad = 16;
// The segments one barcode contributes to one sweep tile.  The groups of a tile are consecutive
// (ascending barcode); a tile's first group is what tile_ref points at.
struct TileRef {
    uint32_t tile;            // global sweep tile index
    uint32_t seg_lo, seg_hi;  // segments [seg_lo, seg_hi)
    uint32_t bc;              // barcode index of every segment of the group
};

// Device-side CIGAR walk (front_end.hip.inc): what the device needs to know about one chosen
// mapping.  The host resolves names, picks the best mapping and lays out the emit order; the
// CIGAR text itself is tokenised and walked on the GPU.
struct alignas(16) MapPlan {
    uint32_t cg_off, cg_len;   // the cg:Z: value inside the uploaded PAF text
    uint32_t emit0;            // batch-wide emit index of the mapping's first emitted base
    uint32_t span;             // |tend - tstart| = emitted bases the CIGAR must produce
    uint64_t site0;            // padded global site of the first emitted base
    uint32_t q0;               // blob index of the first query base the walk reads
    uint32_t q_need;           // qend - qstart = query bases the CIGAR must consume
    uint32_t seq_b, seq_len;   // the read's bytes in the blob
    uint32_t room;             // contig length - tlo: emitting past it is an IndexError
    uint32_t ops_cap;          // upper bound of its emit runs (cg_len / 2 + 1)
    uint32_t flags;            // bits 0-7 barcode, bit 8 reverse strand, bit 9 read holds a byte other than ACGT
    uint32_t seg_cap;          // upper bound of its tile segments
    int32_t q_rel;             // index of q0 inside the read (for the bounds checks; may be out of range)
    uint32_t g_first;          // index of the (tile, barcode) group of the mapping's first tile
};
static_assert(sizeof(MapPlan) == 64, "MapPlan is uploaded as is");
constexpr uint32_t kPlanRev = 1u << 8;
constexpr uint32_t kPlanCheckBases = 1u << 9;
constexpr uint32_t kPlanBroadcast = 1u << 10;      // the PAF columns select ONE base of the read: it stands under every query-consuming run

// Outcome of the device walk per mapping (read back by the host before anything is ingested).
enum : uint32_t {
    kWalkOk = 0,
    kWalkBadCigar = 1,        // malformed CIGAR (no digits before an operation / trailing digits)   [bit flags]
    kWalkBadOp = 2,           // operation letter outside MIDNSHP=XB
    kWalkOutsideRead = 4,     // an aligned run reads outside the read
    kWalkQueryMismatch = 8,   // consumed query bases != qend - qstart
    kWalkSpanMismatch = 16,   // emitted reference bases != tend - tstart
    kWalkParseMask = 0xff,
    kWalkRangeEnd = 1u << 8,  // IndexError class: emits past the end of the contig
    kWalkRangeBase = 1u << 9  // IndexError class: base other than A/C/G/T in an aligned run
};

// The emit runs are written by the parser's threads into the caller's buffer (ParseInput::
// ops_buf) as one dense chunk per thread; on the device the chunks are laid back to back.
struct OpsChunk {
    const EmitOp *host;
    size_t n;
    size_t dev_off;           // index of the chunk's first run in the device array
};

struct ParsedBatch {
    std::vector<OpsChunk> chunks;
    size_t n_ops = 0;
    std::vector<TileSeg> segs;             // grouped by tile; op_lo/op_hi are device indices
    std::vector<TileRef> tiles;            // (tile, barcode) groups, ascending (host walk; the device walk builds the list in HBM from `marks`)
    size_t n_groups = 0;                   // how many of them
    size_t n_touched_tiles = 0;            // distinct tiles among them
    // device walk: one bit per (tile, barcode) key = tile * nbarcodes + barcode that some mapping touches, and per 64-bit word
    // the number of bits set in front of it — 11 KB for chr20+21 where the group list itself is 218 KB (build_groups_kernel)
    std::vector<uint64_t> marks;
    std::vector<uint32_t> rank;
    uint64_t total_emit = 0;
    std::vector<uint64_t> emitted_per_contig;   // indexed by contig add order
    int32_t n_rec = 0;
    // device walk: the host stops after the plan pre-pass; `tiles` holds the (tile, barcode) groups
    // with empty segment ranges, the emit runs and segments are produced on the device
    std::vector<MapPlan> plans;
    std::vector<int32_t> plan_read;        // batch index of the read per plan (error messages)
    std::vector<int64_t> plan_gi;          // record-order index per plan (error precedence)
    int pre_code = 0; std::string pre_msg; int64_t pre_gi = -1;      // first KeyError / ValueError class failure of the pre-pass
    int64_t pre_range_gi = -1; std::string pre_range_msg;            // first IndexError class failure of the pre-pass
    size_t ops_cap = 0, segs_cap = 0;      // capacities the device buffers need
    bool any_check_bases = false;          // some plan was flagged kPlanCheckBases after early_walk had been called
    // Empty again, but with the vectors' memory kept: a batch's plans / groups are ~1 MB, fresh from mmap with every call they
    // cost a page fault per 4 KB — 0.15-0.2 ms of the SERIAL part of a lone update's staging (round 6).
    void reset() {
        chunks.clear(); n_ops = 0; segs.clear(); tiles.clear(); n_groups = 0; n_touched_tiles = 0; total_emit = 0; emitted_per_contig.clear(); n_rec = 0;
        marks.clear(); rank.clear();
        plans.clear(); plan_read.clear(); plan_gi.clear(); pre_code = 0; pre_msg.clear(); pre_gi = -1; pre_range_gi = -1; pre_range_msg.clear();
        ops_cap = segs_cap = 0; any_check_bases = false;
    }
};

struct ParseInput {
    const char *paf; size_t paf_len;
    const char *names; const int64_t *name_off;
    const int64_t *seq_off;
    const int32_t *barcodes;
    int32_t n_reads;
    int32_t min_len;
    int32_t nbarcodes;
    bool summary_only = false;   // choose mappings and fill the summary, no CIGAR walk
    const char *seqs = nullptr;  // the read blob seq_off indexes (null: read bases are not validated)
    EmitOp *ops_buf = nullptr;   // storage for the emit runs: ops_capacity_for(paf_len) entries
    size_t ops_cap = 0;
    int32_t n_threads = 0;       // 0: parse_threads()
    bool device_walk = false;    // fill ParsedBatch::plans / groups only; the CIGAR walk runs on the GPU
    const uint8_t *read_dirty = nullptr;   // per read: 1 if it holds a byte other than A/C/G/T (null: unknown, assume 1)
    const int64_t *seq_len = nullptr;      // per read: its length where seq_off is PADDED (the device blob keeps every read on a byte boundary: two bases per byte); null: seq_off[r + 1] - seq_off[r]
    int64_t n_tiles = 0, paf_base = 0;     // device walk: tile count of the engine (group marking)
    // Work of the caller that is independent of the line parse (gathering the reads into the upload
    // buffer, copying the text) joins pass 1's parallel region as `extra_n` more tasks;
    // `after_pass1` runs on the calling thread right after that region (start the uploads).
    std::function<void(int)> extra_fn;
    int extra_n = 0;
    int extra_first = 0;         // how many of them the workers take BEFORE the line tasks (where the caller takes no task itself: streamed grouping)
    std::function<void()> after_pass1;
    // device walk: called with the finished plans / groups BEFORE the caller's extra tasks are collected
    // (the plans carry no kPlanCheckBases flags yet: ParsedBatch::any_check_bases tells whether a
    // second look at the bases is needed once the reads have been examined)
    std::function<void(ParsedBatch &)> early_walk;
};

// Persistent worker threads for the host front end (creating 40 threads per batch cost more than
// the parsing they did): runs fn(0) .. fn(n_tasks - 1), the caller takes part.
void pool_run(int n_tasks, const std::function<void(int)> &fn);

size_t ops_capacity_for(size_t paf_len);
int cpu_budget();                // hardware threads, affinity mask and cgroup CPU quota, whichever is smallest
int parse_threads();             // BOSSX_PARSE_THREADS, default min(8, hardware threads)

// Parses the PAF text, picks the best mapping per read and expands CIGARs into emit runs.
// Returns BOSSX_OK or an error code with `err` filled.  Nothing is produced on error.
// The class of a CIGAR failure, in the reference's order of checks (paf_host.cpp): BOSSX_OK or the code.
int check_cigar_text(const char *cg, size_t n, int64_t q_len, int64_t span, bool q_ok, bool t_ok, std::string &msg);

int parse_paf_batch(const ParseInput &in, const std::vector<ContigInfo> &contigs,
                    const std::unordered_map<std::string, int32_t> &contig_index,
                    bossx_batch_summary *summary, ParsedBatch &out, std::string &err);

}  // namespace bossx
