// bossx engine: C-ABI (include/bossx.h) over the HIP kernels in kernels.hip.inc.
// gfx950 only; no CPU fallback — every entry point needs the device.
#include <hip/hip_runtime.h>

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is loaded on first use (dlopen), see RcclApi

#include "engine.hpp"
#include "kernels.hip.inc"
#include "front_end.hip.inc"
#include <execinfo.h>
#include <csignal>
#include <unistd.h>

using namespace bossx;

// The names of the last launches, in a ring: what BOSSX_BACKTRACE=1 prints next to the native frames when the process dies (a GPU memory
// fault arrives on a runtime thread, long after the launch that caused it returned: the frames alone say nothing about the kernel).
namespace { const char *g_launch_ring[64]; std::atomic<unsigned> g_launch_at{0}; }
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                     \
    do {                                                                                                \
        g_launch_ring[g_launch_at.fetch_add(1u, std::memory_order_relaxed) & 63u] = #kernel;            \
        kernel<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                    \
    } while (0)

struct bossx_engine {
    bossx_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream: the benefit chain of an update runs next to that update's sweep
    hipStream_t stream2 = nullptr;
    hipStream_t stream_up = nullptr;   // slice-wise uploads of a batch being staged (issued by the worker threads)
    hipEvent_t ev_up = nullptr;
    // ... three more queues for them: sixteen workers handing their slices to ONE stream queued up behind its lock
    static constexpr int kUpStreams = 4;
    hipStream_t stream_ups[kUpStreams] = {nullptr, nullptr, nullptr, nullptr};   // [0] = stream_up
    hipEvent_t ev_ups[kUpStreams] = {nullptr, nullptr, nullptr, nullptr};         // [0] = ev_up
    hipStream_t stream_txt = nullptr;  // the PAF text goes up on its own: the device walk needs nothing else
    hipEvent_t ev_txt = nullptr;
    hipEvent_t ev_walk = nullptr;      // behind the copy of the device walk's totals into host memory (the staging waits for THIS, not for the stream)
    // Everything a batch being staged runs on the device (plan upload, CIGAR walk, code expansion) has a stream of
    // its own: it touches the slot's buffers and the staging scratch only, never the site state, so the NEXT batch
    // can be staged while the update of the current one (sweep, chain) is still running on `stream`.  The consumer
    // (bossx_ingest_staged) makes `stream` wait for the slot's `ev_ready`.
    hipStream_t stream_stage = nullptr;       // the stream the staging's kernels go to: stream_stage_own, or the main stream (bossx_stage_stream)
    hipStream_t stream_stage_own = nullptr;
    bool stage_on_main = false;
    hipEvent_t ev_begin = nullptr, ev_chain = nullptr, ev_sweep = nullptr, ev_fhat = nullptr;
    // The read-start posterior depends on nothing but the batch's read starts: its launches (the counts' atomic adds, the terms, the
    // scaling: ~30 us behind a gap) run on a stream of their own NEXT TO the sweep and the chain, and the histogram waits for one event
    // (round 4 queued them behind the chain on the main stream, on the update's critical path).
    hipStream_t stream_fhat = nullptr; hipEvent_t ev_fhat_side = nullptr;
    double *h_fhat_pin = nullptr;      // page-locked staging of the compact f-hat
    double *d_rs_counts = nullptr; int64_t rs_windows = 0;     // read-start counts resident in HBM (bossx_fhat_reset / _add)
    int64_t *d_rs_keys = nullptr; size_t rs_keys_cap = 0;
    int64_t *h_rs_keys_pin = nullptr; hipEvent_t ev_rs_keys = nullptr;   // page-locked staging of a batch's keys
    unsigned long long *d_rs_sums = nullptr;
    uint32_t *d_tile_done = nullptr;   // [n_tiles] sweep -> chain hand-off flags (epoch stamped)
    uint32_t *d_tile_order = nullptr;  // [n_tiles] block -> tile for publishing launches: every contig's two ends first
    uint16_t *d_tile_contig = nullptr; // [n_tiles] tile -> local contig
    uint32_t epoch = 0;
    bool overlap_ok = false;           // decided at finalize (BOSSX_OVERLAP / BOSSX_NO_OVERLAP / size); cleared after a chain time-out
    bool host_armed = false;           // the host has seen ctrl.any_on set
    bool max_bits_clear = false;       // the sweep's prep launch zeroed ctrl.max_bits and no chain has run since
    int chain_ch = 256;                // bins per pipeline step of the chain kernel (fill_chain_params)
    bool sweep_published = false;      // the last sweep launch publishes its tiles (tile_done flags, agent-scope bin stores)
    bool sweep_in_flight = false;      // update_begin enqueued a sweep that no update has consumed yet
    bool chain_on_stream2 = false;     // update_benefit put the chain on stream2 (ev_chain pending)
    ChainParams last_chain{}; size_t last_chain_lds = 0;   // what that launch ran with (settle_chain reruns it serially after a time-out)
    bool finalized = false;
    bool lut_set = false;
    bool all_local = true;
    bool matrix_chain = false;      // FP64 matrix-core recurrence passed its start-up self-test
    bool chain_flow = true;         // barrier-free chain kernel (benefit_chain_flow_kernel); cleared by BOSSX_CHAIN_BARRIER=1 or after it aborted
    bool last_chain_live = false;   // the last chain launch ran next to its sweep
    bool chain_flow_fits = true;    // its LDS (buffers + the ring for the current windows) fits a CU
    int chain_flow_bufs = 4;        // difference buffers it is launched with (5 when the LDS allows)
    int chain_flow_ce = 2;          // the chain wave stores the carry into every CE-th step, the tail waves rebuild the others (BOSSX_FLOW_CE=1: all of them)
    bool chain_gc = false;          // BOSSX_CHAIN_GC=1: carries leave the chain wave through global stores (benefit_chain_flow_kernel<..., GC>) — measured slower (profiles/r03_chain_gc_experiment.txt), kept as an experiment
    double *d_carry_ring = nullptr; size_t carry_ring_cap = 0;
    // The serial recurrence chunk-parallel (chain_candidates_kernel -> chain_stitch_kernel -> benefit_chain_kernel<SEG>): the
    // default chain; the serial kernel is enqueued behind it and runs only if a segment fails its check.  BOSSX_CHAIN_SPEC=0: serial only.
    bool chain_spec = true;
    bool last_chain_spec = false;      // the last chain launch was the chunk-parallel one
    int32_t spec_mismatches = 0;       // launches in which a segment's end value differed from the stitched one (each costs a serial rerun)
    int32_t spec_recent_fail = 0, spec_since_decay = 0;      // ... of those, the recent ones (note_spec_result)
    bool upd_launched = false, upd_done = false, upd_mirrored = false;    // bossx_update_launch / _collect
    const uint8_t *mirror_valid = nullptr;     // the caller's buffer that holds d_strat's masks byte for byte (the last mask launch mirrored into it)
    double spec_est_us = 0, spec_est_serial_us = 0;     // what finalize expects of the two forms of the chain (microseconds)
    double spec_plain_share = 0;       // of all chunks the stitch adds the plain way, the fraction that falls on the longest chain's wave
    int32_t spec_pause = 0;            // updates left on the serial chain: too many chunks had to be added the plain way last time
    int64_t spec_plain_total = 0, spec_paused_updates = 0, spec_launches = 0;
    int64_t *d_chunk_off = nullptr; int64_t spec_total = 0, spec_max_segs = 0;
    int32_t spec_seg_chunks = kSpecSegChunks;   // chunks per segment block of the chain's final pass (finalize)
    double *d_spec_tab = nullptr, *d_spec_starts = nullptr;
    double *d_spec_sup = nullptr; int64_t *d_sup_off = nullptr; int64_t spec_sup_total = 0;      // chain_compose_kernel: one super-row per group of spec_seg_chunks chunks
    bool spec_no_compose = getenv("BOSSX_NO_COMPOSE") != nullptr;
    unsigned long long *d_spec_stats = nullptr;
    unsigned long long *d_cand_probe = nullptr; size_t cand_probe_waves = 0;      // BOSSX_CAND_PROBE builds only
    unsigned long long *d_spec_hash = nullptr;     // [rows] input hash of every table row (0: never built)
    unsigned long long *d_row_meta = nullptr;      // [rows] the stamp + window every table row is good for (chain_candidates_kernel's quick way out)
    // RULE: EVERY writer of d_ds stamps the tiles whose bin sums it writes — the sweep kernels do (SweepParams::stamp,
    // one new stamp per launch), anything else goes through stamp_contig_tiles() (bossx_import which = 3).  A writer that forgets
    // leaves stale table rows standing; only the segment check would notice (a serial rerun per update: bossx_chain_stats'
    // failed_checks, always collected).
    uint32_t *d_tile_stamp = nullptr;              // [n_tiles] stamp of the sweep launch that last wrote the tile's bin sums
    uint32_t stamp_counter = 0;                    // stamps handed out to sweep launches so far
    double sweep_tile_share = 1.0;                 // share of the tiles the last update's sweep rewrote
    bool tile_share_fresh = false;                 // ... and no chain has been launched since that sweep (a chain without a sweep in front of it does not trust the share)
    int32_t nb = 1;

    // native multi-GPU driver (bossx_dist_init): RCCL communicator of this engine's device
    ncclComm_t comm = nullptr;
    int32_t dist_rank = 0, dist_world = 1;
    bool dist_armed = false;           // "some strategy is on" has been seen globally (sticky)
    int64_t n_collectives = 0;
    uint8_t *d_gather = nullptr; size_t gather_cap = 0;         // bossx_dist_allgather: (world + 1) slots in HBM ...
    uint8_t *h_gather_pin = nullptr; size_t gather_pin_cap = 0; // ... and page-locked on the host

    std::vector<ContigInfo> contigs;                    // add order (rejected included)
    std::unordered_map<std::string, int32_t> index;     // name -> add order
    std::vector<int32_t> filt;                          // add-order indices of non-rejected contigs
    std::vector<std::vector<uint8_t>> host_codes;       // per contig until finalize
    int64_t n_sites_all = 0;                            // Reference.n_sites
    int64_t Gp = 0, B = 0, rows = 0, NBK = 0, n_tiles = 0, strat_bytes = 0;
    double score0 = 0, ent0 = 0;

    // device state
    uint8_t *d_state = nullptr;        // counters + state bytes, tile-major (engine.hpp: kTileStride bytes per (tile, barcode))
    uint8_t *d_touched = nullptr, *d_strat = nullptr, *d_bucket_on = nullptr;
    uint8_t *d_strat_bits = nullptr;   // packed masks (allocated on first use)
    double *d_entropy = nullptr, *d_ds = nullptr, *d_benefit = nullptr;
    void *d_convert = nullptr; size_t convert_cap = 0;     // scratch of bossx_export / _import (convert_field): bounded, persistent
    uint8_t *d_bcode = nullptr; int64_t Bc = 0;      // [nb][2][Bc] exponent code per element of d_benefit (threshold_hist_kernel -> strategy_mask_kernel)
    bool codes_valid = false;                        // d_bcode belongs to the benefits and the normaliser the device-side pick will see
    double *d_lut_score = nullptr, *d_lut_ent = nullptr, *d_fhat = nullptr;
    std::vector<char> raw_blob;       // byte-per-base copy of a batch's reads (BOSSX_HOST_WALK / BOSSX_CHECK_DEVICE_WALK only)
    unsigned long long *d_bucket_sums = nullptr, *d_stats = nullptr;
    unsigned long long *d_stats_rep = nullptr;     // kHistRep replicas of the histogram's global sums in limb form (kernels.hip.inc: PickParams); d_stats holds the (lo, hi) form where a host asks for it
    uint32_t *d_drop_count = nullptr;
    int64_t fhat_cap = 0;
    int32_t *d_err = nullptr;
    uint8_t *d_result = nullptr; size_t result_bytes = 0;   // [Ctrl | err (16 B) | contig_on]: d_ctrl, d_err, d_contig_on point into it
    Ctrl *d_ctrl = nullptr;
    uint8_t *d_contig_on = nullptr;
    long long *d_limbs = nullptr;       // multi-GPU: SUM-reducible statistics
    double *d_tails = nullptr;          // multi-GPU: last n_filt rows of every block, + 1 slot for the normaliser
    double dist_tc = 0.0; bool dist_pick_fused = false;   // short form: dist_finish's mask kernel picks the threshold
    bool norm_in_tails = false;         // bossx_dist_tails ran: dist_hist takes the (reduced) normaliser from that slot
    // contig tables (device)
    int64_t *d_tile_off = nullptr, *d_site_off = nullptr, *d_length = nullptr, *d_bin_off = nullptr,
            *d_row_off = nullptr, *d_strat_off = nullptr, *d_bucket_off = nullptr;
    int32_t *d_drop_thr = nullptr;
    uint8_t *d_local = nullptr;
    // staged batches (device-resident inputs); `slot` selects the current one
    struct Staged {
        EmitOp *d_ops = nullptr; size_t ops_cap = 0;
        uint32_t *d_tiles = nullptr; size_t tiles_cap = 0;
        uint8_t *d_blob = nullptr; size_t blob_cap = 0;
        uint32_t blob_bits = 4;          // bits per base of the blob as uploaded (2: nothing but A C G T in the batch)
        TileSeg *d_segs = nullptr; size_t segs_cap = 0;
        TileRef *d_tilerefs = nullptr; size_t tilerefs_cap = 0;
        uint8_t *d_codes = nullptr; size_t codes_cap = 0;          // per emitted base (expand_codes_kernel)
        TilePiece *d_pieces = nullptr; size_t pieces_cap = 0;      // per segment
        uint32_t n_segs = 0;
        hipEvent_t ev_ready = nullptr;   // recorded on stream_stage behind the slot's last staging kernel
        // Work enqueued on the MAIN stream may still read this slot's buffers (the sweep that applies it, or the fallback scatter
        // of flush_pending): `busy` is set by whoever enqueues such work, `ev_free` is recorded behind it, and the next
        // staging into the slot waits for that event — per slot (one remembered slot was not enough: two ingests before one
        // sweep leave the first slot's scatter in flight while the second slot is remembered)
        bool busy = false;
        hipEvent_t ev_free = nullptr;
        bool ev_free_recorded = false;   // ev_free stands behind the CURRENT content's last reader (an event left from an earlier batch proves nothing: ADVICE r4)
        int32_t *d_err = nullptr;        // the slot's own error word (a base other than A/C/G/T met while ITS codes were expanded)
        ParsedBatch pb;
        bool valid = false;
        bool emit_tiles_built = false;
    };
    std::vector<Staged> slots = std::vector<Staged>(1);
    int32_t slot = 0;
    int32_t pending_slot = -1;      // staged batch whose increments the next sweep applies
    bool pending_err_unmerged = false;   // ... and whose own error word has not joined d_err yet
    double pending_emit = 0, pending_ops = 0;
    bool touched_dirty = false;     // the `touched` byte array holds flags the next sweep must read
    bool full_sweep_needed = true;  // bin sums / bucket sums are not current everywhere (start, import, preload): sweep every tile
    bool dz_fresh = false;          // the last sweep may have zeroed sites by dropout for the first time, anywhere (see launch_sweep)
    std::vector<uint8_t> dz_fresh_k;   // ... or in these contigs only: their threshold moved in the last sweep
    std::unordered_map<const void *, size_t> lds_granted;   // dynamic LDS each chain kernel has been cleared for ON THIS DEVICE
    std::vector<int32_t> last_thr;  // dropout threshold each contig was last swept with
    uint32_t *d_tile_ref = nullptr;
    // persistent sweep blocks: hand-out counters of the launches of one update (zeroed by its prep launch), behind them
    // the count of 16-byte counter vectors the ingesting tiles wrote back (8-byte aligned); blocks per launch by instantiation
    uint32_t *d_work_ctr = nullptr; int32_t n_work_ctr = 0, work_ctr_used = 0;
    uint32_t sweep_grid[2][2] = {{0, 0}, {0, 0}};      // [INGEST][ENT]: CUs x resident blocks
    uint32_t sweep_chunk = 0;           // BOSSX_SWEEP_CHUNK: consecutive items per hand-out (0: by launch size)
    double sweep_bytes_base = 0;        // bytes of the last sweep without its counter write-back (added when the count is read)
    // device CIGAR walk (front_end.hip.inc): staging scratch shared by all slots
    char *h_paf_pin = nullptr; size_t paf_pin_cap = 0;        // PAF text, page-locked
    char *d_paf = nullptr; size_t d_paf_cap = 0;
    uint8_t *h_plan_pin = nullptr; size_t plan_pin_cap = 0;   // MapPlan[] + TileRef[] + read-back block
    MapPlan *d_plans = nullptr; size_t d_plans_cap = 0;
    unsigned long long *d_walk_probe = nullptr; size_t d_walk_probe_cap = 0;    // -DBOSSX_WALK_PROBE builds
    uint64_t nibble_repacks = 0;       // batches packed a second time, four bits per base (a byte other than A C G T in a read with a mapping)
    uint32_t walk_token = 0; uint64_t walk_spin_timeouts = 0;   // the token the walk stores behind its totals; spins that gave up (0 expected)
    bool walk_zeroed = false;          // d_walk is all zero (filled behind the batch before)
    uint8_t *d_marks = nullptr; size_t d_marks_cap = 0;       // bitmap of the batch's (tile, barcode) keys + per-word ranks (build_groups_kernel)
    uint32_t *d_walk = nullptr; size_t d_walk_cap = 0;        // n_runs | walk_err | ops_off | group_count | group_cursor | totals
    uint32_t *d_lane_scan = nullptr; size_t d_lane_scan_cap = 0;   // per mapping and lane: exclusive prefixes of the walk (pass 1 -> pass 2)
    std::vector<uint8_t> read_dirty;                          // per read: holds a byte other than A/C/G/T
    // pinned scratch
    void *h_pin = nullptr; size_t pin_cap = 0;
    void *h_blob_pin = nullptr; size_t blob_pin_cap = 0;
    EmitOp *h_ops_pin = nullptr; size_t ops_pin_cap = 0;   // parser output, pinned (H2D at PCIe rate)
    std::vector<int32_t> drop_thr_host;
    // timing
    bool timing = false;
    int timing_only = -1;       // >= 0: only this kernel is bracketed by events (every event pair is two marker packets between the update's kernels)
    hipEvent_t ev0[BOSSX_K_COUNT]{}, ev1[BOSSX_K_COUNT]{};
    bool ev_pending[BOSSX_K_COUNT]{};
    float ms_last[BOSSX_K_COUNT]{};
    double ms_total[BOSSX_K_COUNT]{};
    int64_t launches[BOSSX_K_COUNT]{};
    double bytes_last[BOSSX_K_COUNT]{};
};

namespace {

constexpr size_t kStatWords = size_t(BOSSX_HIST_BINS) * 3 + 2;
constexpr size_t kStatRepWords = size_t(kHistRep) * size_t(kLimbWords);      // the histogram's limb replicas (kernels.hip.inc: PickParams)
// page-locked blocks handed out by bossx_host_alloc: device-writable host memory (process-wide:
// bossx_host_free has no engine)
std::mutex g_host_mutex;
std::vector<std::pair<uint8_t *, size_t>> g_host_blocks;
constexpr int32_t kNoResult = 0x7fffffff;    // sentinel in the pinned result block's error field

int fail(bossx_engine *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(h, BOSSX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dev_alloc(bossx_engine *h, T **p, size_t count, bool zero = false) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, count * sizeof(T), h->stream));
    return BOSSX_OK;
}

int ensure_pin(bossx_engine *h, size_t bytes) {
    if (h->pin_cap >= bytes) return BOSSX_OK;
    if (h->h_pin) HIPCHK(hipHostFree(h->h_pin));
    h->h_pin = nullptr; h->pin_cap = 0;
    size_t cap = std::max<size_t>(bytes, 1 << 20);
    HIPCHK(hipHostMalloc(&h->h_pin, cap, hipHostMallocDefault));
    h->pin_cap = cap;
    return BOSSX_OK;
}

template <typename T>
int upload_vec(bossx_engine *h, T **dptr, const std::vector<T> &v) {
    int rc = dev_alloc(h, dptr, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BOSSX_OK;
}

ContigTable table_of(const bossx_engine *h) {
    ContigTable t;
    t.tile_off = h->d_tile_off; t.site_off = h->d_site_off; t.length = h->d_length;
    t.bin_off = h->d_bin_off; t.row_off = h->d_row_off; t.strat_off = h->d_strat_off;
    t.bucket_off = h->d_bucket_off; t.drop_thr = h->d_drop_thr; t.local = h->d_local;
    t.n = int32_t(h->filt.size());
    return t;
}

// A chain that bossx_update_benefit put on the second stream must be ordered before anything on
// the main stream that reads its results (benefit, running maximum).
int join_chain(bossx_engine *h) {
    if (h->chain_on_stream2) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_chain, 0));
        h->chain_on_stream2 = false;
    }
    return BOSSX_OK;
}

void time_begin(bossx_engine *h, int k, hipStream_t stream = nullptr) {
    if (!h->timing || (h->timing_only >= 0 && k != h->timing_only)) return;
    if (h->ev_pending[k]) {      // collect the previous launch before re-recording
        hipEventSynchronize(h->ev1[k]);
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev0[k], h->ev1[k]) == hipSuccess) { h->ms_last[k] = ms; h->ms_total[k] += ms; }
        h->ev_pending[k] = false;
    }
    hipEventRecord(h->ev0[k], stream ? stream : h->stream);
}

void time_end(bossx_engine *h, int k, double bytes, hipStream_t stream = nullptr) {
    h->launches[k]++;
    h->bytes_last[k] = bytes;
    if (!h->timing || (h->timing_only >= 0 && k != h->timing_only)) return;
    hipEventRecord(h->ev1[k], stream ? stream : h->stream);
    h->ev_pending[k] = true;
}

void time_collect(bossx_engine *h) {
    for (int k = 0; k < BOSSX_K_COUNT; ++k) {
        if (!h->ev_pending[k]) continue;
        hipEventSynchronize(h->ev1[k]);
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev0[k], h->ev1[k]) == hipSuccess) { h->ms_last[k] = ms; h->ms_total[k] += ms; }
        h->ev_pending[k] = false;
    }
}

SweepParams sweep_params(bossx_engine *h) {
    SweepParams P;
    P.st = SiteState{h->d_state, h->nb}; P.touched = h->d_touched; P.entropy = h->d_entropy;
    P.tile_ref = h->d_tile_ref; P.tiles = nullptr; P.n_groups = 0; P.pieces = nullptr; P.codes = nullptr;
    P.tile_contig = h->d_tile_contig;
    P.err_flag = h->d_err; P.use_touched = h->touched_dirty ? 1 : 0;
    P.probe = getenv("BOSSX_SWEEP_PROBE") ? h->d_stats + kStatWords + 80 : nullptr;
    if (h->pending_slot >= 0) {
        const bossx_engine::Staged &st = h->slots[size_t(h->pending_slot)];
        P.tiles = st.d_tilerefs; P.n_groups = uint32_t(st.pb.n_groups); P.pieces = st.d_pieces; P.codes = st.d_codes;
    }
    P.ds = h->d_ds; P.bucket_sums = h->d_bucket_sums; P.n_tiles = h->n_tiles;
    P.lut_score = h->d_lut_score; P.lut_ent = h->d_lut_ent; P.ct = table_of(h);
    P.Gp = h->Gp; P.B = h->B; P.NBK = h->NBK; P.nb = h->nb;
    P.score0 = h->score0; P.tiny = std::numeric_limits<double>::min();
    P.tile_done = h->d_tile_done; P.epoch = h->epoch;
    P.tile_stamp = h->d_tile_stamp; P.stamp = ++h->stamp_counter;      // (every set of sweep parameters its own stamp: monotonic is all that matters)
    P.publish = h->sweep_published ? 1 : 0;
    P.dense = 0; P.ingest_only = 0; P.ingest_first = 0; P.tile_base = 0;
    P.order = h->d_tile_order;
    P.work_ctr = nullptr; P.n_work = 0; P.chunk = 1;
    P.wb_count = h->timing ? reinterpret_cast<unsigned long long *>(h->d_work_ctr + h->n_work_ctr) : nullptr;
    return P;
}

// Rare path: a second batch arrives (or a slot is re-staged) before the sweep that would have
// applied the pending one.  Apply the pending batch with the global-atomic scatter kernel; it
// marks `touched`, which the next sweep then reads.
// Derived entropy (kernels.hip.inc: ent_save_site): what the passes that modify patterns without looking them up, and the export, need.
EntSave ent_save_of(const bossx_engine *h) {
    return (h->d_entropy && h->lut_set) ? EntSave{h->d_entropy, h->d_lut_ent, h->d_touched, h->Gp} : EntSave{nullptr, nullptr, nullptr, 0};
}

int flush_pending(bossx_engine *h) {
    if (h->pending_slot < 0) return BOSSX_OK;
    bossx_engine::Staged &st = h->slots[size_t(h->pending_slot)];
    ParsedBatch &pb = st.pb;
    const uint32_t n_tiles = uint32_t((pb.total_emit + kEmitTile - 1) / kEmitTile);
    if (!st.emit_tiles_built) {                // emit-order tiling is only needed on this path
        if (size_t(n_tiles) + 1 > st.tiles_cap) {
            if (st.d_tiles) HIPCHK(hipFree(st.d_tiles));
            st.d_tiles = nullptr;
            st.tiles_cap = (size_t(n_tiles) + 1) * 9 / 8 + 64;
            int rc = dev_alloc(h, &st.d_tiles, st.tiles_cap);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(emit_tiles_kernel, dim3(n_tiles / 256 + 1), dim3(256), 0, h->stream, st.d_ops,
                           uint32_t(pb.n_ops), n_tiles, st.d_tiles);
        st.emit_tiles_built = true;
    }
    if (h->pending_err_unmerged && st.d_err) hipLaunchKernelGGL(merge_err_kernel, dim3(1), dim3(1), 0, h->stream, st.d_err, h->d_err);
    h->pending_err_unmerged = false;
    time_begin(h, BOSSX_K_INGEST);
    const EntSave X = ent_save_of(h);
    if (X.E)         // derived entropy: the scatter modifies patterns without looking the sites up — their entropies are saved first
        hipLaunchKernelGGL(ingest_scatter_kernel<true>, dim3(n_tiles), dim3(256), 0, h->stream, st.d_ops, st.d_tiles,
                           uint32_t(pb.n_ops), pb.total_emit, st.d_blob, st.blob_bits, SiteState{h->d_state, h->nb}, h->d_touched, h->d_err, X);
    hipLaunchKernelGGL(ingest_scatter_kernel<false>, dim3(n_tiles), dim3(256), 0, h->stream, st.d_ops, st.d_tiles,
                       uint32_t(pb.n_ops), pb.total_emit, st.d_blob, st.blob_bits, SiteState{h->d_state, h->nb},
                       h->d_touched, h->d_err, X);
    time_end(h, BOSSX_K_INGEST, 6.0 * double(pb.total_emit) + 16.0 * double(pb.n_ops));
    HIPCHK(hipGetLastError());
    if (!st.ev_free) HIPCHK(hipEventCreateWithFlags(&st.ev_free, hipEventDisableTiming));
    HIPCHK(hipEventRecord(st.ev_free, h->stream));      // the scatter is the last reader of this slot's buffers
    st.busy = true; st.ev_free_recorded = true;
    h->pending_slot = -1;
    h->touched_dirty = true;
    return BOSSX_OK;
}

// One field of one (contig, barcode) between the tile-major site state and a contiguous host array of
// c.length elements: `field_off` = byte offset of the field inside a tile record (a counter plane or
// the state bytes), `elem` = bytes per site.  Whole tiles travel as one 2-D copy (pitch = the
// record stride), the contig's last, partial tile as a 1-D one.  Synchronous.
int copy_site_field(bossx_engine *h, const ContigInfo &c, int32_t b, int field_off, int elem, void *host, bool to_device) {
    const int64_t row = int64_t(kTileSites) * elem;
    const int64_t full = c.length / kTileSites, rest = c.length - full * kTileSites;
    uint8_t *dev = h->d_state + (size_t(c.tile_off) * size_t(h->nb) + size_t(b)) * size_t(kTileStride) + size_t(field_off);
    const size_t dpitch = size_t(h->nb) * size_t(kTileStride);
    uint8_t *hp = static_cast<uint8_t *>(host);
    if (full > 0) {
        if (to_device) HIPCHK(hipMemcpy2D(dev, dpitch, hp, size_t(row), size_t(row), size_t(full), hipMemcpyHostToDevice));
        else HIPCHK(hipMemcpy2D(hp, size_t(row), dev, dpitch, size_t(row), size_t(full), hipMemcpyDeviceToHost));
    }
    if (rest > 0) {
        uint8_t *d2 = dev + size_t(full) * dpitch;
        if (to_device) HIPCHK(hipMemcpy(d2, hp + full * row, size_t(rest) * size_t(elem), hipMemcpyHostToDevice));
        else HIPCHK(hipMemcpy(hp + full * row, d2, size_t(rest) * size_t(elem), hipMemcpyDeviceToHost));
    }
    return BOSSX_OK;
}

// bossx_export / bossx_import of the fields whose device layout differs from the reference's (counter planes: reference-relative;
// entropy: per-tile lane order): converted on the device through ONE bounded scratch buffer, a stretch of the contig per pass
// (a whole-contig temporary was 2.5 GB per barcode for chr1 — a checkpoint next to a nearly full HBM could fail).
int convert_field(bossx_engine *h, const ContigInfo &c, int32_t which, void *host, bool to_device) {
    const int64_t L = c.length, nb = h->nb;
    const int64_t rows = which == 0 ? nb * 5 : 1;                   // rows of the scratch per pass (entropy: one barcode at a time)
    const int64_t elem = which == 0 ? 2 : 8;
    // a fixed BYTE budget, whatever the number of rows (a floor of 1 Mi sites per pass made the scratch of a 96-barcode
    // engine 480 x 1 Mi x 2 B = 1 GB per call); ONE buffer kept on the engine, not a hipMalloc / hipFree per call
    constexpr int64_t kBudget = int64_t(32) << 20;
    const int64_t chunk = std::min<int64_t>(L, std::max<int64_t>(4096, kBudget / (rows * elem)));
    const size_t need = size_t(rows * chunk * elem);
    if (need > h->convert_cap) {
        if (h->d_convert) (void)hipFree(h->d_convert);
        h->d_convert = nullptr; h->convert_cap = 0;
        HIPCHK(hipMalloc(&h->d_convert, need));
        h->convert_cap = need;
    }
    void *tmp = h->d_convert;
    hipError_t err = hipSuccess;
    auto keep = [&](hipError_t e) { if (err == hipSuccess) err = e; };
    for (int64_t pass_b = 0; pass_b < (which == 0 ? 1 : nb) && err == hipSuccess; ++pass_b)
        for (int64_t s0 = 0; s0 < L && err == hipSuccess; s0 += chunk) {
            const int64_t n = std::min(chunk, L - s0);
            const dim3 grid(uint32_t(std::min<int64_t>((n + 255) / 256, 8192)));
            auto copy_rows = [&](bool h2d) {
                for (int64_t r = 0; r < rows; ++r) {
                    uint8_t *hp = static_cast<uint8_t *>(host) + ((which == 0 ? r : pass_b) * L + s0) * elem;
                    uint8_t *dp = static_cast<uint8_t *>(tmp) + r * n * elem;
                    keep(h2d ? hipMemcpyAsync(dp, hp, size_t(n * elem), hipMemcpyHostToDevice, h->stream)
                             : hipMemcpyAsync(hp, dp, size_t(n * elem), hipMemcpyDeviceToHost, h->stream));
                }
            };
            if (to_device) copy_rows(true);
            if (which == 0)
                hipLaunchKernelGGL(planes_convert_kernel, grid, dim3(256), 0, h->stream, SiteState{h->d_state, h->nb}, h->nb, c.site_off, s0, n,
                                   static_cast<uint16_t *>(tmp), to_device ? 1 : 0);
            else if (!to_device && ent_save_of(h).E)      // most entropies are derived from the counters (export_entropy_kernel)
                hipLaunchKernelGGL(export_entropy_kernel, grid, dim3(256), 0, h->stream, SiteState{h->d_state, h->nb}, ent_save_of(h), int32_t(pass_b), c.site_off, s0, n,
                                   static_cast<double *>(tmp));
            else
                hipLaunchKernelGGL(entropy_convert_kernel, grid, dim3(256), 0, h->stream, h->d_entropy + pass_b * h->Gp, c.site_off, s0, n,
                                   static_cast<double *>(tmp), to_device ? 1 : 0);
            keep(hipGetLastError());
            if (!to_device) copy_rows(false);
            keep(hipStreamSynchronize(h->stream));      // the scratch is reused by the next pass; the host buffer is the caller's
        }
    HIPCHK(err);
    return BOSSX_OK;
}

// A writer of d_ds other than the sweep (see the rule at bossx_engine::d_tile_stamp): all tiles of the contig get a fresh stamp.
int stamp_contig_tiles(bossx_engine *h, const ContigInfo &c) {
    if (!h->d_tile_stamp || c.remote || c.n_tiles == 0) return BOSSX_OK;
    ++h->stamp_counter;
    std::vector<uint32_t> st(size_t(c.n_tiles), h->stamp_counter);
    HIPCHK(hipMemcpy(h->d_tile_stamp + c.tile_off, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return BOSSX_OK;
}

int check_contig(bossx_engine *h, int32_t c, bool need_filt) {
    if (!h || c < 0 || c >= int32_t(h->contigs.size())) return fail(h, BOSSX_E_INVALID, "contig index out of range");
    if (need_filt && h->contigs[size_t(c)].rejected) return fail(h, BOSSX_E_INVALID, "contig is rejected");
    if (need_filt && h->contigs[size_t(c)].remote) return fail(h, BOSSX_E_INVALID, "contig is remote (owned by another device)");
    if (!h->finalized) return fail(h, BOSSX_E_INVALID, "engine not finalized");
    return BOSSX_OK;
}

}  // namespace

extern "C" {

const char *bossx_version(void) { return "bossx 0.1.0 (gfx950)"; }

const char *bossx_last_error(const bossx_engine *h) { return h ? h->err.c_str() : "null engine"; }

extern "C++" { namespace {
// The staging stream outranks the update's: the host WAITS for the CIGAR walk's totals (0.13 ms of kernels), and when a
// batch is staged ahead those kernels would otherwise queue behind the thousands of blocks of the running update's chain.
hipError_t create_stage_stream(hipStream_t *s) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
        hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) == hipSuccess)
        return hipSuccess;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
} }

extern "C++" { namespace {
// BOSSX_BACKTRACE=1: a crash inside the library prints the native frames (module + offset: resolve with addr2line on the same .so)
void crash_backtrace(int sig) {
    void *frames[48];
    const int n = backtrace(frames, 48);
    const char msg[] = "[bossx] fatal signal, native frames:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    const char msg2[] = "[bossx] the last launches, oldest first:\n";
    (void)!write(2, msg2, sizeof(msg2) - 1);
    const unsigned at = g_launch_at.load();
    for (unsigned k = at > 40u ? at - 40u : 0u; k < at; ++k) {
        const char *nm = g_launch_ring[k & 63u];
        if (nm) { (void)!write(2, "   ", 3); (void)!write(2, nm, strlen(nm)); (void)!write(2, "\n", 1); }
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
} }

int bossx_create(const bossx_config *cfg, bossx_engine **out) {
    if (getenv("BOSSX_BACKTRACE")) { signal(SIGFPE, crash_backtrace); signal(SIGSEGV, crash_backtrace); signal(SIGABRT, crash_backtrace); }
    if (!cfg || !out) return BOSSX_E_INVALID;
    *out = nullptr;
    std::unique_ptr<bossx_engine> e(new bossx_engine());
    bossx_engine *h = e.get();
    h->cfg = *cfg;
    if (cfg->nbarcodes < 1 || cfg->nbarcodes > 255) return BOSSX_E_INVALID;
    h->nb = cfg->nbarcodes;
    int ndev = 0;
    hipError_t er = hipGetDeviceCount(&ndev);
    if (er != hipSuccess || ndev <= 0) {
        // no CPU fallback by design
        fprintf(stderr, "bossx: no HIP device available (%s)\n", er == hipSuccess ? "count = 0" : hipGetErrorString(er));
        return BOSSX_E_HIP;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) return BOSSX_E_HIP;
    if (cfg->stream) {
        h->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return BOSSX_E_HIP;
        h->own_stream = true;
    }
    for (int k = 0; k < BOSSX_K_COUNT; ++k) {
        if (hipEventCreate(&h->ev0[k]) != hipSuccess || hipEventCreate(&h->ev1[k]) != hipSuccess) return BOSSX_E_HIP;
    }
    if (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_begin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_sweep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fhat, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream_fhat, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fhat_side, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream_txt, hipStreamNonBlocking) != hipSuccess ||
        create_stage_stream(&h->stream_stage_own) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_txt, hipEventDisableTiming) != hipSuccess) return BOSSX_E_HIP;
    h->stream_stage = h->stream_stage_own;
    h->stream_ups[0] = h->stream_up; h->ev_ups[0] = h->ev_up;
    for (int i = 1; i < bossx_engine::kUpStreams; ++i)
        if (hipStreamCreateWithFlags(&h->stream_ups[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_ups[i], hipEventDisableTiming) != hipSuccess) return BOSSX_E_HIP;
    *out = e.release();
    return BOSSX_OK;
}

extern "C++" { namespace { void rccl_destroy(ncclComm_t comm); } }

void bossx_destroy(bossx_engine *h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    if (h->comm) { rccl_destroy(h->comm); h->comm = nullptr; }
    if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    if (h->stream_up) { hipStreamSynchronize(h->stream_up); hipStreamDestroy(h->stream_up); }
    if (h->ev_up) hipEventDestroy(h->ev_up);
    for (int i = 1; i < bossx_engine::kUpStreams; ++i) {
        if (h->stream_ups[i]) { hipStreamSynchronize(h->stream_ups[i]); hipStreamDestroy(h->stream_ups[i]); }
        if (h->ev_ups[i]) hipEventDestroy(h->ev_ups[i]);
    }
    if (h->stream_txt) { hipStreamSynchronize(h->stream_txt); hipStreamDestroy(h->stream_txt); }
    if (h->stream_stage_own) { hipStreamSynchronize(h->stream_stage_own); hipStreamDestroy(h->stream_stage_own); }
    if (h->ev_txt) hipEventDestroy(h->ev_txt);
    if (h->ev_walk) hipEventDestroy(h->ev_walk);
    if (h->ev_begin) hipEventDestroy(h->ev_begin);
    if (h->ev_chain) hipEventDestroy(h->ev_chain);
    if (h->ev_sweep) hipEventDestroy(h->ev_sweep);
    if (h->ev_fhat) hipEventDestroy(h->ev_fhat);
    if (h->stream_fhat) { hipStreamSynchronize(h->stream_fhat); hipStreamDestroy(h->stream_fhat); }
    if (h->ev_fhat_side) hipEventDestroy(h->ev_fhat_side);
    if (h->h_fhat_pin) hipHostFree(h->h_fhat_pin);
    if (h->d_rs_counts) hipFree(h->d_rs_counts);
    if (h->d_rs_keys) hipFree(h->d_rs_keys);
    if (h->h_rs_keys_pin) hipHostFree(h->h_rs_keys_pin);
    if (h->ev_rs_keys) hipEventDestroy(h->ev_rs_keys);
    if (h->d_rs_sums) hipFree(h->d_rs_sums);
    if (h->d_tile_done) hipFree(h->d_tile_done);
    if (h->d_tile_order) hipFree(h->d_tile_order);
    if (h->d_carry_ring) hipFree(h->d_carry_ring);
#ifdef BOSSX_CAND_PROBE
    if (h->d_cand_probe) {
        std::vector<unsigned long long> cp(h->cand_probe_waves * 8, 0);
        if (hipMemcpy(cp.data(), h->d_cand_probe, cp.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t w = 0; w < h->cand_probe_waves; ++w) if (cp[w * 8 + 7]) for (int i = 0; i < 8; ++i) sum[i] += double(cp[w * 8 + size_t(i)]);
            if (sum[7] > 0)
                fprintf(stderr, "[bossx] candidates probe, last launch, per wave that reached the matrix-core loop (%.0f of %zu waves; ticks, %.2f per ns): loads + window sums %.0f, wait + differences + hash %.0f, scan + residue rule + edges %.0f, head / cut pieces %.0f, matrix core %.0f; whole wave %.0f = %.2f us\n",
                        sum[7], h->cand_probe_waves, sum[5] / (sum[6] * 10.0), sum[0] / sum[7], sum[1] / sum[7], sum[2] / sum[7], sum[3] / sum[7], sum[4] / sum[7], sum[5] / sum[7], sum[6] / sum[7] * 0.01);
        }
        hipFree(h->d_cand_probe);
    }
#endif
    if (h->d_spec_stats && getenv("BOSSX_SPEC_STATS")) {
        unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st2[2] = {0, 0};
#ifdef BOSSX_STITCH_PROBE
        unsigned long long pr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpy(pr, h->d_spec_stats + 72, sizeof(pr), hipMemcpyDeviceToHost) == hipSuccess && pr[8])
            fprintf(stderr, "[bossx] stitch probe: plain evaluations: %llu ticks loading + rounding-grid test, %llu ticks in the add loop, %llu steps of four bins (%.0f ticks per step)\n", pr[6], pr[7], pr[8], double(pr[7]) / double(pr[8]));
        if (hipMemcpy(pr, h->d_spec_stats + 72, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[bossx] stitch probe (clock64 ticks, all launches): slowest wave ever %llu; sums over waves: cut rows %llu, other slow rows %llu, pieces %llu, whole waves %llu; most cut-row time on one wave (cumulative max) %llu\n", pr[0], pr[1], pr[2], pr[3], pr[4], pr[5]);
#endif
        {
            long long ml[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (hipMemcpy(ml, h->d_spec_stats + 84, sizeof(ml), hipMemcpyDeviceToHost) == hipSuccess && ml[0]) {
                double got, want, from;
                memcpy(&got, &ml[5], 8); memcpy(&want, &ml[6], 8); memcpy(&from, &ml[7], 8);
                fprintf(stderr, "[bossx] first failed segment check: contig %lld, strand %lld, window %lld (= %lld bins), the segment that ends at chunk %lld: started from %a, ended on %a, the stitch had predicted %a\n",
                        ml[1], ml[2], ml[3], ml[8], ml[4], from, got, want);
            }
        }
        if (hipMemcpy(st2, h->d_spec_stats + 69, sizeof(st2), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[bossx] stitch: %llu cut rows walked piece by piece, %llu stretches evaluated on one rounding grid (each a round trip to the bin sums)\n", st2[1], st2[0]);
        {
            unsigned long long br[3] = {0, 0, 0};
            if (hipMemcpy(br, h->d_spec_stats + 12, sizeof(br), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[bossx] strided rows: %llu built; the stitch looked up %llu, %llu did not decide (evaluated from the exact value)\n", br[0], br[1], br[2]);
#ifdef BOSSX_STRIDED_DEBUG
            unsigned long long dbg[2] = {0, 0};
            if (hipMemcpy(dbg, h->d_spec_stats + 15, sizeof(dbg), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[bossx] strided debug: %llu strided super-row steps walked again chunk by chunk, %llu differ\n", dbg[0], dbg[1]);
#endif
        }
        if (hipMemcpy(st, h->d_spec_stats, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[bossx] stitch: %llu groups stepped through their super-row, %llu composed but out of reach, %llu not composed\n", st[9], st[10], st[11]);
        if (hipMemcpy(st, h->d_spec_stats, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[bossx] chunk-parallel chain: %llu (window, chunk) tables built + %llu left standing (inputs unchanged), %llu plain, %llu identity, %.1f candidates each; %llu chunks added plainly by the stitch; %lld updates on the serial chain meanwhile; %d mismatches (plain because: climbs more than 4 binades %llu, at most %llu; start not a normal positive number %llu)\n",
                    st[0], st[8], st[1], st[2], st[0] > st[1] + st[2] ? double(st[3]) / double(st[0] - st[1] - st[2]) : 0.0, st[4], (long long)h->spec_paused_updates, h->spec_mismatches, st[5], st[7], st[6]);
    }
    if (h->d_chunk_off) hipFree(h->d_chunk_off);
    if (h->d_spec_tab) hipFree(h->d_spec_tab);
    if (h->d_spec_starts) hipFree(h->d_spec_starts);
    if (h->d_spec_sup) hipFree(h->d_spec_sup);
    if (h->d_sup_off) hipFree(h->d_sup_off);
    if (h->d_spec_stats) hipFree(h->d_spec_stats);
    if (h->d_spec_hash) hipFree(h->d_spec_hash);
    if (h->d_row_meta) hipFree(h->d_row_meta);
    if (h->d_tile_stamp) hipFree(h->d_tile_stamp);
    if (h->d_tile_contig) hipFree(h->d_tile_contig);
    void *ptrs[] = {h->d_state, h->d_touched, h->d_strat, h->d_bucket_on, h->d_entropy, h->d_ds,
                    h->d_benefit, h->d_lut_score, h->d_lut_ent, h->d_fhat, h->d_bucket_sums, h->d_drop_count,
                    h->d_stats, h->d_tails /* base of the tails + result block */, h->d_tile_off, h->d_site_off, h->d_length, h->d_bin_off, h->d_row_off,
                    h->d_strat_off, h->d_bucket_off, h->d_drop_thr, h->d_local, h->d_limbs,
                    h->d_strat_bits, h->d_bcode, h->d_convert, h->d_stats_rep};
    for (void *p : ptrs) if (p) hipFree(p);
    for (auto &st : h->slots) { if (st.ev_ready) hipEventDestroy(st.ev_ready); if (st.ev_free) hipEventDestroy(st.ev_free); if (st.d_err) hipFree(st.d_err); if (st.d_ops) hipFree(st.d_ops); if (st.d_tiles) hipFree(st.d_tiles); if (st.d_blob) hipFree(st.d_blob); if (st.d_segs) hipFree(st.d_segs); if (st.d_tilerefs) hipFree(st.d_tilerefs); if (st.d_codes) hipFree(st.d_codes); if (st.d_pieces) hipFree(st.d_pieces); }
    if (h->d_tile_ref) hipFree(h->d_tile_ref);
    if (h->d_work_ctr) hipFree(h->d_work_ctr);
    if (h->h_pin) hipHostFree(h->h_pin);
    if (h->h_paf_pin) hipHostFree(h->h_paf_pin);
    if (h->h_plan_pin) hipHostFree(h->h_plan_pin);
    if (h->h_gather_pin) hipHostFree(h->h_gather_pin);
    if (h->d_gather) hipFree(h->d_gather);
    if (h->d_paf) hipFree(h->d_paf);
    if (h->d_plans) hipFree(h->d_plans);
    if (h->d_marks) hipFree(h->d_marks);
    if (h->d_walk) hipFree(h->d_walk);
    if (h->d_lane_scan) hipFree(h->d_lane_scan);
    if (h->h_blob_pin) hipHostFree(h->h_blob_pin);
    if (h->h_ops_pin) hipHostFree(h->h_ops_pin);
    for (int k = 0; k < BOSSX_K_COUNT; ++k) { if (h->ev0[k]) hipEventDestroy(h->ev0[k]); if (h->ev1[k]) hipEventDestroy(h->ev1[k]); }
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
}

int bossx_add_contig(bossx_engine *h, const char *name, const char *seq, int64_t length, int32_t flags) {
    const int32_t rejected = flags & BOSSX_CONTIG_REJECTED;
    const bool remote = (flags & BOSSX_CONTIG_REMOTE) != 0;
    if (!h || !name) return BOSSX_E_INVALID;
    if (h->finalized) return fail(h, BOSSX_E_INVALID, "add_contig after finalize");
    ContigInfo c;
    std::string nm(name);                       // Contig.name: first token (reference.py:29)
    size_t b = nm.find_first_not_of(" \t\n\r"), e2 = nm.find_last_not_of(" \t\n\r");
    nm = b == std::string::npos ? std::string() : nm.substr(b, e2 - b + 1);
    size_t sp = nm.find(' ');
    if (sp != std::string::npos) nm = nm.substr(0, sp);
    c.name = nm;
    c.rejected = rejected != 0;
    std::vector<uint8_t> codes;
    if (c.rejected) {
        c.length = 4;                           // Contig(seq="ACGT", rej=True), reference.py:337
    } else if (remote) {
        if (length < 1) return fail(h, BOSSX_E_INVALID, "remote contig needs its length");
        c.length = length;
        c.remote = true;
    } else {
        if (!seq || length < 1) return fail(h, BOSSX_E_INVALID, "contig needs a sequence");
        c.length = length;
        codes.resize(size_t(length));
        for (int64_t i = 0; i < length; ++i) {  // reference.py:46-68 after .upper()
            switch (seq[i]) {
                case 'C': case 'c': codes[size_t(i)] = 1; break;
                case 'G': case 'g': codes[size_t(i)] = 2; break;
                case 'T': case 't': codes[size_t(i)] = 3; break;
                default: codes[size_t(i)] = 0; break;
            }
        }
    }
    h->index[c.name] = int32_t(h->contigs.size());
    h->contigs.push_back(c);
    h->host_codes.push_back(std::move(codes));
    return BOSSX_OK;
}

int bossx_finalize(bossx_engine *h, double score0, double ent0) {
    if (!h) return BOSSX_E_INVALID;
    if (h->finalized) return fail(h, BOSSX_E_INVALID, "already finalized");
    HIPCHK(hipSetDevice(h->cfg.device));
    h->score0 = score0; h->ent0 = ent0;
    const int64_t nb = h->nb;
    std::vector<int64_t> tile_off{0}, site_off, length, bin_off{0}, row_off{0}, strat_off, bucket_off{0};
    std::vector<uint8_t> local;
    int64_t site = 0, sbytes = 0;
    h->n_sites_all = 0;
    for (size_t i = 0; i < h->contigs.size(); ++i) {
        ContigInfo &c = h->contigs[i];
        h->n_sites_all += c.length;
        if (c.rejected) continue;
        c.filt_index = int32_t(h->filt.size());
        h->filt.push_back(int32_t(i));
        c.site_off = site;
        c.n_tiles = c.remote ? 0 : (c.length + kTileSites - 1) / kTileSites;
        c.tile_off = tile_off.back();
        c.T = c.length / kWindow;
        c.bin_off = bin_off.back();
        c.row_off = row_off.back();
        c.strat_off = sbytes;
        c.n_buckets = c.length / kBucket + 1;
        c.bucket_off = bucket_off.back();
        site += c.n_tiles * kTileSites;
        sbytes += c.remote ? 0 : c.T * 2 * nb;
        tile_off.push_back(c.tile_off + c.n_tiles);
        site_off.push_back(c.site_off);
        length.push_back(c.length);
        bin_off.push_back(c.bin_off + c.T + 1);
        row_off.push_back(c.row_off + c.T);
        strat_off.push_back(c.strat_off);
        bucket_off.push_back(c.bucket_off + c.n_buckets);
        local.push_back(c.remote ? 0 : 1);
        if (c.remote) h->all_local = false;
    }
    if (h->filt.empty()) return fail(h, BOSSX_E_INVALID, "no non-rejected contig");
    if (site == 0) site = kTileSites;       // a rank that owns no contig still gets valid buffers
    h->Gp = site; h->n_tiles = tile_off.back(); h->B = bin_off.back(); h->rows = row_off.back();
    h->NBK = bucket_off.back(); h->strat_bytes = sbytes;
    if (uint64_t(h->Gp) >= (1ull << 40)) return fail(h, BOSSX_E_INVALID, "reference too large");

    int rc;
    if ((rc = dev_alloc(h, &h->d_state, size_t(std::max<int64_t>(h->n_tiles, 1)) * size_t(nb) * size_t(kTileStride) + 256, true))) return rc;
    if ((rc = dev_alloc(h, &h->d_touched, size_t(h->Gp), true))) return rc;
    if (h->cfg.track_entropy) {
        if ((rc = dev_alloc(h, &h->d_entropy, size_t(nb * h->Gp)))) return rc;
    }
    if ((rc = dev_alloc(h, &h->d_ds, size_t(nb * h->B), true))) return rc;
    if ((rc = dev_alloc(h, &h->d_benefit, size_t(nb * 2 * h->B), true))) return rc;
    h->Bc = (h->B + 15) & ~int64_t(15);
    if ((rc = dev_alloc(h, &h->d_bcode, size_t(nb * 2 * h->Bc) + 16, true))) return rc;
    if ((rc = dev_alloc(h, &h->d_strat, size_t(h->strat_bytes)))) return rc;
    HIPCHK(hipMemsetAsync(h->d_strat, 1, size_t(h->strat_bytes), h->stream));       // reference.py:118
    if ((rc = dev_alloc(h, &h->d_tile_done, size_t(h->n_tiles > 0 ? h->n_tiles : 1), true))) return rc;
    {
        // Sweep order for launches that a chain runs next to: the k-th tile from the start and the
        // k-th tile from the end of EVERY contig come before any (k+1)-th one — each contig has a
        // forward and a reverse strand walk waiting for exactly those tiles.
        std::vector<std::pair<int64_t, uint32_t>> key;
        key.reserve(size_t(h->n_tiles));
        for (int32_t fi : h->filt) {
            const ContigInfo &c = h->contigs[size_t(fi)];
            for (int64_t t = 0; t < c.n_tiles; ++t)
                key.emplace_back(std::min(t, c.n_tiles - 1 - t), uint32_t(c.tile_off + t));
        }
        // tile -> contig (one load instead of a binary search over the contig table), and the identity the
        // sweep's early loads rely on: a tile's first site is tile * kTileSites
        std::vector<uint16_t> tc((size_t(h->n_tiles > 0 ? h->n_tiles : 1) + 3) & ~size_t(1), 0);     // (read as dwords by the sweep)
        if (h->filt.size() > 65535) return fail(h, BOSSX_E_INVALID, "more than 65535 contigs");
        for (size_t k = 0; k < h->filt.size(); ++k) {
            const ContigInfo &c = h->contigs[size_t(h->filt[k])];
            if (c.n_tiles && c.site_off != c.tile_off * kTileSites) return fail(h, BOSSX_E_INVALID, "internal: site / tile geometry");
            for (int64_t t = 0; t < c.n_tiles; ++t) tc[size_t(c.tile_off + t)] = uint16_t(k);
        }
        if ((rc = dev_alloc(h, &h->d_tile_contig, tc.size()))) return rc;
        HIPCHK(hipMemcpy(h->d_tile_contig, tc.data(), tc.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        std::sort(key.begin(), key.end());
        std::vector<uint32_t> order(key.size());
        for (size_t i = 0; i < key.size(); ++i) order[i] = key[i].second;
        if ((rc = dev_alloc(h, &h->d_tile_order, order.size() ? order.size() : 1))) return rc;
        if (!order.empty())
            HIPCHK(hipMemcpy(h->d_tile_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // The chain of an update runs next to that update's sweep (measured: -5 % per update at 4.6 Mb
    // now that the host's read-length step no longer covers the sweep, -2-3 % at 110-390 Mb where
    // the chain is 5-10x longer than the sweep, -30 % at 50 Mb x 8 barcodes).
    // BOSSX_NO_OVERLAP=1 keeps them back to back.
    h->overlap_ok = getenv("BOSSX_NO_OVERLAP") == nullptr;
    h->chain_flow = getenv("BOSSX_CHAIN_BARRIER") == nullptr;
    if (const char *e = getenv("BOSSX_FLOW_CE")) h->chain_flow_ce = atoi(e) == 1 ? 1 : 2;
    if (const char *e = getenv("BOSSX_CHAIN_GC")) h->chain_gc = atoi(e) != 0;
    if (const char *e = getenv("BOSSX_CHAIN_SPEC")) h->chain_spec = atoi(e) != 0;
    if (h->chain_spec) {
        std::vector<int64_t> off(h->filt.size() + 1, 0);
        int64_t max_bins = 0;
        for (size_t k = 0; k < h->filt.size(); ++k) {
            const int64_t bins = h->contigs[size_t(h->filt[k])].T + 1;
            off[k + 1] = off[k] + (bins + kSpecL - 1) / kSpecL;
            max_bins = std::max(max_bins, bins);
        }
        h->spec_total = off.back();
        // Segment length of the final pass: a segment block costs ~4 us + 6.8 us per chunk and has a CU to itself (its step
        // arrays take 99 KB of LDS), so what a launch costs is ROUNDS of CUs x the block time — 546 blocks of four chunks at
        // chr20+21 are three rounds (93 us), 242 blocks of nine chunks one (65 us).  The cheapest length up to 32 chunks:
        {
            int cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            double best = 0.0;
            for (int32_t cch = kSpecSegChunks; cch <= 32; ++cch) {
                int64_t blocks = 0;
                for (size_t k = 0; k < h->filt.size(); ++k) {
                    const ContigInfo &c = h->contigs[size_t(h->filt[k])];
                    if (c.remote) continue;
                    blocks += ((c.T + 1 + kSpecL - 1) / kSpecL + cch - 1) / cch;
                }
                blocks *= int64_t(nb) * 2;
                const double cost = std::ceil(double(blocks) / double(cus)) * (4.0 + 6.8 * double(cch));
                if (cch == kSpecSegChunks || cost < best) { best = cost; h->spec_seg_chunks = cch; }
            }
            if (const char *e = getenv("BOSSX_SEG_CHUNKS")) h->spec_seg_chunks = std::max(1, atoi(e));
        }
        const int64_t seg_bins = int64_t(h->spec_seg_chunks) * kSpecL;
        h->spec_max_segs = (max_bins + seg_bins - 1) / seg_bins;
        // Does it pay?  Measured on MI355X (profiles/r03_*): the serial kernel walks 3.9 ns per bin of the longest contig, one
        // block per (contig, barcode, strand) and a CU each; the chunk-parallel form costs 5.2 ns per (window, strand, chunk)
        // table, 0.27 us per chunk of the longest contig for the stitch and 22 us per round of 256 segments.  Many short
        // chains (10 x 5 Mb x 8 barcodes: 160 blocks of 50 k bins) are better off serial; BOSSX_CHAIN_SPEC=2 forces the form.
        {
            // (this device's own contigs only: a rank of a multi-GPU run registers every contig but walks its own)
            int64_t segs = 0, chunks = 0, longest = 0, n_local = 0;
            for (size_t k = 0; k < h->filt.size(); ++k) {
                const ContigInfo &c = h->contigs[size_t(h->filt[k])];
                if (c.remote) continue;
                segs += (c.T + 1 + seg_bins - 1) / seg_bins; chunks += (c.T + 1 + kSpecL - 1) / kSpecL;
                longest = std::max<int64_t>(longest, c.T + 1); ++n_local;
            }
            const double blocks = double(n_local) * nb * 2;
            h->spec_est_serial_us = 3.9e-3 * double(longest) * std::ceil(blocks / 256.0);
            // (round 4: four windows per matrix operation in the candidates, ~0.08 us per chunk in the stitch)
            h->spec_est_us = 1.6e-3 * double(chunks) * nb * 2 * BOSSX_NWIN + 0.08 * double((longest + kSpecL - 1) / kSpecL) +
                             (4.0 + 6.8 * double(h->spec_seg_chunks)) * std::ceil(double(segs) * nb * 2 / 256.0) + 30.0;
            h->spec_plain_share = chunks > 0 ? double((longest + kSpecL - 1) / kSpecL) / (double(chunks) * nb * 2 * BOSSX_NWIN) : 0.0;
            const char *e = getenv("BOSSX_CHAIN_SPEC");
            if (!(e && atoi(e) == 2) && h->spec_est_us > 0.7 * h->spec_est_serial_us) h->chain_spec = false;
        }
        if (h->chain_spec) {
        const size_t rows = size_t(nb) * 2 * BOSSX_NWIN * size_t(h->spec_total);
        if ((rc = upload_vec(h, &h->d_chunk_off, off))) return rc;
        if (hipMalloc(reinterpret_cast<void **>(&h->d_spec_tab), rows * kSpecRow * sizeof(double) + 64) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&h->d_spec_starts), rows * sizeof(double) + 64) != hipSuccess) {
            (void)hipGetLastError();
            h->chain_spec = false;          // (the serial chain needs no scratch)
        } else if ((rc = dev_alloc(h, &h->d_spec_stats, 96, true)) || (rc = dev_alloc(h, &h->d_spec_hash, rows + 8, true)) ||
                   (rc = dev_alloc(h, &h->d_row_meta, rows + 8, true)) || (rc = dev_alloc(h, &h->d_tile_stamp, size_t(h->n_tiles) + 8, true))) return rc;
        if (h->chain_spec && h->spec_seg_chunks <= kStitchBatch) {
            std::vector<int64_t> soff(off.size(), 0);
            for (size_t k = 0; k + 1 < off.size(); ++k) soff[k + 1] = soff[k] + (off[k + 1] - off[k] + h->spec_seg_chunks - 1) / h->spec_seg_chunks;
            h->spec_sup_total = soff.back();
            if ((rc = upload_vec(h, &h->d_sup_off, soff))) return rc;
            if ((rc = dev_alloc(h, &h->d_spec_sup, size_t(nb) * 2 * BOSSX_NWIN * size_t(h->spec_sup_total) * kSupRow + 8, true))) return rc;
        }
        }
    }
    if (h->chain_flow_ce != 2) h->chain_gc = false;
    if ((rc = dev_alloc(h, &h->d_bucket_on, size_t(nb * h->NBK), true))) return rc;
    if ((rc = dev_alloc(h, &h->d_bucket_sums, size_t(nb * h->NBK), true))) return rc;
    if ((rc = dev_alloc(h, &h->d_drop_count, size_t(h->n_tiles), true))) return rc;
    if ((rc = dev_alloc(h, &h->d_tile_ref, size_t(h->n_tiles), true))) return rc;
    h->n_work_ctr = int32_t((h->filt.size() + 8 + 1) & ~size_t(1));       // one per launch of an update: re-swept contigs + the tile / group launches
    if ((rc = dev_alloc(h, &h->d_work_ctr, size_t(h->n_work_ctr) + 4, true))) return rc;
    {
        // persistent sweep blocks: as many as are resident at once (BOSSX_SWEEP_BLOCKS_PER_CU overrides the occupancy query)
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, h->cfg.device));
        const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const bool multi = h->nb > 1;
        const void *fn[2][2] = {{multi ? reinterpret_cast<const void *>(site_sweep_kernel<false, false, true>) : reinterpret_cast<const void *>(site_sweep1_kernel<false, true>),
                                 multi ? reinterpret_cast<const void *>(site_sweep_kernel<false, false, true>) : reinterpret_cast<const void *>(site_sweep1_kernel<false, true>)},
                                {multi ? reinterpret_cast<const void *>(site_sweep_kernel<true, false, true>) : reinterpret_cast<const void *>(site_sweep1_kernel<true, true>),
                                 multi ? reinterpret_cast<const void *>(site_sweep_kernel<true, false, true>) : reinterpret_cast<const void *>(site_sweep1_kernel<true, true>)}};
        if (const char *ce = getenv("BOSSX_SWEEP_CHUNK")) h->sweep_chunk = uint32_t(std::max(atoi(ce), 0));
        const char *e = getenv("BOSSX_SWEEP_BLOCKS_PER_CU");
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                int per_cu = 0;
                if (e && atoi(e) > 0) per_cu = atoi(e);
                else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn[a][b], 256, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
                h->sweep_grid[a][b] = uint32_t(cus) * uint32_t(per_cu);
                if (getenv("BOSSX_SWEEP_DEBUG")) fprintf(stderr, "[bossx] sweep<%d,%d>: %d CUs x %d resident blocks\n", a, b, cus, per_cu);
            }
    }
    if ((rc = dev_alloc(h, &h->d_stats, size_t(BOSSX_HIST_BINS * 3 + 4 + 128), true))) return rc;
    if ((rc = dev_alloc(h, &h->d_stats_rep, kStatRepWords + 8, true))) return rc;
    // One allocation: [halo tails | control block | error flag | per-contig switches].  The tails sit
    // right before the control block, whose first field is the running maximum: the multi-GPU MAX
    // all-reduce over "tails + 1 double" reduces the normaliser in place.  The part from the
    // control block on is what an update copies back (one D2H copy).
    {
        const size_t n_t = h->filt.size() * h->filt.size() * 2 * size_t(nb);
        const size_t bytes = (sizeof(Ctrl) + 16 + h->filt.size() + 7) & ~size_t(7);   // whole 8-byte words
        uint8_t *base = nullptr;
        if ((rc = dev_alloc(h, &base, n_t * sizeof(double) + bytes, true))) return rc;
        h->d_tails = reinterpret_cast<double *>(base);
        h->d_result = base + n_t * sizeof(double);
        h->d_ctrl = reinterpret_cast<Ctrl *>(h->d_result);
        h->d_err = reinterpret_cast<int32_t *>(h->d_result + sizeof(Ctrl));
        h->d_contig_on = h->d_result + sizeof(Ctrl) + 16;
        h->result_bytes = bytes;
    }
    if ((rc = dev_alloc(h, &h->d_limbs, size_t(BOSSX_HIST_BINS + 1) * 5, true))) return rc;
    if ((rc = dev_alloc(h, &h->d_lut_score, size_t(BOSSX_NCOMP) * 4 + 4, true))) return rc;   // + score0, tiny, 0.0 (site_sweep_kernel)
    if ((rc = dev_alloc(h, &h->d_lut_ent, size_t(BOSSX_NCOMP) * 4, true))) return rc;
    if ((rc = upload_vec(h, &h->d_tile_off, tile_off))) return rc;
    if ((rc = upload_vec(h, &h->d_site_off, site_off))) return rc;
    if ((rc = upload_vec(h, &h->d_length, length))) return rc;
    if ((rc = upload_vec(h, &h->d_bin_off, bin_off))) return rc;
    if ((rc = upload_vec(h, &h->d_row_off, row_off))) return rc;
    if ((rc = upload_vec(h, &h->d_strat_off, strat_off))) return rc;
    if ((rc = upload_vec(h, &h->d_bucket_off, bucket_off))) return rc;
    if ((rc = upload_vec(h, &h->d_local, local))) return rc;
    std::vector<int32_t> thr(h->filt.size(), -1);
    if ((rc = upload_vec(h, &h->d_drop_thr, thr))) return rc;
    // reference base codes into the site-state bytes (bits 0-1), replicated per barcode
    for (int32_t fi : h->filt) {
        const ContigInfo &c = h->contigs[size_t(fi)];
        if (c.remote) continue;
        const std::vector<uint8_t> &codes = h->host_codes[size_t(fi)];
        for (int64_t b = 0; b < nb; ++b)
            if ((rc = copy_site_field(h, c, int32_t(b), kTileMetaOff, 1, const_cast<uint8_t *>(codes.data()), true))) return rc;
    }
    if (h->d_entropy) {
        // fill with ent0 (reference.py:105)
        std::vector<double> fill(size_t(1) << 20, ent0);
        const size_t total = size_t(nb * h->Gp);
        for (size_t o = 0; o < total; o += fill.size()) {
            const size_t nfill = std::min(fill.size(), total - o);
            HIPCHK(hipMemcpy(h->d_entropy + o, fill.data(), nfill * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    h->host_codes.clear(); h->host_codes.shrink_to_fit();
    // binomial table C(q, k), k = 2..5
    uint32_t bn[4][36];
    for (int k = 2; k <= 5; ++k)
        for (int q = 0; q < 36; ++q) {
            uint64_t v = 1;
            if (q < k) v = 0;
            else for (int i = 0; i < k; ++i) v = v * uint64_t(q - i) / uint64_t(i + 1);
            bn[k - 2][q] = uint32_t(v);
        }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_binom), bn, sizeof(bn)));
    HIPCHK(hipStreamSynchronize(h->stream));
    // matrix-core recurrence: use it only if it reproduces sequential v_add_f64 bit for bit
    if (!getenv("BOSSX_NO_MATRIX_CHAIN")) {
        HIPCHK(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
        hipLaunchKernelGGL(mfma_selftest_kernel, dim3(64), dim3(64), 0, h->stream, 0x5eedULL, h->d_err);
        int32_t bad = 1;
        HIPCHK(hipMemcpyAsync(&bad, h->d_err, sizeof(bad), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->matrix_chain = (bad == 0);
        HIPCHK(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
        if (!h->matrix_chain) fprintf(stderr, "bossx: FP64 matrix-core recurrence self-test failed (%d); using the vector-ALU chain\n", bad);
    }
    h->finalized = true;
    return BOSSX_OK;
}

int bossx_set_lut(bossx_engine *h, const double *score, const double *entropy, int64_t n) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "set_lut before finalize");
    if (n != int64_t(BOSSX_NCOMP) * 4 || !score || !entropy) return fail(h, BOSSX_E_INVALID, "LUT must have NCOMP*4 entries");
    HIPCHK(hipSetDevice(h->cfg.device));
    // The caller's tables are indexed like the reference's: rank(A, C, G, T, deletion) * 4 + reference base.  On the device
    // the counter planes are reference-relative and ranked rare-first (engine.hpp, comp_rank): entry
    // rank(deletion, plane 3, plane 2, plane 1, plane 0) * 4 + ref with plane k = count of base (ref + k) & 3 holds the
    // SAME value — a permutation, every entry bit for bit the caller's.  (The four reference bases cannot share one entry:
    // numpy sums the genotypes in a fixed order, so relabelling the bases changes last bits — 42,122 of the 278,256
    // haploid rows and 2,610 of the diploid ones are bit-equal across the four bases, tests/test_host_logic.py.)
    {
        auto C = [](uint32_t q, uint32_t k) { uint64_t v = 1; if (q < k) return uint64_t(0); for (uint32_t i = 0; i < k; ++i) v = v * (q - i) / (i + 1); return v; };
        uint64_t bn[6][40];
        for (uint32_t k = 0; k < 6; ++k) for (uint32_t q = 0; q < 40; ++q) bn[k][q] = C(q, k);
        auto rank5 = [&](const uint32_t x[5]) {
            const uint32_t p1 = x[0], p2 = p1 + x[1], p3 = p2 + x[2], p4 = p3 + x[3], p5 = p4 + x[4];
            return size_t(bn[1][p1] + bn[2][p2 + 1] + bn[3][p3 + 2] + bn[4][p4 + 3] + bn[5][p5 + 4]);
        };
        std::vector<double> ds, de;
        ds.resize(size_t(n)); de.resize(size_t(n));
        size_t seen = 0;
        uint32_t c[5];
        for (c[0] = 0; c[0] < BOSSX_MAXCOV; ++c[0])
            for (c[1] = 0; c[0] + c[1] < BOSSX_MAXCOV; ++c[1])
                for (c[2] = 0; c[0] + c[1] + c[2] < BOSSX_MAXCOV; ++c[2])
                    for (c[3] = 0; c[0] + c[1] + c[2] + c[3] < BOSSX_MAXCOV; ++c[3])
                        for (c[4] = 0; c[0] + c[1] + c[2] + c[3] + c[4] < BOSSX_MAXCOV; ++c[4]) {
                            const size_t src = rank5(c);
                            for (uint32_t r = 0; r < 4; ++r) {
                                const uint32_t x[5] = {c[4], c[(r + 3) & 3], c[(r + 2) & 3], c[(r + 1) & 3], c[r]};
                                const size_t dst = rank5(x);
                                ds[dst * 4 + r] = score[src * 4 + r];
                                de[dst * 4 + r] = entropy[src * 4 + r];
                                ++seen;
                            }
                        }
        if (seen != size_t(n)) return fail(h, BOSSX_E_INVALID, "internal: composition count");
        HIPCHK(hipMemcpy(h->d_lut_score, ds.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_lut_ent, de.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice));
    }
    // the three values of the scores array that are not table entries (site_sweep_kernel selects an index, not a value)
    const double extra[4] = {h->score0, std::numeric_limits<double>::min(), 0.0, 0.0};
    HIPCHK(hipMemcpy(h->d_lut_score + size_t(n), extra, sizeof(extra), hipMemcpyHostToDevice));
    h->lut_set = true;
    return BOSSX_OK;
}

extern "C++" {
namespace {

template <typename F>
void run_threads(int nt, F &&fn) {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(fn, t);
    fn(0);
    for (auto &th : pool) th.join();
}

// The read blob of the device holds four bits per base (engine.hpp: kNibBad): a read's bytes -> nibbles, low nibble first, from a
// byte boundary on (an odd read's last high nibble is kNibBad: never addressed).  Returns true if some byte is not A / C / G / T.
struct NibTable { uint8_t t[256]; constexpr NibTable() : t() { for (int i = 0; i < 256; ++i) t[i] = uint8_t(kNibBad); t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3;
                                                                 t['0'] = 4; t['1'] = 5; t['2'] = 6; t['3'] = 7; t['4'] = 8; t['7'] = 9; } };
constexpr NibTable kNib{};
bool pack_read_scalar(const char *p, size_t n, uint8_t *dst) {
    unsigned dirty = 0;
    size_t i = 0;
    for (; i + 2 <= n; i += 2) {
        const unsigned a = kNib.t[static_cast<unsigned char>(p[i])], b = kNib.t[static_cast<unsigned char>(p[i + 1])];
        dirty |= (a | b) >> 2;
        dst[i >> 1] = uint8_t(a | (b << 4));
    }
    if (i < n) { const unsigned a = kNib.t[static_cast<unsigned char>(p[i])]; dirty |= a >> 2; dst[i >> 1] = uint8_t(a | (kNibBad << 4)); }
    return dirty != 0;
}
// 32 bytes per step: a byte is one of A C G T iff a 16-entry table indexed by its low nibble (A = 0x41, C = 0x43, T = 0x54,
// G = 0x47) gives the byte back (the other entries hold a byte whose own low nibble differs from their index: they never match);
// a second table over the same index gives the code.  A block with any other byte (rare: digits, N) goes through the scalar table.
__attribute__((target("avx2"))) bool pack_read_avx2(const char *p, size_t n, uint8_t *dst) {
    const __m256i lut = _mm256_setr_epi8(-1, 0x41, -1, 0x43, 0x54, -1, -1, 0x47, -1, -1, -1, -1, -1, -1, -1, 0,
                                         -1, 0x41, -1, 0x43, 0x54, -1, -1, 0x47, -1, -1, -1, -1, -1, -1, -1, 0);
    const __m256i code = _mm256_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i low = _mm256_set1_epi8(0x0f), pair = _mm256_set1_epi16(0x1001);
    bool dirty = false;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i));
        const __m256i lo = _mm256_and_si256(v, low);
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, lo), v)) != -1) { dirty |= pack_read_scalar(p + i, 32, dst + (i >> 1)); continue; }
        const __m256i w = _mm256_maddubs_epi16(_mm256_shuffle_epi8(code, lo), pair);                      // even + 16 * odd, per pair of bytes
        const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(w, w), 0x08);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + (i >> 1)), _mm256_castsi256_si128(pk));
    }
    if (i < n) dirty |= pack_read_scalar(p + i, n - i, dst + (i >> 1));
    return dirty;
}
// Two bits per base (engine.hpp), from a byte boundary on: true — and the read's bytes are void — if some byte is not A / C / G / T
// (the whole batch is then packed again, as nibbles).
bool pack_read2_scalar(const char *p, size_t n, uint8_t *dst) {
    unsigned dirty = 0;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const unsigned a = kNib.t[static_cast<unsigned char>(p[i])], b = kNib.t[static_cast<unsigned char>(p[i + 1])];
        const unsigned c = kNib.t[static_cast<unsigned char>(p[i + 2])], d = kNib.t[static_cast<unsigned char>(p[i + 3])];
        dirty |= (a | b | c | d) >> 2;
        dst[i >> 2] = uint8_t((a & 3u) | ((b & 3u) << 2) | ((c & 3u) << 4) | ((d & 3u) << 6));
    }
    if (i < n) {
        unsigned v = 0;
        for (size_t k = i; k < n; ++k) { const unsigned a = kNib.t[static_cast<unsigned char>(p[k])]; dirty |= a >> 2; v |= (a & 3u) << ((k - i) * 2); }
        dst[i >> 2] = uint8_t(v);
    }
    return dirty != 0;
}
__attribute__((target("avx2"))) bool pack_read2_avx2(const char *p, size_t n, uint8_t *dst) {
    const __m256i lut = _mm256_setr_epi8(-1, 0x41, -1, 0x43, 0x54, -1, -1, 0x47, -1, -1, -1, -1, -1, -1, -1, 0,
                                         -1, 0x41, -1, 0x43, 0x54, -1, -1, 0x47, -1, -1, -1, -1, -1, -1, -1, 0);
    const __m256i code = _mm256_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i low = _mm256_set1_epi8(0x0f), pair = _mm256_set1_epi16(0x0401), quad = _mm256_set1_epi32(0x00100001);
    bool dirty = false;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i));
        const __m256i lo = _mm256_and_si256(v, low);
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, lo), v)) != -1) { dirty = true; continue; }
        const __m256i w = _mm256_maddubs_epi16(_mm256_shuffle_epi8(code, lo), pair);      // even + 4 * odd, per pair of bytes
        const __m256i q = _mm256_madd_epi16(w, quad);                                      // ... + 16 * the next pair: one byte per four bases, in 32-bit lanes
        const __m256i h16 = _mm256_packus_epi32(q, q);
        const __m256i b8 = _mm256_packus_epi16(h16, h16);                                  // bytes 0..3 in the low 128-bit lane's first dword, 4..7 in the high lane's
        const uint32_t lo4 = uint32_t(_mm256_extract_epi32(b8, 0)), hi4 = uint32_t(_mm256_extract_epi32(b8, 4));
        const uint64_t out = uint64_t(lo4) | (uint64_t(hi4) << 32);
        memcpy(dst + (i >> 2), &out, 8);
    }
    if (i < n) dirty |= pack_read2_scalar(p + i, n - i, dst + (i >> 2));
    return dirty;
}
bool pack_read2(const char *p, size_t n, uint8_t *dst) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? pack_read2_avx2(p, n, dst) : pack_read2_scalar(p, n, dst);
}
bool pack_read(const char *p, size_t n, uint8_t *dst) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? pack_read_avx2(p, n, dst) : pack_read_scalar(p, n, dst);
}

template <typename T>
int grow_dev(bossx_engine *h, T **p, size_t *cap, size_t need, size_t slack) {
    if (need <= *cap) return BOSSX_OK;
    if (getenv("BOSSX_STAGE_TIMING")) fprintf(stderr, "[bossx] grow_dev: %zu -> %zu elements of %zu bytes\n", *cap, need, sizeof(T));
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t c = need * 5 / 4 + slack;      // (a quarter of headroom: a batch a little larger than every one before must not cost a hipFree — a device-wide wait — and a hipMalloc inside an update)
    int rc = dev_alloc(h, p, c);
    if (rc) return rc;
    *cap = c;
    return BOSSX_OK;
}

template <typename T>
int grow_pin(bossx_engine *h, T **p, size_t *cap, size_t need) {
    if (need <= *cap) return BOSSX_OK;
    if (getenv("BOSSX_STAGE_TIMING")) fprintf(stderr, "[bossx] grow_pin: %zu -> %zu elements of %zu bytes\n", *cap, need, sizeof(T));
    if (*p) HIPCHK(hipHostFree(*p));
    *p = nullptr; *cap = 0;
    const size_t c = need * 5 / 4 + 4096;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(p), c * sizeof(T), hipHostMallocDefault));
    *cap = c;
    return BOSSX_OK;
}

}  // namespace
}  // extern "C++"

namespace {

// Host (page-locked, device-mapped) -> device copy of the staging: the engine's own kernel (front_end.hip.inc: upload_kernel)
// unless BOSSX_ENGINE_COPIES=1 asks for hipMemcpyAsync (the copy engine; what rounds 1-5 used).  Callable from the worker threads.
hipError_t upload_async(void *dst, const void *src_pinned, size_t bytes, hipStream_t stream, size_t want_blocks = 0) {
    static const bool engine_copies = getenv("BOSSX_ENGINE_COPIES") != nullptr;
    if (bytes == 0) return hipSuccess;
    if (engine_copies) return hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, stream);
    // enough waves to keep PCIe reads in flight (each thread holds four 16-byte loads), not more than the copy needs
    const size_t vec = (bytes + 15) / 16;
    // (few blocks: what a launch keeps in flight — blocks x 256 threads x 64 bytes — queues up IN FRONT of every other upload's requests on
    // the one PCIe link; with 48 blocks per launch and five launches at once the 250 KB of plans took 112 us to cross, behind 4 MB of reads)
    static const size_t max_blocks = getenv("BOSSX_UPLOAD_BLOCKS") ? size_t(std::max(atoi(getenv("BOSSX_UPLOAD_BLOCKS")), 1)) : 8;
    // (`want_blocks`: a small upload the device waits for — the plans — asks for all of its bytes in ONE round of requests, so it queues once
    // behind what the bulk uploads have in flight instead of once per round)
    const uint32_t blocks = uint32_t(std::min<size_t>(std::max<size_t>((vec + 1023) / 1024, 1), want_blocks ? want_blocks : max_blocks));
    hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint8_t *>(dst), static_cast<const uint8_t *>(src_pinned), bytes);
    return hipGetLastError();
}

const char *walk_message(uint32_t e) {
    if (e & kWalkBadCigar) return "no CIGAR operation, or a run of 10^9 bases or more";
    if (e & kWalkOutsideRead) return "CIGAR walks outside the read";
    if (e & kWalkQueryMismatch) return "CIGAR does not consume qend - qstart query bases";
    if (e & kWalkSpanMismatch) return "CIGAR does not span tend - tstart reference bases";
    if (e & kWalkRangeEnd) return "mapping extends past the end of its contig";
    if (e & kWalkRangeBase) return "base other than A/C/G/T inside an aligned segment";
    return "";
}

// Host walk (paf_host.cpp): the whole front end on the host, emit runs and segments uploaded.
// Kept as the checker of the device walk (BOSSX_CHECK_DEVICE_WALK=1) and as an escape hatch
// (BOSSX_HOST_WALK=1).
int stage_host_walk(bossx_engine *h, bossx_engine::Staged &st, ParseInput in, bossx_batch_summary *summary, ParsedBatch &pb) {
    const size_t ops_need = ops_capacity_for(in.paf_len);
    if (ops_need > h->ops_pin_cap) {
        if (h->h_ops_pin) HIPCHK(hipHostFree(h->h_ops_pin));
        h->h_ops_pin = nullptr; h->ops_pin_cap = 0;
        const size_t cap = ops_need * 5 / 4;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_ops_pin), cap * sizeof(EmitOp), hipHostMallocDefault));
        h->ops_pin_cap = cap;
    }
    in.ops_buf = h->h_ops_pin; in.ops_cap = h->ops_pin_cap;
    in.device_walk = false;
    std::string err;
    int rc = parse_paf_batch(in, h->contigs, h->index, summary, pb, err);
    if (rc) return fail(h, rc, err);
    if ((rc = grow_dev(h, &st.d_ops, &st.ops_cap, pb.n_ops, 1024))) return rc;
    if ((rc = grow_dev(h, &st.d_segs, &st.segs_cap, pb.segs.size(), 64))) return rc;
    if ((rc = grow_dev(h, &st.d_tilerefs, &st.tilerefs_cap, pb.tiles.size(), 64))) return rc;
    if (pb.n_ops) {
        HIPCHK(hipMemcpyAsync(st.d_segs, pb.segs.data(), pb.segs.size() * sizeof(TileSeg), hipMemcpyHostToDevice, h->stream_stage));
        HIPCHK(hipMemcpyAsync(st.d_tilerefs, pb.tiles.data(), pb.tiles.size() * sizeof(TileRef), hipMemcpyHostToDevice, h->stream_stage));
        for (const OpsChunk &ck : pb.chunks)
            HIPCHK(hipMemcpyAsync(st.d_ops + ck.dev_off, ck.host, ck.n * sizeof(EmitOp), hipMemcpyHostToDevice, h->stream_stage));
    }
    HIPCHK(hipStreamSynchronize(h->stream_stage));
    return BOSSX_OK;
}

// BOSSX_CHECK_DEVICE_WALK=1: the device walk's emit runs must equal the host walk's exactly, and its
// segments group by group (as sets: their order inside a group is not defined).
int check_device_walk(bossx_engine *h, bossx_engine::Staged &st, const ParseInput &in0, const ParsedBatch &dev, uint32_t n_segs) {
    ParseInput in = in0;
    std::vector<EmitOp> buf(ops_capacity_for(in.paf_len));
    in.ops_buf = buf.data(); in.ops_cap = buf.size(); in.device_walk = false;
    in.extra_n = 0; in.extra_fn = nullptr; in.after_pass1 = nullptr;
    ParsedBatch pb;
    std::string err;
    int rc = parse_paf_batch(in, h->contigs, h->index, nullptr, pb, err);
    if (rc) return fail(h, BOSSX_E_INVALID, "device walk check: the host walk fails where the device walk passed: " + err);
    if (pb.n_ops != dev.n_ops || pb.segs.size() != n_segs || pb.tiles.size() != dev.n_groups || pb.total_emit != dev.total_emit ||
        pb.n_touched_tiles != dev.n_touched_tiles)
        return fail(h, BOSSX_E_INVALID, "device walk check: counts differ (runs " + std::to_string(dev.n_ops) + " vs " + std::to_string(pb.n_ops) +
                                           ", segments " + std::to_string(n_segs) + " vs " + std::to_string(pb.segs.size()) +
                                           ", groups " + std::to_string(dev.n_groups) + " vs " + std::to_string(pb.tiles.size()) + ")");
    std::vector<EmitOp> ops(pb.n_ops), dops(pb.n_ops);
    for (const OpsChunk &ck : pb.chunks) memcpy(ops.data() + ck.dev_off, ck.host, ck.n * sizeof(EmitOp));
    std::vector<TileSeg> dsegs(n_segs);
    std::vector<TileRef> dgroups(dev.n_groups);
    if (pb.n_ops) HIPCHK(hipMemcpy(dops.data(), st.d_ops, pb.n_ops * sizeof(EmitOp), hipMemcpyDeviceToHost));
    if (n_segs) HIPCHK(hipMemcpy(dsegs.data(), st.d_segs, n_segs * sizeof(TileSeg), hipMemcpyDeviceToHost));
    if (!dgroups.empty()) HIPCHK(hipMemcpy(dgroups.data(), st.d_tilerefs, dgroups.size() * sizeof(TileRef), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ops.size(); ++i) {
        EmitOp a = ops[i], b = dops[i];
        if (a.meta & kOpDel) a.qpos = b.qpos = 0;
        if (memcmp(&a, &b, sizeof(EmitOp)))
            return fail(h, BOSSX_E_INVALID, "device walk check: emit run " + std::to_string(i) + " differs");
    }
    auto key = [](const TileSeg &x) { return std::make_tuple(x.e_lo, x.e_hi, x.op_lo, x.op_hi); };
    for (size_t g = 0; g < dgroups.size(); ++g) {
        const TileRef &a = pb.tiles[g], &b = dgroups[g];
        if (a.tile != b.tile || a.bc != b.bc || a.seg_hi - a.seg_lo != b.seg_hi - b.seg_lo || a.seg_lo != b.seg_lo)
            return fail(h, BOSSX_E_INVALID, "device walk check: group " + std::to_string(g) + " differs");
        std::vector<TileSeg> x(pb.segs.begin() + a.seg_lo, pb.segs.begin() + a.seg_hi), y(dsegs.begin() + b.seg_lo, dsegs.begin() + b.seg_hi);
        auto lt = [&](const TileSeg &p, const TileSeg &q) { return key(p) < key(q); };
        std::sort(x.begin(), x.end(), lt); std::sort(y.begin(), y.end(), lt);
        for (size_t i = 0; i < x.size(); ++i)
            if (key(x[i]) != key(y[i])) return fail(h, BOSSX_E_INVALID, "device walk check: segments of group " + std::to_string(g) + " differ");
    }
    return BOSSX_OK;
}

// Parse + upload one batch into the selected slot.  `seqs` is the read blob; with `seq_ptrs` the
// reads are still separate strings and are gathered into `seqs` (page-locked) here.  The gather, the
// look for bytes other than A/C/G/T, the copy of the PAF text into page-locked memory and the line
// parse are ONE parallel region; the uploads start as soon as it ends and overlap with the rest of
// the host work.
// `seq_off`: PADDED base offsets of the reads in the blob (every read on an even index: two bases per byte on the device);
// `seq_len`: their lengths; `seq_ptrs`: where each read's bytes are now (the caller's strings: borrowed for the call).
int stage_core(bossx_engine *h, const char *paf, size_t paf_len, const char *names, const int64_t *name_off,
               const int64_t *seq_off, const int64_t *seq_len, const char *const *seq_ptrs, const int32_t *barcodes, int32_t n_reads,
               int32_t min_len, bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases) {
    const bool timing = getenv("BOSSX_STAGE_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    // the previous batch's uploads and walk may still use the staging scratch; a sweep enqueued on the main
    // stream may still read THIS slot's buffers (not when the caller stages ahead into another slot)
    HIPCHK(hipStreamSynchronize(h->stream_stage));
    {
        // bossx_stage_stream: a batch that is consumed right away is staged on the main stream — the sweep then follows the expansion in
        // ONE queue (a wait across queues stood between them: ~25 us of an idle GPU per lone update); BOSSX_STAGE_OWN_STREAM=1: never
        hipStream_t want = (h->stage_on_main && !getenv("BOSSX_STAGE_OWN_STREAM")) ? h->stream : h->stream_stage_own;
        if (want != h->stream_stage) { h->stream_stage = want; HIPCHK(hipStreamSynchronize(want)); }
    }
    {
        bossx_engine::Staged &cur = h->slots[size_t(h->slot)];
        // (busy without a recorded event: the batch was handed to the main stream, bossx_ingest_staged, and its reader — the sweep or the
        // fallback scatter — has not been enqueued yet: only the stream itself can be waited for)
        if (cur.busy) { if (cur.ev_free && cur.ev_free_recorded) HIPCHK(hipEventSynchronize(cur.ev_free)); else HIPCHK(hipStreamSynchronize(h->stream)); cur.busy = false; }
    }
    const size_t blob_bytes = n_reads > 0 ? size_t(seq_off[n_reads]) : 0;
    if (blob_bytes >= (size_t(1) << 32)) return fail(h, BOSSX_E_RANGE, "read blob larger than 4 GiB");
    if (paf_len >= (size_t(1) << 32)) return fail(h, BOSSX_E_RANGE, "PAF text larger than 4 GiB");
    if (h->pending_slot == h->slot) {     // the slot still feeds the next sweep: apply it now
        if ((rc = flush_pending(h))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    bossx_engine::Staged &st = h->slots[size_t(h->slot)];
    st.valid = false;
    st.emit_tiles_built = false;
    // Inputs are borrowed for the call only: the worker threads start asynchronous copies out of the
    // caller's text (and, on the bossx_stage_batch path, out of its pageable read blob), so no way out
    // of this function may leave one in flight.
    struct UploadGuard {
        bossx_engine *h; bool ok = false;
        ~UploadGuard() {
            if (ok) return;
            hipStreamSynchronize(h->stream_txt);
            for (int i = 0; i < bossx_engine::kUpStreams; ++i) hipStreamSynchronize(h->stream_ups[i]);
            hipStreamSynchronize(h->stream_stage);
        }
    } upload_guard{h};
    if ((rc = grow_dev(h, &st.d_blob, &st.blob_cap, blob_bytes / 2 + 1024, 0))) return rc;
    // (the page-locked buffer is free: every upload out of it was waited for by the stage stream, synchronised above)
    if (blob_bytes / 2 + 64 > h->blob_pin_cap) {
        if (h->h_blob_pin) HIPCHK(hipHostFree(h->h_blob_pin));
        h->h_blob_pin = nullptr; h->blob_pin_cap = 0;
        const size_t cap = (blob_bytes / 2 + 64) * 5 / 4;
        HIPCHK(hipHostMalloc(&h->h_blob_pin, cap, hipHostMallocDefault));
        h->blob_pin_cap = cap;
    }
    uint8_t *const packed = static_cast<uint8_t *>(h->h_blob_pin);
    ParseInput in{paf ? paf : "", paf ? paf_len : 0, names, name_off, seq_off, barcodes, n_reads, min_len, h->nb};
    in.seq_len = seq_len;
    const bool host_walk = getenv("BOSSX_HOST_WALK") != nullptr;
    // the host walk (and the comparison of the two walks) looks at the bases themselves: a byte-per-base copy at the same offsets
    const bool raw_copy = host_walk || getenv("BOSSX_CHECK_DEVICE_WALK") != nullptr;
    if (raw_copy) h->raw_blob.assign(blob_bytes + 64, 'A');
    char *const seqs = raw_copy ? h->raw_blob.data() : nullptr;
    in.seqs = seqs;
    if (!host_walk) {
        if ((rc = grow_pin(h, &h->h_paf_pin, &h->paf_pin_cap, in.paf_len + 64))) return rc;
        if ((rc = grow_dev(h, &h->d_paf, &h->d_paf_cap, in.paf_len + 64, 4096))) return rc;
    }
    // two bits per base on the way up (engine.hpp) unless a read turns out to hold something else than A C G T; BOSSX_BLOB_NIBBLES=1: round 5's four
    const bool two_bits = !getenv("BOSSX_BLOB_NIBBLES") && !getenv("BOSSX_HOST_WALK");
    const int blob_shift = two_bits ? 2 : 1;
    st.blob_bits = two_bits ? 2u : 4u;
    // ---- the caller's share of pass 1's parallel region --------------------------------------
    h->read_dirty.assign(size_t(n_reads), 0);
    const int n_g = blob_bytes > (size_t(1) << 20) ? parse_threads() : (n_reads > 0 ? 1 : 0);
    // the PAF text in slices of ~384 KB (round 5: four slices behind the line tasks — the device walk then waited for the text's
    // upload, which started when the lines were done): copied and handed to the upload stream before anything else
    const int n_c = host_walk ? 0 : (in.paf_len > (size_t(1) << 20) ? int(std::min<size_t>(32, (in.paf_len + (size_t(384) << 10) - 1) / (size_t(384) << 10))) : (in.paf_len ? 1 : 0));
    in.extra_n = n_g + n_c;
    // The text slices come first; the gather of the reads BEHIND the line tasks: the device walk needs the text and the plans (lines ->
    // grouping -> plans, all on the host's critical path), the 12 MB of reads (0.3 ms of PCIe) are wanted only when the walk is over.
    // (Gather first was tried: alternated update by update in one process — scripts/ab_inproc.py — it costs 0.1 ms; BOSSX_GATHER_FIRST=1.)
    in.extra_first = getenv("BOSSX_TEXT_LAST") ? 0 : (getenv("BOSSX_GATHER_FIRST") ? n_c + n_g : n_c);
    // Every slice goes up as soon as it is gathered (the worker that filled it issues the copy, on a
    // stream of its own): the PCIe transfer of the 24 MB of reads overlaps with the gather instead
    // of following it.  The text slices come first and travel on a third stream: the device walk
    // waits for THEM only (it reads the bases of a mapping only where the host saw a byte other
    // than A/C/G/T in the read), the sweep that applies the batch for the reads.
    std::atomic<int> up_fail{0}, any_dirty{0}, txt_done{0};
    int n_up = bossx_engine::kUpStreams;
    if (const char *e = getenv("BOSSX_UP_STREAMS")) n_up = std::min(std::max(atoi(e), 1), int(bossx_engine::kUpStreams));
    const int dev = h->cfg.device;
    in.extra_fn = [&, n_g, n_c, dev, n_up](int t) {
        static thread_local int dev_set = -1;
        if (dev_set != dev) { if (hipSetDevice(dev) != hipSuccess) up_fail.store(1); dev_set = dev; }
        if (t < n_c) {
            const size_t lo = in.paf_len * size_t(t) / size_t(n_c), hi = in.paf_len * size_t(t + 1) / size_t(n_c);
            const auto c0_ = std::chrono::steady_clock::now();
            memcpy(h->h_paf_pin + lo, in.paf + lo, hi - lo);
            const auto c1_ = std::chrono::steady_clock::now();
            if (hi > lo && upload_async(h->d_paf + lo, h->h_paf_pin + lo, hi - lo, h->stream_txt) != hipSuccess) up_fail.store(1);
            if (timing) {
                const double a_ = std::chrono::duration<double, std::milli>(c1_ - c0_).count(), b_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c1_).count();
                if (a_ + b_ > 2.0) fprintf(stderr, "[bossx] text slice %d: memcpy %.2f ms, hipMemcpyAsync %.2f ms\n", t, a_, b_);
            }
            txt_done.fetch_add(1, std::memory_order_release);
        } else {                  // reads [b, e) of a byte-balanced slice: gather (if still scattered) and look at the bases
            const int g = t - n_c;
            const size_t lo_b = blob_bytes * size_t(g) / size_t(n_g), hi_b = blob_bytes * size_t(g + 1) / size_t(n_g);
            const int64_t *bp = std::lower_bound(seq_off, seq_off + n_reads, int64_t(lo_b));
            const int64_t *ep = g + 1 == n_g ? seq_off + n_reads : std::lower_bound(seq_off, seq_off + n_reads, int64_t(hi_b));
            const int32_t i0 = int32_t(bp - seq_off), i1 = int32_t(ep - seq_off);
            bool dirty = false;
            for (int32_t i = i0; i < i1; ++i) {
                const size_t len = size_t(seq_len[i]);
                if (raw_copy) memcpy(seqs + seq_off[i], seq_ptrs[i], len);
                // (what is not A C G T is looked at again where it is aligned — after the whole batch has been packed again, as nibbles)
                const bool d = two_bits ? pack_read2(seq_ptrs[i], len, packed + (seq_off[i] >> 2)) : pack_read(seq_ptrs[i], len, packed + (seq_off[i] >> 1));
                h->read_dirty[size_t(i)] = d ? 1 : 0;
                dirty |= d;
            }
            if (dirty) any_dirty.store(1, std::memory_order_relaxed);
            if (i1 > i0 && seq_off[i1] > seq_off[i0] &&
                upload_async(st.d_blob + (seq_off[i0] >> blob_shift), packed + (seq_off[i0] >> blob_shift), size_t(seq_off[i1] - seq_off[i0]) >> blob_shift,
                             h->stream_ups[g % n_up]) != hipSuccess)
                up_fail.store(1);
        }
    };
    hipError_t up_err = hipSuccess;
    // The reads may still be on their way when the walk runs — unless it has bases to look at, or the
    // blob is the caller's own memory (borrowed for the call only) rather than our page-locked copy.
    bool reads_awaited = false;
    auto await_reads = [&]() {
        if (reads_awaited || up_err != hipSuccess) return;
        for (int i = 0; i < bossx_engine::kUpStreams && up_err == hipSuccess; ++i) up_err = hipStreamWaitEvent(h->stream_stage, h->ev_ups[i], 0);
        reads_awaited = true;
    };
    const bool defer_reads = !host_walk && !getenv("BOSSX_NO_DEFER_READS");
    in.after_pass1 = [&]() {      // everything staged on the upload streams precedes what follows on the main one
        if (up_fail.load()) { up_err = hipErrorUnknown; return; }
        up_err = hipEventRecord(h->ev_txt, h->stream_txt);
        if (up_err == hipSuccess) up_err = hipStreamWaitEvent(h->stream_stage, h->ev_txt, 0);
        for (int i = 0; i < bossx_engine::kUpStreams && up_err == hipSuccess; ++i) up_err = hipEventRecord(h->ev_ups[i], h->stream_ups[i]);
        if (!defer_reads || any_dirty.load()) await_reads();
    };
    ParsedBatch pb = std::move(st.pb);      // (the slot's previous batch is gone: its vectors' memory is reused — ParsedBatch::reset)
    uint32_t dev_n_segs = 0;
    bool expand_enqueued = false;
    const auto t_pre = std::chrono::steady_clock::now();
    if (host_walk) {
        if ((rc = stage_host_walk(h, st, in, summary, pb))) return rc;
        HIPCHK(up_err);
    } else {
        // ---- host: lines -> records -> best mapping per read -> plans + (tile, barcode) groups ----
        in.device_walk = true;
        in.read_dirty = h->read_dirty.data();
        in.n_tiles = h->n_tiles;
        // ---- device walk, launched from INSIDE the parse as soon as the plans exist (early_walk): it needs
        // the text and the plans, not the reads — the workers are still gathering and uploading those
        // (the gather into page-locked memory is the long pole of the host side) ----------------------
        int early_rc = BOSSX_OK;
        uint32_t n_plans = 0, n_groups = 0;
        WalkParams W{};
        uint32_t *back = nullptr;
        uint32_t back_token = 0;          // != 0: the walk itself stores its totals to `back`, this token last
        size_t plan_bytes = 0, group_bytes = 0;
        auto t_launched = std::chrono::steady_clock::now(), t_plans = t_launched;
        in.early_walk = [&](ParsedBatch &pbe) {
            t_plans = std::chrono::steady_clock::now();
            n_plans = uint32_t(pbe.plans.size()); n_groups = uint32_t(pbe.n_groups);
            if (!n_plans) return;
            auto go = [&]() -> int {
                int rc2;
                const auto g0_ = std::chrono::steady_clock::now();
                auto lap = [&](const char *what) {      // (BOSSX_STAGE_TIMING: which step of a slow walk launch took the time)
                    if (!timing) return;
                    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0_).count();
                    if (ms > 2.0) fprintf(stderr, "[bossx] walk launch: %.2f ms in, behind %s\n", ms, what);
                };
                for (int sp = 0; txt_done.load(std::memory_order_acquire) < n_c; ++sp) { if (sp < 2000) _mm_pause(); else std::this_thread::yield(); }     // the text slices have been handed to the DMA engine (spin, then yield: see WorkPool::start)
                lap("the wait for the text slices");
                if (up_fail.load()) return fail(h, BOSSX_E_HIP, "upload of the PAF text failed");
                HIPCHK(hipEventRecord(h->ev_txt, h->stream_txt));
                const bool two_pass = getenv("BOSSX_TWO_PASS_WALK") != nullptr;       // the round-5 form of the walk (count, scan, emit)
                if (two_pass) HIPCHK(hipStreamWaitEvent(h->stream_stage, h->ev_txt, 0));      // (single pass: only the walk itself waits for the text)
                lap("event record + stream wait");
                if ((rc2 = grow_dev(h, &st.d_ops, &st.ops_cap, pbe.ops_cap, 1024))) return rc2;
                if ((rc2 = grow_dev(h, &st.d_segs, &st.segs_cap, pbe.segs_cap, 64))) return rc2;
                if ((rc2 = grow_dev(h, &st.d_tilerefs, &st.tilerefs_cap, size_t(n_groups), 64))) return rc2;
                // (the plans and the bitmap of groups in ONE device buffer, as they lie in the page-locked one: one upload)
                if ((rc2 = grow_dev(h, &h->d_plans, &h->d_plans_cap, size_t(n_plans) + (((pbe.marks.size() * 3 + 3) / 2) * sizeof(uint64_t) + 128) / sizeof(MapPlan) + 2, 64))) return rc2;
                if (two_pass && (rc2 = grow_dev(h, &h->d_lane_scan, &h->d_lane_scan_cap, size_t(n_plans) * 64 * 3, 4096))) return rc2;
                if ((rc2 = grow_dev(h, &st.d_tiles, &st.tiles_cap, size_t(pbe.total_emit / kEmitTile) + 2, 64))) return rc2;
                const size_t n_walk_front = (size_t(n_plans) * 3 + 1 + size_t(n_groups) * 2 + 4 + 1) & ~size_t(1);     // (what follows is 64-bit words)
                const size_t n_walk = n_walk_front + 2 * (size_t(n_plans) / 64 + 1) + 2 * size_t(n_plans);          // (group_state | base_state)
                // (a regrown buffer is not zero — and may well sit at the address of the one just freed: the capacity tells, not the pointer.
                // Compared by pointer, a follower of the walk read a stale base out of the new buffer's tail and wrote its runs into the blue.)
                { const size_t was_cap = h->d_walk_cap; if ((rc2 = grow_dev(h, &h->d_walk, &h->d_walk_cap, n_walk, 64))) return rc2; if (h->d_walk_cap != was_cap) { h->walk_zeroed = false; if (getenv("BOSSX_POISON_GROWN")) HIPCHK(hipMemset(h->d_walk, 0xff, h->d_walk_cap * sizeof(uint32_t))); } }     // (tests: a regrown buffer full of ones — what is not zeroed again shows)
                // the groups travel as the bitmap of touched (tile, barcode) keys + the rank of every word: build_groups_kernel writes the list
                const size_t n_words = pbe.marks.size();
                const size_t marks_bytes = n_words * sizeof(uint64_t), rank_bytes = ((n_words + 1) * sizeof(uint32_t) + 7) & ~size_t(7);
                plan_bytes = size_t(n_plans) * sizeof(MapPlan); group_bytes = marks_bytes + rank_bytes;
                if ((rc2 = grow_pin(h, &h->h_plan_pin, &h->plan_pin_cap, plan_bytes + group_bytes + size_t(n_plans) * 4 + 64))) return rc2;
                const bool split_plan_upload = getenv("BOSSX_SPLIT_PLAN_UPLOAD") != nullptr;     // (round 5: two uploads of eight blocks)
                if (split_plan_upload) { if ((rc2 = grow_dev(h, &h->d_marks, &h->d_marks_cap, group_bytes + 64, 4096))) return rc2; }
                uint8_t *const d_marks = split_plan_upload ? h->d_marks : reinterpret_cast<uint8_t *>(h->d_plans) + plan_bytes;
                lap("the buffer checks");
                memcpy(h->h_plan_pin, pbe.plans.data(), plan_bytes);
                memcpy(h->h_plan_pin + plan_bytes, pbe.marks.data(), marks_bytes);
                memcpy(h->h_plan_pin + plan_bytes + marks_bytes, pbe.rank.data(), (n_words + 1) * sizeof(uint32_t));
                lap("the copies into the page-locked plan buffer");
                if (split_plan_upload) {
                    HIPCHK(upload_async(h->d_plans, h->h_plan_pin, plan_bytes, h->stream_stage));
                    HIPCHK(upload_async(d_marks, h->h_plan_pin + plan_bytes, group_bytes, h->stream_stage));
                } else {
                    HIPCHK(upload_async(h->d_plans, h->h_plan_pin, plan_bytes + group_bytes, h->stream_stage, 64));
                }
                lap("the plans' upload");
                if (n_words)
                    hipLaunchKernelGGL(build_groups_kernel, dim3(uint32_t((n_words + 255) / 256)), dim3(256), 0, h->stream_stage,
                                       reinterpret_cast<const unsigned long long *>(d_marks), reinterpret_cast<const uint32_t *>(d_marks + marks_bytes),
                                       uint32_t(n_words), uint32_t(h->nb), st.d_tilerefs);
                lap("the groups' upload");
                // (the walk's words are zeroed BEHIND the batch that used them — see the end of this function — so no fill stands between
                // the plans' upload and the walk; a fresh or regrown buffer, or a batch that left early, is zeroed here)
                if (!h->walk_zeroed) HIPCHK(hipMemsetAsync(h->d_walk, 0, n_walk * sizeof(uint32_t), h->stream_stage));
                h->walk_zeroed = false;
                lap("the hipMemsetAsync");
                W.plans = h->d_plans; W.n_plans = n_plans; W.paf = h->d_paf; W.blob = st.d_blob;
                W.ops = st.d_ops; W.groups = st.d_tilerefs; W.n_groups = n_groups; W.segs = st.d_segs;
                W.emit_tile_op = st.d_tiles; W.nb = h->nb;
                const dim3 grid((n_plans + 3) / 4), block(256);
                if (two_pass) {
                    W.n_runs = h->d_walk; W.walk_err = W.n_runs + n_plans; W.ops_off = W.walk_err + n_plans;
                    W.group_count = W.ops_off + n_plans + 1; W.group_cursor = W.group_count + n_groups;
                    W.totals = W.group_cursor + n_groups;
                    W.lane_scan = h->d_lane_scan; W.scan_state = nullptr; W.group_state = nullptr;
                    hipLaunchKernelGGL(cigar_walk_kernel<false>, grid, block, 0, h->stream_stage, W);
                    hipLaunchKernelGGL(walk_scan_kernel, dim3(1), dim3(1024), 0, h->stream_stage, W);
                    hipLaunchKernelGGL(cigar_walk_kernel<true>, grid, block, 0, h->stream_stage, W);
                } else {
                    // single pass (front_end.hip.inc: cigar_walk_fused_kernel): the look-back's state in the words of the two-pass
                    // walk's run counts and offsets; the segment counts per group from the plans alone — launched BEFORE the
                    // stream waits for the text
                    W.scan_state = reinterpret_cast<unsigned long long *>(h->d_walk);
                    W.group_state = reinterpret_cast<unsigned long long *>(h->d_walk + n_walk_front);
                    W.base_state = W.group_state + (size_t(n_plans) / 64 + 1);
                    W.n_runs = nullptr; W.ops_off = nullptr; W.lane_scan = nullptr;
                    W.walk_err = h->d_walk + 2 * size_t(n_plans);
                    W.group_count = h->d_walk + 3 * size_t(n_plans) + 1; W.group_cursor = W.group_count + n_groups;
                    W.totals = W.group_cursor + n_groups;
                    if (!st.d_err) { if ((rc2 = dev_alloc(h, &st.d_err, 1, false))) return rc2; }
                    W.expand_err = st.d_err;
                    back = reinterpret_cast<uint32_t *>(h->h_plan_pin + plan_bytes + group_bytes);
                    if (!getenv("BOSSX_WALK_STORE_KERNEL")) {         // (the round-5 way back: a launch behind the walk + an event)
                        if (++h->walk_token == 0) h->walk_token = 1;
                        __atomic_store_n(back + 3, 0u, __ATOMIC_RELEASE);
                        W.host_back = back; W.host_token = h->walk_token;
                        back_token = h->walk_token;
                    }
                    hipLaunchKernelGGL(plan_groups_kernel, dim3((n_plans + 255) / 256), dim3(256), 0, h->stream_stage, W);
                    hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(1024), 0, h->stream_stage, W);
                    HIPCHK(hipStreamWaitEvent(h->stream_stage, h->ev_txt, 0));
#ifdef BOSSX_WALK_PROBE
                    if ((rc2 = grow_dev(h, &h->d_walk_probe, &h->d_walk_probe_cap, size_t(n_plans) * 8, 64))) return rc2;
                    HIPCHK(hipMemsetAsync(h->d_walk_probe, 0, size_t(n_plans) * 64, h->stream_stage));
                    W.probe = h->d_walk_probe;
#endif
                    hipLaunchKernelGGL(cigar_walk_fused_kernel, grid, block, 0, h->stream_stage, W);
                }
                HIPCHK(hipGetLastError());
                back = reinterpret_cast<uint32_t *>(h->h_plan_pin + plan_bytes + group_bytes);
                if (!back_token) hipLaunchKernelGGL(store_host_kernel, dim3(1), dim3(64), 0, h->stream_stage, W.totals, back, 4u);
                HIPCHK(hipGetLastError());
                if (!h->ev_walk) HIPCHK(hipEventCreateWithFlags(&h->ev_walk, hipEventDisableTiming));
                HIPCHK(hipEventRecord(h->ev_walk, h->stream_stage));
                // what the expansion writes (sized from what the host knows: every emitted base, the upper bound of the segments)
                if ((rc2 = grow_dev(h, &st.d_codes, &st.codes_cap, size_t(pbe.total_emit) + kCodePad + size_t(kExpandTile) + 32, 4096))) return rc2;
                if ((rc2 = grow_dev(h, &st.d_pieces, &st.pieces_cap, pbe.segs_cap, 64))) return rc2;
                if (!st.d_err) { if ((rc2 = dev_alloc(h, &st.d_err, 1, false))) return rc2; }
                lap("the three launches + the copy back");
                return BOSSX_OK;
            };
            early_rc = go();
            t_launched = std::chrono::steady_clock::now();
        };
        std::string err;
        rc = parse_paf_batch(in, h->contigs, h->index, summary, pb, err);
        if (rc) return fail(h, rc, err);                  // (upload_guard drains the copies)
        if (early_rc) return early_rc;
        HIPCHK(up_err);
        const auto t1 = std::chrono::steady_clock::now();
        uint32_t totals[4] = {0, 0, 0, 0};
        std::vector<uint32_t> walk_err;
        if (n_plans) {
            if (pb.any_check_bases) {
                // some read holds a byte other than A/C/G/T: look at the bases of its aligned runs, now that the
                // reads are on their way (the plans go up again, with their flags)
                await_reads(); HIPCHK(up_err);
                HIPCHK(hipStreamSynchronize(h->stream_stage));              // the first copy of the plans has left the staging buffer
                if (st.blob_bits == 2u) {
                    // ... and the reads go up again, as nibbles: two bits have no room for the digits the reference counts, nor for "refused"
                    // (rare — a basecaller writes A C G T — so one thread and one upload: ~3 ms for a 4000-read batch)
                    for (int i = 0; i < bossx_engine::kUpStreams; ++i) HIPCHK(hipStreamSynchronize(h->stream_ups[i]));
                    for (int32_t i = 0; i < n_reads; ++i) pack_read(seq_ptrs[i], size_t(seq_len[i]), packed + (seq_off[i] >> 1));
                    HIPCHK(upload_async(st.d_blob, packed, blob_bytes / 2, h->stream_stage, 64));
                    st.blob_bits = 4u;
                    ++h->nibble_repacks;
                }
                memcpy(h->h_plan_pin, pb.plans.data(), plan_bytes);
                HIPCHK(upload_async(h->d_plans, h->h_plan_pin, plan_bytes, h->stream_stage));
                hipLaunchKernelGGL(check_bases_kernel, dim3((n_plans + 3) / 4), dim3(256), 0, h->stream_stage, W);
                HIPCHK(hipGetLastError());
                back_token = 0;             // (the verdict now includes the bases: it comes back behind this launch)
                hipLaunchKernelGGL(store_host_kernel, dim3(1), dim3(64), 0, h->stream_stage, W.totals, back, 4u);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(h->ev_walk, h->stream_stage));
            }
            await_reads(); HIPCHK(up_err);                 // (ordered before the kernels that read the bases)
            // ---- per-base codes and per-segment pieces, enqueued BEFORE the host looks at the walk's verdict (round 6): the kernel
            // takes the counts from the walk's totals in HBM and does nothing if the walk refused the batch, so the GPU goes from the
            // walk straight into the expansion (round 5: walk -> copy back -> host wakes up -> launch: ~35 us of an idle GPU)
            if (pb.total_emit) {
                ExpandParams X;
                X.ops = st.d_ops; X.n_ops = 0; X.total_emit = uint32_t(pb.total_emit); X.totals = W.totals;
                X.blob = st.d_blob; X.blob_bits = st.blob_bits; X.codes = st.d_codes; X.segs = st.d_segs; X.n_segs = 0; X.pieces = st.d_pieces;
                if (!W.expand_err) HIPCHK(hipMemsetAsync(st.d_err, 0, sizeof(int32_t), h->stream_stage));       // (single pass: group_scan_kernel has zeroed it)
                X.err_flag = st.d_err;
                X.code_blocks = (X.total_emit + uint32_t(kExpandTile) - 1u) / uint32_t(kExpandTile);
                X.tile_op = st.d_tiles;
                hipLaunchKernelGGL(expand_codes_kernel, dim3(X.code_blocks + uint32_t((pb.segs_cap + 255) / 256)), dim3(256), 0, h->stream_stage, X);
                HIPCHK(hipGetLastError());
                expand_enqueued = true;
            }
            // the walk's totals: its own stores into page-locked memory, the token last (the host spins on that word: no event, no wake-up;
            // a store that does not show within a millisecond — memory that is not coherent mid-launch — is waited for the round-5 way)
            bool got = false;
            if (back_token) {
                const auto spin0 = std::chrono::steady_clock::now();
                for (uint32_t sp = 0; !got; ++sp) {
                    got = __atomic_load_n(back + 3, __ATOMIC_ACQUIRE) == back_token;
                    if (got) break;
                    _mm_pause();
                    if ((sp & 255u) == 255u && std::chrono::steady_clock::now() - spin0 > std::chrono::milliseconds(1)) break;
                }
                if (!got) ++h->walk_spin_timeouts;
            }
            if (!got) HIPCHK(hipEventSynchronize(h->ev_walk));
            memcpy(totals, back, sizeof(totals));
            if (back_token) totals[3] = 0;
            if (totals[2]) {
                walk_err.resize(n_plans);
                HIPCHK(hipMemcpy(walk_err.data(), W.walk_err, size_t(n_plans) * sizeof(uint32_t), hipMemcpyDeviceToHost));
            }
        } else {
            await_reads(); HIPCHK(up_err);
            HIPCHK(hipStreamSynchronize(h->stream_stage));
        }
        // ---- failures: the first ValueError / KeyError class one in record order; the IndexError
        // class (raised later in the reference, inside _effect_increments) only if nothing else failed
        auto plan_name = [&](uint32_t i) {
            const int32_t r = pb.plan_read[i];
            return std::string(names + name_off[r], size_t(name_off[r + 1] - name_off[r]));
        };
        // (CIGAR text is tokenised like the reference's re.findall: what it skips is skipped; what remains are a
        // shape mismatch — ValueError, sequences.py:790 — and the span assertion, sequences.py:732)
        // The device walk DETECTS; which exception the reference raises for the mapping — its checks come in a
        // fixed order — is check_cigar_text's to say (paf_host.cpp), from the same text and plan.
        for (uint32_t i = 0; i < uint32_t(walk_err.size()); ++i)
            if (walk_err[i] & kWalkParseMask) {
                const MapPlan &mp = pb.plans[i];
                std::string msg;
                int code = check_cigar_text(in.paf + (size_t(mp.cg_off) - size_t(in.paf_base)), mp.cg_len, int64_t(mp.q_need), int64_t(mp.span), true, true, msg);
                if (!code) { code = BOSSX_E_INVALID; msg = "internal: the device walk refuses what the host check passes (" + std::string(walk_message(walk_err[i] & kWalkParseMask)) + ")"; }
                return fail(h, code, "read '" + plan_name(i) + "': " + msg);
            }
        if (pb.pre_code) return fail(h, pb.pre_code, pb.pre_msg);
        {
            int64_t best_gi = pb.pre_range_gi;
            std::string msg = pb.pre_range_msg;
            for (uint32_t i = 0; i < uint32_t(walk_err.size()); ++i)
                if (walk_err[i] && (best_gi < 0 || pb.plan_gi[i] < best_gi)) {
                    best_gi = pb.plan_gi[i];
                    msg = "read '" + plan_name(i) + "': " + walk_message(walk_err[i]);
                    break;
                }
            if (best_gi >= 0) return fail(h, BOSSX_E_RANGE, msg);
        }
#ifdef BOSSX_WALK_PROBE
        if (W.probe) {
            // 100-MHz stamps per mapping: 0 start | 1 plan + step back | 2 count loop | 3 wave scans + checks | 4 base known | 5 emit loop | 6 end
            HIPCHK(hipStreamSynchronize(h->stream_stage));
            std::vector<unsigned long long> pr(size_t(n_plans) * 8);
            HIPCHK(hipMemcpy(pr.data(), W.probe, pr.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t00 = ~0ull;
            for (uint32_t i = 0; i < n_plans; ++i) if (pr[size_t(i) * 8]) t00 = std::min(t00, pr[size_t(i) * 8]);
            double mx[7] = {0}, mean[7] = {0}; uint32_t arg[7] = {0}; double dmx[7] = {0}; uint32_t n_ok = 0;
            for (uint32_t i = 0; i < n_plans; ++i) {
                const unsigned long long *q = &pr[size_t(i) * 8];
                if (!q[6]) continue;
                ++n_ok;
                for (int k = 0; k < 7; ++k) {
                    const double at = double(q[k] - t00) * 0.01;
                    if (at > mx[k]) mx[k] = at;
                    if (k) { const double dd = double(q[k] - q[k - 1]) * 0.01; mean[k] += dd; if (dd > dmx[k]) { dmx[k] = dd; arg[k] = i; } }
                    else { mean[0] += at; if (at > dmx[0]) { dmx[0] = at; arg[0] = i; } }
                }
            }
            fprintf(stderr, "[walk probe] %u of %u mappings emitted; latest stamp per phase (us from the first wave's start): start %.1f | plan %.1f | count %.1f | scans %.1f | base %.1f | emit %.1f | end %.1f\n",
                    n_ok, n_plans, mx[0], mx[1], mx[2], mx[3], mx[4], mx[5], mx[6]);
            fprintf(stderr, "[walk probe]   phase mean / longest (us), CIGAR bytes of the longest: plan %.1f / %.1f (%u) | count %.1f / %.1f (%u) | scans %.1f / %.1f | base wait %.1f / %.1f (mapping %u) | emit %.1f / %.1f (%u) | pieces %.1f / %.1f (%u)\n",
                    mean[1] / n_ok, dmx[1], pb.plans[arg[1]].cg_len, mean[2] / n_ok, dmx[2], pb.plans[arg[2]].cg_len, mean[3] / n_ok, dmx[3],
                    mean[4] / n_ok, dmx[4], arg[4], mean[5] / n_ok, dmx[5], pb.plans[arg[5]].cg_len, mean[6] / n_ok, dmx[6], pb.plans[arg[6]].cg_len);
        }
#endif
        pb.n_ops = totals[0];
        dev_n_segs = totals[1];
        if (getenv("BOSSX_CHECK_DEVICE_WALK") && (rc = check_device_walk(h, st, in, pb, totals[1]))) return rc;
        if (timing) {
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[bossx] stage_batch: before the parse %.2f, lines -> plans %.2f ms, text wait + plan upload + walk launches %.2f, rest of the host parse (collecting the gather) %.2f, waiting for the device walk %.2f ms (%u mappings, %u runs, %u segments, %u groups)\n",
                    std::chrono::duration<double, std::milli>(t_pre - t0).count(), std::chrono::duration<double, std::milli>(t_plans - t_pre).count(),
                    std::chrono::duration<double, std::milli>(t_launched - t_plans).count(), std::chrono::duration<double, std::milli>(t1 - t_launched).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count(), n_plans, totals[0], totals[1], n_groups);
        }
        pb.plans.clear();
        pb.plan_read.clear(); pb.plan_gi.clear();
    }
    // ---- per-base codes and per-segment pieces: what the sweep's gather reads (asynchronous; the batch
    // has passed every check, and the read blob's upload is ordered before this point) --------------
    st.n_segs = host_walk ? uint32_t(pb.segs.size()) : dev_n_segs;
    if (pb.n_ops && !expand_enqueued) {
        if ((rc = grow_dev(h, &st.d_codes, &st.codes_cap, size_t(pb.total_emit) + kCodePad + size_t(kExpandTile) + 32, 4096))) return rc;
        if ((rc = grow_dev(h, &st.d_pieces, &st.pieces_cap, size_t(st.n_segs), 64))) return rc;
        ExpandParams X;
        X.ops = st.d_ops; X.n_ops = uint32_t(pb.n_ops); X.total_emit = uint32_t(pb.total_emit); X.totals = nullptr;
        X.blob = st.d_blob; X.blob_bits = st.blob_bits; X.codes = st.d_codes; X.segs = st.d_segs; X.n_segs = st.n_segs; X.pieces = st.d_pieces;
        // (the slot's own error word: a batch staged ahead must not raise in the update of the batch before it)
        if (!st.d_err) { if ((rc = dev_alloc(h, &st.d_err, 1, false))) return rc; }
        HIPCHK(hipMemsetAsync(st.d_err, 0, sizeof(int32_t), h->stream_stage));
        X.err_flag = st.d_err;
        X.code_blocks = (X.total_emit + uint32_t(kExpandTile) - 1u) / uint32_t(kExpandTile);
        if (host_walk) {        // (the device walk's second pass wrote the tile -> run table itself)
            if ((rc = grow_dev(h, &st.d_tiles, &st.tiles_cap, size_t(X.code_blocks) + 2, 64))) return rc;
            hipLaunchKernelGGL(emit_tiles_kernel, dim3(X.code_blocks / 256 + 1), dim3(256), 0, h->stream_stage, st.d_ops, X.n_ops, X.code_blocks, st.d_tiles);
        }
        X.tile_op = st.d_tiles;
        hipLaunchKernelGGL(expand_codes_kernel, dim3(X.code_blocks + (X.n_segs + 255u) / 256u), dim3(256), 0, h->stream_stage, X);
        HIPCHK(hipGetLastError());
    }
    if (n_rec) *n_rec = pb.n_rec;
    if (aligned_bases) *aligned_bases = int64_t(pb.total_emit);
    if (!st.ev_ready) HIPCHK(hipEventCreateWithFlags(&st.ev_ready, hipEventDisableTiming));
    HIPCHK(hipEventRecord(st.ev_ready, h->stream_stage));
    if (!host_walk && h->d_walk && !h->walk_zeroed && !getenv("BOSSX_WALK_ZERO_IN_FRONT")) {       // (behind ev_ready: nobody waits for it)
        HIPCHK(hipMemsetAsync(h->d_walk, 0, h->d_walk_cap * sizeof(uint32_t), h->stream_stage));
        h->walk_zeroed = true;
    }
    if (timing) fprintf(stderr, "[bossx] stage_batch: %.2f ms inside the staging core\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    st.pb = std::move(pb);
    st.valid = true;
    upload_guard.ok = true;
    return BOSSX_OK;
}

}  // namespace

int bossx_stage_batch(bossx_engine *h, const char *paf, size_t paf_len, const char *names,
                      const int64_t *name_off, const char *seqs, const int64_t *seq_off,
                      const int32_t *barcodes, int32_t n_reads, int32_t min_len,
                      bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "stage_batch before finalize");
    if (n_reads < 0 || (n_reads > 0 && (!names || !name_off || !seqs || !seq_off))) return fail(h, BOSSX_E_INVALID, "bad batch arrays");
    HIPCHK(hipSetDevice(h->cfg.device));
    // (the blob is the caller's: it is only read — packed, like the strings of bossx_stage_batch_ptrs, into the engine's page-locked buffer)
    std::vector<int64_t> poff(size_t(n_reads) + 1, 0), plen(size_t(n_reads), 0);
    std::vector<const char *> ptrs(static_cast<size_t>(n_reads), nullptr);
    for (int32_t i = 0; i < n_reads; ++i) {
        plen[size_t(i)] = seq_off[i + 1] - seq_off[i];
        if (plen[size_t(i)] < 0) return fail(h, BOSSX_E_INVALID, "seq_off must not decrease");
        ptrs[size_t(i)] = seqs + seq_off[i];
        poff[size_t(i) + 1] = poff[size_t(i)] + ((plen[size_t(i)] + 3) & ~int64_t(3));
    }
    return stage_core(h, paf, paf_len, names, name_off, poff.data(), plen.data(), ptrs.data(), barcodes, n_reads, min_len,
                      summary, n_rec, aligned_bases);
}

int bossx_stage_batch_ptrs(bossx_engine *h, const char *paf, size_t paf_len, const char *const *name_ptrs,
                           const int64_t *name_lens, const char *const *seq_ptrs, const int64_t *seq_lens,
                           const int32_t *barcodes, int32_t n_reads, int32_t min_len,
                           bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "stage_batch before finalize");
    if (n_reads < 0 || (n_reads > 0 && (!name_ptrs || !name_lens || !seq_ptrs || !seq_lens))) return fail(h, BOSSX_E_INVALID, "bad batch arrays");
    const auto t_entry = std::chrono::steady_clock::now();
    struct EntryTimer { std::chrono::steady_clock::time_point t; ~EntryTimer() { if (getenv("BOSSX_STAGE_TIMING")) fprintf(stderr, "[bossx] stage_batch: %.2f ms inside bossx_stage_batch_ptrs\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); } } entry_timer{t_entry};
    HIPCHK(hipSetDevice(h->cfg.device));
    std::vector<int64_t> name_off(size_t(n_reads) + 1, 0), seq_off(size_t(n_reads) + 1, 0);
    for (int32_t i = 0; i < n_reads; ++i) {
        name_off[size_t(i) + 1] = name_off[size_t(i)] + name_lens[i];
        seq_off[size_t(i) + 1] = seq_off[size_t(i)] + ((seq_lens[i] + 3) & ~int64_t(3));      // (every read on a base index that is a multiple of four: a byte boundary at two bits per base, and at four)
    }
    std::string names(size_t(name_off[size_t(n_reads)]), '\0');
    for (int32_t i = 0; i < n_reads; ++i) memcpy(&names[size_t(name_off[size_t(i)])], name_ptrs[i], size_t(name_lens[i]));
    // the sequences are packed into pinned memory inside stage_core's parallel region: one pass, and half the bytes cross PCIe
    return stage_core(h, paf, paf_len, names.data(), name_off.data(), seq_off.data(), seq_lens, seq_ptrs,
                      barcodes, n_reads, min_len, summary, n_rec, aligned_bases);
}

int bossx_pack_reads(const char *bases, int64_t n, uint8_t *dst, int32_t *dirty) {
    if (n < 0 || (n > 0 && (!bases || !dst))) return BOSSX_E_INVALID;
    // BOSSX_PACK_SCALAR: the table alone (what the vector path is held to)
    const bool d = getenv("BOSSX_PACK_SCALAR") ? pack_read_scalar(bases, size_t(n), dst) : pack_read(bases, size_t(n), dst);
    if (dirty) *dirty = d ? 1 : 0;
    return BOSSX_OK;
}

int bossx_pack_reads2(const char *bases, int64_t n, uint8_t *dst, int32_t *dirty) {
    if (n < 0 || (n > 0 && (!bases || !dst))) return BOSSX_E_INVALID;
    const bool d = getenv("BOSSX_PACK_SCALAR") ? pack_read2_scalar(bases, size_t(n), dst) : pack_read2(bases, size_t(n), dst);
    if (dirty) *dirty = d ? 1 : 0;
    return BOSSX_OK;
}

int bossx_paf_summary(bossx_engine *h, const char *paf, size_t paf_len, const char *const *name_ptrs,
                      const int64_t *name_lens, int32_t n_reads, int32_t min_len,
                      bossx_batch_summary *summary, int32_t *n_rec) {
    if (!h || !h->finalized || !summary) return fail(h, BOSSX_E_INVALID, "bad paf_summary call");
    if (n_reads < 0 || (n_reads > 0 && (!name_ptrs || !name_lens))) return fail(h, BOSSX_E_INVALID, "bad batch arrays");
    std::vector<int64_t> name_off(size_t(n_reads) + 1, 0), seq_off(size_t(n_reads) + 1, 0);
    for (int32_t i = 0; i < n_reads; ++i) name_off[size_t(i) + 1] = name_off[size_t(i)] + name_lens[i];
    std::string names(size_t(name_off[size_t(n_reads)]), '\0');
    for (int32_t i = 0; i < n_reads; ++i) memcpy(&names[size_t(name_off[size_t(i)])], name_ptrs[i], size_t(name_lens[i]));
    ParseInput in{paf ? paf : "", paf ? paf_len : 0, names.data(), name_off.data(), seq_off.data(), nullptr, n_reads, min_len, h->nb};
    in.summary_only = true;
    ParsedBatch pb;
    std::string err;
    int rc = parse_paf_batch(in, h->contigs, h->index, summary, pb, err);
    if (rc) return fail(h, rc, err);
    if (n_rec) *n_rec = pb.n_rec;
    return BOSSX_OK;
}

int bossx_stage_stream(bossx_engine *h, int32_t on_main) {
    if (!h) return BOSSX_E_INVALID;
    h->stage_on_main = on_main != 0;
    return BOSSX_OK;
}

int bossx_select_batch(bossx_engine *h, int32_t slot) {
    if (!h || slot < 0 || slot >= 256) return fail(h, BOSSX_E_INVALID, "batch slot must be in [0, 256)");
    if (size_t(slot) >= h->slots.size()) h->slots.resize(size_t(slot) + 1);
    h->slot = slot;
    return BOSSX_OK;
}

int bossx_ingest_staged(bossx_engine *h) {
    if (!h || !h->slots[size_t(h->slot)].valid) return fail(h, BOSSX_E_INVALID, "no staged batch in the selected slot");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc;
    if (h->pending_slot >= 0 && (rc = flush_pending(h))) return rc;   // two batches before one sweep
    const bossx_engine::Staged &st = h->slots[size_t(h->slot)];
    const ParsedBatch &pb = st.pb;
    for (size_t i = 0; i < h->contigs.size(); ++i) h->contigs[i].cov_total += pb.emitted_per_contig[i];
    if (pb.total_emit == 0) return BOSSX_OK;
    if (st.ev_ready) HIPCHK(hipStreamWaitEvent(h->stream, st.ev_ready, 0));     // staged on stream_stage, consumed on the main stream
    h->pending_err_unmerged = st.d_err != nullptr;     // (joins the engine's error word in the sweep's prep launch — or in front of the fallback scatter)
    h->slots[size_t(h->slot)].busy = true;        // (until the sweep that applies it — or a fallback scatter — is behind ev_free)
    h->slots[size_t(h->slot)].ev_free_recorded = false;
    // the increments are applied by the next sweep, tile by tile (site_sweep_kernel prologue);
    // its prep launch marks the touched tiles
    h->pending_slot = h->slot;
    h->pending_emit = double(pb.total_emit);
    h->pending_ops = double(pb.n_ops);
    return BOSSX_OK;
}

int bossx_ingest_paf(bossx_engine *h, const char *paf, size_t paf_len, const char *names,
                     const int64_t *name_off, const char *seqs, const int64_t *seq_off,
                     const int32_t *barcodes, int32_t n_reads, int32_t min_len,
                     bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases) {
    int rc = bossx_stage_batch(h, paf, paf_len, names, name_off, seqs, seq_off, barcodes, n_reads, min_len,
                               summary, n_rec, aligned_bases);
    if (rc) return rc;
    return bossx_ingest_staged(h);
}

namespace {
// the entropy-tracking variant is a separate instantiation: the default (no entropy array) carries no code for it
// One launch over `n_items` work items (tiles or groups): persistent blocks that take items from a counter, except
// when the launch publishes tiles to a chain running next to it (those keep one block per item, in walk order).
#define LAUNCH_SWEEP(ING, grid, block, lds, stream, P)                                                    \
    do {                                                                                                  \
        const uint32_t n_items_ = (grid).x;                                                               \
        const bool ent_ = h->d_entropy != nullptr;                                                        \
        (P).n_work = n_items_;                                                                            \
        (P).work_ctr = ((P).publish || getenv("BOSSX_SWEEP_ONE_PER_BLOCK") || h->work_ctr_used >= h->n_work_ctr) ? nullptr : h->d_work_ctr + h->work_ctr_used++; \
        const uint32_t res_ = h->sweep_grid[ING ? 1 : 0][ent_ ? 1 : 0];                                   \
        (P).chunk = h->sweep_chunk ? h->sweep_chunk : std::min<uint32_t>(8u, std::max<uint32_t>(1u, n_items_ / (2u * res_))); \
        /* (the one-barcode kernel splits the items evenly over its resident blocks; the several-barcode one hands runs out) */ \
        const uint32_t blocks_ = (P).work_ctr ? (h->nb > 1 ? std::min((n_items_ + (P).chunk - 1u) / (P).chunk, res_) : std::min(n_items_, res_)) : n_items_; \
        if (h->nb > 1) {                                                                                  \
            /* (derived entropy: nothing entropy-specific is compiled into the scoring any more) */     \
            hipLaunchKernelGGL((site_sweep_kernel<ING, false, true>), dim3(blocks_), block, lds, stream, P); \
        } else {                                                                                          \
            /* (one instantiation: the entropy stores are guarded by the array's presence at run time — the variant compiled */ \
            /* without them came out of the register allocator 25 registers fatter and spilling) */     \
            hipLaunchKernelGGL((site_sweep1_kernel<ING, true>), dim3(blocks_), block, lds, stream, P);    \
        }                                                                                                 \
    } while (0)
int launch_sweep(bossx_engine *h) {
    // dropout thresholds of this update: mean depth per contig (reference.py:157-158, 174-176)
    std::vector<int32_t> &thr = h->drop_thr_host;    // member: must outlive the async copy
    thr.resize(h->filt.size());
    for (size_t k = 0; k < h->filt.size(); ++k) {
        const ContigInfo &c = h->contigs[size_t(h->filt[k])];
        const double mean = double(c.cov_total) / double(c.length * int64_t(h->nb));
        thr[k] = (!c.remote && mean > 5) ? int32_t(mean / 8) : -1;
    }
    const size_t n_groups = h->pending_slot >= 0 ? h->slots[size_t(h->pending_slot)].pb.n_groups : 0;
    const size_t n_touched = h->pending_slot >= 0 ? h->slots[size_t(h->pending_slot)].pb.n_touched_tiles : 0;
    // With several barcodes a touched tile costs nb scoring passes at the ingest variant's low
    // occupancy (LDS staging, 4 blocks/CU): split the work instead — an ingest-only launch applies
    // the increments and leaves `touched` flags, then the plain variant scores
    // every tile.  (With one barcode the fused form is faster: 0.17 vs 0.36 ms for E. coli.)
    const char *split_env = getenv("BOSSX_SPLIT_INGEST");
    const bool split = n_groups > 0 && (split_env ? atoi(split_env) != 0 : h->nb >= 3);
    // INCREMENTAL sweep.  A tile that receives no base keeps every value the sweep would recompute:
    // scores are a function of (coverage, site state, dropout threshold of the contig), the bin sums
    // and the bucket totals follow.  So when the batch touches a small part of the reference and no
    // contig's dropout threshold moved (it is int(mean depth / 8), reference.py:174-176), only the
    // touched tiles are swept; the others keep their bin sums, and the bucket sums are kept current
    // by per-tile differences.  The reference recomputes everything every update (sequences.py:419,
    // reference.py:157-161,196-199); the results are identical.  BOSSX_INCREMENTAL=0 / 1 forces it off /
    // on whenever legal; by default it is used when fewer than half of the tiles are touched (above
    // that the one dense launch over all tiles is used: chr20+21, 29 % touched, 0.59 -> 0.28 ms).
    if (h->last_thr.size() != thr.size()) h->last_thr.assign(thr.size(), INT32_MIN);
    bool thr_changed = false;
    for (size_t k = 0; k < thr.size(); ++k) thr_changed = thr_changed || thr[k] != h->last_thr[k];
    const char *inc_env = getenv("BOSSX_INCREMENTAL");
    const bool want_inc = inc_env ? atoi(inc_env) != 0 : n_touched * 2 < size_t(h->n_tiles);
    // The reference looks every site that dropout zeroed up again at the NEXT update (`scores == 0.0`,
    // sequences.py:433-441) and writes its entropy then.  New zeros only appear in a sweep whose
    // threshold moved (or after an import / preload): coverage only grows, so under an unchanged
    // threshold no site falls to or below it — and thresholds are per contig.  The update after such a sweep
    // therefore sweeps once more, when the entropy array is kept, every tile of the contigs whose threshold
    // had moved (everything after an import / preload) — an untouched tile would otherwise hold the entropy
    // (and the SCORED bit) of its freshly zeroed sites back until it next receives a base.  (Round 3 swept
    // the whole reference again for ONE contig's threshold: at GRCh38, 27 contigs crossing int(mean / 8) at
    // different updates, a 10-ms sweep of 3.1 Gb behind every one of them.)
    const bool dz_on = h->d_entropy && !getenv("BOSSX_NO_DZ_RESWEEP");
    const bool dz_resweep = h->dz_fresh && dz_on;
    if (h->dz_fresh_k.size() != thr.size()) h->dz_fresh_k.assign(thr.size(), 0);
    // A contig whose own threshold moved is swept whole; the others keep their untouched tiles (27 contigs
    // cross int(mean / 8) at 27 different updates: a 150-kb scaffold must not cost a sweep of 3.1 Gb).
    const bool full = h->full_sweep_needed || h->touched_dirty || split || !want_inc || dz_resweep;
    std::vector<size_t> resweep;                 // local contigs swept whole although the update is incremental
    if (!full)
        for (size_t k = 0; k < thr.size(); ++k)
            if ((thr[k] != h->last_thr[k] || (dz_on && h->dz_fresh_k[k])) && !h->contigs[size_t(h->filt[k])].remote) resweep.push_back(k);
    bool any_thr = false;
    for (int32_t t : thr) any_thr = any_thr || t >= 0;
    h->dz_fresh = any_thr && (h->full_sweep_needed || h->touched_dirty);
    for (size_t k = 0; k < thr.size(); ++k) h->dz_fresh_k[k] = (thr[k] >= 0 && thr[k] != h->last_thr[k]) ? 1 : 0;
    // The chain of this update may run NEXT TO the sweep (second stream, tiles handed over as they
    // are published).  Measured on MI355X it no longer pays by default: the concurrent chain variant
    // is ~8 % slower than the serial one (agent-scope loads, flag polling), a publishing sweep is
    // slower than a plain one (ordered tile hand-out, write-through bin stores), chain blocks take
    // CUs (LDS) from the sweep, and the incremental sweep has shrunk what there is to hide — E. coli
    // 0.42 vs 0.47 ms of kernels, chr20+21 3.26 vs 3.32, but 10 x 5 Mb x 8 barcodes 2.3 vs 2.0 the
    // other way.  So: off unless BOSSX_OVERLAP=1 (always) or BOSSX_OVERLAP=auto (where the sweep
    // is estimated at >= 0.3 of the chain: 2.5 TB/s for the swept tiles, 4.4 ns per bin of the longest
    // contig).  BOSSX_NO_OVERLAP=1 rules it out.
    const char *ov_env = getenv("BOSSX_OVERLAP");
    bool publish = h->overlap_ok && h->host_armed && ov_env != nullptr;
    if (publish && !strcmp(ov_env, "auto")) {
        int64_t longest = 0;
        for (int32_t fi : h->filt) if (!h->contigs[size_t(fi)].remote) longest = std::max(longest, h->contigs[size_t(fi)].T + 1);
        const double swept_tiles = full ? double(h->n_tiles) : double(n_touched);
        const double tile_bytes = double(kTileSites) * h->nb * 11.0;
        // (a tile that receives bases costs ~2.5 tiles, an ingested base ~3 ps on top)
        const double sweep_ms = (swept_tiles + 1.5 * double(n_touched)) * tile_bytes / 2.5e9 + (h->pending_slot >= 0 ? h->pending_emit : 0.0) * 3e-9;
        const double chain_ms = double(longest) * 4.4e-6;
        publish = sweep_ms >= 0.3 * chain_ms;
    }
    if (!resweep.empty()) publish = false;       // (a chain next to the sweep is told about whole-reference or touched-tile sweeps only)
    {
        // one small launch installs the thresholds and marks the tiles.  The bin sums need no
        // clearing: every bin of a local contig is rewritten by every sweep of its tile, the others stay zero.
        PrepParams PR;
        PR.drop_thr = h->d_drop_thr; PR.n_thr = 0;
        PR.max_bits = &h->d_ctrl->max_bits;
        PR.tiles = nullptr; PR.n_tiles = 0; PR.tile_ref = h->d_tile_ref;
        PR.tile_done = h->d_tile_done; PR.n_all = h->n_tiles; PR.full = full ? 1 : 0;
        PR.mark = publish ? 1 : 0;
        PR.work_ctr = h->d_work_ctr; PR.n_ctr = h->n_work_ctr + 4;      // (+ the write-back counts behind them)
        PR.slot_err = nullptr; PR.err = h->d_err;
        if (h->pending_slot >= 0 && h->pending_err_unmerged) { PR.slot_err = h->slots[size_t(h->pending_slot)].d_err; h->pending_err_unmerged = false; }
        h->work_ctr_used = 0;
        if (h->pending_slot >= 0) {
            const bossx_engine::Staged &st = h->slots[size_t(h->pending_slot)];
            PR.tiles = st.d_tilerefs; PR.n_tiles = uint32_t(st.pb.n_groups);
        }
        if (thr.size() <= 32) {
            PR.n_thr = int32_t(thr.size());
            for (size_t k = 0; k < thr.size(); ++k) PR.thr[k] = thr[k];
        } else {
            HIPCHK(hipMemcpyAsync(h->d_drop_thr, thr.data(), thr.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        }
        const int64_t blocks = std::min<int64_t>((std::max<int64_t>(full ? h->n_tiles : 0, int64_t(PR.n_tiles)) + 255) / 256, 1024);
        h->max_bits_clear = true;
        hipLaunchKernelGGL(sweep_prep_kernel, dim3(uint32_t(std::max<int64_t>(blocks, 1))), dim3(256), 0, h->stream, PR);
        // everything up to here precedes a chain that update_benefit may put on stream2 next to
        // the sweep (it needs max_bits cleared, and the previous update's readers finished)
        HIPCHK(hipEventRecord(h->ev_begin, h->stream));
    }
    ++h->epoch;                                  // stamps the tile flags of this sweep
    if (h->epoch == 0xffffffffu) h->epoch = 1;   // never the 'pending' value
    h->sweep_published = publish;                // tiles are published only if a chain will run next to this sweep
    SweepParams P = sweep_params(h);
    // how much of the reference this update's sweep rewrites (tiles of the batch, whole contigs that are swept again): where it is a small
    // part, the chain's candidates kernel looks at the tiles' stamps first (its quick way out); where most chunks change it would only
    // pay a round trip in front of its loads
    {
        double tiles = full ? double(h->n_tiles) : double(h->pending_slot >= 0 ? h->slots[size_t(h->pending_slot)].pb.n_groups : 0);
        if (!full) for (size_t k : resweep) tiles += double(h->contigs[size_t(h->filt[k])].n_tiles);
        h->sweep_tile_share = h->n_tiles > 0 ? tiles / double(h->n_tiles) : 1.0;
        h->tile_share_fresh = true;
    }
    time_begin(h, BOSSX_K_SWEEP);
    double resweep_sites = 0, resweep_bins = 0;
    if (!full) {
        // the contigs whose threshold moved, whole (the plain variant leaves the tiles that receive bases to the launch below) ...
        for (size_t k : resweep) {
            const ContigInfo &c = h->contigs[size_t(h->filt[k])];
            P.tile_base = c.tile_off;
            if (c.n_tiles) LAUNCH_SWEEP(false, dim3(uint32_t(c.n_tiles)), dim3(256), 0, h->stream, P);
            resweep_sites += double(c.length); resweep_bins += double(c.T + 1);
        }
        P.tile_base = 0;
        // ... and the tiles that receive bases: one block per (tile, barcode) group, the first group of a tile does the tile
        if (n_groups) LAUNCH_SWEEP(true, dim3(uint32_t(n_groups)), dim3(256), 0, h->stream, P);
    } else if (split) {
        P.ingest_only = 1;
        LAUNCH_SWEEP(true, dim3(uint32_t(n_groups)), dim3(256), 0, h->stream, P);
        P.ingest_only = 0; P.tiles = nullptr; P.n_groups = 0; P.use_touched = 1;
        LAUNCH_SWEEP(false, dim3(uint32_t(h->n_tiles)), dim3(256), 0, h->stream, P);
    } else if (n_touched * 2 >= size_t(h->n_tiles) && n_touched > 0) {
        // most tiles receive bases: one launch over all tiles, each block looks its tile up
        P.dense = 1;
        LAUNCH_SWEEP(true, dim3(uint32_t(h->n_tiles)), dim3(256), 0, h->stream, P);
    } else if (P.publish && n_groups) {
        // a chain is waiting for tiles in walk order: the touched ones are scattered along it, so
        // they go first (DONE marks keep the plain launch off them), then everything else in order
        P.ingest_first = 1;
        LAUNCH_SWEEP(true, dim3(uint32_t(n_groups)), dim3(256), 0, h->stream, P);
        LAUNCH_SWEEP(false, dim3(uint32_t(h->n_tiles)), dim3(256), 0, h->stream, P);
    } else {
        if (h->n_tiles > 0)
            LAUNCH_SWEEP(false, dim3(uint32_t(h->n_tiles)), dim3(256), 0, h->stream, P);
        if (n_groups)    // one block per (tile, barcode) group; the first group of a tile does the tile
            LAUNCH_SWEEP(true, dim3(uint32_t(n_groups)), dim3(256), 0, h->stream, P);
    }
    h->last_thr = thr;
    h->full_sweep_needed = false;
    // algorithmic bytes: per site*barcode 10 B counters + 1 B state read; per 100-site bin 8 B
    // downsampled score write; per ingested base 1 B code read + 2 B counter write-back
    // (entropy / state write-backs of changed sites are data dependent and not
    // counted; the counter READ of an ingested base is already in the 10 B/site)
    double sites = 0;
    for (int32_t fi : h->filt) if (!h->contigs[size_t(fi)].remote) sites += double(h->contigs[size_t(fi)].length);
    double bins = double(h->B);
    if (!full) {       // incremental: the swept tiles only (tiles that are both in a re-swept contig and touched count once too many: an upper bound)
        sites = double(n_touched) * kTileSites + resweep_sites; bins = double(n_touched) * kTileBins + resweep_bins;
    }
    double bytes = sites * h->nb * 11.0 + bins * h->nb * 8.0;
    if (h->touched_dirty || split) bytes += sites;
    if (h->pending_slot >= 0) bytes += 1.0 * h->pending_emit;      // one code byte per ingested base (the emit runs are read by expand_codes_kernel, not by the sweep)
    // the counter write-back is counted at the granularity it happens at — 16-byte vectors of changed counters, counted by the
    // kernel itself while timing is on — and added when the count is read (bossx_kernel_bytes)
    h->sweep_bytes_base = bytes;
    time_end(h, BOSSX_K_SWEEP, bytes);
    HIPCHK(hipGetLastError());
    if (P.probe) {
        unsigned long long pr[24];
        HIPCHK(hipMemcpy(pr, P.probe, sizeof(pr), hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(P.probe, 0, sizeof(pr)));
        // (site_sweep1_kernel: seven words — tiles | top..ingested | ..gate passed | ..scored | barrier | phase B | behind it;
        //  the barcoded template: six — items | top..gathered | ..scored | barrier | phase B | tail)
        for (int o = 0; o <= 8; o += 8)
            if (pr[o]) {
                double tot = 0;
                fprintf(stderr, "[sweep probe] %s: %llu tiles (wave 0 of each block), cycles per tile:", o ? "ingest" : "plain", pr[o]);
                for (int i = 1; i < 7; ++i) { fprintf(stderr, " %.0f", double(pr[o + i]) / double(pr[o])); tot += double(pr[o + i]) / double(pr[o]); }
                fprintf(stderr, " | total %.0f\n", tot);
            }
    }
    if (h->pending_slot >= 0) {
        bossx_engine::Staged &st = h->slots[size_t(h->pending_slot)];
        if (!st.ev_free) HIPCHK(hipEventCreateWithFlags(&st.ev_free, hipEventDisableTiming));
        HIPCHK(hipEventRecord(st.ev_free, h->stream));  // the sweep launches above are the last readers of this slot's buffers
        st.busy = true; st.ev_free_recorded = true;
    }
    h->pending_slot = -1;
    h->touched_dirty = false;
    return BOSSX_OK;
}
}  // namespace

namespace { int settle_chain(bossx_engine *h); }

int bossx_sweep(bossx_engine *h) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "sweep before finalize");
    if (!h->lut_set) return fail(h, BOSSX_E_INVALID, "sweep before set_lut");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    return launch_sweep(h);
}

int bossx_get_bucket_sums(bossx_engine *h, int32_t contig, uint64_t *dst) {
    int rc = check_contig(h, contig, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    const ContigInfo &c = h->contigs[size_t(contig)];
    const int64_t nfull = c.length / kBucket;
    for (int32_t b = 0; b < h->nb; ++b)
        HIPCHK(hipMemcpyAsync(dst + int64_t(b) * nfull, h->d_bucket_sums + int64_t(b) * h->NBK + c.bucket_off,
                              size_t(nfull) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    int32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, h->d_err, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (flag) {
        HIPCHK(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
        return fail(h, BOSSX_E_RANGE, "a read contains a base other than A/C/G/T inside an aligned segment");
    }
    return BOSSX_OK;
}

int bossx_set_bucket_switches(bossx_engine *h, int32_t contig, const uint8_t *sw) {
    int rc = check_contig(h, contig, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    const ContigInfo &c = h->contigs[size_t(contig)];
    std::vector<uint8_t> plane(size_t(c.n_buckets));
    for (int32_t b = 0; b < h->nb; ++b) {
        for (int64_t i = 0; i < c.n_buckets; ++i) plane[size_t(i)] = sw[i * h->nb + b] ? 1 : 0;
        HIPCHK(hipMemcpy(h->d_bucket_on + int64_t(b) * h->NBK + c.bucket_off, plane.data(), plane.size(), hipMemcpyHostToDevice));
    }
    return BOSSX_OK;
}

namespace {

int launch_sweep(bossx_engine *h);

// What the chunk-parallel chain left in the error block of the result (words 1 and 2), read by whoever fetched it.
void note_spec_result(bossx_engine *h, const int32_t *herr) {
    if (!h->last_chain_spec) return;
    // chunks the stitch had to add the plain way (sums that climb more than four binades inside a chunk: capped
    // next to uncapped regions).  Each costs ~4 us on ONE wave; beyond ~2 % of a chain's chunks the serial kernel wins.
    // chunks the stitch had to add the plain way (a start in another binade than the approximation, sums that climb more than
    // four pieces can follow: capped next to uncapped regions).  Each costs ~2.2 us on ONE wave (matrix core, four bins per
    // dependent operation); what counts is the longest chain's share of them.  The serial kernel takes over only where the
    // estimate says it would be faster (round 3 paused at 2 % of the chunks: a fifth of a long run's launches).
    const int64_t plain = herr[1];
    h->spec_plain_total += plain;
    const double plain_us = 3.3 * double(herr[3]);          // the most evaluations from the exact value on ONE chain: that wave is what the launch waits for
    (void)h->spec_plain_share;
    if (h->spec_est_us + plain_us > 0.9 * h->spec_est_serial_us && !getenv("BOSSX_SPEC_NO_PAUSE")) h->spec_pause = 8;      // (the switch: tests)
    // launches whose serial fallback ran (cumulative on the device).  A failed check costs one serial chain on top of the
    // chunk-parallel one: where three of the last sixty-four launches needed it the serial kernel is the faster form
    // (BOSSX_SPEC_KEEP=1: tests that want every launch to try)
    if (herr[2] != h->spec_mismatches) { h->spec_recent_fail += herr[2] - h->spec_mismatches; h->spec_mismatches = herr[2]; }
    if ((++h->spec_since_decay & 63) == 0 && h->spec_recent_fail > 0) --h->spec_recent_fail;
    if (h->spec_recent_fail >= 3 && !getenv("BOSSX_SPEC_KEEP")) h->chain_spec = false;
}

int fill_chain_params(bossx_engine *h, const int32_t *windows, const double *mult, ChainParams &P, size_t &lds) {
    int32_t wmax = 0;
    for (int k = 0; k < BOSSX_NWIN; ++k) { P.w[k] = windows[k]; wmax = std::max(wmax, windows[k]); }
    for (int i = 0; i < 10; ++i) P.m[i] = mult[i];
    for (int32_t fi : h->filt) {
        const ContigInfo &c = h->contigs[size_t(fi)];
        for (int k = 0; k < BOSSX_NWIN; ++k)
            if (windows[k] < 1 || windows[k] > c.T + 1)       // Bottleneck: window must be in [1, n]
                return fail(h, BOSSX_E_WINDOW, "Moving window (=" + std::to_string(windows[k]) + ") must between 1 and " +
                                                  std::to_string(c.T + 1) + ", inclusive");
    }
    // pipeline step: 256 bins when every chain block gets a CU of its own and the LDS allows, else 128
    auto lds_need = [&](int ch, int32_t &ring_out) {
        int32_t ring = 256;
        while (ring < wmax + 2 * ch) ring <<= 1;
        ring_out = ring;
        return size_t(ring) * 8 + 2 * 2 * size_t(kChainRows) * size_t(ch + 2) * 8 + 1024;     // ring + difference / sum tiles
    };
    int32_t ring = 0;
    const size_t n_blocks = h->filt.size() * size_t(h->nb) * 2;
    h->chain_ch = 256;
    // (the barrier kernel ran two 128-bin blocks per CU when more blocks than CUs were launched; the
    // barrier-free one hands chunks over through LDS counters, whose latency 128-bin chunks do not amortise)
    // static LDS of the barrier-free kernel: NBD difference buffers, NBD-1 carry buffers, flags
    const bool gc = h->chain_gc;
    auto flow_static = [gc](int nbd) {      // (with the carries in global memory the carry buffers leave the LDS)
        return size_t(nbd) * kChainRows * (256 + 2) * 8 + (gc ? size_t(1) : size_t(nbd - 1)) * kChainRows * (256 / 4 + 2) * 8 + 2048;
    };
    {
        int32_t r256 = 0; lds_need(256, r256);
        const size_t cap = size_t(160) * 1024;
        // (five buffers measured the same as four at 111 Mb with LDS carries: four leave more room for the ring of
        // long-read windows; with global carries the tails run a chunk further behind the chain wave: five when they fit)
        const bool want5 = gc ? getenv("BOSSX_FLOW_BUFS4") == nullptr : getenv("BOSSX_FLOW_BUFS5") != nullptr;
        h->chain_flow_bufs = flow_static(5) + size_t(r256) * 8 <= cap && want5 ? 5 : 4;
        h->chain_flow_fits = flow_static(4) + size_t(r256) * 8 <= cap;
    }
    const bool flow = h->matrix_chain && h->chain_flow && h->chain_flow_fits && !getenv("BOSSX_CHAIN_128");
    if ((n_blocks > 256 && !flow) || getenv("BOSSX_CHAIN_128") || lds_need(256, ring) > 160 * 1024) h->chain_ch = 128;
    if (lds_need(h->chain_ch, ring) > 160 * 1024)
        return fail(h, BOSSX_E_WINDOW, "read-length window exceeds the LDS ring (reads longer than ~1 Mb in the 95th percentile)");
    P.ds = h->d_ds; P.benefit = h->d_benefit; P.ctrl = h->d_ctrl; P.ct = table_of(h);
    P.B = h->B; P.nb = h->nb; P.ring = ring; P.gate = 0;
    P.tile_done = nullptr; P.epoch = h->epoch; P.wait_ticks = 200000000ll;   // 2 s
    P.never_ready = 0;
    P.zero_stats = nullptr; P.n_zero = 0;
    P.probe = getenv("BOSSX_CHAIN_PROBE") ? reinterpret_cast<long long *>(h->d_stats + kStatWords + 8) : nullptr;
    P.mismatch_log = (getenv("BOSSX_SPEC_STATS") && h->d_spec_stats) ? reinterpret_cast<long long *>(h->d_spec_stats + 84) : nullptr;
    P.carry_ring = nullptr;
    if (h->chain_gc) {      // one ring per chain block
        const size_t need = n_blocks * size_t(kCarryRing) * (256 / 4 / 2) * 64;
        if (need > h->carry_ring_cap) {
            HIPCHK(hipStreamSynchronize(h->stream));
            if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
            if (h->d_carry_ring) HIPCHK(hipFree(h->d_carry_ring));
            h->d_carry_ring = nullptr; h->carry_ring_cap = 0;
            int rc2 = dev_alloc(h, &h->d_carry_ring, need, true);
            if (rc2) return rc2;
            h->carry_ring_cap = need;
        }
        P.carry_ring = h->d_carry_ring;
    }
    P.max_limit = std::min<int64_t>(h->B, h->n_sites_all / kWindow);
    lds = size_t(ring) * sizeof(double);
    return BOSSX_OK;
}

extern "C++" {
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: the size each
// instantiation has been cleared for is remembered per ENGINE (= per device), not per process.
void grant_lds(bossx_engine *h, const void *fn, size_t lds) {
    size_t &allowed = h->lds_granted[fn];
    if (lds > allowed) {
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        allowed = lds;
    }
}
template <bool MATRIX, bool LIVE, int CH>
void launch_chain_variant(bossx_engine *h, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const ChainParams &P) {
    grant_lds(h, reinterpret_cast<const void *>(benefit_chain_kernel<MATRIX, LIVE, CH>), lds);
    hipLaunchKernelGGL((benefit_chain_kernel<MATRIX, LIVE, CH>), grid, block, lds, stream, P);
}
template <bool LIVE, int CH, int NBD, int CE>
void launch_chain_flow_ce(bossx_engine *h, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const ChainParams &P) {
    grant_lds(h, reinterpret_cast<const void *>(benefit_chain_flow_kernel<LIVE, CH, NBD, CE>), lds);
    hipLaunchKernelGGL((benefit_chain_flow_kernel<LIVE, CH, NBD, CE>), grid, block, lds, stream, P);
}
template <bool LIVE, int CH, int NBD>
void launch_chain_flow(bossx_engine *h, int ce, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const ChainParams &P) {
    if (P.carry_ring) {
        grant_lds(h, reinterpret_cast<const void *>(benefit_chain_flow_kernel<LIVE, CH, NBD, 2, true>), lds);
        hipLaunchKernelGGL((benefit_chain_flow_kernel<LIVE, CH, NBD, 2, true>), grid, block, lds, stream, P);
    } else if (ce == 1) launch_chain_flow_ce<LIVE, CH, NBD, 1>(h, grid, block, lds, stream, P);
    else launch_chain_flow_ce<LIVE, CH, NBD, 2>(h, grid, block, lds, stream, P);
}
}  // extern "C++"

void launch_chain(bossx_engine *h, const ChainParams &P0, size_t lds, hipStream_t stream = nullptr) {
    h->codes_valid = false;          // (the benefits change: the exponent codes of the last histogram pass no longer describe them)
    // the share of rewritten tiles describes the sweep of THIS update: a chain launched without one (bossx_benefit twice, a rerun) takes
    // the long way through the row hashes
    if (!h->tile_share_fresh) h->sweep_tile_share = 1.0;
    h->tile_share_fresh = false;
    if (!stream) stream = h->stream;
    ChainParams P = P0;
    P.rerun_bit = 0;
    time_begin(h, BOSSX_K_BENEFIT, stream);
    h->last_chain_spec = false;
    const bool live0 = P.tile_done != nullptr;
    if (!live0 && h->chain_spec && getenv("BOSSX_CHAIN_SERIAL_NOW")) { /* tests: this launch on the serial kernel, nothing counted */ }
    else if (!live0 && h->chain_spec && h->spec_pause > 0) { --h->spec_pause; ++h->spec_paused_updates; }
    else if (!live0 && h->chain_spec && h->matrix_chain && h->chain_ch == 256 && h->spec_total > 0) {
        // chunk-parallel: candidate tables -> stitched start values -> every segment at once (kernels.hip.inc)
        SpecParams Q;
        P.spec_chunk_off = h->d_chunk_off; P.spec_total = h->spec_total; P.spec_starts = h->d_spec_starts; P.seg_chunks = h->spec_seg_chunks;
        Q.C = P; Q.chunk_off = h->d_chunk_off; Q.total_chunks = h->spec_total; Q.tab = h->d_spec_tab; Q.starts = h->d_spec_starts;
        Q.stats = getenv("BOSSX_SPEC_STATS") ? h->d_spec_stats : nullptr;
        Q.probe = nullptr;
        Q.hash = getenv("BOSSX_SPEC_NO_SKIP") ? nullptr : h->d_spec_hash;
        // (a chunk is ~51 tiles and a row looks one window back: with a tenth of the tiles rewritten hardly a row is left standing)
        const bool stamps = !getenv("BOSSX_SPEC_NO_STAMPS") && (h->sweep_tile_share < 0.05 || getenv("BOSSX_SPEC_STAMPS"));
        Q.row_meta = h->d_row_meta; Q.tile_stamp = stamps ? h->d_tile_stamp : nullptr; Q.stamp_now = h->stamp_counter;
        Q.strict = getenv("BOSSX_SPEC_STRICT") ? atoi(getenv("BOSSX_SPEC_STRICT")) : 0;
#ifdef BOSSX_CAND_PROBE
        {
            const size_t waves = size_t(h->spec_total) * size_t((BOSSX_NWIN + kCandWin - 1) / kCandWin) * size_t(h->nb * 2);
            if (!h->d_cand_probe && hipMalloc(&h->d_cand_probe, waves * 64) == hipSuccess) h->cand_probe_waves = waves;
            if (h->d_cand_probe && hipMemsetAsync(h->d_cand_probe, 0, waves * 64, stream) == hipSuccess) Q.probe = h->d_cand_probe;
        }
#endif
        hipLaunchKernelGGL(chain_candidates_kernel, dim3(uint32_t(h->spec_total), (BOSSX_NWIN + kCandWin - 1) / kCandWin, uint32_t(h->nb * 2)), dim3(64), 0, stream, Q);
        Q.sup = (h->d_spec_sup && !h->spec_no_compose) ? h->d_spec_sup : nullptr; Q.sup_off = h->d_sup_off; Q.sup_total = h->spec_sup_total; Q.group = h->spec_seg_chunks;
        if (Q.sup)          // every group of seg_chunks rows composed into one super-row, all groups at once: the stitch walks groups
            hipLaunchKernelGGL(chain_compose_kernel, dim3(uint32_t(h->spec_sup_total), BOSSX_NWIN, uint32_t(h->nb * 2)), dim3(64),
                               size_t(h->spec_seg_chunks) * size_t(comp_row(h->spec_seg_chunks)) * sizeof(double), stream, Q);
        hipLaunchKernelGGL(chain_stitch_kernel, dim3(uint32_t(h->filt.size() * size_t(h->nb) * 2 * BOSSX_NWIN)), dim3(64), 0, stream, Q);
        if (getenv("BOSSX_SPEC_SELFTEST") && h->spec_total > h->spec_seg_chunks)      // (the first contig needs a second segment)
            hipLaunchKernelGGL(chain_spoil_kernel, dim3(1), dim3(1), 0, stream, h->d_spec_starts, int64_t(h->spec_seg_chunks));
        // the segments: the barrier-free pipeline where its LDS fits (round 5: 20-34 cycles per four bins against the barrier kernel's
        // 47-58), the barrier kernel otherwise (BOSSX_SEG_BARRIER=1 / BOSSX_CHAIN_BARRIER=1: always)
        const dim3 seg_grid(uint32_t(h->spec_max_segs), uint32_t(h->filt.size() * size_t(h->nb) * 2));
        if (h->chain_flow && h->chain_flow_fits && !getenv("BOSSX_SEG_BARRIER")) {
            grant_lds(h, reinterpret_cast<const void *>(benefit_chain_flow_kernel<false, 256, 4, 2, false, true>), lds);
            hipLaunchKernelGGL((benefit_chain_flow_kernel<false, 256, 4, 2, false, true>), seg_grid, dim3(kChainThreads), lds, stream, P);
        } else {
            grant_lds(h, reinterpret_cast<const void *>(benefit_chain_kernel<true, false, 256, true>), lds);
            hipLaunchKernelGGL((benefit_chain_kernel<true, false, 256, true>), seg_grid, dim3(kChainThreads), lds, stream, P);
        }
        h->last_chain_spec = true;
        ++h->spec_launches;
        // ... and behind them the serial chain, which does nothing unless a segment failed its check
        hipLaunchKernelGGL(chain_rerun_kernel, dim3(1), dim3(1), 0, stream, h->d_ctrl, 0);
        P.rerun_bit = kSpecMismatch;
    }
    const dim3 grid(uint32_t(h->filt.size() * size_t(h->nb) * 2)), block(kChainThreads);
    const bool live = P.tile_done != nullptr;
    const int ch = h->chain_ch;
    const bool use_flow = h->matrix_chain && h->chain_flow && ch == 256 && h->chain_flow_fits;
    const size_t fixed = use_flow ? size_t(h->chain_flow_bufs) * kChainRows * (256 + 2) * 8 + (P.carry_ring ? size_t(1) : size_t(h->chain_flow_bufs - 1)) * kChainRows * (256 / 4 + 2) * 8 + 2048
                                  : 2 * 2 * size_t(kChainRows) * size_t(ch + 2) * 8;     // static tiles of this instantiation
    if (live && grid.x <= 8 && fixed + lds < size_t(140) * 1024) {
        // A few long chains next to a running sweep: ask for enough LDS that no sweep block fits on
        // the chain's CU (less than the smallest sweep block's 22 KB stays free), otherwise the
        // sweep's waves share the chain wave's SIMD and slow the recurrence by ~6 % while they run.
        lds = size_t(140) * 1024 - fixed;
    }
    h->last_chain_live = live;
    if (h->matrix_chain && h->chain_flow && ch == 256 && h->chain_flow_fits) {
        // as many buffers between the stages as the LDS holds next to the ring of the current windows
        const int ce = h->chain_flow_ce;
        if (h->chain_flow_bufs == 5) { if (live) launch_chain_flow<true, 256, 5>(h, ce, grid, block, lds, stream, P); else launch_chain_flow<false, 256, 5>(h, ce, grid, block, lds, stream, P); }
        else { if (live) launch_chain_flow<true, 256, 4>(h, ce, grid, block, lds, stream, P); else launch_chain_flow<false, 256, 4>(h, ce, grid, block, lds, stream, P); }
    } else if (h->matrix_chain) {
        if (ch == 256) { if (live) launch_chain_variant<true, true, 256>(h, grid, block, lds, stream, P); else launch_chain_variant<true, false, 256>(h, grid, block, lds, stream, P); }
        else { if (live) launch_chain_variant<true, true, 128>(h, grid, block, lds, stream, P); else launch_chain_variant<true, false, 128>(h, grid, block, lds, stream, P); }
    } else {
        if (ch == 256) { if (live) launch_chain_variant<false, true, 256>(h, grid, block, lds, stream, P); else launch_chain_variant<false, false, 256>(h, grid, block, lds, stream, P); }
        else { if (live) launch_chain_variant<false, true, 128>(h, grid, block, lds, stream, P); else launch_chain_variant<false, false, 128>(h, grid, block, lds, stream, P); }
    }
    if (P.rerun_bit) hipLaunchKernelGGL(chain_rerun_kernel, dim3(1), dim3(1), 0, stream, h->d_ctrl, 1);
    // algorithmic bytes: read the downsampled scores once per direction, write both strands
    time_end(h, BOSSX_K_BENEFIT, double(h->B) * h->nb * (2 * 8.0 + 2 * 8.0), stream);
}

}  // namespace

namespace {
// For every consumer of the chain's results other than bossx_update (which has its own retry): a
// chain that ran next to the sweep may have given up waiting for it (kernels serialised by a
// profiler, or chain blocks starving the sweep) — its benefit array and running maximum are then
// void.  Wait for it, and if it timed out rerun it on the main stream, after the sweep, and keep
// the serial schedule from now on.  A no-op (no synchronisation) when no such chain is pending.
int settle_chain(bossx_engine *h) {
    if (!h->chain_on_stream2) return BOSSX_OK;
    HIPCHK(hipStreamSynchronize(h->stream2));
    h->chain_on_stream2 = false;
    int32_t err = 0;
    HIPCHK(hipMemcpy(&err, &h->d_ctrl->err, sizeof(err), hipMemcpyDeviceToHost));
    if (!(err & 4)) return BOSSX_OK;
    err &= ~4;
    HIPCHK(hipMemcpyAsync(&h->d_ctrl->err, &err, sizeof(err), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(&h->d_ctrl->max_bits, 0, sizeof(unsigned long long), h->stream));
    if (h->last_chain_live) h->overlap_ok = false; else h->chain_flow = false;
    ChainParams CP = h->last_chain;
    CP.tile_done = nullptr;                 // the main stream is behind the sweep: no waiting
    launch_chain(h, CP, h->last_chain_lds);
    HIPCHK(hipGetLastError());
    return BOSSX_OK;
}
}  // namespace

int bossx_benefit(bossx_engine *h, const int32_t *windows, const double *mult, double *max_benefit) {
    if (!h || !h->finalized || !windows || !mult) return fail(h, BOSSX_E_INVALID, "bad benefit call");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    ChainParams P;
    size_t lds = 0;
    int rc = fill_chain_params(h, windows, mult, P, lds);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(&h->d_ctrl->max_bits, 0, sizeof(unsigned long long), h->stream));
    h->max_bits_clear = false;
    launch_chain(h, P, lds);
    HIPCHK(hipGetLastError());
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, &h->d_ctrl->max_bits, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double mx;
    memcpy(&mx, &bits, sizeof(mx));
    if (max_benefit) *max_benefit = mx;
    if (P.probe) {
        long long pr[72];
        HIPCHK(hipMemcpy(pr, P.probe, sizeof(pr), hipMemcpyDeviceToHost));
        fprintf(stderr, "[chain probe] chunks=%lld\n", pr[64]);
        for (int w = 0; w < 16; ++w)
            fprintf(stderr, "  wave %d: total %lld  A %lld  B %lld  C %lld (cycles; per chunk %.0f / %.0f / %.0f / %.0f)\n", w, pr[w * 4],
                    pr[w * 4 + 1], pr[w * 4 + 2], pr[w * 4 + 3], double(pr[w * 4]) / double(pr[64]), double(pr[w * 4 + 1]) / double(pr[64]),
                    double(pr[w * 4 + 2]) / double(pr[64]), double(pr[w * 4 + 3]) / double(pr[64]));
    }
    return BOSSX_OK;
}

namespace {

int upload_fhat(bossx_engine *h, const bossx_fhat_desc *fh) {
    const int64_t target = h->n_sites_all / kWindow;
    if (fh->target != target) return fail(h, BOSSX_E_INVALID, "fhat target does not match Reference.n_sites // 100");
    if (fh->rep != 20) return fail(h, BOSSX_E_INVALID, "fhat repeat factor must be 20");
    int rc;
    if (fh->n_windows * 2 > h->fhat_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_fhat) HIPCHK(hipFree(h->d_fhat));
        if (h->h_fhat_pin) HIPCHK(hipHostFree(h->h_fhat_pin));
        h->h_fhat_pin = nullptr;
        h->fhat_cap = fh->n_windows * 2 + 64;
        if ((rc = dev_alloc(h, &h->d_fhat, size_t(h->fhat_cap)))) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_fhat_pin), size_t(h->fhat_cap) * sizeof(double), hipHostMallocDefault));
    }
    // through page-locked staging: a copy from pageable memory is serviced by the host when the
    // stream reaches it (45 us after the chain ended, measured); this one is a plain DMA
    const size_t bytes = size_t(fh->n_windows) * 2 * sizeof(double);
    HIPCHK(hipEventSynchronize(h->ev_fhat));        // the previous upload has left the staging buffer
    memcpy(h->h_fhat_pin, fh->fhat_c, bytes);
    HIPCHK(hipMemcpyAsync(h->d_fhat, h->h_fhat_pin, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_fhat, h->stream));
    return BOSSX_OK;
}

// BOSSX_UPDATE_FHAT_RESIDENT: the posterior from the resident counts, straight into d_fhat
struct FhatModel { int64_t n_windows, target_rs; double alpha, den, expected, on_target; };
int build_fhat(bossx_engine *h, const FhatModel *up) {
    if (!h->d_rs_counts || h->rs_windows != up->n_windows) return fail(h, BOSSX_E_INVALID, "resident f-hat: bossx_fhat_reset has not installed counts for this many windows");
    int rc;
    if (up->n_windows * 2 > h->fhat_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_fhat) HIPCHK(hipFree(h->d_fhat));
        if (h->h_fhat_pin) HIPCHK(hipHostFree(h->h_fhat_pin));
        h->h_fhat_pin = nullptr;
        h->fhat_cap = up->n_windows * 2 + 64;
        if ((rc = dev_alloc(h, &h->d_fhat, size_t(h->fhat_cap)))) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_fhat_pin), size_t(h->fhat_cap) * sizeof(double), hipHostMallocDefault));
    }
    FhatParams P;
    P.counts = h->d_rs_counts; P.fhat = h->d_fhat; P.sums = h->d_rs_sums;
    P.n = up->n_windows; P.rep = 20; P.d = up->target_rs - 20 * up->n_windows;
    P.alpha = up->alpha; P.den = up->den; P.expected = up->expected; P.on_target = up->on_target;
    // on the side stream, behind the batch's own fhat_add; the main stream (where the histogram follows) waits for the result.  Not when
    // the chain ran next to the sweep on stream2 (opt-in): that schedule orders itself through ev_fhat on the main stream.
    const bool side = !h->chain_on_stream2 && h->stream_fhat != nullptr && !getenv("BOSSX_FHAT_MAIN_STREAM");
    hipStream_t fs = side ? h->stream_fhat : h->stream;
    if (!side && h->ev_rs_keys) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_rs_keys, 0));      // (this batch's counts were added on the side stream)
    HIPCHK(hipMemsetAsync(h->d_rs_sums, 0, 6 * sizeof(unsigned long long), fs));
    // (few blocks: every wave ends with atomics on the same three accumulators — 2048 blocks spent 45 us queueing there)
    const uint32_t blocks = uint32_t(std::min<int64_t>((up->n_windows * 2 + 255) / 256, 128));
    hipLaunchKernelGGL(fhat_terms_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, fs, P);
    hipLaunchKernelGGL(fhat_scale_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, fs, P);
    HIPCHK(hipGetLastError());
    if (side) {
        HIPCHK(hipEventRecord(h->ev_fhat_side, fs));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_fhat_side, 0));
    }
    HIPCHK(hipEventRecord(h->ev_fhat, h->stream));
    return BOSSX_OK;
}

// The statistics land in the limb replicas (d_stats_rep); `clear`: zero them first (the fused update lets the chain kernel do it).
int launch_hist(bossx_engine *h, const bossx_fhat_desc *fh, int gate, bool clear = true) {
    const int64_t target = h->n_sites_all / kWindow;
    if (clear) HIPCHK(hipMemsetAsync(h->d_stats_rep, 0, kStatRepWords * sizeof(unsigned long long), h->stream));
    HistParams P;
    P.benefit = h->d_benefit; P.fhat_c = h->d_fhat;
    P.replicas = h->d_stats_rep;
    P.code = getenv("BOSSX_NO_CODES") ? nullptr : h->d_bcode; P.Bc = h->Bc;
    P.ct = table_of(h); P.B = h->B; P.target = target; P.dpad = target > h->B ? target - h->B : 0;
    P.target_rs = fh->target_rs; P.d2 = target - fh->target_rs;
    P.fexp = fh->n_windows * fh->rep; P.d1 = fh->target_rs - P.fexp;
    P.nb = h->nb; P.all_local = h->all_local ? 1 : 0; P.gate = gate; P.ctrl = h->d_ctrl;
    if (P.d2 < 0) P.d2 = 0;   // trimmed instead of padded: indices unchanged
    if (P.d1 < 0) P.d1 = 0;
    // every block loops over spans of 4096 positions; no more blocks than stay resident (two per CU: 59 KB of LDS each)
    const int64_t spans = (target + kHistSpan - 1) / kHistSpan;
    static const int64_t cap = getenv("BOSSX_HIST_BLOCKS") ? atoll(getenv("BOSSX_HIST_BLOCKS")) : 1024;       // (experiments)
    const int64_t per_row = std::max<int64_t>(1, cap / (int64_t(h->nb) * 2));
    const int64_t blocks = std::min<int64_t>(spans, per_row);
    P.probe = nullptr;
#ifdef BOSSX_HIST_PROBE
    P.probe = reinterpret_cast<long long *>(h->d_stats + kStatWords + 8);
#endif
    time_begin(h, BOSSX_K_HIST);
    hipLaunchKernelGGL(threshold_hist_kernel, dim3(uint32_t(std::max<int64_t>(blocks, 1)), uint32_t(h->nb * 2)), dim3(kHistThreads), 0, h->stream, P);
#ifdef BOSSX_HIST_PROBE
    {
        long long pr[24];
        (void)hipStreamSynchronize(h->stream);
        if (hipMemcpy(pr, P.probe, sizeof(pr), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[hist probe] %lld blocks/row; block 0: span in LDS %lld, walked %lld, sums %lld, flush issued %lld | middle block: %lld %lld %lld %lld (cycles)\n",
                    (long long)blocks, pr[1], pr[2], pr[3], pr[4], pr[9], pr[10], pr[11], pr[12]);
    }
#endif
    // algorithmic bytes: every element read once, one code byte written
    time_end(h, BOSSX_K_HIST, double(target) * h->nb * 2 * (8.0 + (P.code ? 1.0 : 0.0)));
    HIPCHK(hipGetLastError());
    h->codes_valid = P.code != nullptr;
    return BOSSX_OK;
}

// The replicas' sums where somebody wants them: as limbs for the multi-GPU all-reduce (d_limbs) and / or in the C-ABI's (lo, hi)
// form (d_stats: counts | fgrid | ubar) for a host that reads the statistics.
void launch_fold(bossx_engine *h, bool to_limbs, bool to_canon, int gate) {
    hipLaunchKernelGGL(fold_limbs_kernel, dim3((BOSSX_HIST_BINS + 1 + 255) / 256), dim3(256), 0, h->stream, h->d_stats_rep, int32_t(kHistRep),
                       to_limbs ? h->d_limbs : nullptr, to_canon ? h->d_stats : nullptr, h->d_ctrl, gate);
}

// `use_codes`: the threshold in the control block was picked ON THE DEVICE from the histogram pass that also left d_bcode (the
// kernel then compares one code byte per element instead of a double wherever Ctrl::thr_code allows)
int launch_mask(bossx_engine *h, int gate, bool with_tails = false, const PickParams *pick = nullptr,
                unsigned long long *host_result = nullptr, uint8_t *host_strat = nullptr, bool use_codes = false, bool delta = false) {
    MaskParams P;
    // (only what changes crosses PCIe — if the caller's buffer is the one this engine mirrored the masks into last time, nothing has
    // written d_strat since, and the caller says it has not written into the buffer either: BOSSX_UPDATE_STRAT_DELTA)
    P.host_delta = (delta && host_strat && host_strat == h->mirror_valid && !getenv("BOSSX_NO_MASK_DELTA")) ? 1 : 0;
    h->mirror_valid = host_strat;
    P.do_pick = pick ? 1 : 0;
    if (pick) P.pick = *pick; else P.pick = PickParams{};
    P.host_result = host_result; P.dev_result = reinterpret_cast<const unsigned long long *>(h->d_result);
    P.result_words = int32_t(h->result_bytes / 8);
    P.host_strat = host_strat;
    P.benefit = h->d_benefit; P.bucket_on = h->d_bucket_on; P.strat = h->d_strat; P.ct = table_of(h);
    P.code = (use_codes && h->codes_valid) ? h->d_bcode : nullptr; P.Bc = h->Bc;
    P.B = h->B; P.NBK = h->NBK; P.rows = h->rows; P.nb = h->nb; P.gate = gate; P.ctrl = h->d_ctrl;
    P.tails = with_tails ? h->d_tails : nullptr; P.tail_k = int32_t(h->filt.size());
    const int64_t span = int64_t(256) * kMaskRun;
    // (with the threshold choice in every block — ~90 loads per lane of its first wave — one block per CU for small references, each
    // looping over its spans; more where there are many spans per block to pay for it: at GRCh38 256 blocks were ONE wave per SIMD and
    // the kernel waited for its own loads, 585 us for 124 MB)
    const int64_t spans = (h->rows + span - 1) / span;
    const int64_t blocks = std::min<int64_t>(spans, pick ? std::min<int64_t>(std::max<int64_t>(spans / 4, 256), 2048) : 4096);
    time_begin(h, BOSSX_K_MASK);
    if (pick) hipLaunchKernelGGL(strategy_mask_kernel<true>, dim3(uint32_t(std::max<int64_t>(blocks, 1))), dim3(256), 0, h->stream, P);
    else hipLaunchKernelGGL(strategy_mask_kernel<false>, dim3(uint32_t(std::max<int64_t>(blocks, 1))), dim3(256), 0, h->stream, P);
    // algorithmic bytes: per row * strand * barcode one code byte (or one double) read, one mask byte written
    time_end(h, BOSSX_K_MASK, double(h->rows) * h->nb * 2 * (P.code ? 2.0 : 9.0));
    HIPCHK(hipGetLastError());
    return BOSSX_OK;
}

}  // namespace

int bossx_fhat_reset(bossx_engine *h, const double *counts, int64_t n_windows) {
    if (!h || !h->finalized || n_windows < 0) return fail(h, BOSSX_E_INVALID, "bad fhat_reset call");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->stream_fhat) HIPCHK(hipStreamSynchronize(h->stream_fhat));
    int rc;
    if (n_windows != h->rs_windows || !h->d_rs_counts) {
        if (h->d_rs_counts) HIPCHK(hipFree(h->d_rs_counts));
        h->d_rs_counts = nullptr; h->rs_windows = 0;
        if ((rc = dev_alloc(h, &h->d_rs_counts, size_t(std::max<int64_t>(n_windows * 2, 1))))) return rc;
        h->rs_windows = n_windows;
    }
    if (!h->d_rs_sums && (rc = dev_alloc(h, &h->d_rs_sums, size_t(8)))) return rc;
    const size_t bytes = size_t(n_windows) * 2 * sizeof(double);
    if (counts) HIPCHK(hipMemcpy(h->d_rs_counts, counts, bytes, hipMemcpyHostToDevice));
    else if (bytes) HIPCHK(hipMemset(h->d_rs_counts, 0, bytes));
    return BOSSX_OK;
}

int bossx_fhat_add(bossx_engine *h, const int64_t *keys, int32_t n_keys) {
    if (!h || !h->finalized || n_keys < 0 || (n_keys > 0 && !keys)) return fail(h, BOSSX_E_INVALID, "bad fhat_add call");
    if (!h->d_rs_counts) return fail(h, BOSSX_E_INVALID, "fhat_add before fhat_reset");
    if (n_keys == 0) return BOSSX_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc;
    if (!h->ev_rs_keys) HIPCHK(hipEventCreateWithFlags(&h->ev_rs_keys, hipEventDisableTiming));
    if (size_t(n_keys) > h->rs_keys_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->stream_fhat) HIPCHK(hipStreamSynchronize(h->stream_fhat));
        if (h->d_rs_keys) HIPCHK(hipFree(h->d_rs_keys));
        if (h->h_rs_keys_pin) HIPCHK(hipHostFree(h->h_rs_keys_pin));
        h->d_rs_keys = nullptr; h->h_rs_keys_pin = nullptr; h->rs_keys_cap = 0;
        const size_t cap = size_t(n_keys) * 2 + 1024;
        if ((rc = dev_alloc(h, &h->d_rs_keys, cap))) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_rs_keys_pin), cap * sizeof(int64_t), hipHostMallocDefault));
        h->rs_keys_cap = cap;
    }
    // the caller's array is only borrowed for the call: the keys go through page-locked staging
    // (the previous batch's copy has left it: the event is long signalled in practice)
    HIPCHK(hipEventSynchronize(h->ev_rs_keys));
    memcpy(h->h_rs_keys_pin, keys, size_t(n_keys) * sizeof(int64_t));
    hipStream_t fs = h->stream_fhat ? h->stream_fhat : h->stream;
    HIPCHK(upload_async(h->d_rs_keys, h->h_rs_keys_pin, size_t(n_keys) * sizeof(int64_t), fs));
    hipLaunchKernelGGL(fhat_add_kernel, dim3(uint32_t((n_keys + 255) / 256)), dim3(256), 0, fs,
                       h->d_rs_counts, h->d_rs_keys, n_keys, h->rs_windows * 2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev_rs_keys, fs));      // the keys have left the staging buffer AND the counts hold them
    return BOSSX_OK;
}

int bossx_histogram(bossx_engine *h, double normaliser, const bossx_fhat_desc *fh, int64_t *counts,
                    uint64_t *fgrid_fx, uint64_t *ubar0_fx) {
    if (!h || !h->finalized || !fh || !fh->fhat_c || !counts || !fgrid_fx || !ubar0_fx) return fail(h, BOSSX_E_INVALID, "bad histogram call");
    if (!(normaliser > 0)) return fail(h, BOSSX_E_EMPTY, "no non-zero benefit (np.max of an empty array)");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    int rc = upload_fhat(h, fh);
    if (rc) return rc;
    unsigned long long bits;
    memcpy(&bits, &normaliser, sizeof(bits));
    HIPCHK(hipMemcpyAsync(&h->d_ctrl->max_bits, &bits, sizeof(bits), hipMemcpyHostToDevice, h->stream));
    h->max_bits_clear = false;
    if ((rc = launch_hist(h, fh, 0))) return rc;
    launch_fold(h, false, /*to_canon=*/true, 0);
    std::vector<unsigned long long> host(kStatWords);
    HIPCHK(hipMemcpyAsync(host.data(), h->d_stats, kStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < BOSSX_HIST_BINS; ++i) counts[i] = int64_t(host[size_t(i)]);
    memcpy(fgrid_fx, host.data() + BOSSX_HIST_BINS, size_t(BOSSX_HIST_BINS) * 2 * sizeof(uint64_t));
    memcpy(ubar0_fx, host.data() + BOSSX_HIST_BINS * 3, 2 * sizeof(uint64_t));
    return BOSSX_OK;
}

int bossx_apply_threshold(bossx_engine *h, double threshold) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "bad apply_threshold call");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    HIPCHK(hipMemcpyAsync(&h->d_ctrl->threshold, &threshold, sizeof(double), hipMemcpyHostToDevice, h->stream));
    return launch_mask(h, 0);
}

namespace {
void launch_buckets(bossx_engine *h, double threshold) {
    BucketParams P;
    P.bucket_sums = h->d_bucket_sums; P.bucket_on = h->d_bucket_on; P.contig_on = h->d_contig_on;
    P.ctrl = h->d_ctrl; P.ct = table_of(h); P.NBK = h->NBK; P.nb = h->nb; P.threshold = threshold;
    const int64_t blocks = std::min<int64_t>((h->NBK * h->nb + 255) / 256, 1024);
    hipLaunchKernelGGL(bucket_switch_kernel, dim3(uint32_t(std::max<int64_t>(blocks, 1))), dim3(256), 0, h->stream, P);
}
}  // namespace

int bossx_update_begin(bossx_engine *h, double bucket_threshold) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "bad update_begin call");
    if (!h->lut_set) return fail(h, BOSSX_E_INVALID, "update before set_lut");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    int rc = launch_sweep(h);      // records ev_begin between its prep launch and the sweep proper
    if (rc) return rc;
    launch_buckets(h, bucket_threshold);
    HIPCHK(hipEventRecord(h->ev_sweep, h->stream));     // sweep + bucket switches of this update are behind this
    HIPCHK(hipGetLastError());
    h->sweep_in_flight = true;
    return BOSSX_OK;
}

int bossx_device_ptr(bossx_engine *h, int32_t which, void **ptr, size_t *bytes) {
    if (!h || !h->finalized || !ptr || !bytes) return fail(h, BOSSX_E_INVALID, "bad device_ptr call");
    switch (which) {
        case BOSSX_PTR_ARMED: *ptr = &h->d_ctrl->any_on; *bytes = sizeof(int32_t); break;
        case BOSSX_PTR_NORMALISER: *ptr = &h->d_ctrl->max_bits; *bytes = sizeof(unsigned long long); break;
        case BOSSX_PTR_LIMBS: *ptr = h->d_limbs; *bytes = size_t(BOSSX_HIST_BINS + 1) * 5 * sizeof(long long); break;
        case BOSSX_PTR_TAILS: *ptr = h->d_tails; *bytes = (h->filt.size() * h->filt.size() * 2 * size_t(h->nb) + 1) * sizeof(double); break;
        default: return fail(h, BOSSX_E_INVALID, "unknown device pointer selector");
    }
    return BOSSX_OK;
}

static void launch_tails(bossx_engine *h) {
    MaskParams P;
    P.benefit = h->d_benefit; P.bucket_on = h->d_bucket_on; P.strat = h->d_strat; P.ct = table_of(h);
    P.B = h->B; P.NBK = h->NBK; P.rows = h->rows; P.nb = h->nb; P.gate = 1; P.ctrl = h->d_ctrl;
    P.tails = nullptr; P.tail_k = int32_t(h->filt.size());
    P.do_pick = 0; P.pick = PickParams{};
    P.host_result = nullptr; P.dev_result = nullptr; P.result_words = 0; P.host_strat = nullptr; P.host_delta = 0;
    h->mirror_valid = nullptr;
    hipLaunchKernelGGL(export_tails_kernel, dim3(64), dim3(256), 0, h->stream, P, h->d_tails);
}

int bossx_fhat_build(bossx_engine *h, int64_t n_windows, int64_t target_rs, double alpha, double den, double expected, double on_target) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "bad fhat_build call");
    HIPCHK(hipSetDevice(h->cfg.device));
    const FhatModel fm{n_windows, target_rs, alpha, den, expected, on_target};
    return build_fhat(h, &fm);
}

int bossx_dist_hist(bossx_engine *h, const bossx_fhat_desc *fh) {
    if (!h || !h->finalized || !fh) return fail(h, BOSSX_E_INVALID, "bad dist_hist call");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    int rc = BOSSX_OK;
    if (fh->fhat_c) rc = upload_fhat(h, fh);         // NULL: the posterior bossx_fhat_build left in HBM
    else if (!h->d_fhat || fh->n_windows * 2 > h->fhat_cap) rc = fail(h, BOSSX_E_INVALID, "dist_hist without f-hat: call bossx_fhat_build first");
    if (rc) return rc;
    if ((rc = launch_hist(h, fh, 1))) return rc;
    launch_fold(h, /*to_limbs=*/true, /*to_canon=*/false, 1);      // this device's sums, ready for the SUM all-reduce
    HIPCHK(hipGetLastError());
    return BOSSX_OK;
}

int bossx_dist_pick(bossx_engine *h, double tc) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "bad dist_pick call");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    if (h->norm_in_tails) {
        // short form: the halo rows are already exchanged, so nothing separates the threshold choice
        // from the masks — bossx_dist_finish's mask kernel picks it itself (one launch fewer)
        h->dist_tc = tc;
        h->dist_pick_fused = true;
        h->norm_in_tails = false;
        return BOSSX_OK;
    }
    PickParams PP{};
    PP.limbs = h->d_limbs; PP.n_rep = 1; PP.ctrl = h->d_ctrl; PP.tc = tc; PP.gate = 1;
    hipLaunchKernelGGL(threshold_pick_kernel, dim3(1), dim3(64), 0, h->stream, PP);
    launch_tails(h);
    HIPCHK(hipGetLastError());
    return BOSSX_OK;
}

int bossx_dist_tails(bossx_engine *h) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "bad dist_tails call");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    launch_tails(h);          // the normaliser already sits behind the tails (ctrl.max_bits)
    h->norm_in_tails = true;
    HIPCHK(hipGetLastError());
    return BOSSX_OK;
}

// Device-to-host copy of every mask: as bytes (Contig.strat layout) or packed 8:1.
static int copy_masks(bossx_engine *h, uint8_t *dst, bool bits) {
    if (!bits) {
        HIPCHK(hipMemcpyAsync(dst, h->d_strat, size_t(h->strat_bytes), hipMemcpyDeviceToHost, h->stream));
        return BOSSX_OK;
    }
    const int64_t nout = (h->strat_bytes + 7) / 8;
    if (!h->d_strat_bits) {
        int rc = dev_alloc(h, &h->d_strat_bits, size_t(nout > 0 ? nout : 1));
        if (rc) return rc;
    }
    if (nout == 0) return BOSSX_OK;
    const int64_t blocks = std::min<int64_t>((nout + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_strat_kernel, dim3(unsigned(blocks)), dim3(256), 0, h->stream,
                       h->d_strat, h->d_strat_bits, h->strat_bytes);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, h->d_strat_bits, size_t(nout), hipMemcpyDeviceToHost, h->stream));
    return BOSSX_OK;
}

namespace { int dist_finish_impl(bossx_engine *h, uint8_t *strat_all, uint8_t *contig_on, bossx_update_result *res, bool bits, int mode = 0); }

int bossx_dist_finish(bossx_engine *h, uint8_t *strat_all, uint8_t *contig_on, bossx_update_result *res) {
    return dist_finish_impl(h, strat_all, contig_on, res, false);
}

namespace {
// mode 0: masks + results, waited for; 1: enqueued only (bossx_dist_update_launch); 2: wait and read (bossx_dist_update_collect)
int dist_finish_impl(bossx_engine *h, uint8_t *strat_all, uint8_t *contig_on, bossx_update_result *res, bool bits, int mode) {
    if (!h || !h->finalized || !res) return fail(h, BOSSX_E_INVALID, "bad dist_finish call");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = BOSSX_OK;
    if (mode != 2) {
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    h->sweep_in_flight = false;
    if (h->dist_pick_fused) {
        PickParams PP{};
        PP.limbs = h->d_limbs; PP.n_rep = 1; PP.ctrl = h->d_ctrl; PP.tc = h->dist_tc; PP.gate = 1;
        rc = launch_mask(h, 1, true, &PP, nullptr, nullptr, /*use_codes=*/true);
        h->dist_pick_fused = false;
    } else {
        rc = launch_mask(h, 1, true, nullptr, nullptr, nullptr, /*use_codes=*/true);
    }
    if (rc) return rc;
    }
    const size_t need = h->result_bytes;
    if ((rc = ensure_pin(h, need + 64))) return rc;
    char *pin = static_cast<char *>(h->h_pin);
    Ctrl *hc = reinterpret_cast<Ctrl *>(pin);
    int32_t *herr = reinterpret_cast<int32_t *>(pin + sizeof(Ctrl));
    uint8_t *hon = reinterpret_cast<uint8_t *>(pin + sizeof(Ctrl) + 16);
    if (mode != 2) {
        HIPCHK(hipMemcpyAsync(pin, h->d_result, h->result_bytes, hipMemcpyDeviceToHost, h->stream));
        if (strat_all && (rc = copy_masks(h, strat_all, bits))) return rc;
    }
    if (mode == 1) return BOSSX_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    note_spec_result(h, herr);
    if (*herr) {
        HIPCHK(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
        return fail(h, BOSSX_E_RANGE, "a read contains a base other than A/C/G/T inside an aligned segment");
    }
    if (contig_on) {
        for (size_t i = 0; i < h->contigs.size(); ++i) contig_on[i] = 0;
        for (size_t k = 0; k < h->filt.size(); ++k) contig_on[size_t(h->filt[k])] = hon[k];
    }
    res->updated = hc->any_on; res->any_on = hc->any_on;
    res->strat_size = hc->strat_size; res->n_bins = hc->n_bins;
    res->threshold = hc->threshold; res->ubar0 = hc->ubar0;
    res->argmax_margin = hc->argmax_margin; res->thr_code = hc->thr_code;
    memcpy(&res->normaliser, &hc->max_bits, sizeof(double));
    if (res->updated && (hc->err & 2)) {
        HIPCHK(hipMemsetAsync(&h->d_ctrl->err, 0, sizeof(int32_t), h->stream));
        return fail(h, BOSSX_E_EMPTY, "no non-zero benefit (np.max of an empty array)");
    }
    return BOSSX_OK;
}
}  // namespace

// ---- native collectives driver ------------------------------------------------------------------------
extern "C++" {
namespace {
// librccl through dlopen: libbossx.so stays loadable (and single-GPU runs stay possible) where no RCCL is
// installed, and inside a torch process the already loaded copy is the one that is found.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // BOSSX_RCCL_LIB: another library with RCCL's entry points (tests/rccl_loopback: ranks that are threads of one
        // process on one device — the native driver's N > 1 path on a one-GPU box)
        if (const char *e = getenv("BOSSX_RCCL_LIB")) {
            if (!(lib = dlopen(e, RTLD_NOW | RTLD_LOCAL))) { err = std::string("BOSSX_RCCL_LIB: ") + (dlerror() ? dlerror() : e); return false; }
        } else {
            const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char *n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        }
        if (!lib) { err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !AllGather || !GetErrorString) { lib = nullptr; return false; }
        return true;
    }
};
RcclApi &rccl() { static RcclApi api; return api; }
void rccl_destroy(ncclComm_t comm) { if (rccl().lib && comm) rccl().CommDestroy(comm); }

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(h, BOSSX_E_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_));    \
    } while (0)

// in-place all-reduce on the engine's stream (ordered between its kernels)
int dist_allreduce(bossx_engine *h, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op) {
    NCCLCHK(rccl().AllReduce(buf, buf, count, dt, op, h->comm, h->stream));
    ++h->n_collectives;
    return BOSSX_OK;
}
}  // namespace
}  // extern "C++"

int bossx_dist_unique_id(uint8_t *id) {
    if (!id || !rccl().load()) return BOSSX_E_HIP;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return BOSSX_E_HIP;
    static_assert(sizeof(u) == BOSSX_NCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof(u));
    return BOSSX_OK;
}

int bossx_dist_init(bossx_engine *h, const uint8_t *id, int32_t rank, int32_t world) {
    if (!h || !h->finalized || !id || world < 1 || rank < 0 || rank >= world) return fail(h, BOSSX_E_INVALID, "bad dist_init call");
    if (!rccl().load()) return fail(h, BOSSX_E_HIP, rccl().err);
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->comm) { rccl_destroy(h->comm); h->comm = nullptr; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCLCHK(rccl().CommInitRank(&h->comm, world, u, rank));
    h->dist_rank = rank; h->dist_world = world; h->dist_armed = false;
    // the protocol consumes the chain through stage-wise entry points: keep it after the sweep
    h->overlap_ok = false;
    return BOSSX_OK;
}

int64_t bossx_dist_collectives(const bossx_engine *h) { return h ? h->n_collectives : 0; }

// `bytes` bytes of every rank, in rank order, into recv_all[world * bytes] (host memory on both sides): the per-batch
// summaries of sharded reads (mapping columns + read lengths, parallel.py account_batch) travel through the engine's
// own communicator and stream — page-locked staging, one RCCL all-gather, one synchronisation.
int bossx_dist_allgather(bossx_engine *h, const void *send, void *recv_all, size_t bytes) {
    if (!h || !h->comm || !send || !recv_all || !bytes) return fail(h, BOSSX_E_INVALID, "bad dist_allgather call (bossx_dist_init first)");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t world = size_t(h->dist_world), total = (world + 1) * bytes;
    int rc;
    if ((rc = grow_dev(h, &h->d_gather, &h->gather_cap, total, 256))) return rc;
    if ((rc = grow_pin(h, &h->h_gather_pin, &h->gather_pin_cap, total))) return rc;
    memcpy(h->h_gather_pin, send, bytes);
    // On a side stream: the caller has just put this update's sweep on the engine's stream (update_begin) so that the exchange
    // and the host's bookkeeping overlap with it; on that stream the host would wait for the whole sweep (10 ms at GRCh38).  The
    // communicator is used by one stream at a time: every other collective of an update is issued after this call has returned, and
    // every rank makes the calls in the same order.
    hipStream_t gs = h->stream2 ? h->stream2 : h->stream;
    HIPCHK(hipMemcpyAsync(h->d_gather, h->h_gather_pin, bytes, hipMemcpyHostToDevice, gs));
    NCCLCHK(rccl().AllGather(h->d_gather, h->d_gather + bytes, bytes, ncclInt8, h->comm, gs));
    ++h->n_collectives;
    HIPCHK(hipMemcpyAsync(h->h_gather_pin + bytes, h->d_gather + bytes, world * bytes, hipMemcpyDeviceToHost, gs));
    HIPCHK(hipStreamSynchronize(gs));
    memcpy(recv_all, h->h_gather_pin + bytes, world * bytes);
    return BOSSX_OK;
}

int bossx_chain_stats(const bossx_engine *h, int64_t out[4]) {
    if (!h || !out) return BOSSX_E_INVALID;
    out[0] = h->spec_launches; out[1] = h->spec_paused_updates; out[2] = h->spec_plain_total; out[3] = h->spec_mismatches;
    return BOSSX_OK;
}

int bossx_chain_counters(bossx_engine *h, int64_t out[8]) {
    if (!h || !out) return BOSSX_E_INVALID;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!h->d_spec_stats) return BOSSX_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long st[16];
    HIPCHK(hipMemcpy(st, h->d_spec_stats, sizeof(st), hipMemcpyDeviceToHost));
    out[0] = int64_t(st[0]); out[1] = int64_t(st[8]); out[2] = int64_t(st[1]); out[3] = int64_t(st[12]); out[4] = int64_t(st[13]); out[5] = int64_t(st[14]);
    out[6] = int64_t(st[9]); out[7] = int64_t(st[10] + st[11]);
    return BOSSX_OK;
}

int bossx_dist_chain(bossx_engine *h, const int32_t *windows, const double *mult) {
    if (!h || !h->finalized || !h->comm || !windows || !mult) return fail(h, BOSSX_E_INVALID, "bad dist_chain call (bossx_dist_init first)");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc;
    if (!h->dist_armed && (rc = dist_allreduce(h, &h->d_ctrl->any_on, 1, ncclInt32, ncclMax))) return rc;      // core.py:111 is a global decision
    return bossx_update_benefit(h, windows, mult);                                                            // gated on the (now global) flag
}

namespace {
int dist_update_run(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                    bossx_update_result *res, int mode) {
    if (!h || !h->finalized || !up || !res) return fail(h, BOSSX_E_INVALID, "bad dist_update call");
    if (!h->comm) return fail(h, BOSSX_E_INVALID, "dist_update before bossx_dist_init");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc;
    if (mode == 2) {
        const bool have_inputs = up->fhat_c != nullptr || ((up->flags & BOSSX_UPDATE_FHAT_RESIDENT) != 0);
        rc = dist_finish_impl(h, strat_all, contig_on, res, (up->flags & BOSSX_UPDATE_STRAT_BITS) != 0, 2);
        if (rc) return rc;
        if (res->any_on) h->dist_armed = true;
        if (!have_inputs) res->updated = 0;
        return BOSSX_OK;
    }
    if (mode == 1 && h->chain_on_stream2) mode = 0;
    if (!(up->flags & BOSSX_UPDATE_SWEEP_DONE) && (rc = bossx_update_begin(h, up->bucket_threshold))) return rc;
    const bool fhat_resident = (up->flags & BOSSX_UPDATE_FHAT_RESIDENT) != 0 && up->fhat_c == nullptr;
    const bool have_strategy_inputs = up->fhat_c != nullptr || fhat_resident;
    if (!(up->flags & BOSSX_UPDATE_BENEFIT_DONE)) {
        if (!h->dist_armed && (rc = dist_allreduce(h, &h->d_ctrl->any_on, 1, ncclInt32, ncclMax))) return rc;
        if (have_strategy_inputs && (rc = bossx_update_benefit(h, up->windows, up->mult))) return rc;
    }
    if (have_strategy_inputs) {
        // halo rows + the normaliser behind them: ONE MAX all-reduce (non-negative, one contributor each: exact)
        if ((rc = bossx_dist_tails(h))) return rc;
        const size_t n_t = h->filt.size() * h->filt.size() * 2 * size_t(h->nb) + 1;
        if ((rc = dist_allreduce(h, h->d_tails, n_t, ncclFloat64, ncclMax))) return rc;
        bossx_fhat_desc fh{up->fhat_c, up->n_windows, 20, up->target_rs, h->n_sites_all / kWindow};
        if (fhat_resident && (rc = bossx_fhat_build(h, up->n_windows, up->target_rs, up->fhat_alpha, up->fhat_den, up->fhat_expected, up->fhat_on_target))) return rc;
        if ((rc = bossx_dist_hist(h, &fh))) return rc;
        if ((rc = dist_allreduce(h, h->d_limbs, size_t(BOSSX_HIST_BINS + 1) * 5, ncclInt64, ncclSum))) return rc;
        if ((rc = bossx_dist_pick(h, up->tc))) return rc;
    }
    rc = dist_finish_impl(h, strat_all, contig_on, res, (up->flags & BOSSX_UPDATE_STRAT_BITS) != 0, mode);
    if (rc) return rc;
    if (mode == 1) { h->upd_launched = true; return BOSSX_OK; }
    if (res->any_on) h->dist_armed = true;
    if (!have_strategy_inputs) res->updated = 0;
    return BOSSX_OK;
}
}  // namespace

int bossx_dist_update(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                      bossx_update_result *res) {
    if (h && h->upd_launched) return fail(h, BOSSX_E_INVALID, "bossx_dist_update while a launched update has not been collected");
    return dist_update_run(h, up, strat_all, contig_on, res, 0);
}

int bossx_dist_update_launch(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                             bossx_update_result *res) {
    if (h && h->upd_launched) return fail(h, BOSSX_E_INVALID, "an update is already launched: collect it first");
    if (h) h->upd_done = false;
    int rc = dist_update_run(h, up, strat_all, contig_on, res, 1);
    if (rc == BOSSX_OK && h && !h->upd_launched) h->upd_done = true;
    if (rc != BOSSX_OK && h) { h->upd_launched = false; h->upd_done = false; }
    return rc;
}

int bossx_dist_update_collect(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                              bossx_update_result *res) {
    if (!h) return BOSSX_E_INVALID;
    if (h->upd_done) { h->upd_done = false; return BOSSX_OK; }
    if (!h->upd_launched) return fail(h, BOSSX_E_INVALID, "no launched update to collect");
    h->upd_launched = false;
    return dist_update_run(h, up, strat_all, contig_on, res, 2);
}

int bossx_host_alloc(bossx_engine *h, size_t bytes, void **ptr) {
    if (!h || !ptr) return fail(h, BOSSX_E_INVALID, "bad host_alloc call");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        g_host_blocks.emplace_back(static_cast<uint8_t *>(*ptr), bytes);
    }
    return BOSSX_OK;
}

int bossx_host_free(void *ptr) {
    if (!ptr) return BOSSX_OK;
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        for (size_t i = 0; i < g_host_blocks.size(); ++i)
            if (g_host_blocks[i].first == ptr) { g_host_blocks.erase(g_host_blocks.begin() + long(i)); break; }
    }
    return hipHostFree(ptr) == hipSuccess ? BOSSX_OK : BOSSX_E_HIP;
}

int bossx_set_overlap(bossx_engine *h, int32_t on) {
    if (!h) return BOSSX_E_INVALID;
    h->overlap_ok = on != 0 && getenv("BOSSX_NO_OVERLAP") == nullptr;
    return BOSSX_OK;
}

int bossx_arm(bossx_engine *h) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "bad arm call");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int32_t one = 1;
    HIPCHK(hipMemcpyAsync(&h->d_ctrl->any_on, &one, sizeof(one), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->host_armed = true;
    return BOSSX_OK;
}

int bossx_get_max(bossx_engine *h, double *max_benefit) {
    if (!h || !h->finalized || !max_benefit) return fail(h, BOSSX_E_INVALID, "bad get_max call");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, &h->d_ctrl->max_bits, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(max_benefit, &bits, sizeof(double));
    return BOSSX_OK;
}

int bossx_update_benefit(bossx_engine *h, const int32_t *windows, const double *mult) {
    if (!h || !h->finalized || !windows || !mult) return fail(h, BOSSX_E_INVALID, "bad update_benefit call");
    HIPCHK(hipSetDevice(h->cfg.device));
    ChainParams CP;
    size_t lds = 0;
    int rc = fill_chain_params(h, windows, mult, CP, lds);
    if (rc) return rc;
    CP.gate = 1;
    CP.zero_stats = h->d_stats_rep; CP.n_zero = int32_t(kStatRepWords);    // for the histogram of bossx_update
    if (h->overlap_ok && h->sweep_published && h->sweep_in_flight) {
        // The strategy is switched on (the gate is known to be open) and this update's sweep is in
        // flight on the main stream: run the chain NEXT TO it on stream2.  The sweep hands tiles
        // out from both contig ends and publishes each tile's bin sums (tile_done == epoch); the
        // chain's prefetch wave waits for the tiles of a chunk before reading it.
        HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_begin, 0));
        if (!h->max_bits_clear) HIPCHK(hipMemsetAsync(&h->d_ctrl->max_bits, 0, sizeof(unsigned long long), h->stream2));
        CP.tile_done = h->d_tile_done; CP.epoch = h->epoch;
        if (getenv("BOSSX_OVERLAP_SELFTEST")) { CP.never_ready = 1; CP.wait_ticks = 500000; }   // never satisfied: exercises the time-out path
        launch_chain(h, CP, lds, h->stream2);
        HIPCHK(hipEventRecord(h->ev_chain, h->stream2));
        h->chain_on_stream2 = true;
        h->last_chain = CP; h->last_chain_lds = lds;
    } else {
        if (!h->max_bits_clear) HIPCHK(hipMemsetAsync(&h->d_ctrl->max_bits, 0, sizeof(unsigned long long), h->stream));
        launch_chain(h, CP, lds);
    }
    h->max_bits_clear = false;
    HIPCHK(hipGetLastError());
    return BOSSX_OK;
}

// One whole decision update enqueued back to back (no host round trip between kernels):
// sweep -> bucket switches -> benefit chain -> threshold statistics -> threshold choice ->
// masks, then one device-to-host copy of all masks and the control block.
namespace {
// mode 0: the whole update; 1: enqueue it and return (bossx_update_launch); 2: wait for what mode 1 enqueued and
// read the results (bossx_update_collect).
int update_run(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
               bossx_update_result *res, int64_t *counts, uint64_t *fgrid_fx, uint64_t *ubar0_fx, int mode) {
    if (!h || !h->finalized || !up || !res) return fail(h, BOSSX_E_INVALID, "bad update call");
    if (!h->lut_set) return fail(h, BOSSX_E_INVALID, "update before set_lut");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc;
    if (mode == 1 && h->chain_on_stream2) mode = 0;       // a chain next to the sweep has its own retry: no split
    if (mode == 1) h->upd_done = false;
    if (mode == 0 && h->upd_launched) return fail(h, BOSSX_E_INVALID, "bossx_update while a launched update has not been collected");
    bossx_fhat_desc fh{up->fhat_c, up->n_windows, 20, up->target_rs, h->n_sites_all / kWindow};
    ChainParams CP;
    size_t lds = 0;
    const bool fhat_resident = (up->flags & BOSSX_UPDATE_FHAT_RESIDENT) != 0 && up->fhat_c == nullptr;
    const bool have_strategy_inputs = up->fhat_c != nullptr || fhat_resident;
    if (have_strategy_inputs) {
        if ((rc = fill_chain_params(h, up->windows, up->mult, CP, lds))) return rc;
    }
    if (mode != 2 && !(up->flags & BOSSX_UPDATE_SWEEP_DONE)) {
        if ((rc = launch_sweep(h))) return rc;
        launch_buckets(h, up->bucket_threshold);
    }
    h->sweep_in_flight = false;
    const size_t need = h->result_bytes + 16 + (counts ? kStatWords * 8 : 0);
    if ((rc = ensure_pin(h, need + 64))) return rc;
    char *pin = static_cast<char *>(h->h_pin);
    Ctrl *hc = reinterpret_cast<Ctrl *>(pin);
    int32_t *herr = reinterpret_cast<int32_t *>(pin + sizeof(Ctrl));
    uint8_t *hon = reinterpret_cast<uint8_t *>(pin + sizeof(Ctrl) + 16);
    unsigned long long *hst = reinterpret_cast<unsigned long long *>(pin + ((h->result_bytes + 15) & ~size_t(15)));
    bool chain_done = (up->flags & BOSSX_UPDATE_BENEFIT_DONE) != 0;
    bool mirrored = false;
    if (mode == 2) { chain_done = true; mirrored = h->upd_mirrored; }
    // A chain that ran next to the sweep lives on stream2: the rest of the update follows it IN THAT
    // QUEUE (no cross-queue signal between the chain and the histogram); the sweep and the bucket
    // switches it also depends on finished long ago (ev_sweep).
    hipStream_t const main_stream = h->stream;
    struct Restore { bossx_engine *h; hipStream_t s; ~Restore() { h->stream = s; } } restore{h, main_stream};
    // f-hat goes up on the (idle) main stream right away, while the chain still runs
    const FhatModel fm{up->n_windows, up->target_rs, up->fhat_alpha, up->fhat_den, up->fhat_expected, up->fhat_on_target};
    if (mode != 2 && have_strategy_inputs && (rc = fhat_resident ? build_fhat(h, &fm) : upload_fhat(h, &fh))) return rc;
    if (mode != 2 && have_strategy_inputs && chain_done && h->chain_on_stream2 && (up->flags & BOSSX_UPDATE_SWEEP_DONE)) {
        // ev_fhat was recorded on the main stream behind the sweep and the bucket switches of this
        // update, so it covers them as well: one (long signalled) cross-queue dependency
        HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fhat, 0));
        h->stream = h->stream2;
        h->chain_on_stream2 = false;        // same queue now: ordered behind the chain
    }
    for (int attempt = 0;; ++attempt) {
        const bool enqueue = !(mode == 2 && attempt == 0);      // (collect: attempt 0 was enqueued by the launch call)
        if (enqueue && have_strategy_inputs) {
            if (!chain_done) {
                if (!h->max_bits_clear) HIPCHK(hipMemsetAsync(&h->d_ctrl->max_bits, 0, sizeof(unsigned long long), h->stream));
                h->max_bits_clear = false;
                CP.gate = 1;
                CP.zero_stats = h->d_stats_rep; CP.n_zero = int32_t(kStatRepWords);
                launch_chain(h, CP, lds);
            } else if (h->chain_on_stream2) {
                HIPCHK(hipStreamWaitEvent(h->stream, h->ev_chain, 0));     // the chain ran next to the sweep
            }
            h->chain_on_stream2 = false;
            // the chain kernel cleared the statistics; the mask kernel picks the threshold itself
            if ((rc = launch_hist(h, &fh, 1, /*clear=*/false))) return rc;
            if (counts) launch_fold(h, false, /*to_canon=*/true, 1);      // (a caller that wants the statistics: the (lo, hi) form in d_stats)
            PickParams PP{};
            PP.limbs = reinterpret_cast<const long long *>(h->d_stats_rep); PP.n_rep = kHistRep; PP.ctrl = h->d_ctrl; PP.tc = up->tc; PP.gate = 1;
            // the mask kernel's block 0 writes the result block into the pinned buffer itself; the
            // sentinel tells whether it got that far (it returns early while nothing is switched on)
            *herr = kNoResult;
            // small references: the kernel mirrors its mask bytes into the caller's buffer when that is
            // device-writable (from bossx_host_alloc) and holds the byte form — no copy afterwards
            uint8_t *mirror = nullptr;
            // (round 6: up to 8 MB — the vector stores of the mask kernel post 2.2 MB of chr20+21 masks across PCIe in ~40 us,
            // less than the copy engine's submission alone used to cost, and no copy-engine submission is left in a lone update)
            static const int64_t mirror_max = getenv("BOSSX_MIRROR_MAX") ? atoll(getenv("BOSSX_MIRROR_MAX")) : (int64_t(8) << 20);
            // (one barcode only beyond 1 MB: there the kernel stores whole 16-byte vectors; several barcodes go byte by byte — fine into HBM, not across PCIe)
            if (strat_all && !(up->flags & BOSSX_UPDATE_STRAT_BITS) && h->strat_bytes <= (h->nb == 1 ? mirror_max : std::min<int64_t>(mirror_max, int64_t(1) << 20))) {
                std::lock_guard<std::mutex> lock(g_host_mutex);
                for (const auto &blk : g_host_blocks)
                    if (strat_all >= blk.first && strat_all + h->strat_bytes <= blk.first + blk.second) mirror = strat_all;
            }
            mirrored = mirror != nullptr;
            if ((rc = launch_mask(h, 1, false, &PP, reinterpret_cast<unsigned long long *>(pin), mirror, /*use_codes=*/true, (up->flags & BOSSX_UPDATE_STRAT_DELTA) != 0))) return rc;
        }
        HIPCHK(hipGetLastError());
        // results
        if (enqueue) {
            if (!have_strategy_inputs)
                HIPCHK(hipMemcpyAsync(pin, h->d_result, h->result_bytes, hipMemcpyDeviceToHost, h->stream));
            if (counts) HIPCHK(hipMemcpyAsync(hst, h->d_stats, kStatWords * 8, hipMemcpyDeviceToHost, h->stream));
            if (strat_all && have_strategy_inputs && !mirrored && (rc = copy_masks(h, strat_all, (up->flags & BOSSX_UPDATE_STRAT_BITS) != 0))) return rc;
        }
        if (mode == 1) {            // everything is in the queue: the caller does other work and collects later
            h->upd_launched = true; h->upd_mirrored = mirrored;
            return BOSSX_OK;
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        if (have_strategy_inputs && *herr == kNoResult) {      // the kernel left early: fetch the block the ordinary way
            HIPCHK(hipMemcpyAsync(pin, h->d_result, h->result_bytes, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        if (have_strategy_inputs && (hc->err & 4) && attempt == 0) {
            // The concurrent chain gave up waiting for the sweep (kernels serialised by a profiler):
            // rerun it after the sweep, and stay serial from now on.
            const int32_t cleared = 0;      // also drops the pick kernel's 'empty' flag of the aborted attempt
            HIPCHK(hipMemcpyAsync(&h->d_ctrl->err, &cleared, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (h->last_chain_live) h->overlap_ok = false; else h->chain_flow = false;
            chain_done = false;
            HIPCHK(hipStreamSynchronize(main_stream));
            h->stream = main_stream;       // rerun on the main stream, after the sweep
            continue;
        }
        break;
    }
    if (hc->any_on) h->host_armed = true;
    if (have_strategy_inputs) note_spec_result(h, herr);
    if (*herr) {
        HIPCHK(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
        return fail(h, BOSSX_E_RANGE, "a read contains a base other than A/C/G/T inside an aligned segment");
    }
    if (contig_on) {
        for (size_t i = 0; i < h->contigs.size(); ++i) contig_on[i] = 0;
        for (size_t k = 0; k < h->filt.size(); ++k) contig_on[size_t(h->filt[k])] = hon[k];
    }
    res->updated = hc->any_on && have_strategy_inputs;
    res->any_on = hc->any_on;
    res->strat_size = hc->strat_size; res->n_bins = hc->n_bins;
    res->threshold = hc->threshold; res->ubar0 = hc->ubar0;
    res->argmax_margin = hc->argmax_margin; res->thr_code = hc->thr_code;
    memcpy(&res->normaliser, &hc->max_bits, sizeof(double));
    if (res->updated && (hc->err & 2)) {
        HIPCHK(hipMemsetAsync(&h->d_ctrl->err, 0, sizeof(int32_t), h->stream));
        return fail(h, BOSSX_E_EMPTY, "no non-zero benefit (np.max of an empty array)");
    }
    if (counts) {
        for (int i = 0; i < BOSSX_HIST_BINS; ++i) counts[i] = int64_t(hst[i]);
        if (fgrid_fx) memcpy(fgrid_fx, hst + BOSSX_HIST_BINS, size_t(BOSSX_HIST_BINS) * 2 * sizeof(uint64_t));
        if (ubar0_fx) memcpy(ubar0_fx, hst + BOSSX_HIST_BINS * 3, 2 * sizeof(uint64_t));
    }
    return BOSSX_OK;
}
}  // namespace

int bossx_update(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                 bossx_update_result *res, int64_t *counts, uint64_t *fgrid_fx, uint64_t *ubar0_fx) {
    return update_run(h, up, strat_all, contig_on, res, counts, fgrid_fx, ubar0_fx, 0);
}

int bossx_update_launch(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                        bossx_update_result *res, int64_t *counts, uint64_t *fgrid_fx, uint64_t *ubar0_fx) {
    if (h && h->upd_launched) return fail(h, BOSSX_E_INVALID, "an update is already launched: collect it first");
    int rc = update_run(h, up, strat_all, contig_on, res, counts, fgrid_fx, ubar0_fx, 1);
    if (rc == BOSSX_OK && h && !h->upd_launched) h->upd_done = true;       // (ran to the end after all: nothing left to collect)
    if (rc != BOSSX_OK && h) { h->upd_launched = false; h->upd_done = false; }
    return rc;
}

int bossx_update_collect(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                         bossx_update_result *res, int64_t *counts, uint64_t *fgrid_fx, uint64_t *ubar0_fx) {
    if (!h) return BOSSX_E_INVALID;
    if (h->upd_done) { h->upd_done = false; return BOSSX_OK; }          // the launch call already filled everything
    if (!h->upd_launched) return fail(h, BOSSX_E_INVALID, "no launched update to collect");
    h->upd_launched = false;
    return update_run(h, up, strat_all, contig_on, res, counts, fgrid_fx, ubar0_fx, 2);
}

int bossx_get_strat(bossx_engine *h, int32_t contig, uint8_t *dst) {
    int rc = check_contig(h, contig, false);
    if (rc) return rc;
    const ContigInfo &c = h->contigs[size_t(contig)];
    if (c.rejected) { dst[0] = 0; return BOSSX_OK; }
    if (c.remote) return fail(h, BOSSX_E_INVALID, "contig is remote (owned by another device)");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(dst, h->d_strat + c.strat_off, size_t(c.T * 2 * h->nb), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return BOSSX_OK;
}

int bossx_get_strat_bits(bossx_engine *h, uint8_t *dst) {
    if (!h || !h->finalized || !dst) return fail(h, BOSSX_E_INVALID, "bad get_strat_bits call");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = copy_masks(h, dst, true);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return BOSSX_OK;
}

int32_t bossx_n_contigs(const bossx_engine *h) { return h ? int32_t(h->contigs.size()) : 0; }
int64_t bossx_contig_length(const bossx_engine *h, int32_t c) {
    return (h && c >= 0 && c < int32_t(h->contigs.size())) ? h->contigs[size_t(c)].length : -1;
}
int64_t bossx_n_sites(const bossx_engine *h) { return h ? h->n_sites_all : 0; }
int64_t bossx_merged_bins(const bossx_engine *h) { return h ? h->B : 0; }
int32_t bossx_matrix_chain(const bossx_engine *h) { return (h && h->matrix_chain) ? 1 : 0; }
int64_t bossx_strat_bytes(const bossx_engine *h) { return h ? h->strat_bytes : 0; }
int64_t bossx_strat_bits_bytes(const bossx_engine *h) { return h ? (h->strat_bytes + 7) / 8 : 0; }
int64_t bossx_strat_offset(const bossx_engine *h, int32_t c) {
    if (!h || c < 0 || c >= int32_t(h->contigs.size()) || h->contigs[size_t(c)].rejected || h->contigs[size_t(c)].remote) return -1;
    return h->contigs[size_t(c)].strat_off;
}

int bossx_export(bossx_engine *h, int32_t contig, int32_t which, void *dst, size_t dst_bytes) {
    int rc = check_contig(h, contig, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    { int jrc = settle_chain(h); if (jrc) return jrc; }
    if (h->pending_slot >= 0 && (rc = flush_pending(h))) return rc;   // make staged increments visible
    const ContigInfo &c = h->contigs[size_t(contig)];
    const int64_t L = c.length, nb = h->nb, nbin = c.T + 1;
    auto need = [&](size_t n) { return dst_bytes >= n; };
    switch (which) {
        case 0: {
            if (!need(size_t(nb * 5 * L) * 2)) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            // the counter planes are reference-relative on the device: un-rotated into the reference's A C G T deletion order
            if ((rc = convert_field(h, c, 0, dst, false))) return rc;
            break;
        }
        case 1: {
            if (!need(size_t(nb * L) * 8)) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            double *tmp = nullptr;
            if ((rc = dev_alloc(h, &tmp, size_t(nb * L)))) return rc;
            SweepParams P = sweep_params(h);
            const int64_t blocks = std::min<int64_t>((L + 255) / 256, 4096);
            hipLaunchKernelGGL(export_scores_kernel, dim3(uint32_t(blocks)), dim3(256), 0, h->stream, P, c.filt_index, tmp);
            hipError_t e1 = hipMemcpyAsync(dst, tmp, size_t(nb * L) * 8, hipMemcpyDeviceToHost, h->stream);
            hipError_t e2 = hipStreamSynchronize(h->stream);
            hipFree(tmp);
            HIPCHK(e1); HIPCHK(e2);
            break;
        }
        case 2: {
            if (!h->d_entropy) return fail(h, BOSSX_E_INVALID, "entropy tracking is off");
            if (!need(size_t(nb * L) * 8)) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            if ((rc = convert_field(h, c, 2, dst, false))) return rc;
            break;
        }
        case 3: {
            if (!need(size_t(nb * nbin) * 8)) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            for (int64_t b = 0; b < nb; ++b)
                HIPCHK(hipMemcpyAsync(static_cast<double *>(dst) + b * nbin, h->d_ds + b * h->B + c.bin_off,
                                      size_t(nbin) * 8, hipMemcpyDeviceToHost, h->stream));
            break;
        }
        case 4: {
            if (!need(size_t(nb * 2 * nbin) * 8)) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            for (int64_t p = 0; p < nb * 2; ++p)
                HIPCHK(hipMemcpyAsync(static_cast<double *>(dst) + p * nbin, h->d_benefit + p * h->B + c.bin_off,
                                      size_t(nbin) * 8, hipMemcpyDeviceToHost, h->stream));
            break;
        }
        case 5: {
            if (!need(size_t(nb * L))) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int64_t b = 0; b < nb; ++b)
                if ((rc = copy_site_field(h, c, int32_t(b), kTileMetaOff, 1, static_cast<uint8_t *>(dst) + b * L, false))) return rc;
            break;
        }
        case 6: {
            if (!need(size_t(L))) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            HIPCHK(hipMemcpyAsync(dst, h->d_touched + c.site_off, size_t(L), hipMemcpyDeviceToHost, h->stream));
            break;
        }
        case 8: {   // last min(nbin, n_filt) rows of additional_benefit: halo rows other devices need
            const int64_t K = std::min<int64_t>(nbin, int64_t(h->filt.size()));
            if (!need(size_t(nb * 2 * K) * 8)) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            for (int64_t pl = 0; pl < nb * 2; ++pl)
                HIPCHK(hipMemcpyAsync(static_cast<double *>(dst) + pl * K, h->d_benefit + pl * h->B + c.bin_off + nbin - K,
                                      size_t(K) * 8, hipMemcpyDeviceToHost, h->stream));
            break;
        }
        case 7: {
            if (!need(size_t(nb * c.n_buckets))) return fail(h, BOSSX_E_INVALID, "export buffer too small");
            for (int64_t b = 0; b < nb; ++b)
                HIPCHK(hipMemcpyAsync(static_cast<uint8_t *>(dst) + b * c.n_buckets, h->d_bucket_on + b * h->NBK + c.bucket_off,
                                      size_t(c.n_buckets), hipMemcpyDeviceToHost, h->stream));
            break;
        }
        default:
            return fail(h, BOSSX_E_INVALID, "unknown export selector");
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return BOSSX_OK;
}

int bossx_import(bossx_engine *h, int32_t contig, int32_t which, const void *src, size_t src_bytes) {
    int rc = check_contig(h, contig, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->pending_slot >= 0 && (rc = flush_pending(h))) return rc;
    ContigInfo &c = h->contigs[size_t(contig)];
    const int64_t L = c.length, nb = h->nb;
    h->full_sweep_needed = true;            // state changed behind the sweep's back: every tile again
    switch (which) {
        case 0: {
            if (src_bytes != size_t(nb * 5 * L) * 2) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            // A C G T deletion planes -> the reference-relative planes of the site state (the state bytes hold the reference bases)
            if ((rc = convert_field(h, c, 0, const_cast<void *>(src), true))) return rc;
            unsigned long long *d_tot = h->d_stats;
            HIPCHK(hipMemsetAsync(d_tot, 0, sizeof(unsigned long long), h->stream));
            hipLaunchKernelGGL(contig_total_kernel, dim3(1024), dim3(256), 0, h->stream, SiteState{h->d_state, h->nb}, h->nb,
                               c.site_off, L, d_tot);
            unsigned long long tot = 0;
            HIPCHK(hipMemcpyAsync(&tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            c.cov_total = tot;
            break;
        }
        case 2: {
            if (!h->d_entropy) return fail(h, BOSSX_E_INVALID, "entropy tracking is off");
            if (src_bytes != size_t(nb * L) * 8) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            if ((rc = convert_field(h, c, 2, const_cast<void *>(src), true))) return rc;
            break;
        }
        case 3: {
            // the downsampled scores (Contig.scores_ds, reference.py:225-231) as the input of bossx_benefit: a parity hook for the
            // move_sum chain alone (tests/golden/g_movesum.npz).  The next sweep overwrites them.
            const int64_t nbin = L / kWindow + 1;
            if (src_bytes != size_t(nb * nbin) * 8) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int64_t b = 0; b < nb; ++b)
                HIPCHK(hipMemcpy(h->d_ds + b * h->B + c.bin_off, static_cast<const double *>(src) + b * nbin, size_t(nbin) * 8,
                                 hipMemcpyHostToDevice));
            // (bin sums written behind the sweep's back: every tile of the contig counts as rewritten — chain_candidates_kernel's stamps)
            if ((rc = stamp_contig_tiles(h, c))) return rc;
            break;
        }
        case 5: {
            if (src_bytes != size_t(nb * L)) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int64_t b = 0; b < nb; ++b)
                if ((rc = copy_site_field(h, c, int32_t(b), kTileMetaOff, 1,
                                          const_cast<uint8_t *>(static_cast<const uint8_t *>(src) + b * L), true))) return rc;
            break;
        }
        case 6: {
            if (src_bytes != size_t(L)) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            HIPCHK(hipMemcpy(h->d_touched + c.site_off, src, size_t(L), hipMemcpyHostToDevice));
            h->touched_dirty = true;
            break;
        }
        case 7: {
            if (src_bytes != size_t(nb * c.n_buckets)) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            bool any = false;
            for (size_t i = 0; i < src_bytes; ++i) any = any || static_cast<const uint8_t *>(src)[i];
            for (int64_t b = 0; b < nb; ++b)
                HIPCHK(hipMemcpy(h->d_bucket_on + b * h->NBK + c.bucket_off, static_cast<const uint8_t *>(src) + b * c.n_buckets,
                                 size_t(c.n_buckets), hipMemcpyHostToDevice));
            if (any) {
                const uint8_t one = 1;
                const int32_t one32 = 1;
                HIPCHK(hipMemcpy(h->d_contig_on + c.filt_index, &one, 1, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(&h->d_ctrl->any_on, &one32, sizeof(one32), hipMemcpyHostToDevice));
            }
            break;
        }
        case 9: {
            if (src_bytes != size_t(c.T * 2 * nb)) return fail(h, BOSSX_E_INVALID, "import size mismatch");
            HIPCHK(hipMemcpy(h->d_strat + c.strat_off, src, src_bytes, hipMemcpyHostToDevice));
            h->mirror_valid = nullptr;
            break;
        }
        default:
            return fail(h, BOSSX_E_INVALID, "unknown import selector");
    }
    return BOSSX_OK;
}

int bossx_preload_coverage(bossx_engine *h, double depth, uint64_t seed) {
    if (!h || !h->finalized) return fail(h, BOSSX_E_INVALID, "preload before finalize");
    if (!(depth >= 0) || depth > 60) return fail(h, BOSSX_E_INVALID, "depth must be in [0, 60]");
    HIPCHK(hipSetDevice(h->cfg.device));
    for (int32_t fi : h->filt) {
        ContigInfo &c = h->contigs[size_t(fi)];
        if (c.remote) continue;
        const int64_t blocks = std::min<int64_t>((c.length + 255) / 256, 8192);
        hipLaunchKernelGGL(preload_kernel, dim3(uint32_t(blocks)), dim3(256), 0, h->stream, SiteState{h->d_state, h->nb},
                           h->d_touched, h->nb, c.site_off, c.length, depth, seed, ent_save_of(h));
        unsigned long long *d_tot = h->d_stats;
        HIPCHK(hipMemsetAsync(d_tot, 0, sizeof(unsigned long long), h->stream));
        hipLaunchKernelGGL(contig_total_kernel, dim3(1024), dim3(256), 0, h->stream, SiteState{h->d_state, h->nb}, h->nb,
                           c.site_off, c.length, d_tot);
        unsigned long long tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        c.cov_total = tot;
    }
    HIPCHK(hipGetLastError());
    h->touched_dirty = true;
    h->full_sweep_needed = true;
    return BOSSX_OK;
}

int bossx_enable_timing(bossx_engine *h, int32_t on) {
    if (!h) return BOSSX_E_INVALID;
    h->timing = on != 0;
    h->timing_only = on >= 2 ? on - 2 : -1;      // 2 + k: events around kernel k alone (BOSSX_K_*)
    if (h->timing_only >= BOSSX_K_COUNT) h->timing_only = -1;
    return BOSSX_OK;
}

int bossx_kernel_ms(bossx_engine *h, float *ms_last, double *ms_total, int64_t *launches) {
    if (!h) return BOSSX_E_INVALID;
    time_collect(h);
    for (int k = 0; k < BOSSX_K_COUNT; ++k) {
        if (ms_last) ms_last[k] = h->ms_last[k];
        if (ms_total) ms_total[k] = h->ms_total[k];
        if (launches) launches[k] = h->launches[k];
    }
    return BOSSX_OK;
}

int bossx_kernel_bytes(bossx_engine *h, double *bytes_last) {
    if (!h || !bytes_last) return BOSSX_E_INVALID;
    if (h->timing && h->finalized && h->d_work_ctr) {
        // the last sweep's counter write-back: 16 bytes per vector it actually wrote
        HIPCHK(hipSetDevice(h->cfg.device));
        time_collect(h);
        unsigned long long wb[2] = {0, 0};
        HIPCHK(hipMemcpy(wb, h->d_work_ctr + h->n_work_ctr, sizeof(wb), hipMemcpyDeviceToHost));
        // 16 bytes per counter vector written back, 8 per entropy value written (its table entry, like a score's, comes from
        // the L2-resident table and is not counted; round 4 first counted 16 here — and quoted more bytes than the PMC
        // counters saw cross the HBM interface)
        h->bytes_last[BOSSX_K_SWEEP] = h->sweep_bytes_base + 16.0 * double(wb[0]) + 8.0 * double(wb[1]);
    }
    for (int k = 0; k < BOSSX_K_COUNT; ++k) bytes_last[k] = h->bytes_last[k];
    return BOSSX_OK;
}

int bossx_synchronize(bossx_engine *h) {
    if (!h) return BOSSX_E_INVALID;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
    return BOSSX_OK;
}

}  // extern "C"
