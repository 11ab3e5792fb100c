#!/usr/bin/env python3
"""Benchmark of the BOSS-RUNS decision update on MI355X (BASELINE.json metric:
"decision-update wall-clock (ms) + Mbp scored/sec, 4000-read batch").

One step = ONE decision update as SURVEY.md §8(d) defines it: from "PAF batch text + read
sequences in host memory" to "all contig masks in host memory" — the read-length step
(`rl_dist.update`, boss/core.py:106) and `BossRuns.process_batch_paf` (= process_batch_runs
minus the mapper call, boss/runs/core.py:214-224) on a FRESH 4000-read batch every step: native
PAF/CIGAR front end, upload, fused per-site sweep with coverage ingestion, bucket switches, exact
move_sum benefit chain, threshold statistics and choice, masks, device-to-host copy.  The npz
write is not part of the step (SURVEY §8d reports it separately).

`value` = reference positions whose score state is brought up to date per second
(G * nbarcodes / t_update), summed over ranks; `ms_per_step` = t_update.  The same K updates on
batches already parsed and resident in HBM are reported beside it as `kernels_only_ms`.

    python bench.py                               # N=1, chr20+chr21 110 Mb ploidy 2 (BASELINE configs[2])
    python bench.py --workload ecoli|barcoded|shard390|grch38
    python bench.py --gpus N                      # starts its own N ranks (one per GPU, RCCL) and relays rank 0's line
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N     # the same under an outer launcher
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GPU_CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md: peak engine clock
CHAIN_FLOOR_CYCLES = 5.1    # one dependent FP64 matrix op (20.4 cycles with accumulate forwarding) per 4 recurrence steps: scripts/mfma_chain_floor.hip

WORKLOADS = {
    # name: (contig lengths, names, ploidy, nbarcodes, reject, preload depth)
    "ecoli": ([4_641_652], ["ecoli_K12"], 1, 1, None, 0.0),
    "chr20_21": ([64_444_167, 46_709_983, 16_569], ["chr20", "chr21", "MT"], 2, 1, "MT", 8.0),
    "shard390": ([248_956_422, 138_394_717], ["chr1", "chr9"], 2, 1, None, 8.0),
    "barcoded": ([5_000_000] * 10, ["bac%02d" % i for i in range(10)], 1, 8, None, 0.0),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # (0.25 s of timed region at chr20+21; the driver passes its own K)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS) + ["grch38"])
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true",
                    help="skip the extra sweep-kernel measurement on a 200 Mb contig (HBM-sized working set)")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the short runs of the other single-GPU workloads (`workloads` object)")
    ap.add_argument("--prepare-only", action="store_true",
                    help="generate the batches (into BOSSX_BATCH_CACHE) and exit: no GPU is touched")
    ap.add_argument("--no-entropy", action="store_true",
                    help="do not keep the reference's per-site entropy array current (it is dead state on the strategy path; the "
                         "reference maintains it every update, sequences.py:443,450, and so does the headline)")
    ap.add_argument("--no-cold", action="store_true", help="skip cold_update_ms (ONE update behind 2 s / 10 s of an idle GPU: 20 s of sleeping)")
    ap.add_argument("--no-late", action="store_true", help="skip late_regime (the lone loop once more from ~28x coverage)")
    ap.add_argument("--no-entropy-off-run", action="store_true",
                    help="skip the short second run without the entropy array (`track_entropy_false` object)")
    a = ap.parse_args()
    a.track_entropy = not a.no_entropy
    return a


# ---- synthetic inputs (generated before the GPU is touched: worker processes are forked) -------
_GEN = {}


def _gen_one(job):
    from boss_runs_amd import synth
    key, seed, n_reads, nb = job
    b = synth.make_batch(_GEN[key], n_reads, seed=seed, nbarcodes=nb, extras=False)
    b["read_lengths_arr"] = np.fromiter(b["read_lengths"].values(), dtype=np.int64, count=len(b["read_lengths"]))
    del b["read_lengths"]
    return b


def generate_batches(jobs):
    """jobs: list of (reference key, seed, n_reads, nbarcodes) -> list of batches, in order.
    BOSSX_BATCH_CACHE=<dir>: batches are kept there as pickles and reused — the profiling script
    fills it in a run of its own, so that no process is forked under rocprofv3 (a pool child's
    exit handler inside the profiler's preloaded tool has hung a counter pass for 46 minutes)."""
    import multiprocessing as mp
    import pickle
    cache = os.environ.get("BOSSX_BATCH_CACHE")
    paths = [os.path.join(cache, "batch_%s_%d_%d_%d.pkl" % j) for j in jobs] if cache else []
    if cache and all(os.path.exists(p) for p in paths):
        return [pickle.load(open(p, "rb")) for p in paths]
    n = min(len(jobs), max(1, min(int(os.environ.get("BOSSX_GEN_PROCS", "32")), (os.cpu_count() or 1))))
    if n <= 1:
        out = [_gen_one(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(n) as pool:
            out = pool.map(_gen_one, jobs, chunksize=1)
    if cache:
        os.makedirs(cache, exist_ok=True)
        for p, b in zip(paths, out):
            pickle.dump(b, open(p, "wb"), protocol=4)
    return out


def make_reference(workload, rank):
    from boss_runs_amd import synth
    lens, names = WORKLOADS[workload][0], WORKLOADS[workload][1]
    return synth.make_reference(lens, seed=1 + rank, names=["%s_r%d" % (n, rank) for n in names])


def make_runs(workload, mine, rank, world, device, track_entropy, preload_override=None):
    """N=1: the fused single-GPU `BossRuns`.  N>1: `DistributedBossRuns`; the global reference is
    the per-GPU contig set repeated once per rank (weak scaling), contig-partitioned so that
    every rank owns its own copy, with ONE global threshold per update (the collectives of
    boss_runs_amd/parallel.py are inside the timed region)."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from boss_runs_amd.parallel import DistributedBossRuns
    lens, names, ploidy, nb, reject, preload = WORKLOADS[workload]
    args = BossConfig()
    args.general.name = "bench_%s_r%d" % (workload, rank)
    args.optional.ploidy = ploidy
    args.optional.bucket_threshold = 0        # strategies on from the first update
    args.gpu.device = device
    args.gpu.track_entropy = bool(track_entropy)
    if reject:       # BASELINE configs[2]: reject_refs=MT (the 16.5-kb MT is dropped by the 100-kb filter first, reference.py:330-337)
        args.optional.reject_refs = ",".join("%s_r%d" % (reject, r) for r in range(world))
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    if world == 1 and not os.environ.get("BOSSX_FORCE_COLLECTIVES"):
        runs = BossRuns(args)
        runs.init(contigs=[(n, synth.codes_to_str(c)) for n, c in mine])
    else:
        allc = []
        for r in range(world):
            for n, L in zip(names, lens):
                allc.append(("%s_r%d" % (n, r), L))
        mine_d = {n: synth.codes_to_str(c) for n, c in mine}
        runs = DistributedBossRuns(args)
        # masks stay on their owner rank in the timed loop (the all-gather to rank 0 is only
        # needed when boss.npz is written; halo rows are still exchanged every update)
        runs.init(contigs=[(n, mine_d.get(n, L)) for n, L in allc], sharded_reads=True, gather_masks=False)
        assert all(not runs.contigs[n].remote for n in mine_d if n in runs.contigs), "partition must give each rank its own contigs"
    runs.write_masks = False                  # npz write is reported separately
    runs.log_fractions = False
    if preload_override is not None:
        preload = preload_override
    if preload > 0:
        runs.engine.preload_coverage(preload, seed=7 + rank)
    return runs, nb


def pmc_traffic(workload, kernel_substr, only=None):
    """HBM bytes per launch of `kernel_substr` from the newest committed rocprofv3 PMC summary
    of this workload (profiles/rNN_<workload>_rocprof_summary.json: FETCH_SIZE and WRITE_SIZE
    from separate --pmc passes, in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).
    Returns (bytes, file name, commit the profile was taken at, hash of the kernel sources then)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s_rocprof_summary.json" % workload)))
    if not files:
        return None, None, None, None
    d = json.load(open(files[-1]))
    tot, found = 0.0, False
    _PMC_EXTRA.clear()
    for name, v in d.get("kernels", {}).items():
        if only is not None and only not in name:
            continue
        if kernel_substr in name and "FETCH_SIZE_avg_per_launch" in v and "WRITE_SIZE_avg_per_launch" in v:
            # launches of the variants per update differ (plain + ingest): weight by launches per update
            per_update = v.get("launches_per_update", 1.0)
            tot += (2.0 * v["FETCH_SIZE_avg_per_launch"] + v["WRITE_SIZE_avg_per_launch"]) * 1024.0 * per_update
            found = True
            if v.get("avg_ns_timed_region"):      # the trace's launches of the lone-update loop on their own (summarise_prof.py)
                _PMC_EXTRA["rocprof_avg_launch_ms_timed_region"] = v["avg_ns_timed_region"] / 1e6
                _PMC_EXTRA["rocprof_avg_launch_ms_all_loops"] = v["avg_ns"] / 1e6
    return (tot if found else None), os.path.basename(files[-1]), d.get("commit"), d.get("kernel_sources")


_PMC_EXTRA = {}


def kernel_sources_hash():
    """sha1 over the device side of the library (boss-runs_amd/csrc: the .hip file, its kernel includes,
    the shared header) with `//` comments and blank lines stripped: ties a committed rocprofv3 summary
    to the kernel CODE it measured, whatever was committed next to it afterwards (docs, tests, host
    code, comments)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(REPO, "boss-runs_amd", "csrc")
    for f in ("bossx.hip", "kernels.hip.inc", "front_end.hip.inc", "engine.hpp"):
        h.update(f.encode() + b"\0")
        with open(os.path.join(d, f), "r") as fh:
            for line in fh:
                code = line.split("//", 1)[0].rstrip()
                if code:
                    h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def current_commit():
    p = os.path.join(REPO, ".bossx_commit")       # written before a gpurun call (the GPU box has no .git)
    if os.path.exists(p):
        return open(p).read().strip()
    try:
        import subprocess
        return subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True,
                              text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def large_sweep(device, L=200_000_000, reps=5, depth=8.0):
    """The sweep kernel alone on one 200 Mb contig (working set 2.4 GB >> Infinity Cache):
    `stream` = fresh state (no LUT gathers), `gather` = every site scored (depth-8 preload)."""
    from boss_runs_amd.engine import Engine
    from boss_runs_amd.scoring import SiteScoring
    rng = np.random.default_rng(1)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L, dtype=np.uint8)].tobytes()
    e = Engine(nbarcodes=1, device=device, track_entropy=False)
    e.add_contig("big", seq)
    del seq
    hap = SiteScoring(1)
    e.finalize(hap.score0[0], hap.ent0[0])
    e.set_lut(*hap.tables())
    e.enable_timing(True)
    out = {"sites": L}
    os.environ["BOSSX_INCREMENTAL"] = "0"          # every tile, every time: this measures the full streaming sweep
    for tag in ("stream", "gather"):
        if tag == "gather":
            e.preload_coverage(depth, seed=3)
            e.sweep(); e.synchronize()          # first sweep after the preload touches every site
        ms = []
        for _ in range(reps):
            e.sweep(); e.synchronize()
            st = e.kernel_stats()["site_sweep"]
            ms.append(st["ms_last"])
        m = float(np.median(ms))
        out[tag] = {"ms": m, "algorithmic_bytes": st["bytes_last"], "achieved": st["bytes_last"] / 1e6 / m,
                    "frac": st["bytes_last"] / 1e6 / m / HBM_PEAK_GBS, "unit": "GB/s"}
    os.environ.pop("BOSSX_INCREMENTAL", None)
    e.close()
    return out


def entropy_off_run(workload, contigs, device, batches, warmup, steps):
    """The headline step once more on an engine that does not keep the entropy array (config.gpu.track_entropy = False):
    what the per-site entropy costs an update."""
    runs, nb = make_runs(workload, contigs, 0, 1, device, False)
    R = Runner(workload, runs, nb, batches, False)
    eng = runs.engine
    eng.enable_timing(True)            # BEFORE the warm-up: creating the engine's events cost the first timed update 8-16 ms (see main)
    for b in batches[:warmup]:
        R.step_e2e(b)
    eng.enable_timing(True, only="site_sweep")
    base = eng.kernel_stats()
    eng.synchronize()
    t0 = time.perf_counter()
    for b in batches[warmup:warmup + steps]:
        R.step_e2e(b)
    eng.synchronize()
    dt = time.perf_counter() - t0
    kern = kernel_table(eng.kernel_stats(), base)
    eng.close()
    return {"ms_per_step": 1e3 * dt / steps, "steps": steps, "site_sweep_avg_ms": kern["site_sweep"]["avg_ms"],
            "site_sweep_frac": (kern["site_sweep"]["gbs"] or 0.0) / HBM_PEAK_GBS,
            "note": "same batches, same lone step without the entropy array.  Since round 5 a one-barcode engine derives a looked-up, uncapped "
                    "site's entropy from its counters and writes the array only where a site crosses the cap (DESIGN §3), so the two "
                    "loops run the same kernels; what is left between them is the box and the order of the loops"}


def cold_updates(R, eng, batches, pauses=(2.0, 2.0, 2.0, 2.0, 2.0, 10.0)):
    """The regime of a live run (the reference updates every 60 s, boss/config.py:29): the GPU idle for `pause` seconds, clocks
    down, then ONE lone update from PAF text to masks.  Median over the 2-s pauses, the one 10-s pause on its own."""
    ms = []
    for i, p in enumerate(pauses):
        eng.synchronize()
        time.sleep(p)
        t0 = time.perf_counter()
        R.step_e2e(batches[i % len(batches)])
        eng.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
    short = [m for m, p in zip(ms, pauses) if p < 5]
    return {"after_2s_idle_ms_median": float(np.median(short)), "after_2s_idle_ms": short,
            "after_10s_idle_ms": [m for m, p in zip(ms, pauses) if p >= 5],
            "note": "ONE lone update (PAF text + reads in host memory -> masks in host memory) behind an idle GPU: what a live run's "
                    "60-s cadence sees; the back-to-back loop of ms_per_step keeps the clocks up"}


def late_regime_run(workload, contigs, device, batches, warmup, steps, track_entropy, depth=28.0):
    """The headline loop once more from ~28x coverage: most sites near or beyond the cap of 30 — bins that mix capped and deep
    sites, where the chain's tables are cut into pieces and the stitch evaluates stretches from the exact value — the state a
    long run lives in (the default loop starts at ~8x)."""
    runs, nb = make_runs(workload, contigs, 0, 1, device, track_entropy, preload_override=depth)
    R = Runner(workload, runs, nb, batches, False)
    eng = runs.engine
    eng.enable_timing(True)            # BEFORE the warm-up (as in main and entropy_off_run)
    for b in batches[:warmup]:
        R.step_e2e(b)
    eng.enable_timing(True, only="site_sweep")      # (the timed loop as in main: the sweep's events only; the per-kernel times from a second, instrumented loop)
    base = eng.kernel_stats()
    cs0 = eng.chain_stats()
    eng.synchronize()
    t0 = time.perf_counter()
    for b in batches[warmup:warmup + steps]:
        R.step_e2e(b)
    eng.synchronize()
    dt = time.perf_counter() - t0
    sweep = kernel_table(eng.kernel_stats(), base)["site_sweep"]
    cs1 = eng.chain_stats()
    eng.enable_timing(True)
    base = eng.kernel_stats()
    n_inst = min(steps, 5)
    eng.synchronize()
    t0 = time.perf_counter()
    for b in batches[warmup:warmup + n_inst]:
        R.step_e2e(b)
    eng.synchronize()
    dt_inst = time.perf_counter() - t0
    kern = kernel_table(eng.kernel_stats(), base)
    eng.close()
    return {"preload_depth": depth, "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "site_sweep_avg_ms": sweep["avg_ms"], "benefit_chain_avg_ms": kern["benefit_chain"]["avg_ms"],
            "instrumented_ms_per_step": 1e3 * dt_inst / n_inst,
            "kernels_ms_per_update": float(sum(v["avg_ms"] * v["launches"] for v in kern.values()) / max(n_inst, 1)),
            "benefit_chain_form": {k: cs1[k] - cs0[k] for k in cs0}}


def kernel_table(stats, base):
    kern = {}
    for k, v in stats.items():
        n = v["launches"] - base[k]["launches"]
        ms = (v["ms_total"] - base[k]["ms_total"]) / max(n, 1)
        kern[k] = dict(avg_ms=ms, launches=n, bytes=v["bytes_last"],
                       gbs=(v["bytes_last"] / 1e9) / (ms / 1e3) if ms > 0 else None)
    return kern


class Runner:
    """One workload on this rank: end-to-end steps on fresh batches, and the same updates on
    resident batches."""

    def __init__(self, workload, runs, nb, batches, distributed):
        self.w, self.runs, self.nb, self.batches, self.dist = workload, runs, nb, batches, distributed
        self.eng = runs.engine

    def ahead(self, nxt):
        """`lookahead` argument naming batch `nxt` (None: none)."""
        if nxt is None or os.environ.get("BOSSX_NO_LOOKAHEAD"):
            return None
        return (nxt["paf"], nxt["seqs"], nxt["barcodes"] if self.nb > 1 else None)

    def step_e2e(self, b, nxt=None):
        """One decision update, PAF text + read strings in host memory -> masks in host memory.  `nxt`: the
        batch of the next step — parsed / uploaded / walked while the GPU runs this step's chain
        (BossRuns.process_batch_paf(lookahead=...)); every step then still does one parse and one update."""
        bcs = b["barcodes"] if self.nb > 1 else None
        if self.dist:
            self.runs.process_batch_paf(b["paf"], b["seqs"], barcodes=bcs, read_lengths=b["read_lengths_arr"], lookahead=self.ahead(nxt))
        else:
            self.runs.rl_dist.update(b["read_lengths_arr"])        # boss/core.py:106
            self.runs.process_batch_paf(b["paf"], b["seqs"], barcodes=bcs, lookahead=self.ahead(nxt))

    def run_e2e(self, sel, tail=None):
        """The steps of `sel`, each staging its successor ahead (`tail` behind the last one)."""
        for i, b in enumerate(sel):
            self.step_e2e(b, sel[i + 1] if i + 1 < len(sel) else tail)

    def prime(self, b):
        """What the step before `b` would have done for it (untimed): stage it ahead."""
        if self.ahead(b) is not None:
            self.runs._stage_ahead(self.ahead(b))

    def stage(self, batches):
        """Parse + upload into numbered slots (untimed): inputs resident in HBM."""
        summ, t = [], []
        for i, b in enumerate(batches):
            self.eng.select_batch(i)
            t0 = time.perf_counter()
            summ.append(self.eng.stage_batch(b["paf"], b["seqs"], barcodes=b["barcodes"] if self.nb > 1 else None))
            t.append(time.perf_counter() - t0)
        return summ, t

    def step_resident(self, i, b, summ):
        runs, eng = self.runs, self.eng
        eng.ingest_staged(slot=i)
        if self.dist:          # sweep starts now; the exchange / host bookkeeping overlap with it
            runs.begin_update()
            runs.account_batch(summ, b["read_lengths_arr"], len(b["seqs"]))
        else:
            eng.update_begin(runs.args.optional.bucket_threshold)
            runs.rl_dist.update(b["read_lengths_arr"])
            runs.launch_benefit()    # the chain needs only the read-length windows
            runs._account_reads(summ, len(b["seqs"]))
        runs.update_wrapper()


def timed(fn_barrier, steps_fn):
    import gc
    # (no gc.collect() here: what it frees goes back to the allocator, trimmed heaps back to the OS, and the first update after it pays
    # the page faults of its ~20 MB of parse buffers again — 9 ms seen on the first timed update; the collection is done before the warm-up)
    gc.disable()                 # no collector pause inside the timed region
    fn_barrier()
    t0 = time.perf_counter()
    steps_fn()
    fn_barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    return dt


def cpu_baseline_like_for_like(runs, contigs, workload, batch):
    """ONE update of the oracle (structure-faithful numpy port of the reference, oracle/) on the
    host, starting from exactly the state the GPU engine holds (coverage, scores, switches, masks,
    read-start counts and the read-length histogram are exported from it), on the same fresh batch
    the GPU then processes — so the two time the same work and their results can be compared."""
    from boss_runs_amd import synth
    from oracle.pipeline import OracleRuns
    lens, names, ploidy, nb, reject, preload = WORKLOADS[workload]
    o = OracleRuns([(n, synth.codes_to_str(c)) for n, c in contigs], ploidy=ploidy, nbarcodes=nb, bucket_threshold=0,
                   reject_refs={"%s_r0" % reject} if reject else ())
    for name, oc in o.contigs_filt.items():
        pc = runs.contigs[name]
        oc.coverage[:] = pc.coverage
        oc.scores[:] = pc.scores
        oc.bucket_switches[:] = pc.bucket_switches
        oc.switched_on[:] = pc.switched_on
        oc.strat[:] = pc.strat
        o.read_starts.read_starts[name][:] = runs.read_starts.read_starts[name]
    o.rl_dist.read_lengths[:] = runs.rl_dist.read_lengths
    o.timings = {}
    rl = dict(zip(batch["seqs"].keys(), batch["read_lengths_arr"].tolist()))
    t0 = time.perf_counter()
    o.process_batch(batch["paf"], batch["seqs"], read_lengths=rl, barcodes=batch["barcodes"] if nb > 1 else None)
    t_cpu = time.perf_counter() - t0
    # the GPU on the same batch from the same state
    t0 = time.perf_counter()
    runs.rl_dist.update(batch["read_lengths_arr"])
    runs.process_batch_paf(batch["paf"], batch["seqs"], barcodes=batch["barcodes"] if nb > 1 else None)
    t_gpu = time.perf_counter() - t0
    same = runs.threshold == o.threshold
    for name, oc in o.contigs_filt.items():
        same = same and bool(np.array_equal(runs.contigs[name].strat, oc.strat))
    return o, t_cpu, t_gpu, bool(same)


def other_workload(name, device, batches, steps, warmup, track_entropy):
    """Short run of another single-GPU workload: end-to-end and resident-batch update times."""
    contigs = _GEN[name]
    runs, nb = make_runs(name, contigs, 0, 1, device, track_entropy)
    R = Runner(name, runs, nb, batches, False)
    eng = runs.engine
    G = sum(c.length for c in runs.contigs_filt.values())
    eng.enable_timing(True)            # BEFORE the warm-up (the first record of an event costs milliseconds)
    for b in batches[:warmup]:
        R.step_e2e(b)
    eng.enable_timing(True, only="site_sweep")      # the timed lone loop as in main: the sweep's events only
    base = eng.kernel_stats()
    sel = batches[warmup:warmup + steps]
    dt = timed(eng.synchronize, lambda: [R.step_e2e(b) for b in sel])          # lone updates, like the headline
    kern = kernel_table(eng.kernel_stats(), base)
    eng.enable_timing(True)            # the resident loop fully instrumented: the chain's time comes from there
    summ, _ = R.stage(sel)
    base_r = eng.kernel_stats()
    dtr = timed(eng.synchronize, lambda: [R.step_resident(i, b, s) for i, (b, s) in enumerate(zip(sel, summ))])
    kern["benefit_chain"] = kernel_table(eng.kernel_stats(), base_r)["benefit_chain"]
    eng.enable_timing(False)
    ms = 1e3 * dt / len(sel)
    out = {"workload": "%s: %s bp, ploidy %d, nbarcodes %d" % (name, "+".join(str(x) for x in WORKLOADS[name][0][:3]) +
                                                              ("+..." if len(WORKLOADS[name][0]) > 3 else ""),
                                                              WORKLOADS[name][2], nb),
           "steps": len(sel), "ms_per_step": ms, "kernels_only_ms": 1e3 * dtr / len(sel),
           "value_mbp_per_s": G * nb / 1e6 / (dt / len(sel)),
           "site_sweep": {"avg_ms": kern["site_sweep"]["avg_ms"], "frac_of_hbm_peak": (kern["site_sweep"]["gbs"] or 0.0) / HBM_PEAK_GBS},
           "benefit_chain_ms": kern["benefit_chain"]["avg_ms"],
           "benefit_chain_form": {k: v for k, v in eng.chain_stats().items()}}
    eng.close()
    return out


GRCH38_EXTRA = [("MT", 16_569), ("scaf_150k", 150_000), ("scaf_250k", 250_000), ("scaf_400k", 400_000)]
from boss_runs_amd.synth import GRCH38_LENS as _GRCH38_LENS      # noqa: E402  (numpy only)
GRCH38_LENGTHS = list(_GRCH38_LENS) + [L for _, L in GRCH38_EXTRA]


def grch38_contigs():
    """24 chromosome lengths of GRCh38 + MT (dropped: < 100 kb) + three scaffolds >= 100 kb."""
    from boss_runs_amd import synth
    names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY"]
    return list(zip(names, synth.GRCH38_LENS)) + GRCH38_EXTRA


def _grch38_codes(index, length):
    return np.random.default_rng(9000 + index).integers(0, 4, size=length, dtype=np.uint8)


def run_grch38(a, rank, world, local_rank, steps=None, warmup=None):
    """STRONG scaling of the 3.1 Gb reference (north_star: "1/2/4/8-GPU scaling reported on a 3 Gb
    synthetic reference"): contigs partitioned over the ranks (boss_runs_amd/parallel.py), the 4000
    reads of a batch sharded with them (a rank parses and ingests the reads that map to its own
    contigs), ONE global threshold per update through the in-stream RCCL collectives.  With one
    rank the whole genome is on one GPU (34 GB without the entropy array) and no collective runs.
    Returns the result dict on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.parallel import DistributedBossRuns, partition_contigs, shard_balance
    from boss_runs_amd.runs import BossRuns
    steps = steps or a.steps
    warmup = warmup if warmup is not None else a.warmup
    allc = grch38_contigs()
    kept = [(i, n, L) for i, (n, L) in enumerate(allc) if L >= 100_000]
    owner = partition_contigs([L for _, _, L in kept], world)
    mine = [(i, n, L) for (i, n, L), o in zip(kept, owner) if o == rank]
    G_total = sum(L for _, _, L in kept)
    G_mine = sum(L for _, _, L in mine)
    t0 = time.perf_counter()
    codes = {n: _grch38_codes(i, L) for i, n, L in mine}
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_reads = max(1, int(round(a.reads * G_mine / G_total))) if mine else 0
    key = "grch38_r%d" % rank
    _GEN[key] = [(n, codes[n]) for _, n, _ in mine]
    n_b = warmup + steps
    batches = generate_batches([(key, 31000 * (rank + 1) + i, n_reads, 1) for i in range(n_b)]) if mine else \
        [dict(paf="", seqs={}, barcodes={}, read_lengths_arr=np.zeros(0, dtype=np.int64)) for _ in range(n_b)]
    t_gen = time.perf_counter() - t0
    args = BossConfig()
    args.general.name = "bench_grch38_r%d" % rank
    args.optional.ploidy = 2
    args.optional.bucket_threshold = 0
    args.gpu.device = local_rank
    args.gpu.track_entropy = bool(a.track_entropy)
    args.gpu.mask_format = "bits"      # 62 MB of mask bytes per update at 3.1 Gb: the packed form (masks.py) is what leaves the GPU
    contig_arg = [(n, acgt[codes[n]].tobytes() if n in codes else L) for n, L in allc]
    # (BOSSX_FORCE_COLLECTIVES=1 with the torchrun variables set: the multi-GPU protocol on ONE rank —
    # what every rank of an N-GPU run executes, measurable on a single GPU)
    distributed = world > 1 or bool(os.environ.get("BOSSX_FORCE_COLLECTIVES") and dist.is_initialized())
    if not distributed:
        runs = BossRuns(args)
        runs.init(contigs=contig_arg)
    else:
        runs = DistributedBossRuns(args)
        runs.init(contigs=contig_arg, sharded_reads=True, gather_masks=False)
    del contig_arg
    runs.write_masks = False
    runs.log_fractions = False
    runs.engine.preload_coverage(8.0, seed=17 + rank)
    R = Runner("grch38", runs, 1, batches, distributed)
    eng = runs.engine

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()
    for b in batches[:warmup]:
        R.step_e2e(b)
    eng.enable_timing(True)
    base = eng.kernel_stats()
    sel = batches[warmup:]
    elapsed = timed(barrier, lambda: [R.step_e2e(b) for b in sel])          # lone updates, like the headline
    kern = kernel_table(eng.kernel_stats(), base)
    eng.enable_timing(False)
    shard = torch.tensor([float(G_mine), float(n_reads), elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        allsh = [torch.zeros_like(shard) for _ in range(world)]
        dist.all_gather(allsh, shard)
        elapsed = max(float(t[2]) for t in allsh)
        shards = [{"rank": r, "sites": int(t[0]), "reads_per_batch": int(t[1])} for r, t in enumerate(allsh)]
    else:
        shards = [{"rank": 0, "sites": int(G_mine), "reads_per_batch": n_reads}]
    longest = max(L for _, _, L in kept) // 100 + 1
    out = None
    if rank == 0:
        ms = 1e3 * elapsed / steps
        achieved = kern["site_sweep"]["gbs"] or 0.0
        out = {"workload": "grch38: %d contigs >= 100 kb, %d bp, ploidy 2, %d-read batches sharded with the contigs; "
                           "step = PAF text + reads in host memory -> masks in host memory" % (len(kept), G_total, a.reads),
               "scaling": "strong", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
               "steps": steps, "warmup": warmup, "ms_per_step": ms, "value_mbp_per_s": G_total / 1e6 / (elapsed / steps),
               "shards": shards,
               "partition": "whole contigs; longest-first packing or contiguous runs in FASTA order, whichever leaves the lighter heaviest rank (parallel.partition_contigs)",
               "shard_balance": {"max_over_mean": shard_balance([L for _, _, L in kept], owner, world),
                                 "linear": shard_balance([L for _, _, L in kept], partition_contigs([L for _, _, L in kept], world, "linear"), world),
                                 "lpt": shard_balance([L for _, _, L in kept], partition_contigs([L for _, _, L in kept], world, "lpt"), world)},
               "site_sweep_rank0": {"avg_ms": kern["site_sweep"]["avg_ms"], "frac_of_hbm_peak": achieved / HBM_PEAK_GBS,
                                    "algorithmic_bytes": kern["site_sweep"]["bytes"]},
               "benefit_chain_ms_rank0": kern["benefit_chain"]["avg_ms"],
               "serial_form_floor_ms": longest * CHAIN_FLOOR_CYCLES / GPU_CLOCK_GHZ * 1e-6,
               "note": "serial_form_floor_ms: what the SERIAL walk of the longest contig's move_sum would take at the dependent-op "
                       "latency alone — the bound of the fallback kernel; the chunk-parallel chain is not bound by it",
               "collectives_per_update": (runs.n_collectives / max(n_b, 1)) if distributed else 0,
               "generation_s": t_gen}
    eng.close()
    return out


_REAL_STDOUT = None


def own_stdout():
    """Libraries print to file descriptor 1 (RCCL's version banner at init does): keep the real
    stdout for the ONE JSON line and point descriptor 1 at stderr for everything else."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_json(obj):
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: this process never touches a GPU —
    it starts N fresh ranks (one per GPU, `torch.distributed.run` on 127.0.0.1, RCCL over xGMI) of
    this same script and relays rank 0's single JSON line.  (Under the driver's own
    `python -m torch.distributed.run ... bench.py --gpus N` the ranks already exist: RANK / WORLD_SIZE
    are set and this function is not reached.)"""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if n_dev < a.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (a.gpus, n_dev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N ranks share the host: bound each rank's parser pool and generator pool
    ncpu = os.cpu_count() or 1
    env.setdefault("BOSSX_PARSE_THREADS", str(max(2, min(16, ncpu // a.gpus))))
    env.setdefault("BOSSX_GEN_PROCS", str(max(1, min(32, ncpu // a.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for line in p.stdout:
        if line.lstrip().startswith("{"):
            lines.append(line.strip())
        else:
            sys.stderr.write(line)
    rc = p.wait()
    if rc != 0 or not lines:
        raise SystemExit("bench.py --gpus %d: the ranks exited with code %d and %d result line(s)" % (a.gpus, rc, len(lines)))
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()


def host_cpu_state():
    """CPU time of this process so far and the cgroup's throttling counters (cgroup v2; zeros where absent)."""
    import time
    st = {"cpu_s": time.process_time(), "nr_throttled": 0, "throttled_usec": 0, "quota": None}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in ("nr_throttled", "throttled_usec"):
                st[k] = int(v)
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        st["quota"] = None if q == "max" else int(q) / int(p)
    except (OSError, ValueError):
        pass
    return st


def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and not a.prepare_only:
        return spawn_ranks(a)
    if not a.prepare_only:
        own_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    workload = a.workload or "chr20_21"
    if workload == "grch38":
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the decision-update path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        if world > 1 or (os.environ.get("BOSSX_FORCE_COLLECTIVES") and "RANK" in os.environ):
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        os.chdir(tempfile.mkdtemp(prefix="bossx_bench_"))
        res = run_grch38(a, rank, world, local_rank)
        if rank == 0:
            line = {"metric": "decision-update wall-clock (ms) + Mbp scored/sec, 4000-read batch",
                    "value": res["value_mbp_per_s"], "unit": "Mbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                    "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": "u16+f64", "data": "synthetic", "config": {"workload": res["workload"]}, "commit": current_commit(),
                    "grch38": res}
            emit_json(line)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    from boss_runs_amd import synth

    # ---- inputs: reference + W+K (+1) distinct synthetic batches, before the GPU is initialised ----
    n_b = a.warmup + a.steps
    _GEN[workload] = make_reference(workload, rank)
    nb_main = WORKLOADS[workload][3]
    jobs = [(workload, 1000 * (rank + 1) + i, a.reads, nb_main) for i in range(n_b + 1)]
    others = []
    if world == 1 and not a.no_others:
        others = [w for w in ("ecoli", "barcoded", "shard390") if w != workload]
        for w in others:
            _GEN[w] = make_reference(w, 0)
            jobs += [(w, 5000 + i, a.reads, WORKLOADS[w][3]) for i in range(2 + 5)]
    t0 = time.perf_counter()
    allb = generate_batches(jobs)
    t_gen = time.perf_counter() - t0
    if a.prepare_only:
        return
    batches, extra = allb[:n_b], allb[n_b]
    other_batches, off = {}, n_b + 1
    for w in others:
        other_batches[w] = allb[off:off + 7]
        off += 7
    del allb

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decision-update path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 or (os.environ.get("BOSSX_FORCE_COLLECTIVES") and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    os.chdir(tempfile.mkdtemp(prefix="bossx_bench_"))
    contigs = _GEN[workload]
    runs, nb = make_runs(workload, contigs, rank, world, local_rank, a.track_entropy)
    distributed = hasattr(runs, "account_batch")
    eng = runs.engine
    R = Runner(workload, runs, nb, batches, distributed)
    G = sum(c.length for c in runs.contigs_filt.values() if not getattr(c, "remote", False))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    # ---- the timed region: K end-to-end decision updates on fresh batches, ONE AFTER THE OTHER (SURVEY §8d's t_update:
    # PAF text + read strings in host memory -> masks in host memory; nothing of batch i+1 is touched before update i has
    # returned — a live run's next batch does not exist yet, and in simulation it is the outcome of this update's masks)
    # (the engine's HIP events are switched on BEFORE the warm-up: the first record of an event on a stream costs milliseconds —
    # 8-16 ms seen on the first timed update when they were switched on behind it — and belongs to no update)
    eng.enable_timing(True)          # HIP events on the engine's own streams, collected after the region
    import gc
    gc.collect()
    barrier()                        # (torch's first synchronize initialises its context: before the warm-up, not between it and the timed region)
    for b in batches[:a.warmup]:
        R.step_e2e(b)
    # Inside the timed region only the kernel the roofline is quoted on is bracketed by events (round 6): every event pair is two marker
    # packets between the kernels of an update — with all five kernels bracketed the step measured ~0.1 ms more than the product takes.
    # The other kernels' times come from the same loop run once more, fully instrumented, right behind the timed region (`kernels`).
    eng.enable_timing(True, only="site_sweep")
    barrier()
    base = eng.kernel_stats()
    sel = batches[a.warmup:]
    host_before = host_cpu_state()
    cs0 = eng.chain_stats()
    step_s = []                      # wall time of every update of the timed region (diagnostic: a box's stalls show as outliers)

    def lone_loop():
        for b in sel:
            t_b = time.perf_counter()
            R.step_e2e(b)
            step_s.append(time.perf_counter() - t_b)
    elapsed = timed(barrier, lone_loop)
    host_after = host_cpu_state()
    cs1 = eng.chain_stats()
    stats = eng.kernel_stats()
    kern_timed = kernel_table(stats, base)
    # ... the same K lone updates once more with every kernel bracketed (the state has K more batches: the regime moves on a little)
    eng.enable_timing(True)
    base_i = eng.kernel_stats()
    inst_s = []

    def inst_loop():
        for b in sel[:5]:            # (few: every update moves the state on, and the loops behind this one should see the regime of the timed one)
            t_b = time.perf_counter()
            R.step_e2e(b)
            inst_s.append(time.perf_counter() - t_b)
    elapsed_inst = timed(barrier, inst_loop)
    kern = kernel_table(eng.kernel_stats(), base_i)
    kern_inst_sweep = kern["site_sweep"]
    kern["site_sweep"] = kern_timed["site_sweep"]           # (the roofline kernel: as it ran inside the timed region)
    cs1b = eng.chain_stats()
    # ---- secondary: the steady state of a REPLAY, where the caller already holds batch i+1 and hands it along
    # (process_batch_paf(lookahead=...)): step i parses / uploads / walks batch i+1 while the GPU runs update i
    n_pipe = min(len(sel), 10)
    R.prime(sel[0])
    elapsed_pipe = timed(barrier, lambda: R.run_e2e(sel[:n_pipe], tail=extra))
    cs2 = eng.chain_stats()
    # the CPU port runs ONE update here, from the state the engine holds now — a regime with scored,
    # unscored and capped sites side by side.  (The deeply saturated regime a long run ends in — every site
    # capped, thresholds of 1e-304 — is compared separately: scripts/ecoli_diff.py at 90 / 120 / 150 updates,
    # profiles/r04_ecoli_diff.txt, and tests/test_parity_gpu.py::test_deep_saturation_threshold_vs_oracle: equal.)
    # Behind the timed region, not in front of it: ten seconds of an idle GPU in front of a 50-ms region cost
    # its first steps their clocks.
    cpu_cmp = None
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        eng.enable_timing(False)
        cpu_cmp = cpu_baseline_like_for_like(runs, contigs, workload, extra)
        eng.enable_timing(True)
    cold = None
    if world == 1 and rank == 0 and not a.no_cold:
        eng.enable_timing(False)
        cold = cold_updates(R, eng, sel)
        eng.enable_timing(True)
    # ---- the same updates with the inputs already resident in HBM (parse + upload outside) --------
    summ, t_stage = R.stage(sel)
    base2 = eng.kernel_stats()
    cs3 = eng.chain_stats()
    elapsed_res = timed(barrier, lambda: [R.step_resident(i, b, s) for i, (b, s) in enumerate(zip(sel, summ))])
    cs4 = eng.chain_stats()
    kern_res = kernel_table(eng.kernel_stats(), base2)
    # ---- and once more with every tile swept at every update (BOSSX_INCREMENTAL=0): the streaming form
    # of the sweep kernel on this workload, for the roofline of the kernel as opposed to the update
    os.environ["BOSSX_INCREMENTAL"] = "0"
    base3 = eng.kernel_stats()
    timed(barrier, lambda: [R.step_resident(i, b, s) for i, (b, s) in enumerate(zip(sel, summ))])
    kern_full = kernel_table(eng.kernel_stats(), base3)
    os.environ.pop("BOSSX_INCREMENTAL", None)
    eng.enable_timing(False)
    # event overhead check: the resident loop once more without events
    elapsed_res_noev = timed(barrier, lambda: [R.step_resident(i, b, s) for i, (b, s) in enumerate(zip(sel, summ))])
    aligned = float(np.mean([s["aligned"] for s in summ]))

    if world > 1:
        t = torch.tensor([elapsed, elapsed_res], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_res = float(t[0].item()), float(t[1].item())
        gt = torch.tensor([float(G * nb)], dtype=torch.float64, device="cuda")
        dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        total_sites = float(gt.item())
    else:
        total_sites = float(G * nb)
    ms_per_step = 1e3 * elapsed / a.steps
    value = total_sites / 1e6 / (elapsed / a.steps)

    if rank == 0:
        dom = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"])
        roof_k = "site_sweep"    # the HBM-streaming kernel the roofline is quoted on
        achieved = kern[roof_k]["gbs"] or 0.0
        # (the profiled run also holds the every-tile sweeps of the full_sweep loop: when the timed region
        # swept only the tiles that receive bases, its traffic is that of the ingesting launches alone)
        incremental = kern[roof_k]["bytes"] < 0.99 * kern_full[roof_k]["bytes"]
        traffic, traffic_src, traffic_commit, traffic_ksrc = pmc_traffic(workload, "site_sweep",
                                                           only="_kernel<true" if incremental else None)
        commit = current_commit()
        longest_bins = max(c.length // 100 + 1 for c in runs.contigs_filt.values())
        out = {
            "metric": "decision-update wall-clock (ms) + Mbp scored/sec, 4000-read batch",
            "value": value, "unit": "Mbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16+f64", "data": "synthetic",
            "config": {"workload": "%s: %s bp%s, ploidy %d, nbarcodes %d, %d-read PAF batches (mean 6 kb), per GPU; "
                                   "step = PAF text + reads in host memory -> masks in host memory"
                       % (workload, "+".join("%d" % c[1].shape[0] for c in contigs),
                          (" (reject_refs=%s)" % WORKLOADS[workload][4]) if WORKLOADS[workload][4] else "",
                          WORKLOADS[workload][2], nb, a.reads),
                       "sites_per_gpu": G, "aligned_bases_per_batch": aligned,
                       "track_entropy": bool(a.track_entropy),
                       "parallelism": "contig-sharded x%d, one global threshold" % world,
                       "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
                       "collectives_per_update": (runs.n_collectives / max(getattr(runs, "n_updates", 0), 1)) if distributed else 0},
            "commit": commit,
            "timed_region_s": elapsed,
            "ms_per_step_note": "t_update of SURVEY §8d: one decision update from PAF text + reads in host memory to masks in host "
                                "memory (parse -> upload -> walk -> sweep -> chain -> threshold -> masks), strictly one after the other",
            "lone_update_ms": ms_per_step,
            "pipelined_ms_per_step": 1e3 * elapsed_pipe / n_pipe,
            "pipelined_note": "secondary: a replay that already holds batch i+1 hands it to step i (process_batch_paf(lookahead=...)), "
                              "which parses / uploads / walks it while the GPU runs update i — one parse and one update per step either way",
            "kernels_only_ms": 1e3 * elapsed_res / a.steps,
            "kernels_only_note": "the same K updates with every batch already parsed and resident in HBM "
                                 "(ingest + sweep + buckets + chain + histogram + masks + D2H); "
                                 "a LATER loop of the same batches without the per-kernel HIP events: %.3f ms — later, so on a state that "
                                 "has received 3K more batches: more bins mix capped and deep sites and the chain's tables are cut more "
                                 "often (kernels_resident_loop vs kernels); late_regime times that end of a run on its own"
                                 % (1e3 * elapsed_res_noev / a.steps),
            "cold_update_ms": cold,
            "roofline": {"kernel": roof_k, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_commit": traffic_commit,
                         "traffic_same_commit": bool(commit and traffic_commit and commit[:12] == str(traffic_commit)[:12]),
                         "traffic_same_kernel_sources": bool(traffic_ksrc and traffic_ksrc == kernel_sources_hash()),
                         "algorithmic_bytes": kern[roof_k]["bytes"], "avg_launch_ms": kern[roof_k]["avg_ms"],
                         "rocprof": dict(_PMC_EXTRA, note="kernel-trace durations of the committed profile: the launches of its lone-update loop, and all 53 "
                                         "launches of the script (later loops run on a state that has saturated further: slower launches)"),
                         "pattern_ceiling": {"source": "profiles/r05_rw_pattern.txt (scripts/experiments/rw_pattern.hip: the launch's loads and stores without its arithmetic)",
                                             "memory_system_tb_s_for_this_read_write_mix": [2.7, 3.3], "reads_only_tb_s": [6.1, 6.9],
                                             "note": "a write costs twice a read and by the 64-byte granule it dirties; 0.40 of 8 TB/s is at or beyond what the machine gives the pattern (DESIGN 4.2)"},
                         "full_sweep": {"avg_launch_ms": kern_full[roof_k]["avg_ms"], "algorithmic_bytes": kern_full[roof_k]["bytes"],
                                        "achieved": kern_full[roof_k]["gbs"], "frac": (kern_full[roof_k]["gbs"] or 0.0) / HBM_PEAK_GBS,
                                        "note": "the same batches with every tile swept (BOSSX_INCREMENTAL=0), resident loop"},
                         "note": "site_sweep = the sweep launches of one update (fused CIGAR expansion + coverage "
                                 "increment + scoring + bin sums) as they ran in the timed region — only the tiles that "
                                 "receive bases when fewer than half do (the others keep their bin sums; algorithmic bytes "
                                 "count the swept tiles only); HIP events on the engine stream; full_sweep = every tile "
                                 "swept; roofline_large = the same kernel on an HBM-sized working set"},
            "metric_note": "value = Mbp scored/s (reference positions brought up to date per second); "
                           "ms_per_step = decision-update wall-clock, PAF text in host memory -> masks in host memory",
            "kernels": kern, "kernels_resident_loop": kern_res, "dominant_kernel_by_time": dom,
            "kernels_note": "site_sweep: HIP events inside the timed region (the only kernel bracketed there); the other kernels: the same K lone "
                            "updates run once more right behind it with every kernel bracketed (instrumented_lone_update_ms; its site_sweep: "
                            "%.4f ms)" % kern_inst_sweep["avg_ms"],
            "instrumented_lone_update_ms": 1e3 * elapsed_inst / max(len(inst_s), 1),
            # the chain is bottleneck.move_sum's serial FP64 recurrence (1 % of the data), run chunk-parallel and still exact
            "chain_latency": {"kernel": "benefit_chain", "bound": "matrix-op issue + the stitch's dependent walk over the chunks",
                              "bins_total": int(eng.merged_bins), "bins_longest_contig": int(longest_bins),
                              "ns_per_bin_longest": 1e6 * kern["benefit_chain"]["avg_ms"] / max(longest_bins, 1),
                              "serial_form_floor_ms": longest_bins * CHAIN_FLOOR_CYCLES / GPU_CLOCK_GHZ * 1e-6,
                              "on_fp64_matrix_core": eng.matrix_chain,
                              "note": "candidate tables (four windows per v_mfma_f64_4x4x4) -> stitched start values -> every 4096-bin "
                                      "segment recomputed from its exact start; serial_form_floor_ms is what the SERIAL walk of the "
                                      "longest contig would take at the dependent-op latency alone (5.1 cycles per bin) — the bound "
                                      "of the fallback kernel, not of an update"},
            "host": {"stage_batch_ms_mean": 1e3 * float(np.mean(t_stage)),
                     "lone_update_ms_each": [round(1e3 * t, 3) for t in step_s],
                     "lone_update_ms_median": 1e3 * float(np.median(step_s)),
                     "batch_generation_s": t_gen,
                     # the host side of the timed region: CPUs this container may use (CFS quota), CPU-seconds per wall-second
                     # the process spent inside it, and how long the scheduler held its threads back there (cgroup cpu.stat)
                     "cpu_quota_cores": host_before["quota"],
                     "cpus_busy_in_timed_region": (host_after["cpu_s"] - host_before["cpu_s"]) / max(elapsed, 1e-9),
                     "throttled_ms_in_timed_region": (host_after["throttled_usec"] - host_before["throttled_usec"]) / 1e3,
                     "throttled_periods_in_timed_region": host_after["nr_throttled"] - host_before["nr_throttled"],
                     "note": "stage_batch_ms_mean = native PAF/CIGAR parse + upload of one batch INTO A NEW SLOT (first-use allocations "
                             "included: the resident loop's preparation, not the lone update's staging); lone_update_ms_each = "
                             "the timed region's updates one by one (ms_per_step is their mean plus the closing synchronisation)"},
            "move_sum_on_fp64_matrix_core": eng.matrix_chain,
            "benefit_chain_form": dict(eng.chain_stats(),
                                       per_loop={"timed_lone_updates": {k: cs1[k] - cs0[k] for k in cs0},
                                                 "instrumented_lone_updates": {k: cs1b[k] - cs1[k] for k in cs1},
                                                 "pipelined": {k: cs2[k] - cs1b[k] for k in cs1},
                                                 "resident": {k: cs4[k] - cs3[k] for k in cs3}},
                                       note="chunk-parallel, exact (candidate tables on the matrix core -> stitched start "
                                       "values -> every 4096-bin segment recomputed from its exact start and checked against its "
                                       "successor's; the serial kernel is enqueued behind, gated on a failed check); counters since "
                                       "finalize, over every loop of this script"),
        }
        if cpu_cmp is not None:
            o, t_cpu, t_gpu, same = cpu_cmp
            out["cpu_baseline"] = {
                "value": G * nb / 1e6 / t_cpu, "unit": "Mbp/s", "cores": 1, "kind": "port",
                "sample": "1 full update of the same workload (%s, %d sites) through oracle/ (numpy port of the "
                          "reference) from the engine's exported state right behind the timed region, same region: PAF text -> masks" % (workload, G),
                "ms_per_update": 1e3 * t_cpu, "host_cores_available": os.cpu_count(),
                "stages_ms": {k: 1e3 * v for k, v in o.timings.items()},
                "gpu_ms_same_batch": 1e3 * t_gpu, "masks_and_threshold_equal_to_gpu": same}
            out["speedup_vs_cpu_port"] = (1e3 * t_cpu) / ms_per_step
            del o
        if world == 1 and not a.no_large:
            out["roofline_large"] = large_sweep(local_rank)
        runs.engine.close()
        del runs, eng, R
        if world == 1 and not a.no_late:
            try:
                out["late_regime"] = late_regime_run(workload, contigs, local_rank, batches, a.warmup, min(a.steps, 20), a.track_entropy)
            except Exception as e:      # a side measurement must not lose the main line
                out["late_regime"] = {"error": repr(e)}
        if world == 1 and a.track_entropy and not a.no_entropy_off_run:
            try:
                out["track_entropy_false"] = entropy_off_run(workload, contigs, local_rank, batches, a.warmup, min(a.steps, 10))
            except Exception as e:      # a side measurement must not lose the main line
                out["track_entropy_false"] = {"error": repr(e)}
        if others:
            out["workloads"] = {}
            for w in others:
                try:
                    out["workloads"][w] = other_workload(w, local_rank, other_batches[w], 5, 2, a.track_entropy)
                except Exception as e:      # a side measurement must not lose the main line
                    out["workloads"][w] = {"error": repr(e)}
    else:
        out = None
        runs.engine.close()
    if world > 1 and a.workload is None:
        # N > 1 without an explicit workload (the driver's scaling runs): the line above is the weak-
        # scaling curve of the N = 1 workload; the 3 Gb STRONG-scaling point of the same N rides along
        res = run_grch38(a, rank, world, local_rank, steps=5, warmup=2)
        if rank == 0:
            out["grch38_strong"] = res
    if rank == 0:
        emit_json(with_summary(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")


def with_summary(out):
    """The contract's keys first, then the secondary figures a reader needs next to `ms_per_step` in ONE small object (a log
    that keeps only the head of the line still carries them), then everything else, and the same object once more at the very
    end (for a log that keeps only the tail)."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    summary = {
        "lone_update_ms_median": get(out, "host", "lone_update_ms_median"),
        "cold_update_ms_after_2s_idle": get(out, "cold_update_ms", "after_2s_idle_ms_median"),
        "late_regime_ms_per_step": get(out, "late_regime", "ms_per_step"),
        "late_regime_chain_ms": get(out, "late_regime", "benefit_chain_avg_ms"),
        "kernels_only_ms": out.get("kernels_only_ms"),
        "sweep_launch_ms": get(out, "roofline", "avg_launch_ms"),
        "roofline_frac": get(out, "roofline", "frac"),
        "chain_ms": get(out, "kernels", "benefit_chain", "avg_ms"),
        "threshold_hist_gbs": get(out, "kernels", "threshold_hist", "gbs"),
        "strategy_mask_gbs": get(out, "kernels", "strategy_mask", "gbs"),
        "cpu_port_ms_per_update": get(out, "cpu_baseline", "ms_per_update"),
    }
    ordered = {k: out[k] for k in CONTRACT_KEYS if k in out}
    ordered["summary"] = summary
    for k, v in out.items():
        if k not in ordered:
            ordered[k] = v
    ordered["summary_tail"] = summary
    return ordered


if __name__ == "__main__":
    main()
