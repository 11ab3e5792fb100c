#!/usr/bin/env python3
"""Benchmark of the BOSS-RUNS decision update on MI355X (BASELINE.json metric:
"decision-update wall-clock (ms) + Mbp scored/sec, 4000-read batch").

One step = one decision update on one resident 4000-read batch: coverage-scatter of the
batch (inputs already parsed and uploaded to HBM), the fused per-site sweep, bucket switches,
exact move_sum benefit chain, threshold statistics, threshold choice on the host, mask
kernel and the device-to-host copy of every contig's mask.  `value` = reference positions
whose score state is brought up to date per second (G * nbarcodes / t_update), summed over
ranks; `ms_per_step` is the decision-update wall-clock.

    python bench.py                               # N=1, E. coli 4.6 Mb (BASELINE configs[1])
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
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

WORKLOADS = {
    # name: (contig lengths, names, ploidy, nbarcodes, reject, preload depth)
    "ecoli": ([4_641_652], ["ecoli_K12"], 1, 1, None, 0.0),
    "chr20_21": ([64_444_167, 46_709_983], ["chr20", "chr21"], 2, 1, None, 8.0),
    "shard390": ([248_956_422, 138_394_717], ["chr1", "chr9"], 2, 1, None, 8.0),
    "barcoded": ([5_000_000] * 10, ["bac%02d" % i for i in range(10)], 1, 8, None, 0.0),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ecoli", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-updates", type=int, default=3)
    ap.add_argument("--no-large", action="store_true",
                    help="skip the extra sweep-kernel measurement on a 200 Mb contig (HBM-sized working set)")
    return ap.parse_args()


def make_runs(workload, rank, world, device):
    """N=1: the fused single-GPU `BossRuns`.  N>1: `DistributedBossRuns`; the global reference is
    the per-GPU contig set repeated once per rank (weak scaling), contig-partitioned so that
    every rank owns its own copy, with ONE global threshold per update (the collectives of
    boss_runs_amd/parallel.py are inside the timed region)."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from boss_runs_amd.parallel import DistributedBossRuns
    lens, names, ploidy, nb, reject, preload = WORKLOADS[workload]
    mine = synth.make_reference(lens, seed=1 + rank, names=["%s_r%d" % (n, rank) for n in names])
    args = BossConfig()
    args.general.name = "bench_r%d" % rank
    args.optional.ploidy = ploidy
    args.optional.bucket_threshold = 0        # strategies on from the first update
    args.gpu.device = device
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
        assert all(not runs.contigs[n].remote for n in mine_d), "partition must give each rank its own contigs"
    runs.write_masks = False                  # npz write is reported separately
    runs.log_fractions = False
    if preload > 0:
        runs.engine.preload_coverage(preload, seed=7 + rank)
    return runs, mine, nb


def pmc_traffic(workload, kernel_substr):
    """HBM bytes per launch of `kernel_substr` from the newest committed rocprofv3 PMC summary
    of this workload (profiles/rNN_<workload>_rocprof_summary.json: FETCH_SIZE and WRITE_SIZE
    from separate --pmc passes, in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s_rocprof_summary.json" % workload)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot, found = 0.0, False
    for name, v in d.get("kernels", {}).items():
        if kernel_substr in name and "FETCH_SIZE_avg_per_launch" in v and "WRITE_SIZE_avg_per_launch" in v:
            tot += (2.0 * v["FETCH_SIZE_avg_per_launch"] + v["WRITE_SIZE_avg_per_launch"]) * 1024.0
            found = True
    return (tot if found else None), os.path.basename(files[-1])


def large_sweep(device, L=200_000_000, reps=5, depth=8.0):
    """The sweep kernel alone on one 200 Mb contig (working set 2.4 GB >> Infinity Cache):
    `stream` = fresh state (no LUT gathers), `gather` = every site scored (depth-8 preload)."""
    from boss_runs_amd.engine import Engine
    from boss_runs_amd.scoring import SiteScoring
    rng = np.random.default_rng(1)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L, dtype=np.uint8)].tobytes()
    e = Engine(nbarcodes=1, device=device, track_entropy=True)
    e.add_contig("big", seq)
    del seq
    hap = SiteScoring(1)
    e.finalize(hap.score0[0], hap.ent0[0])
    e.set_lut(*hap.tables())
    e.enable_timing(True)
    out = {"sites": L}
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
    e.close()
    return out


def cpu_baseline(contigs, batches, n_updates):
    """The oracle (structure-faithful numpy port of the reference) on the host cores."""
    from boss_runs_amd import synth
    from oracle.pipeline import OracleRuns
    o = OracleRuns([(n, synth.codes_to_str(c)) for n, c in contigs], bucket_threshold=0)
    times = []
    for b in batches[:n_updates]:
        t0 = time.perf_counter()
        o.process_batch(b["paf"], b["seqs"], read_lengths=b["read_lengths"])
        times.append(time.perf_counter() - t0)
    return o, times


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decision-update path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 or (os.environ.get("BOSSX_FORCE_COLLECTIVES") and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from boss_runs_amd import synth

    os.chdir(tempfile.mkdtemp(prefix="bossx_bench_"))
    runs, contigs, nb = make_runs(a.workload, rank, world, local_rank)
    distributed = hasattr(runs, "account_batch")
    eng = runs.engine
    G = sum(c.length for c in runs.contigs_filt.values() if not getattr(c, "remote", False))

    # ---- inputs: W+K distinct synthetic batches, parsed and resident in HBM before timing ----
    n_b = a.warmup + a.steps
    batches, summaries = [], []
    t_parse = []
    for i in range(n_b):
        b = synth.make_batch(contigs, a.reads, seed=1000 * (rank + 1) + i, nbarcodes=nb, extras=False)
        eng.select_batch(i)
        t0 = time.perf_counter()
        s = eng.stage_batch(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None)
        t_parse.append(time.perf_counter() - t0)
        b["read_lengths_arr"] = np.fromiter(b["read_lengths"].values(), dtype=np.int64, count=len(b["read_lengths"]))
        batches.append(b)
        summaries.append(s)
    aligned = float(np.mean([s["aligned"] for s in summaries]))

    def step(i):
        eng.ingest_staged(slot=i)
        if distributed:          # sweep starts now; the exchange / host bookkeeping overlap with it
            runs.begin_update()
        else:
            eng.update_begin(runs.args.optional.bucket_threshold)
        if distributed:
            runs.account_batch(summaries[i], batches[i]["read_lengths_arr"], len(batches[i]["seqs"]))
        else:
            runs.rl_dist.update(batches[i]["read_lengths_arr"])
            runs.launch_benefit()    # the chain needs only the read-length windows
            runs._account_reads(summaries[i], len(batches[i]["seqs"]))
        runs.update_wrapper()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    for i in range(a.warmup):
        step(i)
    eng.enable_timing(True)
    base = eng.kernel_stats()
    import gc
    gc.collect()
    gc.disable()                 # no collector pause inside the timed region
    barrier()
    t0 = time.perf_counter()
    for i in range(a.warmup, n_b):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    stats = eng.kernel_stats()
    eng.enable_timing(False)
    # PCIe-inclusive path, reported beside (never as) `value`: PAF text + reads in host memory ->
    # masks in host memory, i.e. stage (parse + upload) + update, on three further batches
    t_e2e = []
    if not distributed:
        for i in range(3):
            b = synth.make_batch(contigs, a.reads, seed=777000 + 1000 * rank + i, nbarcodes=nb, extras=False)
            rl = np.fromiter(b["read_lengths"].values(), dtype=np.int64, count=len(b["read_lengths"]))
            eng.select_batch(0)
            t0 = time.perf_counter()
            runs.rl_dist.update(rl)
            runs.process_batch_paf(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None)
            t_e2e.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gt = torch.tensor([float(G * nb)], dtype=torch.float64, device="cuda")
        dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        total_sites = float(gt.item())
    else:
        total_sites = float(G * nb)
    ms_per_step = 1e3 * elapsed / a.steps
    value = total_sites / 1e6 / (elapsed / a.steps)

    if rank == 0:
        kern = {}
        for k, v in stats.items():
            n = v["launches"] - base[k]["launches"]
            ms = (v["ms_total"] - base[k]["ms_total"]) / max(n, 1)
            kern[k] = dict(avg_ms=ms, launches=n, bytes=v["bytes_last"],
                           gbs=(v["bytes_last"] / 1e9) / (ms / 1e3) if ms > 0 else None)
        dom = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"])
        roof_k = "site_sweep"    # the HBM-streaming kernel the roofline is quoted on
        achieved = kern[roof_k]["gbs"] or 0.0
        traffic, traffic_src = pmc_traffic(a.workload, "site_sweep_kernel")
        out = {
            "metric": "decision-update wall-clock (ms) + Mbp scored/sec, 4000-read batch",
            "value": value, "unit": "Mbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16+f64", "data": "synthetic",
            "config": {"workload": "%s: %s, ploidy %d, nbarcodes %d, %d-read PAF batches (mean 6 kb), per GPU"
                       % (a.workload, "+".join("%d" % c[1].shape[0] for c in contigs),
                          WORKLOADS[a.workload][2], nb, a.reads),
                       "sites_per_gpu": G, "aligned_bases_per_batch": aligned,
                       "parallelism": "contig-sharded x%d, one global threshold" % world,
                       "collectives_per_update": (runs.comm.n_collectives / max(n_b, 1)) if distributed else 0},
            "roofline": {"kernel": roof_k, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes": kern[roof_k]["bytes"], "avg_launch_ms": kern[roof_k]["avg_ms"],
                         "note": "site_sweep = sweep<false> + sweep<true> launches (fused CIGAR expansion + "
                                 "coverage increment + scoring + bin sums); at this size the working set is "
                                 "cache resident and the kernel is latency bound; see roofline_large"},
            "metric_note": "value = Mbp scored/s (reference positions brought up to date per second); "
                           "ms_per_step = decision-update wall-clock",
            "kernels": kern, "dominant_kernel_by_time": dom,
            # the chain is a serial FP64 recurrence (1 % of the data): its bound is the dependent
            # matrix-op latency, not HBM — 52 cycles per 4 bins measured (scripts/mfma_f64_probe.hip)
            "chain_latency": {"kernel": "benefit_chain", "bound": "dependent-op latency",
                              "bins": int(eng.merged_bins), "ns_per_bin": 1e6 * kern["benefit_chain"]["avg_ms"] / max(int(eng.merged_bins), 1),
                              "floor_cycles_per_bin": 13.0, "on_fp64_matrix_core": eng.matrix_chain,
                              "runs_next_to_sweep": not os.environ.get("BOSSX_NO_OVERLAP"),
                              "note": "the kernel starts while the sweep of the same update is still running and "
                                      "waits for tiles it has not published yet: its duration includes those "
                                      "waits (0.316 ms = 6.8 ns/bin when it runs after the sweep)"},
            "host": {"stage_batch_ms_mean": 1e3 * float(np.mean(t_parse)),
                     "pcie_inclusive_update_ms": (1e3 * float(np.median(t_e2e))) if t_e2e else None,
                     "pcie_inclusive_note": "PAF text + read strings in host memory -> masks in host memory "
                                            "(threaded parse, upload, update); not the headline value"},
            "move_sum_on_fp64_matrix_core": eng.matrix_chain,
        }
        if not a.no_large and world == 1:
            del runs, eng
            out["roofline_large"] = large_sweep(local_rank)
        if not a.no_cpu_baseline and world == 1:
            _, times = cpu_baseline(contigs, batches, a.cpu_updates)
            t_med = float(np.median(times))
            out["cpu_baseline"] = {
                "value": G * nb / 1e6 / t_med, "unit": "Mbp/s", "cores": 1, "kind": "port",
                "sample": "%d full updates of the same workload through oracle/ (numpy port of the reference; "
                          "median %.2f s per update, PAF parse included)" % (a.cpu_updates, t_med),
                "ms_per_update": 1e3 * t_med, "host_cores_available": os.cpu_count()}
            out["speedup_vs_cpu_port"] = (1e3 * t_med) / ms_per_step
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
