"""Cycle probes of the sweep kernel (BOSSX_SWEEP_PROBE=1 prints them per launch) on a workload:
    python3 scripts/sweep_probe.py [workload]"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 100 + i, 4000, nb) for i in range(6)])
os.chdir(tempfile.mkdtemp())
os.environ["BOSSX_NO_OVERLAP"] = "1"
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, os.environ.get("ENT", "1") == "1")
for b in batches[:3]:
    runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None)
os.environ["BOSSX_SWEEP_PROBE"] = "1"
for b in batches[3:]:
    runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None)
