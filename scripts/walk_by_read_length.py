"""Stage-only runs (PAF text -> emit runs in HBM) of 4000-read batches whose longest read is capped
at different lengths, for `rocprofv3 --kernel-trace --stats`: shows how much of the two CIGAR walk
kernels is the latency of the wave that owns the longest mapping.
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 scripts/walk_by_read_length.py <max_len>"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
from boss_runs_amd import synth

max_len = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
w = "chr20_21"
ref = bench.make_reference(w, 0)
# the same total of aligned bases per batch whatever the cap: mean 6 kb
batches = [synth.make_batch(ref, 4000, seed=300 + i, mean_len=6000.0, max_len=max_len) for i in range(4)]
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, ref, 0, 1, 0, False)
for rep in range(4):
    for b in batches:
        runs.engine.stage_batch(b["paf"], b["seqs"], barcodes=None)
print("max read %d, aligned %.1f Mb per batch" % (max(max(b["read_lengths"].values()) for b in batches),
                                                 sum(b["aligned"] for b in batches) / len(batches) / 1e6))
