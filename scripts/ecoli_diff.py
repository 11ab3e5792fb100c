"""Where do the GPU and the CPU port differ after many E. coli updates?  (run on the GPU box)
    python3 scripts/ecoli_diff.py [n_updates ...]"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from boss_runs_amd import synth
from oracle.pipeline import OracleRuns

w = "ecoli"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 100 + i, 4000, 1) for i in range(12)])
os.chdir(tempfile.mkdtemp())
for n_updates in [int(x) for x in sys.argv[1:]] or [20, 60, 90]:
    runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
    for i in range(n_updates):
        b = batches[i % 10]
        runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"])
    contigs = bench._GEN[w]
    o = OracleRuns([(n, synth.codes_to_str(c)) for n, c in contigs], ploidy=1, nbarcodes=1, bucket_threshold=0)
    for name, oc in o.contigs_filt.items():
        pc = runs.contigs[name]
        oc.coverage[:] = pc.coverage; oc.scores[:] = pc.scores
        oc.bucket_switches[:] = pc.bucket_switches; oc.switched_on[:] = pc.switched_on; oc.strat[:] = pc.strat
        o.read_starts.read_starts[name][:] = runs.read_starts.read_starts[name]
    o.rl_dist.read_lengths[:] = runs.rl_dist.read_lengths
    b = batches[11]
    rl = dict(zip(b["seqs"].keys(), b["read_lengths_arr"].tolist()))
    o.process_batch(b["paf"], b["seqs"], read_lengths=rl)
    runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"])
    name = list(o.contigs_filt)[0]
    pc, oc = runs.contigs[name], o.contigs_filt[name]
    cov_eq = np.array_equal(pc.coverage, oc.coverage)
    sc = np.asarray(pc.scores); so = np.asarray(oc.scores)
    nsc = int((sc != so).sum())
    print("updates %d: depth max %d mean %.1f | coverage equal %s | scores differ at %d sites | threshold gpu %r cpu %r | strat differ %d of %d" % (
        n_updates, int(oc.coverage.sum(axis=1).max()), float(oc.coverage.sum() / oc.coverage.shape[0]), cov_eq, nsc,
        runs.threshold, o.threshold, int((pc.strat != oc.strat).sum()), pc.strat.size), flush=True)
    if nsc:
        idx = np.nonzero((sc != so).reshape(len(sc), -1).any(axis=1))[0][:5]
        for i in idx:
            print("   site %d cov %s gpu %r cpu %r" % (i, oc.coverage[i].ravel().tolist(), sc[i].ravel().tolist(), so[i].ravel().tolist()))
    runs.engine.close()
