"""Soak of the GPU path against the oracle over many seeded random scenarios (not part of the test
suite: ~5 minutes on the GPU box).  Every scenario runs 4 updates; masks, thresholds and benefits
must be bit-identical.  Odd seeds run the chain next to the sweep (BOSSX_OVERLAP=1), every third one
many short contigs (more chain blocks than a few CUs), every fifth long reads (large windows).
    python3 scripts/soak_parity.py [first_seed] [n]"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.runs import BossRuns
from oracle.pipeline import OracleRuns

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
os.chdir(tempfile.mkdtemp())
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    many = seed % 3 == 0
    n_contigs = int(rng.integers(6, 14)) if many else int(rng.integers(1, 4))
    lens = [int(rng.integers(100_000, 160_000 if many else 420_000)) for _ in range(n_contigs)]
    nb = int(rng.choice([1, 1, 2, 4]))
    ploidy = int(rng.choice([1, 2]))
    long_reads = seed % 5 == 0
    n_reads = int(rng.integers(300, 1200))
    mean_len = 12000.0 if long_reads else float(rng.choice([1500.0, 4000.0, 8000.0]))
    if seed % 2:
        os.environ["BOSSX_OVERLAP"] = "1"
    else:
        os.environ.pop("BOSSX_OVERLAP", None)
    names = ["s%d" % i for i in range(n_contigs)]
    contigs = synth.make_reference(lens, seed=seed, names=names)
    strs = [(nm, synth.codes_to_str(c)) for nm, c in contigs]
    args = BossConfig(); args.general.name = "soak%d" % seed
    args.optional.ploidy = ploidy; args.optional.bucket_threshold = 0
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = BossRuns(args); runs.write_masks = False
    runs.init(contigs=strs)
    o = OracleRuns(strs, ploidy=ploidy, nbarcodes=nb, bucket_threshold=0)
    ok = True
    for b in range(4):
        batch = synth.make_batch(contigs, n_reads, seed=seed * 100 + b, mean_len=mean_len, max_len=60000 if long_reads else 30000, nbarcodes=nb)
        bcs = batch["barcodes"] if nb > 1 else None
        try:
            o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"], barcodes=bcs)
            o_err = None
        except Exception as e:           # e.g. Bottleneck's window > contig
            o_err = type(e).__name__
        try:
            runs.rl_dist.update(batch["read_lengths"])
            runs.process_batch_paf(batch["paf"], batch["seqs"], barcodes=bcs)
            g_err = None
        except Exception as e:
            g_err = type(e).__name__
        if o_err or g_err:
            if (o_err is None) != (g_err is None):
                ok = False; print("seed %d batch %d: errors differ: oracle %s gpu %s" % (seed, b, o_err, g_err))
            break
        if runs.threshold != o.threshold:
            ok = False; print("seed %d batch %d: threshold %r vs %r" % (seed, b, runs.threshold, o.threshold))
        for nm, oc in o.contigs.items():
            pc = runs.contigs[nm]
            if not np.array_equal(pc.strat, oc.strat):
                ok = False; print("seed %d batch %d %s: masks differ at %d entries" % (seed, b, nm, int((pc.strat != oc.strat).sum())))
            if not oc.rej and oc.length >= 100_000 and not np.array_equal(pc.additional_benefit, oc.additional_benefit):
                ok = False; print("seed %d batch %d %s: benefits differ" % (seed, b, nm))
        if not ok:
            break
    bad += 0 if ok else 1
    runs.engine.close()
    print("seed %d: %s (%d contigs, nb %d, ploidy %d, mean read %d, %s)%s" % (
        seed, "ok" if ok else "MISMATCH", n_contigs, nb, ploidy, mean_len, "chain next to sweep" if seed % 2 else "serial",
        "" if not (o_err or g_err) else " [stopped at %s]" % (o_err or g_err)), flush=True)
print("%d scenarios, %d mismatching, %.0f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
