"""Where the host time of a lone update goes: cProfile over thirty lone updates (after five), top functions by cumulative time.
   gpurun -- python3 scripts/lone_profile.py [workload]"""
import cProfile, os, pstats, sys, tempfile, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 1000 + i, 4000, nb) for i in range(35)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
R = bench.Runner(w, runs, nb, batches, False)
for b in batches[:5]:
    R.step_e2e(b)
pr = cProfile.Profile()
pr.enable()
for b in batches[5:]:
    R.step_e2e(b)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
