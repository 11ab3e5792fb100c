import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(25)])
os.chdir(tempfile.mkdtemp())
print(bench.late_regime_run(w, bench._GEN[w], 0, batches, 5, 20, True, depth=float(sys.argv[1])))
