import sys, importlib.util
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fz", "tools/fuzz_traversal.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import time
for seed in fz.RESIDUAL_SEEDS[:int(sys.argv[1])]:
    t0=time.time(); rows, un = fz.residual(seed)
    print(seed, "unexcused", un, "rows", len(rows), f"{time.time()-t0:.1f}s")
    for r in rows: print("   ", r)
