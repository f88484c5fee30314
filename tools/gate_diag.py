"""Where does the bench's parity gate differ from the oracle?  (diagnostic, GPU box)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import farneback as OF
wl = dict(bench.WORKLOADS["4k"])
plan = bench.make_plan(256, 16, 0, 1)
plan["frames"] = [0, 17]; plan["pass_starts"] = [0]
job = bench.Job(wl, 16, plan, 256, 2000, 0)
job.calc_pass(0); job.sync()
prev, nxt = job.passes[0]
for i in range(2):
    a, b = job.synth.frame(prev[i]), job.synth.frame(nxt[i])
    ref = OF.calc(a, b, levels=5)
    got = job.fb.get_flow(i)
    d = np.abs(got - ref).max(axis=2)
    tol = 1e-4 * max(1, np.abs(ref).max())
    ys, xs = np.nonzero(d > tol)
    print(f"pair {i}: max|ref| {np.abs(ref).max():.4f} max err {d.max():.3e} tol {tol:.3e} n_over {len(ys)}")
    if len(ys):
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
        iy, ix = np.unravel_index(d.argmax(), d.shape)
        print("  worst at", iy, ix, "ref", ref[iy, ix], "got", got[iy, ix])
    print("  percentiles", np.percentile(d, [50, 99, 99.99, 100]))
    # single-pair path
    from transflow_amd.farneback import Farneback
    fb1 = Farneback(3840, 2160, levels=5)
    one = fb1.calc(a, b)
    print("  single-pair path vs oracle", np.abs(one - ref).max(), " batch vs single", np.abs(one - got).max())
    fb1.close()
