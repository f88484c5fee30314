#!/usr/bin/env python3
"""Every frame pair of bench.py's 4K clip (T = 256: 255 pairs) through the GPU path, against the CPU oracle.

bench.py's gate compares three pairs of the first pass with the oracle before anything is timed and 64 after it
(`parity_wide`); this tool spends a few GPU-minutes on all 255: the two passes the timed loop alternates between, in the
default mode (what is timed: no pixel may lie beyond 1e-4 * max(1, max|ref|)) and in the handles' exact mode (must equal
the oracle bit for bit), the oracle's flows made `threads` pairs at a time on the host's cores.  Then the remap
recurrence over the first `remap_frames` pairs of the clip in clip order (one compositor, as the reference's pipeline
runs it: pipeline.py:562-567) against the numpy oracle fed the GPU's flows and the GPU's own uniform fields: layer state
and every frame bit for bit.
usage (GPU box): python3 tools/clip_parity.py [threads=64] [remap_frames=24] > gpurun_out/r06_clip_parity.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import farneback as OF  # noqa: E402
from oracle import remap_ref as OR  # noqa: E402


def main():
    threads = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("threads=")), 64)
    remap_frames = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("remap_frames=")), 24)
    wl = bench.WORKLOADS["4k"]
    w, h, T, batch = wl["w"], wl["h"], 256, 128
    plan = bench.make_plan(T, batch, 0, 1)
    job = bench.Job(wl, batch, plan, T, seed=2000, device=0, lanes=1)
    frames = {}

    def frame(t):
        if t not in frames:
            frames[t] = job.synth.frame(t)
        return frames[t]

    t0 = time.time()
    got = np.empty((h, w, 2), np.float32)
    modes = {"default": dict(pairs=0, identical=0, differing_px=0, outliers=0, worst=0.0),
             "exact": dict(pairs=0, identical=0, differing_px=0, outliers=0, worst=0.0)}
    seen = set()
    for p, start in enumerate(plan["pass_starts"]):
        prev, nxt = job.passes[p]
        todo = [i for i in range(batch) if start + i not in seen]       # the second pass repeats one pair of the first
        for c0 in range(0, len(todo), threads):
            chunk = todo[c0:c0 + threads]
            refs = OF.calc_batch([frame(prev[i]) for i in chunk], [frame(nxt[i]) for i in chunk], threads, levels=wl["levels"])
            for mode, exact in (("default", False), ("exact", True)):
                job.fb.set_exact(True if exact else None)
                job.calc_pass(p)
                job.check(job.lib.tf_sync())
                m = modes[mode]
                for i, ref in zip(chunk, refs):
                    job.fb.get_flow_into(i, got)
                    m["pairs"] += 1
                    if np.array_equal(got, ref):
                        m["identical"] += 1
                        continue
                    d = np.abs(got - ref).max(axis=2)
                    tol = bench.TOL_REL * max(1.0, float(np.abs(ref).max()))
                    m["differing_px"] += int((d > 0).sum())
                    m["outliers"] += int((d > tol).sum()) + int((~np.isfinite(got)).sum())
                    m["worst"] = max(m["worst"], float(d.max()) / tol)
            job.fb.set_exact(None)
            seen.update(start + i for i in chunk)
            print(f"# pass {p}: pairs {start + chunk[0]} .. {start + chunk[-1]} done, {time.time() - t0:.0f} s", flush=True)
            for t in list(frames):
                if t < start + chunk[0]:
                    del frames[t]
    for mode, m in modes.items():
        print(f"{mode:8s}: {m['pairs']} pairs of the clip ({w}x{h}, levels={wl['levels']}): {m['identical']} bit-identical to the oracle, "
              f"{m['differing_px']} pixels differing in the others (of {(m['pairs'] - m['identical']) * w * h}), largest deviation "
              f"{m['worst']:.2e} of the tolerance, {m['outliers']} beyond it")
    ok = modes["default"]["outliers"] == 0 and modes["exact"]["identical"] == modes["exact"]["pairs"] == 255
    # the remap recurrence over the first pairs of the clip, one compositor in clip order
    from transflow_amd.device import DevBuffer
    from transflow_amd.remap import CompImage
    n = min(remap_frames, batch)
    job.calc_pass(0)
    job.check(job.lib.tf_sync())
    layer, comp = job.make_layer(), CompImage(h, w, (255, 255, 255))
    ora = OR.MoveRefLayer(h, w, OR.LayerParams(reset_mode="random", reset_random_factor=0.5), reset_mask=job.reset_mask,
                          introduction_masks=[np.ones((h, w), bool)])
    white = np.full((h, w, 3), 255, np.uint8)
    ubuf = DevBuffer(h * w * 8)
    same = 0
    for i in range(n):
        flow = job.fb.get_flow(i)
        layer.uniform_dev(bench.SEED_U, ubuf.ptr)
        u = ubuf.download((h, w), np.float64)
        job.remap_pair(layer, comp, i)
        data, rgba = layer.get_state()
        ora.update(OR.post_process(flow.copy(), wl["direction"]), [job.pixmap], u)
        exp = OR.composite(white, [ora.render()])
        same += bool(np.array_equal(data, ora.data) and np.array_equal(rgba, ora.rgba) and np.array_equal(comp.download(), exp))
    print(f"remap   : {same} of {n} consecutive frames of the clip (one recurrence from frame 0): layer state, rgba and frame "
          f"bit-identical to the numpy oracle fed the GPU's flows and uniform fields; out of frame: {bool(layer.out_of_frame())}")
    ok = ok and same == n
    print(f"# {'OK' if ok else 'FAILED'}, {time.time() - t0:.0f} s, oracle on {threads} threads of {os.cpu_count()}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
