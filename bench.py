#!/usr/bin/env python3
"""bench.py -- frames/sec of the Farnebäck + remap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload 4k|1080p|1080p-1level] [--batch B]

A "step" is one pass of the hot path over one batch of B synthetic frame pairs
per GPU: batched Farnebäck over the B pairs, then per pair, in stream order,
post_process -> moveref update (+reset) -> pixmap gather -> render.  Frames,
pixmap and masks are resident in HBM before the timed region; output frames
stay in HBM (the PCIe-inclusive rate is noted in DESIGN.md, never reported as
`value`).  For N > 1 the driver starts one rank per GPU with
torch.distributed.run; every rank runs its own shard of frames (independent
pairs, one remap stream per rank): weak scaling, no data-path
collective; RCCL carries the rendezvous, the one-off broadcast of the shared
pixmap/mask and the barrier + max-over-ranks timing.

Rank 0 prints ONE JSON line (see README/DESIGN for the fields).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: 1080p pair, one Farnebäck scale + remap
    "1080p-1level": dict(w=1920, h=1080, levels=0, direction=1, reset=False),
    # configs[2]: 1080p, full 5-level pyramid, FORWARD accumulator remap
    "1080p": dict(w=1920, h=1080, levels=5, direction=0, reset=False),
    # configs[3]/[4]: 4K, 5-level pyramid, moveref with random reset through a float mask
    "4k": dict(w=3840, h=2160, levels=5, direction=1, reset=True),
}


def synth_frames(h, w, count, seed):
    """multi-scale sine texture + noise; frame t is the texture seen
    through the smooth field t*(u,v)/count, with fresh noise per frame."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.4, 1.0, 6).astype(np.float32)
    fx = rng.uniform(0.004, 0.06, 6).astype(np.float32)
    fy = rng.uniform(0.004, 0.06, 6).astype(np.float32)
    ph = rng.uniform(0, 2 * np.pi, 6).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    u = (3.0 * np.sin(2 * np.pi * yy / h * 2)).astype(np.float32)
    v = (2.0 * np.cos(2 * np.pi * xx / w * 3)).astype(np.float32)
    frames = []
    for t in range(count):
        s = np.float32(t / max(1, count - 1))
        x, y = xx - s * u, yy - s * v
        val = np.zeros((h, w), np.float32)
        for m in range(6):
            val += a[m] * np.sin(np.float32(2 * np.pi) * (fx[m] * x + fy[m] * y) + ph[m])
        noise = np.random.default_rng(seed + 1 + t).standard_normal((h, w), dtype=np.float32) * 6
        frames.append(np.clip(np.rint(128 + 20 * val + noise), 0, 255).astype(np.uint8))
    return frames


class Job:
    """Everything one rank keeps resident for the timed loop."""

    def __init__(self, wl, batch, seed, device, pixmap=None, reset_mask=None):
        from transflow_amd import _lib
        from transflow_amd.farneback import Farneback
        from transflow_amd.remap import CompImage, RemapLayer
        self.lib = _lib.load()
        self.check = _lib.check
        self.wl, self.batch = wl, batch
        w, h = wl["w"], wl["h"]
        self.fb = Farneback(w, h, levels=wl["levels"], frame_slots=batch + 1, max_pairs=batch, device=device)
        for i, f in enumerate(synth_frames(h, w, batch + 1, seed)):
            self.fb.set_frame(i, f)
        rng = np.random.default_rng(1237)
        self.pixmap = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if pixmap is None else pixmap
        if wl["reset"]:
            rm = np.random.default_rng(1238).random((h, w), dtype=np.float32) if reset_mask is None else reset_mask
            self.layer = RemapLayer(h, w, reset_mode="random", reset_random_factor=0.5, reset_mask=rm)
        else:
            self.layer = RemapLayer(h, w)
        self.layer.set_sources([np.ones((h, w), np.uint8)])
        self.comp = CompImage(h, w, (255, 255, 255))
        p = C.c_void_p()
        self.check(self.lib.tf_dev_alloc(C.byref(p), self.pixmap.nbytes))
        self.pixmap_dev = p.value
        self.check(self.lib.tf_dev_upload(C.c_void_p(self.pixmap_dev), C.c_void_p(self.pixmap.ctypes.data),
                                          self.pixmap.nbytes))
        lo, hi = list(range(batch)), list(range(1, batch + 1))
        # FORWARD: (prev, next) = (previous, current); BACKWARD: (current, previous)  cv.py:467-472
        self.prev, self.next = (lo, hi) if wl["direction"] == 0 else (hi, lo)
        self.flow_ptrs = None

    def step(self):
        fb, layer, comp, d = self.fb, self.layer, self.comp, self.wl["direction"]
        fb.calc_slots(self.prev, self.next)
        base = fb.flow_ptr(0)                    # the result buffer alternates with the call's parity
        if self.flow_ptrs is None or self.flow_ptrs[0] != base:
            self.flow_ptrs = [base + i * self.wl["w"] * self.wl["h"] * 8 for i in range(self.batch)]
        for i in range(self.batch):
            if d == 0:
                # FORWARD post_process: the scatter pass here, the rest (source.py:359-362) inside the remap kernel
                layer.step_dev(comp, fb.post_process_scatter(i), self.pixmap_dev, 3, clip_flow=2, seed=20251003)
            else:
                # BACKWARD post_process is the clip alone: folded into the remap kernel
                layer.step_dev(comp, self.flow_ptrs[i], self.pixmap_dev, 3, clip_flow=True, seed=20251003)

    def sync(self):
        self.check(self.lib.tf_sync())

    def prof(self, on, flt=None):
        self.check(self.lib.tf_prof_set_filter(flt.encode() if flt else None))
        self.check(self.lib.tf_prof_enable(1 if on else 0))

    def prof_reset(self):
        self.check(self.lib.tf_prof_reset())

    def prof_report(self):
        buf = C.create_string_buffer(1 << 16)
        self.check(self.lib.tf_prof_report(buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out


def kernel_alg_bytes(name, wl, batch, iterations=3):
    """Algorithmic bytes one step moves through kernel `name` (all its launches)."""
    from transflow_amd import roofline as rf
    sizes = rf.level_sizes(wl["w"], wl["h"], 0.5, wl["levels"])
    n = [a * b for a, b in sizes]
    total = 0
    for k in range(len(n)):
        nc = n[k + 1] if k + 1 < len(n) else 0
        mult = 1
        if name in ("fb_level_rowpass", "fb_level_colpass") and k < 2:   # long blur kernels start at level 2 (pyr_scale 0.5)
            continue
        if name == "fb_flow_iter" and not rf.level_is_fused(n[k], batch):
            continue
        if name in ("fb_update_matrices", "fb_blur_solve") and rf.level_is_fused(n[k], batch):
            continue
        if name == "fb_blur_solve":
            mult = iterations
        elif name in ("fb_update_matrices", "fb_flow_iter"):
            # first launch of a level carries S3 (flow init); the I-1 rebuilds are plain S4 (+S5 when fused)
            total += (iterations - 1) * batch * (68 if name == "fb_update_matrices" else 96) * n[k]
        total += mult * rf.kernel_bytes(name, n[0], n[k], nc, batch)
    return total


def measured_traffic(name, wl, batch, launches_per_step, iterations=3):
    """HBM bytes per launch of kernel `name` from the PMC passes recorded in profiles/ (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, 2*FETCH + WRITE after calibration on this library's
    access widths -- profiles/README.md).  Bytes per level pixel measured at 4K level 0, scaled to the
    pixels this workload's launches cover.  None when no measurement exists for the kernel."""
    from transflow_amd import roofline as rf
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    if name not in table:
        return None
    n = [a * b for a, b in rf.level_sizes(wl["w"], wl["h"], 0.5, wl["levels"])]
    per_level_launches = {"fb_update_matrices": iterations, "fb_blur_solve": iterations,
                          "fb_flow_iter": iterations}.get(name, 1)
    if name == "fb_flow_iter":          # runs on the levels of >= 4M pixels over the batch only
        n = [v for v in n if rf.level_is_fused(v, batch)]
    images = 2 if name in ("fb_polyexp", "fb_level_image") else 1
    total = table[name]["bytes_per_px"] * sum(n) * batch * images * per_level_launches
    return total / max(1.0, launches_per_step)


def copy_ceiling(lib, check, nbytes=1 << 30, reps=8):
    """Measured rate (read + write bytes per second) of a 16-byte-per-lane copy kernel over 1 GiB, HIP
    events on the library stream: the practical HBM ceiling the 8 TB/s spec figure is quoted next to
    (SURVEY 8d)."""
    src, dst, e0, e1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.tf_dev_alloc(C.byref(src), nbytes))
    check(lib.tf_dev_alloc(C.byref(dst), nbytes))
    check(lib.tf_event_create(C.byref(e0)))
    check(lib.tf_event_create(C.byref(e1)))
    check(lib.tf_dev_stream_copy(dst, src, nbytes))
    check(lib.tf_event_record(e0))
    for _ in range(reps):
        check(lib.tf_dev_stream_copy(dst, src, nbytes))
    check(lib.tf_event_record(e1))
    ms = C.c_float()
    check(lib.tf_event_elapsed_ms(e0, e1, C.byref(ms)))
    lib.tf_event_destroy(e0)
    lib.tf_event_destroy(e1)
    check(lib.tf_dev_free(src))
    check(lib.tf_dev_free(dst))
    return 2.0 * nbytes * reps / (ms.value * 1e-3) / 1e9


def cpu_baseline(wl):
    """The oracle (a scalar C port of OpenCV's CPU path + the numpy remap), timed on
    this host on ONE frame pair of the same workload.  Checker code, used here only
    as the reported CPU baseline."""
    from oracle import farneback as OF
    from oracle import remap_ref as OR
    w, h = wl["w"], wl["h"]
    f = synth_frames(h, w, 2, 777)
    a, b = (f[0], f[1]) if wl["direction"] == 0 else (f[1], f[0])
    OF.lib()
    rng = np.random.default_rng(3)
    pixmap = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    prm = OR.LayerParams(reset_mode="random", reset_random_factor=0.5) if wl["reset"] else OR.LayerParams()
    layer = OR.MoveRefLayer(h, w, prm, reset_mask=rng.random((h, w), dtype=np.float32),
                            introduction_masks=[np.ones((h, w), bool)])
    white = np.full((h, w, 3), 255, np.uint8)
    t_fb = t_rm = 0.0
    n = 0
    while t_fb + t_rm < 12.0 and n < 64:   # bounded sample: ~12 s of CPU work
        t0 = time.perf_counter()
        flow = OF.calc(a, b, levels=wl["levels"])
        t1 = time.perf_counter()
        flow = OR.post_process(flow, wl["direction"])
        layer.update(flow, [pixmap], rng.random((h, w)))
        OR.composite(white, [layer.render()])
        t2 = time.perf_counter()
        t_fb += t1 - t0
        t_rm += t2 - t1
        n += 1
    return {"value": n / (t_fb + t_rm), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frame pair(s) of the same workload ({w}x{h}, levels={wl['levels']}): "
                      f"C port of OpenCV Farneback {t_fb / n:.2f} s/frame + numpy remap {t_rm / n:.2f} s/frame, "
                      "single thread; cv2 is not installed on this host"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", default=None, help="WxH: run the chosen workload's configuration at another frame size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed 1080p side measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    group = None
    if world > 1 or args.gpus > 1 or os.environ.get("TF_BENCH_FORCE_DIST"):  # the env var rehearses the RCCL path on 1 GPU
        if world != args.gpus and not os.environ.get("TF_BENCH_FORCE_DIST"):
            raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} ranks (torch.distributed.run); WORLD_SIZE={world}")
        from transflow_amd.batch import Group  # imports torch BEFORE libtfhip.so so one HIP runtime is shared
        group = Group("nccl")

    wl = dict(WORKLOADS[args.workload])
    if args.size:
        wl["w"], wl["h"] = (int(v) for v in args.size.lower().split("x"))
    pixmap = reset_mask = None
    if group is not None:
        # shared inputs come from rank 0 over RCCL (one-off broadcast, outside the timed region)
        rng = np.random.default_rng(1237)
        pixmap = group.broadcast_bytes(rng.integers(0, 256, (wl["h"], wl["w"], 3), dtype=np.uint8)
                                       if rank == 0 else np.zeros((wl["h"], wl["w"], 3), np.uint8))
        if wl["reset"]:
            reset_mask = group.broadcast_bytes(np.random.default_rng(1238).random((wl["h"], wl["w"]), dtype=np.float32)
                                               if rank == 0 else np.zeros((wl["h"], wl["w"]), np.float32))
    job = Job(wl, args.batch, seed=2000 + 17 * rank, device=local_rank, pixmap=pixmap, reset_mask=reset_mask)

    # warmup; the last warmup step is profiled per kernel to find the dominant one (the first step of a
    # process pays one-off costs inside whichever kernel happens to run first)
    n_warm = max(1, args.warmup)
    for i in range(n_warm):
        if i == n_warm - 1:
            job.sync()
            job.prof(True)
            job.prof_reset()
        job.step()
    job.sync()
    per_kernel = job.prof_report()
    job.prof(False)
    fb_kernels = {k: v for k, v in per_kernel.items() if k.startswith("fb_")}
    dominant = max(fb_kernels, key=lambda k: fb_kernels[k][1])
    job.prof_reset()
    job.prof(True, dominant)  # only the dominant kernel is bracketed by events in the timed region

    if group is not None:
        group.barrier()
    job.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.sync()
    t_sync = time.perf_counter() - t0
    if group is not None:
        group.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("TF_BENCH_DEBUG"):
        print(f"[bench] rank {rank}: steps done after {t_sync * 1e3:.2f} ms, closing barrier {(elapsed - t_sync) * 1e3:.2f} ms",
              file=sys.stderr)
    if group is not None:
        elapsed = group.max_over_ranks(elapsed)
    job.prof(False)
    dom_cnt, dom_ms = job.prof_report()[dominant]

    if rank != 0:
        if group is not None:
            group.close()
        return

    from transflow_amd import roofline as rf
    frames = args.steps * args.batch * max(1, world)
    fps = frames / elapsed
    alg_dom = kernel_alg_bytes(dominant, wl, args.batch) * args.steps
    achieved = alg_dom / (dom_ms * 1e-3) / 1e9
    traffic = measured_traffic(dominant, wl, args.batch, dom_cnt / max(1, args.steps))
    step_bytes = args.batch * (rf.farneback_bytes(wl["w"], wl["h"], 0.5, wl["levels"], 3)
                               + rf.remap_bytes(wl["w"], wl["h"], reset_mask=wl["reset"], forward=wl["direction"] == 0))
    out = {
        "metric": "frames/sec (Farneback+remap)", "value": fps, "unit": "frames/s", "n_gpus": max(1, world),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['w']}x{wl['h']} uint8 frame pairs, Farneback pyr_scale=0.5 "
                               f"levels={wl['levels']} winsize=15 iterations=3 poly_n=5 poly_sigma=1.2 flags=0, "
                               f"{'FORWARD' if wl['direction'] == 0 else 'BACKWARD'} post_process, moveref layer "
                               f"(reset {'random p=0.5 through a float mask, u drawn on the GPU' if wl['reset'] else 'off'}), "
                               "1 RGB pixmap source, render",
                   "frame_pairs_per_step_per_gpu": args.batch,
                   "frames_per_step_per_gpu": args.batch + 1,
                   "frame_expansions": "pairs t and t+1 of a step share frame t+1: its pyramid levels and polynomial "
                                       "expansion (A1+A2, functions of the frame alone) are computed once per step and "
                                       "read by both; nothing is kept between steps (TF_FB_NO_SHARE=1 expands per pair "
                                       "and side: other_workloads_untimed_region.unshared_expansions)",
                   "parallelism": f"frames sharded over {max(1, world)} GPU(s), one remap stream per GPU"},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": rf.HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / rf.HBM_PEAK_GBS, "traffic": traffic,
                     "launches": dom_cnt, "avg_launch_ms": dom_ms / max(1, dom_cnt),
                     "algorithmic_bytes_per_launch": alg_dom / max(1, dom_cnt),
                     "whole_step": {"algorithmic_bytes": step_bytes,
                                    "achieved": step_bytes * args.steps / elapsed / 1e9,
                                    "frac": step_bytes * args.steps / elapsed / 1e9 / rf.HBM_PEAK_GBS}},
        "kernels_ms_per_step": {k: round(v[1], 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl)
    if world == 1 and not args.no_extra:
        # the same kernel with nothing beside it: a second handle with everything on one stream (in the
        # timed region above the caller's remap of one call may run beside the next call's kernels)
        os.environ["TF_FB_NO_OVERLAP"] = "1"
        try:
            j = Job(wl, args.batch, seed=2000, device=local_rank)
        finally:
            del os.environ["TF_FB_NO_OVERLAP"]
        for _ in range(2):
            j.step()
        j.sync()
        j.prof_reset()
        j.prof(True, dominant)
        n = 5
        for _ in range(n):
            j.step()
        j.sync()
        j.prof(False)
        cnt, ms = j.prof_report()[dominant]
        ach = kernel_alg_bytes(dominant, wl, args.batch) * n / (ms * 1e-3) / 1e9
        out["roofline"]["measured_copy_ceiling_GBs"] = copy_ceiling(job.lib, job.check)
        out["roofline"]["exclusive"] = {"what": "same kernel, same workload, TF_FB_NO_OVERLAP=1 (one stream, nothing runs beside it); untimed region",
                                        "launches": cnt, "avg_launch_ms": ms / max(1, cnt), "achieved": ach,
                                        "frac": ach / rf.HBM_PEAK_GBS}
        del j
        extra = {}
        for name in ("1080p", "1080p-1level"):
            if name == args.workload:
                continue
            w2 = WORKLOADS[name]
            # the same bytes per call as the main workload: more pairs of the smaller frames
            b2 = max(1, min(64, args.batch * (wl["w"] * wl["h"]) // (w2["w"] * w2["h"])))
            j = Job(w2, b2, seed=2000, device=local_rank)
            for _ in range(3):
                j.step()
            j.sync()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                j.step()
            j.sync()
            dt = time.perf_counter() - t0
            sb = b2 * (rf.farneback_bytes(w2["w"], w2["h"], 0.5, w2["levels"], 3)
                       + rf.remap_bytes(w2["w"], w2["h"], reset_mask=w2["reset"], forward=w2["direction"] == 0))
            extra[name] = {"frames_per_s": n * b2 / dt, "frame_pairs_per_step": b2,
                           "whole_step_frac_of_8TBs": sb * n / dt / 1e9 / 8000.0}
            del j
        # the main workload with one expansion per pair and side, as 16 separate calls would do them
        os.environ["TF_FB_NO_SHARE"] = "1"
        try:
            for _ in range(2):
                job.step()
            job.sync()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                job.step()
            job.sync()
            extra["unshared_expansions"] = {"frames_per_s": n * args.batch / (time.perf_counter() - t0),
                                            "what": f"{args.workload}, TF_FB_NO_SHARE=1"}
        finally:
            del os.environ["TF_FB_NO_SHARE"]
        out["other_workloads_untimed_region"] = extra
    print(json.dumps(out))
    if group is not None:
        group.close()


if __name__ == "__main__":
    main()
