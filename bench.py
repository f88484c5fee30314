#!/usr/bin/env python3
"""bench.py -- frames/sec of the Farnebäck + remap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload 4k|1080p|1080p-1level] [--batch B]

The workload is BASELINE.json configs[4] (and configs[3] at N = 1): ONE clip of T = 256 synthetic 4K
frames, its 255 frame pairs sharded over the N ranks (`shard_range`, one-frame halo).  A "step" is
one pass of the hot path over one batch of B consecutive pairs of the rank's shard: batched
Farnebäck over the B pairs, then per pair, in stream order, post_process -> moveref update
(+ random reset) -> pixmap gather -> render (BACKWARD workloads: the B remap steps as one
tf_remap_steps_dev call -- the same state and frames as B single calls, the parity gate compares
them); consecutive steps walk down the shard and start over at its end.  The rank's frames, the pixmap and the masks are resident in HBM before the timed
region; output frames stay in HBM (the PCIe-inclusive rate is noted in DESIGN.md, never `value`).

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (fresh child
processes, before this process makes any GPU call); under `python -m torch.distributed.run ...
bench.py --gpus N` the ranks already exist and RANK / LOCAL_RANK / WORLD_SIZE come from the
environment.  Either way no rank imports torch: the ranks meet through a rendezvous file + TCP star
(transflow_amd.batch.HostGroup: barrier, max over ranks), the shared pixmap and reset mask come from
rank 0 through RCCL (tf_batch_broadcast), and the gather of finished frames to rank 0
(tf_batch_gather), and of the clip's flows to one remap recurrence on rank 0 (tf_batch_gather_at,
"flows to root"), are exercised and timed after the timed region -- results stay per rank in the
timed region, the path has no data-path collective.  Weak scaling: every rank does
one batch per step.

Before anything is timed, rank 0 runs the parity gate: two pairs of the first batch through the
very calls the timed loop makes, against the CPU oracle; no value is printed if it fails.

Rank 0 prints ONE JSON line (README / DESIGN.md §5 describe the fields).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

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
SEED_U = 20251003          # seed of the on-GPU uniform field of the random reset
TOL_REL = 1e-4             # north_star: float32 (u, v) within 1e-4 relative


class ClipSynth:
    """frame t of a clip of `count` frames = a multi-scale sine texture seen through the
    smooth field s*(u, v), s = t/(count-1), u = 3 sin(4 pi y/h), v = 2 cos(6 pi x/w), plus fresh noise.
    u depends on y alone and v on x alone, so every sine's argument splits into A(x) + B(y) and the
    texture is one [h, 12] x [12, w] product per frame.  Frame t depends on (seed, t, count) only:
    every rank makes exactly the frames of its shard."""

    def __init__(self, h, w, count, seed):
        self.h, self.w, self.count, self.seed = h, w, count, seed
        rng = np.random.default_rng(seed)
        self.a = rng.uniform(0.4, 1.0, 6)
        self.fx = rng.uniform(0.004, 0.06, 6)
        self.fy = rng.uniform(0.004, 0.06, 6)
        self.ph = rng.uniform(0, 2 * np.pi, 6)
        self.x = np.arange(w, dtype=np.float64)
        self.y = np.arange(h, dtype=np.float64)
        self.u = 3.0 * np.sin(2 * np.pi * self.y / h * 2)     # x displacement, a function of y
        self.v = 2.0 * np.cos(2 * np.pi * self.x / w * 3)     # y displacement, a function of x

    def frame(self, t):
        s = t / max(1, self.count - 1)
        two_pi = 2 * np.pi
        left = np.empty((self.h, 12), np.float32)
        right = np.empty((12, self.w), np.float32)
        for m in range(6):
            # sin(2 pi (fx (x - s u(y)) + fy (y - s v(x))) + ph) = sin(A(x) + B(y)),  A carries the phase
            A = two_pi * (self.fx[m] * self.x - self.fy[m] * s * self.v) + self.ph[m]
            B = two_pi * (self.fy[m] * self.y - self.fx[m] * s * self.u)
            left[:, 2 * m], left[:, 2 * m + 1] = self.a[m] * np.cos(B), self.a[m] * np.sin(B)
            right[2 * m], right[2 * m + 1] = np.sin(A), np.cos(A)
        val = left @ right
        noise = np.random.default_rng(self.seed + 1 + t).standard_normal((self.h, self.w), dtype=np.float32)
        val *= 20.0
        val += 128.0
        noise *= 6.0
        val += noise
        np.rint(val, out=val)
        np.clip(val, 0, 255, out=val)
        return val.astype(np.uint8)

    def frames(self, lo, hi):
        """Frames lo..hi-1, made on a few threads (numpy releases the GIL in the heavy parts)."""
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(8, (os.cpu_count() or 2) // 2, hi - lo))
        with ThreadPoolExecutor(workers) as ex:
            return list(ex.map(self.frame, range(lo, hi)))


def pick_lanes(w, h, pairs_per_pass):
    """One lane or two for a rank's passes.  A second lane runs the part-empty launches of one batch -- the coarse levels,
    the whole-column marches of levels whose columns of workgroups do not fill the chip's 768 slots -- beside the full ones
    of the other.  From ~64 pairs of 4K per pass (530 M level-0 pixels) every level down to 1/4 scale fills the chip by
    itself and there is nothing left to fill: 4K x 64 runs 1922 frames/s on one lane, 1905 on two (two batches in flight
    only share the chip); 4K x 32, a rank's whole shard at 8 GPUs, 1776 on one and 1883 on two; 1080p x 64 6763 and 6984
    (profiles/r05_batch_sweep_*.txt, r05_bench_4k_*)."""
    return 1 if pairs_per_pass * w * h >= 64 * 3840 * 2160 else 2


def make_plan(total_frames, batch, rank, world, equal=False):
    """Which pairs, frames and passes rank `rank` of `world` owns.  Shards differ by at
    most one pair, so the ranks' passes may differ in length (T = 256 over 8 ranks: 32 pairs on seven
    ranks, 31 on the last); `equal` (--equal-batches) trims every rank's pass to the shortest."""
    from transflow_amd.batch import batch_starts, frames_needed, pass_pairs, shard_range
    pairs = shard_range(max(0, total_frames - 1), rank, world)
    frames = frames_needed(pairs)
    n = pairs[1] - pairs[0]
    per_pass = pass_pairs(max(0, total_frames - 1), batch, world, equal)[rank]
    return {"rank": rank, "pairs": list(pairs), "frames": list(frames), "n_pairs": n,
            "pairs_per_pass": per_pass, "pass_starts": batch_starts(n, per_pass) if per_pass else []}


class Job:
    """Everything one rank keeps resident for the timed loop."""

    def __init__(self, wl, batch, plan, clip_frames, seed, device, pixmap=None, reset_mask=None, pixmap_dev=None, lanes=2):
        from transflow_amd import _lib
        from transflow_amd.farneback import Farneback
        from transflow_amd.remap import CompImage
        self.lib = _lib.load()
        self.check = _lib.check
        self.wl, self.plan = wl, plan
        w, h = wl["w"], wl["h"]
        f0, f1 = plan["frames"]
        self.batch = plan["pairs_per_pass"]
        if self.batch < 1:
            raise SystemExit(f"rank {plan['rank']}: the clip's {clip_frames - 1} pairs do not reach this rank")
        # two lanes: consecutive batches go to two handles on two call streams and are in flight together (tfhip.h,
        # tf_fb_create_lane): the part-empty launches of one batch run beside the full ones of the other
        self.fb = Farneback(w, h, levels=wl["levels"], frame_slots=f1 - f0, max_pairs=self.batch, device=device, lanes=lanes)
        self.synth = ClipSynth(h, w, clip_frames, seed)
        for i, f in enumerate(self.synth.frames(f0, f1)):
            self.fb.set_frame(i, f)
        self.pixmap = np.random.default_rng(1237).integers(0, 256, (h, w, 3), dtype=np.uint8) if pixmap is None else pixmap
        self.reset_mask = None
        if wl["reset"]:
            self.reset_mask = (np.random.default_rng(1238).random((h, w), dtype=np.float32)
                               if reset_mask is None else reset_mask)
        self.layer = self.make_layer()
        # pair i of a pass paints its own image: the pass's frames stay in HBM, side by side in one buffer,
        # until the next pass (what ONE gather to rank 0 sends)
        from transflow_amd.device import DevBuffer
        self.frame_bytes = h * w * 3
        self.out_frames = DevBuffer(self.batch * self.frame_bytes)
        self.comps = [CompImage(h, w, (255, 255, 255), image_dev=self.out_frames.ptr + i * self.frame_bytes)
                      for i in range(self.batch)]
        if pixmap_dev is None:
            p = C.c_void_p()
            self.check(self.lib.tf_dev_alloc(C.byref(p), self.pixmap.nbytes))
            pixmap_dev = p.value
            self.check(self.lib.tf_dev_upload(C.c_void_p(pixmap_dev), C.c_void_p(self.pixmap.ctypes.data),
                                              self.pixmap.nbytes))
        self.pixmap_dev = pixmap_dev
        # slot lists of every pass.  FORWARD: (prev, next) = (frame t, frame t+1); BACKWARD: swapped (cv.py:467-472)
        self.passes = []
        for s in plan["pass_starts"]:
            lo, hi = list(range(s, s + self.batch)), list(range(s + 1, s + self.batch + 1))
            self.passes.append((lo, hi) if wl["direction"] == 0 else (hi, lo))
        self.n_steps = 0
        self.flow_ptrs = None

    def make_layer(self):
        from transflow_amd.remap import RemapLayer
        h, w = self.wl["h"], self.wl["w"]
        if self.wl["reset"]:
            layer = RemapLayer(h, w, reset_mode="random", reset_random_factor=0.5, reset_mask=self.reset_mask)
        else:
            layer = RemapLayer(h, w)
        layer.set_sources([np.ones((h, w), np.uint8)])
        return layer

    def remap_pair(self, layer, comp, i):
        if self.wl["direction"] == 0:
            # FORWARD post_process: the scatter pass here, the rest (source.py:359-362) inside the remap kernel
            layer.step_dev(comp, self.fb.post_process_scatter(i), self.pixmap_dev, 3, clip_flow=2, seed=SEED_U)
        else:
            # BACKWARD post_process is the clip alone: folded into the remap kernel
            layer.step_dev(comp, self.flow_ptrs[i], self.pixmap_dev, 3, clip_flow=True, seed=SEED_U)

    def calc_pass(self, which):
        prev, nxt = self.passes[which % len(self.passes)]
        self.fb.calc_slots(prev, nxt)
        base = self.fb.flow_ptr(0)                    # the result buffer alternates with the call's parity
        if self.flow_ptrs is None or self.flow_ptrs[0] != base:
            self.flow_ptrs = [base + i * self.wl["w"] * self.wl["h"] * 8 for i in range(self.batch)]

    def remap_pass(self, layer):
        """the remap recurrence over the pass's flows, in order.  BACKWARD: the steps of the pass as ONE call
        (tf_remap_steps_dev: the same state and frames as the single calls, which the parity gate and the GPU tests
        compare it with; the library then knows which of a step's stores the next step overwrites unread).  FORWARD:
        a pair's winner map is made beside its flow just before its step (one map per handle), so step by step."""
        if self.wl["direction"] == 0:
            for i in range(self.batch):
                self.remap_pair(layer, self.comps[i], i)
        else:
            layer.steps_dev(self.comps, self.flow_ptrs, self.pixmap_dev, 3, clip_flow=True, seed=SEED_U)

    def step(self):
        self.calc_pass(self.n_steps)
        self.n_steps += 1
        self.remap_pass(self.layer)

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


def gate_pairs(batch, n_check=2):
    """Which pairs of a batch the gate compares with the oracle: the first and the LAST (n_check = 2).  In the
    one-kernel iteration the pairs of a launch are dealt to the eight XCDs' ticket lists in contiguous runs: pairs 0
    and 1 -- what rounds 1-4 checked -- both sit in list 0, the first and the last pair in lists 0 and 7."""
    n_check = min(n_check, batch)
    if n_check <= 1:
        return [0][:n_check]
    return sorted({round(j * (batch - 1) / (n_check - 1)) for j in range(n_check)})


def parity_gate(job, n_check=3):
    """The calls the timed loop makes (one batched Farnebäck pass over the first batch, shared expansions;
    then the fused remap step per pair with the uniform drawn on the GPU) against the CPU oracle on
    `n_check` pairs of the batch: the first, the middle one and the last (gate_pairs; round 6 -- rounds 1-5 checked
    two).  The exact mode's leg stays at the first and the last.

    Flow, default mode (what is timed): EVERY pixel within TOL_REL * max(1, max|ref|) of the oracle.  (Rounds 2 and 3
    excused a few border pixels -- FarnebackUpdateMatrices' in-frame test is discontinuous in the flow and the kernels'
    per-segment window sums decided it differently from OpenCV's image-long running sums.  Since round 4 the kernels keep
    OpenCV's column sums and compute M without FMA contraction: the allowance is gone; `flow_pixels_differing` counts the
    pixels that differ from the oracle at all.)  With two lanes the same pairs are then computed by the other lane and
    must equal the first lane's BIT FOR BIT.
    Flow, exact mode: the same pairs again with option fb_exact_sums (the window summed in OpenCV's own order along the
    rows too) must equal the oracle BIT FOR BIT.
    Remap: layer state, rgba and frame bit-exact for the GPU's own flow and the very uniform field the kernel drew
    (tf_remap_uniform_dev).  Returns (report, oracle timings) -- the oracle work doubles as the one-thread CPU
    baseline sample."""
    from oracle import farneback as OF
    from oracle import remap_ref as OR
    from transflow_amd.device import DevBuffer
    from transflow_amd.remap import CompImage
    wl = job.wl
    w, h = wl["w"], wl["h"]
    idx = gate_pairs(job.batch, n_check)
    n_check = len(idx)
    OF.lib()
    job.calc_pass(0)
    job.sync()
    f0 = job.plan["frames"][0]
    prev, nxt = job.passes[0]
    layer = job.make_layer()
    comp = CompImage(h, w, (255, 255, 255))
    prm = OR.LayerParams(reset_mode="random", reset_random_factor=0.5) if wl["reset"] else OR.LayerParams()
    ora = OR.MoveRefLayer(h, w, prm, reset_mask=job.reset_mask, introduction_masks=[np.ones((h, w), bool)])
    white = np.full((h, w, 3), 255, np.uint8)
    ubuf = DevBuffer(h * w * 8)
    rep = {"pairs_checked": n_check, "pairs": idx, "flow_max_abs_err": 0.0, "flow_tol": 0.0, "flow_pixels": n_check * h * w,
           "outliers_default": 0, "flow_pixels_differing": 0, "lanes_bit_identical": True,
           "outliers_exact": 0, "exact_max_abs_err": 0.0, "exact_bit_identical": True,
           "flow_ok": True, "remap_bit_exact": True,
           "what": "the first, the middle and the last pair of the first batch (`pairs`: ticket lists 0, 4 and 7 of the one-kernel "
                   "iteration's launches) through the timed loop's own calls vs the CPU oracle.  flow_ok = NO pixel "
                   f"beyond {TOL_REL:g} * max(1, max|ref|) (outliers_default == 0; no allowance), the other lane's flow of "
                   "the same pairs bit-identical to the first lane's, AND the first and the last of them (`exact_pairs`) in the "
                   "handle's exact mode (tf_fb_set_exact: the window summed in OpenCV's own order) bit-identical to the oracle"}
    t_fb = t_rm = 0.0
    refs, firsts = [], []
    for i in idx:
        a, b = job.synth.frame(f0 + prev[i]), job.synth.frame(f0 + nxt[i])
        t0 = time.perf_counter()
        ref = OF.calc(a, b, levels=wl["levels"])
        t_fb += time.perf_counter() - t0
        refs.append(ref)
        got = job.fb.get_flow(i)
        d = np.abs(got - ref).max(axis=2)
        err = float(d.max())
        tol = TOL_REL * max(1.0, float(np.abs(ref).max()))
        over = int((d > tol).sum())
        rep["flow_max_abs_err"] = max(rep["flow_max_abs_err"], err)
        rep["flow_tol"] = max(rep["flow_tol"], tol)
        rep["outliers_default"] += over
        rep["flow_pixels_differing"] += int((d > 0).sum())
        rep["flow_ok"] = rep["flow_ok"] and over == 0 and bool(np.isfinite(got).all())
        firsts.append(got)
        # remap: the oracle is fed the GPU's flow and the GPU's uniform field, so every integer must agree
        layer.uniform_dev(SEED_U, ubuf.ptr)
        u = ubuf.download((h, w), np.float64)
        job.remap_pair(layer, comp, i)
        data, rgba = layer.get_state()
        frame = comp.download()
        t0 = time.perf_counter()
        flow = OR.post_process(got.copy(), wl["direction"])
        ora.update(flow, [job.pixmap], u)
        exp = OR.composite(white, [ora.render()])
        t_rm += time.perf_counter() - t0
        same = np.array_equal(data, ora.data) and np.array_equal(rgba, ora.rgba) and np.array_equal(frame, exp)
        rep["remap_bit_exact"] = rep["remap_bit_exact"] and bool(same)
    # the same batch through the other lane (with one lane: the same handle again, a replay)
    job.calc_pass(0)
    job.sync()
    for j, i in enumerate(idx):
        rep["lanes_bit_identical"] = rep["lanes_bit_identical"] and bool(np.array_equal(job.fb.get_flow(i), firsts[j]))
    job.gate_flows = dict(zip(idx, firsts))      # pass 0, synchronised: what the timed region's re-check compares with
    job.gate_oracle = {"flow_max_abs_err": rep["flow_max_abs_err"], "flow_tol": rep["flow_tol"],
                       "flow_pixels_differing": rep["flow_pixels_differing"], "outliers": rep["outliers_default"]}
    # the first and the last of them with the window summed in OpenCV's own order (the handle's mode, tf_fb_set_exact)
    ex = sorted({0, n_check - 1})                # positions in idx / refs
    rep["exact_pairs"] = [idx[j] for j in ex]
    job.fb.set_exact(True)
    try:
        job.fb.calc_slots([prev[idx[j]] for j in ex], [nxt[idx[j]] for j in ex])
        job.sync()
        for k, j in enumerate(ex):
            got = job.fb.get_flow(k)
            d = np.abs(got - refs[j]).max(axis=2)
            tol = TOL_REL * max(1.0, float(np.abs(refs[j]).max()))
            rep["outliers_exact"] += int((d > tol).sum())
            rep["exact_max_abs_err"] = max(rep["exact_max_abs_err"], float(d.max()))
            rep["exact_bit_identical"] = rep["exact_bit_identical"] and bool(np.array_equal(got, refs[j]))
    finally:
        job.fb.set_exact(None)
    # the pass's remap steps as ONE call (what the timed loop does on a BACKWARD workload) against the same steps one by one:
    # layer state, rgba and every frame must be the same bytes
    if wl["direction"] != 0 and job.batch >= 2:
        m = min(job.batch, 3)
        job.calc_pass(0)
        one, many = job.make_layer(), job.make_layer()
        singles = []
        for i in range(m):
            job.remap_pair(one, comp, i)
            singles.append(comp.download())
        many.steps_dev(job.comps[:m], job.flow_ptrs[:m], job.pixmap_dev, 3, clip_flow=True, seed=SEED_U)
        same = all(np.array_equal(singles[i], job.comps[i].download()) for i in range(m))
        (d1, r1), (d2, r2) = one.get_state(), many.get_state()
        rep["remap_steps_call_equals_single_steps"] = bool(same and np.array_equal(d1, d2) and np.array_equal(r1, r2))
        rep["remap_bit_exact"] = bool(rep["remap_bit_exact"] and rep["remap_steps_call_equals_single_steps"])
        one.close()
        many.close()
    rep["flow_pixels_over_tol"] = rep["outliers_default"]
    rep["flow_ok"] = bool(rep["flow_ok"] and rep["exact_bit_identical"] and rep["lanes_bit_identical"])
    rep["out_of_frame"] = bool(layer.out_of_frame())
    rep["ok"] = bool(rep["flow_ok"] and rep["remap_bit_exact"] and not rep["out_of_frame"])
    ubuf.close()
    layer.close()
    comp.close()
    return rep, {"pairs": n_check, "farneback_s": t_fb, "remap_s": t_rm}


def timed_region_recheck(job, lanes):
    """Are the flows of the overlapped timed region the flows the synchronised gate vouched for?  The flows of the LAST
    timed step (first and last pair of its batch) are still in its lane's result buffer when the region ends: they are
    brought down, the same pass is computed again with nothing beside it -- once per lane, a synchronisation after each,
    exactly how the gate ran pass 0 -- and all must agree bit for bit; where the last step happened to be pass 0 again,
    also with the flows the gate compared with the oracle.  No oracle work: a few milliseconds."""
    last = job.n_steps - 1
    which = last % len(job.passes)
    gate = getattr(job, "gate_flows", None)
    idx = sorted(gate) if gate else gate_pairs(job.batch, 3)      # the gate's own pairs (rank 0); the same rule elsewhere
    timed = [job.fb.get_flow(i) for i in idx]
    same = True
    for _ in range(max(1, lanes)):
        job.calc_pass(which)
        job.sync()
        for i, t in zip(idx, timed):
            same = same and bool(np.array_equal(job.fb.get_flow(i), t))
    out = {"step": last, "pass": which, "pairs": idx, "lanes_recomputed": max(1, lanes),
           "equals_synchronised_recomputation": bool(same),
           "what": "flows of the last step of the overlapped timed region vs the same pass computed alone on each lane"}
    if which == 0 and gate is not None:
        out["equals_gate_flows"] = bool(all(np.array_equal(gate[i], t) for i, t in zip(idx, timed)))
        if out["equals_gate_flows"]:
            # the timed region's flows ARE the arrays the gate compared with the oracle: its verdict holds for them as it stands
            out["vs_oracle"] = dict(getattr(job, "gate_oracle", {}),
                                    what="the last timed step's flows of these pairs are bit for bit the arrays the parity gate "
                                         "compared with the CPU oracle before the timed region: this is their distance from "
                                         "the oracle (max |d|, the tolerance it was held to, pixels that differ at all, pixels "
                                         "beyond the tolerance)")
    out["ok"] = bool(same and out.get("equals_gate_flows", True))
    return out


def host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "usable_cores": usable}


def mem_available_bytes():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return None


def cpu_baseline(job, gate_times):
    """The CPU side of the same workload on this host: the oracle (scalar C port of OpenCV's CPU path +
    the numpy remap) on one thread -- the sample is the parity gate's own oracle work -- and the C port on
    every usable core (one frame pair per OpenMP thread: pairs are independent; fewer threads only where the
    host's free memory does not hold one pair's working set per core, and the line says so).  cv2 is timed
    too where it exists.  Checker code, used here only as the reported baseline."""
    from oracle import farneback as OF
    wl = job.wl
    host = host_description()
    n1 = gate_times["pairs"]
    one = n1 / (gate_times["farneback_s"] + gate_times["remap_s"])
    # working set of one pair in the C port: level image, blur scratch, two flows, M, two expansions + the result
    per_pair = wl["w"] * wl["h"] * 4 * (1 + 1 + 4 + 5 + 10 + 2) + (64 << 20)
    # one pair per thread costs the same wall time whatever the thread count until the memory system saturates
    # (EPYC 9575F: 32 threads 27 s, 256 threads 125 s for 256 pairs): 64 keeps the sample inside the run's budget;
    # TF_BENCH_CPU_THREADS=0 takes every usable core (the line says how many were used either way)
    want = int(os.environ.get("TF_BENCH_CPU_THREADS", 64))
    cores = host["usable_cores"] if want <= 0 else min(host["usable_cores"], want)
    avail = mem_available_bytes()
    if avail is not None:
        cores = min(cores, max(1, int(0.6 * avail / per_pair)))
    cores = max(1, cores)
    f0 = job.plan["frames"][0]
    prev, nxt = job.passes[0]
    have = {}

    def frame(slot):
        if slot not in have:
            have[slot] = job.synth.frame(f0 + slot)
        return have[slot]

    n = cores                                     # one pair per thread; the batch's pairs repeat beyond its length
    fa = [frame(prev[i % job.batch]) for i in range(n)]
    fb_ = [frame(nxt[i % job.batch]) for i in range(n)]
    t0 = time.perf_counter()
    sample_flows = OF.calc_batch(fa, fb_, cores, levels=wl["levels"])
    t_all = time.perf_counter() - t0
    # The sample's oracle flows are those of the first min(cores, pairs per pass) pairs of pass 0: compared with the GPU's
    # flows of the same pass they widen the in-run parity check from the gate's three pairs at no extra oracle time.
    wide = None
    try:
        m = min(n, job.batch)
        job.calc_pass(0)
        job.sync()
        wide = {"pairs": m, "pairs_bit_identical": 0, "pixels_differing": 0, "outliers": 0, "max_err_over_tol": 0.0,
                "what": f"the C port's flows of the multi-core sample (pairs 0 .. {m - 1} of the first pass) against the GPU's "
                        f"flows of the same pass, default mode: NO pixel may lie beyond {TOL_REL:g} * max(1, max|ref|)"}
        got = np.empty((wl["h"], wl["w"], 2), np.float32)
        for i in range(m):
            job.fb.get_flow_into(i, got)
            if np.array_equal(got, sample_flows[i]):
                wide["pairs_bit_identical"] += 1
                continue
            d = np.abs(got - sample_flows[i]).max(axis=2)
            tol = TOL_REL * max(1.0, float(np.abs(sample_flows[i]).max()))
            wide["pixels_differing"] += int((d > 0).sum())
            wide["outliers"] += int((d > tol).sum()) + int((~np.isfinite(got)).sum())
            wide["max_err_over_tol"] = max(wide["max_err_over_tol"], float(d.max()) / tol)
        wide["ok"] = wide["outliers"] == 0
    except Exception as err:   # noqa: BLE001 -- reported, the baseline's numbers stand
        wide = {"error": f"{type(err).__name__}: {err}", "ok": None}
    del sample_flows
    remap_per_frame = gate_times["remap_s"] / n1
    key = "all_cores" if cores == host["usable_cores"] else "multi_core"
    out = {"value": one, "unit": "frames/s", "cores": 1, "kind": "port", "host": host,
           "sample": f"{n1} frame pair(s) of the same workload ({wl['w']}x{wl['h']}, levels={wl['levels']}): C port of "
                     f"OpenCV Farneback {gate_times['farneback_s'] / n1:.2f} s/frame + numpy remap "
                     f"{remap_per_frame:.2f} s/frame, one thread",
           key: {"value": n / (t_all + n * remap_per_frame), "farneback_only_value": n / t_all, "unit": "frames/s",
                 "cores": cores, "usable_cores": host["usable_cores"], "kind": "port",
                 "sample": f"{n} frame pairs, one per OpenMP thread on {cores} of the host's {host['usable_cores']} usable "
                           f"cores, {t_all:.2f} s for the C port of Farneback; the numpy remap ({remap_per_frame:.2f} "
                           "s/frame, serial: it is a recurrence) added per frame"}}
    out["parity_of_the_multi_core_sample"] = wide
    try:
        import cv2
    except ImportError:
        out["opencv"] = "cv2 is not installed on this host"
    else:
        res = {}
        for threads in (1, 0):
            cv2.setNumThreads(threads)
            cv2.calcOpticalFlowFarneback(fa[0], fb_[0], None, 0.5, wl["levels"], 15, 3, 5, 1.2, 0)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                cv2.calcOpticalFlowFarneback(fa[0], fb_[0], None, 0.5, wl["levels"], 15, 3, 5, 1.2, 0)
            res[f"threads_{cv2.getNumThreads()}"] = reps / (time.perf_counter() - t0)
        out["opencv"] = {"version": cv2.__version__, "farneback_frames_per_s": res}
    return out


def copy_ceiling(lib, check, nbytes=1 << 30, reps=8):
    """Measured rate (read + write bytes per second) of a 16-byte-per-lane copy kernel over 1 GiB, HIP
    events on the library stream: the practical HBM ceiling the 8 TB/s spec figure is quoted next to."""
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


def gather_leg(job, host, rccl, plans, reps=3):
    """The gather of BASELINE configs[4], outside the timed region: every rank's frames of its last pass go
    to rank 0 in ONE tf_batch_gather per pass -- rank r sends its pairs_per_pass[r] frames (they sit side
    by side in one buffer), the root posts one receive per rank with that rank's byte count
    (transflow_amd.batch.gather_counts: the ranks' passes may differ in length, every rank derives the same
    counts from the all-gathered plans).  Checked: rank 0 compares the CRC of what arrived from each rank
    (first and last frame of its pass) with the CRCs that rank computed of its own frames."""
    from transflow_amd.batch import gather_counts
    from transflow_amd.device import DevBuffer
    nbytes = job.frame_bytes
    world, rank = host.world, host.rank
    per_pass = [p["pairs_per_pass"] for p in plans]
    assert per_pass[rank] == job.batch
    counts, offsets = gather_counts(per_pass, nbytes)
    total = sum(counts)
    recv = DevBuffer(total) if rank == 0 else None
    own = [zlib.crc32(job.comps[i].download().tobytes()) for i in (0, job.batch - 1)]
    crcs = host.gather(own)

    def once():
        rccl.gather_dev(job.out_frames.ptr, counts[rank], None if recv is None else recv.ptr,
                        counts if rank == 0 else None)

    once()
    job.sync()
    ok = None
    if rank == 0:
        ok = True
        for r in range(world):      # the first and the last frame of what each rank sent
            first = recv.download((nbytes,), np.uint8, offset=offsets[r])
            last = recv.download((nbytes,), np.uint8, offset=offsets[r] + counts[r] - nbytes)
            ok = ok and [zlib.crc32(first.tobytes()), zlib.crc32(last.tobytes())] == list(crcs[r])
    host.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    job.sync()
    host.barrier()
    dt = host.max_over_ranks(time.perf_counter() - t0) / reps
    # the rate a step would have with the gather inside it
    host.barrier()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        job.step()
        once()
    job.sync()
    host.barrier()
    dt_in = host.max_over_ranks(time.perf_counter() - t0) / n
    # ... and with the gather on the communicator's own stream, beside the next step's Farneback call: the library stream
    # only waits for it (on the device) before the next step's remap writes the frames' buffer again
    host.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        job.calc_pass(job.n_steps)
        job.n_steps += 1
        rccl.gather_end()
        job.remap_pass(job.layer)
        rccl.gather_begin(job.out_frames.ptr, counts[rank], None if recv is None else recv.ptr,
                          counts if rank == 0 else None)
    rccl.gather_end()
    job.sync()
    host.barrier()
    dt_beside = host.max_over_ranks(time.perf_counter() - t0) / n
    if recv is not None:
        recv.close()
    into_root = total - counts[0]
    return {"what": "untimed region: each rank's frames of one pass (uint8 RGB, one per pair, side by side) to rank 0 in "
                    "one tf_batch_gather (RCCL send/recv into the root, per-rank byte counts)",
            "frames_per_gather": sum(per_pass), "frames_per_rank": per_pass, "bytes_into_root": into_root,
            "ms": dt * 1e3, "GBs_into_root": into_root / dt / 1e9 if dt > 0 else None, "verified_crc": bool(ok),
            "frames_per_s_with_gather_every_step": sum(per_pass) / dt_in,
            "frames_per_s_with_gather_beside_the_next_step": sum(per_pass) / dt_beside}


def flows_to_root_leg(job, host, rccl, plans):
    """SURVEY.md §8e mode F, outside the timed region: the whole clip's flows go to rank 0, each pass's to the clip
    position of its first pair (transflow_amd.batch.flows_to_root_calls, one tf_batch_gather_at per pass index), and
    rank 0 runs ONE remap recurrence over them in clip order -- the frames the reference's one compositor would paint for
    the clip (pipeline.py:565), which the per-rank streams of the timed region are not beyond each rank's first pair.
    Checked: the CRC of the first and the last flow of every rank's last pass as they sit in the root's clip buffer
    against what that rank computed; and the frame the root's stream paints at the end of rank 0's first pass against
    the frame rank 0 paints from its own flows with a fresh layer (same flows, same order, same state: same bytes)."""
    from transflow_amd.batch import flows_to_root_calls
    from transflow_amd.device import DevBuffer
    if job.wl["direction"] == 0:
        return {"skipped": "a FORWARD workload hands the compositor winner maps made beside the flow buffer "
                           "(Farneback.post_process_scatter); the leg runs on BACKWARD workloads"}
    w, h = job.wl["w"], job.wl["h"]
    flow_bytes = w * h * 8
    world, rank = host.world, host.rank
    calls = flows_to_root_calls(plans, flow_bytes)
    total_pairs = calls[0][0]["recv_capacity"] // flow_bytes
    clip = DevBuffer(total_pairs * flow_bytes) if rank == 0 else None

    def collect(sync_each):
        for k, call in enumerate(calls):
            me = call[rank]
            send = 0
            if me["send_bytes"]:
                job.calc_pass(k)
                send = job.fb.flow_ptr(0)
            rccl.gather_at(send, me["send_bytes"], None if clip is None else clip.ptr, me["recv_bytes"], me["recv_offsets"],
                           me["recv_capacity"])
            if sync_each:
                job.sync()

    def paint(layer, flow_of, n, stop_after=None):
        """the serial remap over n flows; -> CRC of the frame painted for flow `stop_after`"""
        crc = None
        for j in range(n):
            layer.step_dev(job.comps[j % job.batch], flow_of(j), job.pixmap_dev, 3, clip_flow=True, seed=SEED_U)
            if j == stop_after:
                crc = zlib.crc32(job.comps[j % job.batch].download().tobytes())
        return crc

    # what each rank knows of its own: the first and last flow of its last pass; rank 0 also the frame that ends its first pass
    collect(True)
    base = job.fb.flow_ptr(0)
    own = []
    for i in (0, job.batch - 1):
        a = np.empty(flow_bytes, np.uint8)
        job.check(job.lib.tf_dev_download(C.c_void_p(a.ctypes.data), C.c_void_p(base + i * flow_bytes), flow_bytes))
        own.append(zlib.crc32(a.tobytes()))
    own_all = host.gather(own)
    ok_flows = ok_stream = None
    if rank == 0:
        ok_flows = True
        for r, p in enumerate(plans):
            at = (p["pairs"][0] - plans[0]["pairs"][0] + p["pass_starts"][-1]) * flow_bytes
            got = []
            for i in (0, p["pairs_per_pass"] - 1):
                got.append(zlib.crc32(clip.download((flow_bytes,), np.uint8, offset=at + i * flow_bytes).tobytes()))
            ok_flows = ok_flows and got == list(own_all[r])
        job.calc_pass(0)
        mine = job.fb.flow_ptr(0)
        want = paint(job.make_layer(), lambda j: mine + j * flow_bytes, job.batch, job.batch - 1)
        root_layer = job.make_layer()
        got = paint(root_layer, lambda j: clip.ptr + j * flow_bytes, total_pairs, job.batch - 1)
        job.sync()
        ok_stream = got == want
        oob = bool(root_layer.out_of_frame())
    # the rate of the mode: every pass of every rank, the gathers, the root's remap of the clip, end to end
    host.barrier()
    t0 = time.perf_counter()
    collect(False)
    job.sync()
    t_collect = time.perf_counter() - t0
    if rank == 0:     # the clip's remap steps as one call (the frames land in the pass's images in turn)
        job.make_layer().steps_dev([job.comps[j % job.batch] for j in range(total_pairs)],
                                   [clip.ptr + j * flow_bytes for j in range(total_pairs)], job.pixmap_dev, 3, clip_flow=True,
                                   seed=SEED_U)
        job.sync()
    host.barrier()
    dt = host.max_over_ranks(time.perf_counter() - t0)
    t_collect = host.max_over_ranks(t_collect)
    if clip is not None:
        clip.close()
    if rank != 0:
        return None
    into_root = sum(c[r]["send_bytes"] for c in calls for r in range(1, world))
    return {"what": "untimed region, SURVEY 8e mode F: every pass's flows (float32 u, v) to their clip positions on rank 0, one "
                    "tf_batch_gather_at per pass index; rank 0 then runs ONE remap recurrence over the clip in order",
            "clip_pairs": total_pairs, "gathers": len(calls), "bytes_into_root": into_root,
            "ms_passes_and_gathers": t_collect * 1e3, "ms_root_remap": (dt - t_collect) * 1e3, "ms": dt * 1e3,
            "frames_per_s": total_pairs / dt, "verified_flow_crc": bool(ok_flows), "verified_stream": bool(ok_stream),
            "remap_out_of_frame": oob, "ok": bool(ok_flows and ok_stream and not oob)}


FLOWS_TO_ROOT_LEG = flows_to_root_leg
GATHER_LEG = gather_leg     # (a name of its own: tests replace it to see what a failing or hanging side leg does to the line)


def run_with_timeout(fn, seconds):
    """fn() on a helper thread; TimeoutError if it has not returned after `seconds` (the thread is then left behind:
    the process must end with os._exit)."""
    import threading
    box = {}

    def target():
        try:
            box["value"] = fn()
        except BaseException as err:   # noqa: BLE001 -- handed to the caller
            box["error"] = err

    t = threading.Thread(target=target, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        box["stuck"] = True
        raise TimeoutError(f"no answer after {seconds:.0f} s")
    if "error" in box:
        raise box["error"]
    return box["value"]


STUCK_THREADS = False
EXIT_CODE = 0


EXIT_NO_RCCL = 6


def rccl_or_nothing(host, rank, wl, lib, check, timeout=None):
    """The communicator and the shared inputs it brings from rank 0 (one-off broadcast, outside the timed region), or
    nothing on EVERY rank when it did not come up on any one of them: (group, pixmap buffer, pixmap, reset mask, None) or
    (None, None, None, None, the first rank's error).  A communicator that never comes up must not hang the run -- the
    path itself needs no collective -- but the run is then not the one that was asked for: main() prints the line and
    leaves with EXIT_NO_RCCL unless --allow-no-rccl was given."""
    global STUCK_THREADS
    from transflow_amd import batch as B
    w, h = wl["w"], wl["h"]

    def rccl_setup():
        g = B.RcclGroup(host)
        from transflow_amd.device import DevBuffer
        pix_buf = DevBuffer(h * w * 3)
        if rank == 0:
            pix_buf.upload(np.random.default_rng(1237).integers(0, 256, (h, w, 3), dtype=np.uint8))
        g.broadcast_dev(pix_buf.ptr, h * w * 3)
        check(lib.tf_sync())
        mask = None
        if wl["reset"]:
            m_buf = DevBuffer(h * w * 4)
            if rank == 0:
                m_buf.upload(np.random.default_rng(1238).random((h, w), dtype=np.float32))
            g.broadcast_dev(m_buf.ptr, h * w * 4)
            check(lib.tf_sync())
            mask = m_buf.download((h, w), np.float32)
            m_buf.close()
        return g, pix_buf, pix_buf.download((h, w, 3), np.uint8), mask

    rccl = pix_buf = pixmap = reset_mask = rccl_error = None
    try:
        if timeout is None:
            timeout = float(os.environ.get("TF_BENCH_RCCL_TIMEOUT", 180))
        rccl, pix_buf, pixmap, reset_mask = run_with_timeout(rccl_setup, timeout)
    except BaseException as err:    # noqa: BLE001 -- say so loudly and carry on with inputs generated per rank
        if isinstance(err, TimeoutError):
            STUCK_THREADS = True
        rccl_error = f"{type(err).__name__}: {err}"
        print(f"[bench] rank {rank}: RCCL leg failed ({rccl_error}); shared inputs generated per rank instead",
              file=sys.stderr)
        rccl = pix_buf = pixmap = reset_mask = None
    errs = host.allgather(rccl_error)
    if any(errs):                             # all or nothing
        if any("TimeoutError" in e for e in errs if e):
            STUCK_THREADS = True              # a peer never answered: the communicator is abandoned, not destroyed
            if rccl is not None:
                rccl.abandon()
        elif rccl is not None:
            rccl.close()
        rccl = pix_buf = pixmap = reset_mask = None
    return rccl, pix_buf, pixmap, reset_mask, next((e for e in errs if e), None)


def missing_rccl_exit_code(world, asked_for_rccl, rccl_ranks, allow_no_rccl):
    """A run that was to use RCCL (more than one rank, or --rccl) and ended without a communicator of all its ranks has
    measured something else than it was asked to: the line is still printed (rccl_ranks, rccl_error say what happened)
    and the process leaves with EXIT_NO_RCCL, so that nobody mistakes the fallback for the result.  --allow-no-rccl
    (rehearsals on a box with fewer GPUs than ranks) makes it 0 again."""
    if (world > 1 or asked_for_rccl) and rccl_ranks != world and not allow_no_rccl:
        return EXIT_NO_RCCL
    return 0


def step_counters(rf, workload, wl, pairs, steps_per_s):
    """north_star's "rocprof HBM GB/s against the chip's peak" for the WHOLE step: the HBM-side bytes of one step by the
    counters (every kernel of the step: profiles/r06_traffic_step.json, made by tools/traffic_step.py from separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at this pass size) times this run's steps per second."""
    t = rf.profile_step_traffic(workload, wl["w"], wl["h"], wl["levels"], pairs)
    if t is None:
        return {"counter_bytes": None, "counter_GBs": None, "counter_frac": None,
                "counter_source": "no counter table for this workload at this pass size under profiles/"}
    gbs = t["bytes_per_step"] * steps_per_s / 1e9
    return {"counter_bytes": t["bytes_per_step"], "counter_GBs": gbs, "counter_frac": gbs / rf.HBM_PEAK_GBS,
            "counter_source": f"profile constant: HBM-side bytes of one step, all its kernels, by the counters ({t['table']}: "
                              "2*FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over the same step at this pass "
                              "size) x this run's steps per second; not a measurement of this run",
            "counter_kernels_GB_per_step": t["kernels_GB_per_step"]}


def level_fracs(rf, dominant, by_level, wl, pairs):
    """The dominant kernel's launches level by level (timed alone): a level that fills the chip and one that does not
    are different stories."""
    if dominant != "fb_flow_iter":
        return None
    per = rf.built_flow_iter_level_bytes(wl["w"], wl["h"], wl["levels"], pairs)
    out = {}
    for name, (cnt, ms) in sorted(by_level.items()):
        k = int(name.rsplit(".k", 1)[1])
        if k in per and cnt:
            steps = cnt / 3.0                                  # three launches of a level per step
            gbs = per[k] * steps / (ms * 1e-3) / 1e9
            out[f"level{k}"] = {"launches": cnt, "avg_launch_ms": ms / cnt, "achieved": gbs, "frac": gbs / rf.HBM_PEAK_GBS}
    return out


def line_skeleton(args, wl, world, plans):
    """The part of the result line that does not depend on a measurement (the dry run prints it with nulls)."""
    w, h = wl["w"], wl["h"]
    pairs_per_step = sum(p["pairs_per_pass"] for p in plans)
    return {
        "metric": "frames/sec (Farneback+remap)", "value": None, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
        # every rank runs --batch pairs per step: weak; passes capped at the ranks' shards of the ONE clip (the default
        # from 4 GPUs up): a step covers the whole clip whatever N is -- the total is fixed: strong
        "higher_is_better": True,
        "scaling": "weak" if all(p["pairs_per_pass"] == args.batch for p in plans) else "strong",
        "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: one clip of T={args.clip_frames} {w}x{h} uint8 frames sharded over the ranks "
                               f"(rank r owns pairs shard_range({args.clip_frames - 1}, r, {world}) + a one-frame halo), "
                               f"Farneback pyr_scale=0.5 levels={wl['levels']} winsize=15 iterations=3 poly_n=5 poly_sigma=1.2 "
                               f"flags=0, {'FORWARD' if wl['direction'] == 0 else 'BACKWARD'} post_process, moveref layer "
                               f"(reset {'random p=0.5 through a float mask, u drawn on the GPU' if wl['reset'] else 'off'}), "
                               "1 RGB pixmap source, render",
                   "clip_frames": args.clip_frames,
                   "frame_pairs_per_step_per_gpu": [p["pairs_per_pass"] for p in plans],
                   "frame_pairs_per_step": pairs_per_step,
                   "equal_batches": bool(args.equal_batches),
                   "lanes": [args.lanes or pick_lanes(w, h, p["pairs_per_pass"]) for p in plans],
                   "pairs_per_rank": [p["n_pairs"] for p in plans],
                   "frame_expansions": "pairs t and t+1 of a step share frame t+1: its pyramid levels and polynomial "
                                       "expansion (A1+A2, functions of the frame alone) are computed once per step and "
                                       "read by both; nothing is kept between steps",
                   "parallelism": f"frame pairs of one clip sharded over {world} GPU(s), one remap stream per GPU, "
                                  "no data-path collective"},
    }


def main():
    global STUCK_THREADS, EXIT_CODE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=128,
                    help="frame pairs per pass (per GPU), capped at the rank's shard of the clip: 128 at 1 and 2 GPUs, a rank's "
                         "whole shard at 4 (64, one rank 63) and 8 (32, one rank 31).  The more pairs a pass holds, the more "
                         "of the pyramid's levels put at least as many columns of workgroups side by side as the chip has "
                         "slots (768) and march in segments on a full chip like level 0: with 128 pairs of 4K levels 0 - 2 do "
                         "(1152 columns at level 2), with 64 levels 0 - 1, with 32 level 0 alone.  One lane, 4K: 1776 / 1922 / "
                         "1927 / 1973 frames/s at 32 / 64 / 96 / 128 pairs per pass (profiles/r05_batch_sweep_4k.txt)")
    ap.add_argument("--clip-frames", type=int, default=256, help="T: frames of the clip that is sharded over the ranks")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 1, 2),
                    help="Farneback handles per rank that take the batches in turn (2: consecutive batches are in flight "
                         "together on the library's two call streams, tf_fb_create_lane).  0 (default): chosen per rank from "
                         "what a pass holds (pick_lanes): one lane where a pass fills the chip by itself, two below that")
    ap.add_argument("--size", default=None, help="WxH: run the chosen workload's configuration at another frame size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed side measurements")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the three synchronised steps behind the timed region that time the dominant kernel with nothing "
                         "beside it (roofline.alone): under rocprofv3 every launch of the process is then a gate, warm-up or "
                         "timed-region launch, and its per-kernel average is comparable with roofline.avg_launch_ms")
    ap.add_argument("--burn-in", type=int, default=-1,
                    help="untimed passes before the W warm-up steps (-1, the default: passes for about three seconds; 0: none): "
                         "the first process on a fresh box measured up to 2 - 3 %% below the next one")
    ap.add_argument("--no-gate", action="store_true", help="skip the parity gate (the JSON line says so)")
    ap.add_argument("--equal-batches", action="store_true",
                    help="trim every rank's pass to the shortest one (T=256 over 8 ranks: 31 pairs everywhere instead of 32 x 7 + 31)")
    ap.add_argument("--rccl", action="store_true", help="use the RCCL legs (broadcast, gather) even with one rank")
    ap.add_argument("--allow-no-rccl", action="store_true",
                    help="with --gpus N > 1 or --rccl: a run whose communicator did not come up on all N ranks still prints "
                         "its line (rccl_ranks 0, rccl_error) but leaves with exit code 6 -- unless this flag says that the "
                         "fallback (every rank makes the shared inputs itself, no gather legs) is what was wanted, e.g. ranks "
                         "sharing one GPU in a rehearsal")
    ap.add_argument("--dry-run", action="store_true", help="ranks meet, shard the clip and report the plan; no GPU call")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="a run-time option of the library (tf_set_option; include/tfhip.h) for A/B runs; recorded in the JSON line")
    args = ap.parse_args()

    from transflow_amd import batch as B
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher above us: start the ranks ourselves, before this process touches the GPU
        sys.exit(B.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank, local_rank, world = B.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    host = B.HostGroup(rank, world)

    wl = dict(WORKLOADS[args.workload])
    if args.size:
        wl["w"], wl["h"] = (int(v) for v in args.size.lower().split("x"))
    w, h = wl["w"], wl["h"]
    if args.clip_frames < 2:
        raise SystemExit("--clip-frames must be at least 2")
    plan = make_plan(args.clip_frames, args.batch, rank, world, args.equal_batches)
    plans = host.allgather(plan)
    if args.dry_run:
        if rank == 0:
            line = line_skeleton(args, wl, world, plans)     # the keys of a measured line, nulls where a GPU would speak
            line.update({"rccl_ranks": None, "rccl_version": None, "per_rank_frames_per_s": None,
                         "parity_gate": "skipped (--dry-run)", "timed_region_recheck": None, "roofline": None, "cpu_baseline": None,
                         "kernels_ms_per_step": None, "remap_out_of_frame": None, "gather": None, "flows_to_root": None,
                         "gather_verified_crc": None, "flows_to_root_ok": None, "untimed_steps_before_timed_region": None,
                         "parity_wide": None,
                         "dry_run": True, "clip_frames": args.clip_frames, "plans": plans})
            print(json.dumps(line))
        host.close()
        return

    from transflow_amd import _lib
    lib, check = _lib.load(), _lib.check
    device = local_rank
    if os.environ.get("TF_BENCH_SHARE_GPU"):
        # rehearsal on a box with fewer GPUs than ranks: the ranks share the devices there are (RCCL then refuses
        # the communicator -- two ranks on one device -- and the run goes on without its legs, which the line says)
        n_dev = C.c_int()
        check(lib.tf_device_count(C.byref(n_dev)))
        device = local_rank % max(1, n_dev.value)
    check(lib.tf_init(device))
    options = {}
    for item in args.option:
        name, _, value = item.partition("=")
        _lib.set_option(name, int(value))
        options[name] = int(value)
    rccl, rccl_error = None, None
    pixmap = reset_mask = pixmap_dev = pix_buf = None
    if world > 1 or args.rccl:
        rccl, pix_buf, pixmap, reset_mask, rccl_error = rccl_or_nothing(host, rank, wl, lib, check)
        pixmap_dev = pix_buf.ptr if pix_buf is not None else None
    lanes = args.lanes or pick_lanes(w, h, plan["pairs_per_pass"])     # this rank's
    job = Job(wl, args.batch, plan, args.clip_frames, seed=2000, device=device, pixmap=pixmap, reset_mask=reset_mask,
              pixmap_dev=pixmap_dev, lanes=lanes)
    if pix_buf is not None:
        job.pixmap_buffer = pix_buf     # the job gathers from this buffer: it lives as long as the job

    gate, gate_times = None, None
    if rank == 0 and not args.no_gate:
        gate, gate_times = parity_gate(job)
    gate = host.broadcast(gate)
    if gate is not None and not gate["ok"]:
        if rank == 0:
            print(f"[bench] parity gate FAILED, nothing is reported: {json.dumps(gate)}", file=sys.stderr)
        host.close()
        sys.exit(3)

    # Untimed passes before the W warm-up steps: the first process on a fresh box measured up to 2 - 3 % below the next
    # one (1992 -> 2013, 2018 -> 2058 frames/s: the kernels themselves ran slower), and not with ~4 s of passes in front
    # (2059, then 2053 / 2054 / 2071 without).  --burn-in N: N passes; -1 (default): passes for about three seconds; 0: none.
    # The line says how many ran (`burn_in_steps`); the timed region is the K steps after the W warm-up steps either way.
    burn_in_steps = 0
    if args.burn_in >= 0:
        for _ in range(args.burn_in):
            job.step()
        burn_in_steps = args.burn_in
    else:
        t_burn = time.perf_counter()
        while time.perf_counter() - t_burn < 3.0 and burn_in_steps < 2000:
            for _ in range(5):
                job.step()
            job.sync()
            burn_in_steps += 5
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

    job.sync()
    host.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.sync()
    t_rank = time.perf_counter() - t0
    host.barrier()
    elapsed = host.max_over_ranks(time.perf_counter() - t0)
    job.prof(False)
    dom_cnt, dom_ms = job.prof_report()[dominant]
    recheck = host.gather(timed_region_recheck(job, lanes))
    # the same kernel with nothing beside it: a few steps with a synchronisation after each, so neither the other lane's
    # batch nor the previous batch's remap shares the chip with it (in the timed region they do: that is what two lanes
    # are for, and a launch's duration there includes the time it shares)
    from transflow_amd import _lib as _L
    job.prof_reset()
    _L.set_option("prof_levels", 1)          # labels carry the pyramid level: "fb_flow_iter.k0"
    job.prof(True, dominant)
    for _ in range(0 if args.no_alone else 3):
        job.step()
        job.sync()
    job.prof(False)
    _L.set_option("prof_levels", 0)
    alone_report = job.prof_report()
    by_level = {k: v for k, v in alone_report.items() if k.startswith(dominant + ".k")}
    # (a kernel launched without a level label -- fb_w1_vsum, fb_blur_solve_generic -- is reported under its own name)
    alone_src = by_level or {k: v for k, v in alone_report.items() if k == dominant}
    alone_cnt, alone_ms = sum(v[0] for v in alone_src.values()), sum(v[1] for v in alone_src.values())
    if args.no_alone or alone_cnt == 0 or alone_ms <= 0:
        alone_cnt, alone_ms = 0, None           # not measured: the line carries nulls, never Infinity
    rank_fps = host.gather(args.steps * job.batch / t_rank)
    oob = host.gather(bool(job.layer.out_of_frame()))

    # Everything the result line needs from the other ranks has been collected by now.  What follows are
    # untimed side legs: each runs under a time limit and reports its failure in the line instead of losing it.
    leg_errors = {}
    gather, flows_to_root, rccl_version = None, None, None
    if rccl is not None:
        rccl_version = rccl.rccl_version

        def leg():
            g = GATHER_LEG(job, host, rccl, plans)
            f = None
            try:
                f = FLOWS_TO_ROOT_LEG(job, host, rccl, plans)
            except Exception as err:    # noqa: BLE001 -- reported in the line; the gather leg's result stands
                leg_errors["flows_to_root"] = f"{type(err).__name__}: {err}"
            finally:
                rccl.close()
            return g, f

        try:
            gather, flows_to_root = run_with_timeout(leg, float(os.environ.get("TF_BENCH_LEG_TIMEOUT", 240)))
        except TimeoutError as err:
            STUCK_THREADS = True        # a helper thread sits in a call that never returned; the sockets are its
            rccl.abandon()
            leg_errors["gather"] = f"TimeoutError: {err}"
        except BaseException as err:    # noqa: BLE001 -- reported in the line
            leg_errors["gather"] = f"{type(err).__name__}: {err}"
    if rank != 0:
        if not STUCK_THREADS:
            host.close()
        return

    from transflow_amd import roofline as rf
    pairs_per_step = sum(p["pairs_per_pass"] for p in plans)
    frames = args.steps * pairs_per_step
    fps = frames / elapsed
    P = job.batch
    built = rf.built_kernel_bytes(dominant, wl["w"], wl["h"], wl["levels"], P) * args.steps
    model = rf.model_kernel_bytes(dominant, wl["w"], wl["h"], wl["levels"], P) * args.steps
    achieved = built / (dom_ms * 1e-3) / 1e9
    launches_per_step = dom_cnt / max(1, args.steps)
    traffic = rf.profile_traffic(dominant, wl["w"], wl["h"], wl["levels"], P, launches_per_step)
    traffic_source = ("profile constant: B/px from the rocprofv3 FETCH_SIZE/WRITE_SIZE passes under profiles/ scaled to this "
                      "workload's launches, not a measurement of this run")
    # where the whole-step table (round 6) was made for exactly this workload and pass size, the dominant kernel's row of it
    # is the fresher and unscaled figure: its bytes per step over its launches per step
    step_table = rf.profile_step_traffic(args.workload, wl["w"], wl["h"], wl["levels"], P)
    kernel_row = {"fb_flow_iter": "k_flow_iter_pc"}.get(dominant)
    if step_table is not None and kernel_row in step_table.get("kernels_launches_per_step", {}) and \
            step_table["kernels_launches_per_step"][kernel_row] == round(launches_per_step):
        traffic = step_table["kernels_bytes_per_step"][kernel_row] / step_table["kernels_launches_per_step"][kernel_row]
        traffic_source = (f"profile constant: the kernel's row of {step_table['table']} (2*FETCH_SIZE + WRITE_SIZE of separate "
                          "rocprofv3 --pmc passes over the same step at this pass size) over its launches per step, not a "
                          "measurement of this run")
    avg_ms = dom_ms / max(1, dom_cnt)
    step_model = P * (rf.farneback_bytes(w, h, 0.5, wl["levels"], 3)
                      + rf.remap_bytes(w, h, reset_mask=wl["reset"], forward=wl["direction"] == 0))
    step_built = rf.built_step_bytes(w, h, wl["levels"], P, reset_mask=wl["reset"], forward=wl["direction"] == 0)
    per_gpu_s = args.steps / elapsed
    out = line_skeleton(args, wl, world, plans)
    out.update({"value": fps, "ms_per_step": elapsed / args.steps * 1e3, "burn_in_steps": burn_in_steps,
                # `warmup` is the W that was asked for; what ran in front of the timed region is this many untimed steps
                # (about three seconds of passes for the clocks, then the W warm-up steps, the last of them profiled)
                "untimed_steps_before_timed_region": burn_in_steps + n_warm})
    per_launch = built / max(1, dom_cnt)
    timed = {"launches": dom_cnt, "avg_launch_ms": avg_ms, "achieved": achieved, "frac": achieved / rf.HBM_PEAK_GBS}
    alone = None
    if alone_cnt:
        alone_gbs = per_launch * alone_cnt / (alone_ms * 1e-3) / 1e9
        alone = {"what": "the same launches in steps that are synchronised one by one (no other lane, no remap beside "
                         "them), measured right after the timed region",
                 "launches": alone_cnt, "avg_launch_ms": alone_ms / alone_cnt,
                 "by_level": level_fracs(rf, dominant, by_level, wl, P),
                 "achieved": alone_gbs, "frac": alone_gbs / rf.HBM_PEAK_GBS}
    # The kernel's own figure: with one lane the timed region's launches have the chip to themselves and ARE it; with two
    # lanes a launch of the timed region shares the chip with the other lane's batch (its duration includes the time it
    # shares: `overlapped`), and the kernel's own figure is the `alone` measurement.
    own = timed if lanes == 1 else alone
    step_frac = step_built * per_gpu_s / 1e9 / rf.HBM_PEAK_GBS
    out.update({
        "rccl_ranks": world if rccl_version is not None else 0, "rccl_version": rccl_version,
        "per_rank_frames_per_s": rank_fps,
        "parity_gate": gate if gate is not None else "skipped (--no-gate)",
        "timed_region_recheck": {"ok": all(r["ok"] for r in recheck), "per_rank": recheck},
        "roofline": {"bound": "hbm", "kernel": dominant,
                     "achieved": own["achieved"] if own else None, "peak": rf.HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": own["frac"] if own else None,
                     "frac_source": ("timed region (one lane: nothing runs beside a launch)" if lanes == 1 else
                                     "`alone`: the timed workload's launches with nothing beside them, right after the "
                                     "timed region" if alone else "not measured (--no-alone with two lanes): see `overlapped`"),
                     "avg_launch_ms": own["avg_launch_ms"] if own else None,
                     "step_frac": step_frac,     # = whole_step.frac: everything a step launches, built bytes / wall time
                     "frac_overlapped": timed["frac"] if lanes > 1 else None,
                     "overlapped": dict(timed, what="the timed region's own launches: with two lanes each shares the chip "
                                                    "with the other lane's batch and its duration includes that -- how long "
                                                    "a launch took, not how well the kernel uses the memory system")
                                   if lanes > 1 else None,
                     "bytes_model": "bytes the kernel as built must move per launch (DESIGN.md §5): for the one-kernel "
                                    "iteration R0 20 + R1 20 + flow in 8 + flow out 8 = 56 B/px (M never leaves the CU).  "
                                    "SURVEY §8(d)'s stage-once model charges the reference's stages 96 B/px (M stored and "
                                    "read back): on those bytes the same launches read > 1 of peak by construction "
                                    "(`model_work_rate`: a work rate, not utilisation)",
                     "traffic": traffic, "traffic_source": traffic_source,
                     "counter_GBs": traffic / (own["avg_launch_ms"] * 1e-3) / 1e9 if traffic and own else None,
                     "counter_frac": traffic / (own["avg_launch_ms"] * 1e-3) / 1e9 / rf.HBM_PEAK_GBS if traffic and own else None,
                     "launches": dom_cnt,
                     "algorithmic_bytes_per_launch": per_launch,
                     "alone": alone,
                     "model_work_rate": {"what": "SURVEY Appendix C stage-once bytes of the reference's stages (96 B/px per "
                                                 "iteration, M stored and re-read) per second: the reference's work rate, "
                                                 "not HBM utilisation",
                                         "bytes_per_launch": model / max(1, dom_cnt),
                                         "achieved": (model / max(1, dom_cnt)) / (own["avg_launch_ms"] * 1e-3) / 1e9 if own else None,
                                         "frac": (model / max(1, dom_cnt)) / (own["avg_launch_ms"] * 1e-3) / 1e9 / rf.HBM_PEAK_GBS if own else None},
                     "whole_step": dict({"built_bytes": step_built, "achieved": step_built * per_gpu_s / 1e9,
                                         "frac": step_frac,
                                         "model_bytes": step_model, "model_work_rate": step_model * per_gpu_s / 1e9,
                                         "model_frac": step_model * per_gpu_s / 1e9 / rf.HBM_PEAK_GBS},
                                        **step_counters(rf, args.workload, wl, P, per_gpu_s))},
        "kernels_ms_per_step": {k: round(v[1], 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
        "remap_out_of_frame": any(oob),
    })
    if rccl_error:
        out["rccl_error"] = rccl_error
    if options:
        out["library_options"] = options   # not the defaults: an A/B run
    if gather is not None:
        out["gather"] = gather
    if flows_to_root is not None:
        out["flows_to_root"] = flows_to_root
    # the multi-GPU legs' verdicts where nobody can miss them (null: the leg did not run)
    out["gather_verified_crc"] = gather.get("verified_crc") if isinstance(gather, dict) else None
    out["flows_to_root_ok"] = flows_to_root.get("ok") if isinstance(flows_to_root, dict) else None
    leg_limit = float(os.environ.get("TF_BENCH_LEG_TIMEOUT", 240))
    if not args.no_cpu_baseline and gate_times is not None:      # rank 0, at every N (the other ranks are done)
        try:
            out["cpu_baseline"] = run_with_timeout(lambda: cpu_baseline(job, gate_times), leg_limit)
        except BaseException as err:    # noqa: BLE001 -- reported in the line
            STUCK_THREADS = STUCK_THREADS or isinstance(err, TimeoutError)
            leg_errors["cpu_baseline"] = f"{type(err).__name__}: {err}"
    if not args.no_extra and not STUCK_THREADS:      # the copy ceiling at every N (rank 0's GPU), the other workloads at N = 1

        def extras():
            ceiling = copy_ceiling(job.lib, job.check)
            out["roofline"]["measured_copy_ceiling_GBs"] = ceiling
            # above the copy kernel's rate the byte model would be wrong: flagged in the line, never a lost line
            out["roofline"]["exceeds_copy_ceiling"] = bool(max(achieved, (alone or {}).get("achieved", 0.0)) > ceiling * 1.02)
            if world == 1:
                # the bit-identical mode's rate on the same workload (option fb_exact_sums: the window summed in OpenCV's own
                # order along the rows too; its flow was compared bit for bit with the oracle's in the gate)
                job.fb.set_exact(True)
                try:
                    for _ in range(2):                   # both lanes once: each sizes its buffer of column sums at its first call
                        job.step()
                    job.sync()
                    t0 = time.perf_counter()
                    for _ in range(4):
                        job.step()
                    job.sync()
                    out["exact_mode"] = {"frames_per_s": 4 * job.batch / (time.perf_counter() - t0),
                                         "bit_identical": None if gate is None else bool(gate["exact_bit_identical"]),
                                         "what": "the timed workload in the handles' exact mode (tf_fb_set_exact), untimed region; the default mode "
                                                 "differs from the same oracle in parity_gate.flow_pixels_differing pixels"}
                finally:
                    job.fb.set_exact(None)
            extra = {}
            for name in (("1080p", "1080p-1level") if world == 1 else ()):
                if name == args.workload:
                    continue
                w2 = WORKLOADS[name]
                # the same bytes per call as the main workload: more pairs of the smaller frames
                b2 = max(1, min(64, args.batch * (w * h) // (w2["w"] * w2["h"])))
                j = Job(w2, b2, make_plan(b2 + 1, b2, 0, 1), b2 + 1, seed=2000, device=device,
                        lanes=args.lanes or pick_lanes(w2["w"], w2["h"], b2))
                g2 = None
                if not args.no_gate:            # the same gate as the main workload, before its rate is reported
                    g2, _ = parity_gate(j, n_check=1)
                    if not g2["ok"]:
                        extra[name] = {"parity_gate": g2, "frames_per_s": None,
                                       "error": "parity gate failed: no rate is reported for this workload"}
                        continue
                for _ in range(3):
                    j.step()
                j.sync()
                t0 = time.perf_counter()
                n = 20
                for _ in range(n):
                    j.step()
                j.sync()
                dt = time.perf_counter() - t0
                sb = rf.built_step_bytes(w2["w"], w2["h"], w2["levels"], b2, reset_mask=w2["reset"], forward=w2["direction"] == 0)
                extra[name] = {"frames_per_s": n * b2 / dt, "frame_pairs_per_step": b2,
                               "whole_step_frac_of_8TBs": sb * n / dt / 1e9 / rf.HBM_PEAK_GBS,
                               "parity_gate": g2 if g2 is not None else "skipped (--no-gate)"}
                del j
            if world == 1:
                out["other_workloads_untimed_region"] = extra

        try:
            run_with_timeout(extras, leg_limit)
        except BaseException as err:    # noqa: BLE001 -- reported in the line
            STUCK_THREADS = STUCK_THREADS or isinstance(err, TimeoutError)
            leg_errors["extras"] = f"{type(err).__name__}: {err}"
    if leg_errors:
        out["side_leg_errors"] = leg_errors
    # the wide parity check rides on the CPU baseline's multi-core sample (its oracle flows are pass 0's): where nobody can miss it
    wide = (out.get("cpu_baseline") or {}).get("parity_of_the_multi_core_sample") if isinstance(out.get("cpu_baseline"), dict) else None
    out["parity_wide"] = wide
    print(json.dumps(out, allow_nan=False), flush=True)
    if not STUCK_THREADS:
        host.close()
    if out["roofline"].get("exceeds_copy_ceiling"):
        print("[bench] roofline.achieved exceeds the copy ceiling measured in this run: check the byte model", file=sys.stderr)
        EXIT_CODE = 5
    no_rccl = missing_rccl_exit_code(world, args.rccl, out["rccl_ranks"], args.allow_no_rccl)
    if no_rccl:
        print(f"[bench] this run was to use RCCL on {world} rank(s) and had {out['rccl_ranks']}: the line above is the fallback's "
              f"(every rank made the shared inputs itself, no gather legs), exit code {no_rccl}; --allow-no-rccl accepts it",
              file=sys.stderr)
        EXIT_CODE = no_rccl
    if isinstance(wide, dict) and wide.get("ok") is False:      # (last: a parity failure outranks the other codes)
        print(f"[bench] parity FAILED on the wide sample after the timed region: {json.dumps(wide)}", file=sys.stderr)
        EXIT_CODE = 3


def run_as_main():
    main()
    sys.stdout.flush()
    sys.stderr.flush()
    if STUCK_THREADS:      # a helper thread is still inside a call that never returned: leave without joining it,
        os._exit(4)        # and say so (the result line, if this is rank 0, has been printed)
    sys.exit(EXIT_CODE)


if __name__ == "__main__":
    run_as_main()
