#!/usr/bin/env python3
"""bench.py — ORB extract + BF-Hamming match throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of HBM-resident synthetic frames:
detect (gray -> pyramid -> FAST cells -> quadtree -> blur -> orientation + rBRIEF) followed by the
knn-2 Hamming match of every frame against its predecessor (chained across batches), exactly the
call order of RgbdFeatureFrontend (rgbd_feature_frontend.cpp:187,237).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU): every rank runs its own stream
(seed 1234 + 100*rank), no data-path collective is needed for extract+match ("weak" scaling);
timing is barrier + synchronize on both sides and the MAX over ranks.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the context's stream
for the dominant kernel; `cpu_baseline` times the CPU oracle (oracle/, the restatement of the
reference's CPU plugin) on a bounded sample of the same stream on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# integer vector ALU: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz (a wave64 VALU op issues over 4 cycles; the
# 157 TF fp32 spec = this x 2 (packed) x 2 (fma)).
VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12
MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense FP4 via v_mfma_scale_f32_32x32x64_f8f6f4 (MI355X_MICROARCH.md, matrix cores)
STAGE_KERNEL = {"gray": "mslam::k_gray4", "resize": "mslam::k_resize_col", "fast": "mslam::k_fast_cells",
                "quadtree": "mslam::k_quadtree", "blur": "mslam::k_blur", "describe": "mslam::k_describe",
                "match_knn2": "void mslam::k_match_knn2_fp4<4>", "ratio_compact": "mslam::k_ratio_compact"}
if os.environ.get("MSLAM_HIP_MATCHER") == "popcount":  # the xor/popcount matcher instead of the matrix-core one
    STAGE_KERNEL["match_knn2"] = "void mslam::k_match_knn2<8, 1, 8>"


def pmc_traffic(stage, kernels_per_launch, frames_per_launch):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate runs, in KB; on gfx950 FETCH_SIZE counts half of the bytes of coalesced
    streaming reads — MI355X_MICROARCH.md §HBM — hence the factor 2, which the gray kernel's known
    input confirms: 500 frames x 921 600 B = 460.8 MB, FETCH_SIZE reads 230.4 MB).  None when no profile
    is committed.  `kernels_per_launch`: the resize stage is 7 kernels (one per level) timed as one."""
    path = os.path.join(ROOT, "profiles", "r01_e_pmc_fetch_write_per_launch.json")
    try:
        j = json.load(open(path))
        k = STAGE_KERNEL[stage]
        d = j[k]
        fpl = j["_meta"]["frames_per_launch"]
        scale = frames_per_launch / float(fpl.get(k, fpl["default"]))  # traffic is linear in the batch size
    except (OSError, KeyError, ValueError):
        return None
    return int((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024 * scale * kernels_per_launch)


def baseline_metric():
    """the metric string of BASELINE.json (this file sits next to it)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "ORB keypoints extracted+matched /sec, 640\u00d7480 RGB-D, 1/2/4/8 GPU; HBM GB/s vs roofline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1000, help="frames per step (one step = the 1000-frame stream of cfg2)")
    ap.add_argument("--unique", type=int, default=1000, help="distinct synthetic frames kept in HBM (cycled)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--bow", action="store_true",
                    help="cfg3/cfg4: also DBoW3 loop scoring every frame (synthetic k=10 vocabulary) and, with "
                         "N>1, the RCCL all-gather of BoW vectors + cross-stream scoring")
    ap.add_argument("--voc-levels", type=int, default=6, help="vocabulary depth L (k=10): 6 -> 1e6 words")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real multi-GPU run) or gloo (rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serialized", action="store_true",
                    help="also report every stage timed alone (all work on one stream), after the timed region")
    ap.add_argument("--cpu-sample", type=int, default=500, help="frames of the stream timed on the CPU oracle")
    return ap.parse_args()


def stage_bytes(ctx, B, n_kp, n_cand, voc_k=10, voc_L=6, db_entries=64):
    """Algorithmic HBM bytes per LAUNCH for every stage: each input byte read once,
    each output byte written once, for a batch of B frames with n_kp keypoints / n_cand FAST
    candidates in total."""
    w, h, _ = ctx.level_geometry()
    px = [a * b for a, b in zip(w, h)]
    P = sum(px)
    return {
        "gray": B * (3 * px[0] + px[0]),
        "resize": B * sum(px[l - 1] + px[l] for l in range(1, len(px))),
        "fast": B * P + 4 * n_cand,
        "quadtree": 4 * n_cand + 4 * n_kp,
        "blur": B * 2 * P,
        "describe": B * 2 * P + 48 * n_kp,  # reads both planes around each keypoint, writes desc+xy+angle+octave+resp
        "match_knn2": 2 * 32 * n_kp + 16 * n_kp,
        "ratio_compact": 12 * n_kp + 8 * n_kp,
        # bow_tree = n*(32 + L*k*32) + n*12 out; vectors are <= n x (u32, f64)
        "bow_descend": n_kp * (32 + voc_L * voc_k * 32) + 12 * n_kp,
        "bow_vector": 12 * n_kp + 12 * n_kp,
        "bow_score": db_entries * 2 * 12 * n_kp,
    }


def cpu_baseline(frames, n_sample):
    """Oracle (= port of the reference CPU plugin) detect + match on the host cores."""
    import __graft_entry__ as graft
    from concurrent.futures import ThreadPoolExecutor
    orc = graft.load_oracle()
    orc.lib()
    cores = max(1, min(os.cpu_count() or 1, 32))
    n_sample = min(n_sample, len(frames))
    p = orc.params()
    sample = [np.ascontiguousarray(frames[i]) for i in range(n_sample)]

    # detect every frame once (parallel over frames), then match consecutive pairs (parallel over pairs):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        dets = list(ex.map(lambda i: orc.detect(sample[i], p), range(n_sample)))
        list(ex.map(lambda i: orc.match(dets[i]["desc"], dets[i - 1]["desc"]), range(1, n_sample)))
    dt = time.perf_counter() - t0
    n_kp = sum(len(d["xy"]) for d in dets)
    return {"value": n_kp / dt, "unit": "keypoints/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same synthetic 640x480 stream, detect + knn-2 match vs previous frame, "
                      "oracle/ C restatement (scalar; OpenCV's internal SIMD is not reproduced), %d threads over frames"
                      % (n_sample, cores)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    # rehearsal aid: MSLAM_BENCH_DEVICE pins every rank to one GPU (only meaningful with --dist-backend gloo)
    dev = int(os.environ.get("MSLAM_BENCH_DEVICE", local))
    if world > 1:
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(a.dist_backend)
    torch.cuda.set_device(dev)
    red_dev = "cuda" if a.dist_backend == "nccl" else "cpu"

    import synth
    import __graft_entry__ as graft
    pkg = graft.load_package()

    B = a.batch
    n_unique = max(B, (a.unique // B) * B)
    frames = synth.make_stream(n_unique, a.width, a.height, seed=1234 + 100 * rank)
    d_frames = torch.from_numpy(frames).cuda()
    frame_bytes = a.width * a.height * 3

    # capacities scale with the frame area (4096 keypoints / 16384 FAST candidates per level at 640x480)
    area = max(1, -(-a.width * a.height // (640 * 480)))
    ctx = pkg.Context(width=a.width, height=a.height, max_batch=B, max_keypoints=4096 * area,
                      max_candidates=16384 * area, device=dev)
    n_batches = n_unique // B
    cross = None
    if a.bow:
        ctx.bow_load(synth.make_vocabulary(10, a.voc_levels, seed=77))
        from modular_slam_amd.multi_stream import CrossStreamLoopCandidates
        cross = CrossStreamLoopCandidates(k_max=2048)

    def step(i):
        off = (i % n_batches) * B
        ctx.detect_batch_dev(d_frames.data_ptr() + off * frame_bytes, B)
        ctx.match_batch_dev(0.7, True)
        if a.bow:
            ctx.bow_batch_dev(True)
            if world > 1:
                cross.step_gpu(ctx)

    # warm-up; the first pass over each distinct batch also counts its keypoints (the unit of the metric) and
    # FAST candidates, so that no extra launches are needed after the timed region
    counts_per_batch, cand_per_batch = {}, {}

    def count_batch(b):
        ctx.sync()
        v = ctx.batch_view()  # the output set alternates between batches: fetch the view after every detect
        counts_per_batch[b] = int(pkg.read_device(ctx, v.count, (B,), np.int32).sum())
        cand_per_batch[b] = sum(len(ctx.debug_keypoints(pkg.DBG_CANDIDATES, 0, l)) for l in range(8)) * B

    for i in range(max(a.warmup, n_batches)):
        step(i)
        if i % n_batches not in counts_per_batch:
            count_batch(i % n_batches)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    w_eff_pre = max(a.warmup, n_batches)
    ctx.set_profiling(2)  # HIP events around every stage, in place on the stream it is launched on, during the timed steps
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(w_eff_pre + i)
    ctx.sync()  # also surfaces capacity overflows loudly
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timed = ctx.stage_times(cap=8192)  # (stage, ms) of every launch of the timed steps
    ctx.set_profiling(0)

    # units processed: keypoints extracted (and matched against the previous frame) in the timed steps
    w_eff = max(a.warmup, n_batches)
    n_kp = sum(counts_per_batch[(w_eff + i) % n_batches] for i in range(a.steps))

    t_max = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    kp_sum = torch.tensor([float(n_kp)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(kp_sum, op=dist.ReduceOp.SUM)
    dt_max, kp_total = float(t_max.item()), float(kp_sum.item())

    out = None
    if rank == 0:
        # per-stage launch durations measured inside the timed region (HIP events on the launching stream)
        # a step's entries end with its ratio_compact; a very long run stops recording at 8192 entries: keep whole steps
        ends = [i for i, (name, _) in enumerate(timed) if name == "ratio_compact"]
        steps_cov = len(ends)
        timed = timed[:ends[-1] + 1] if ends else timed
        tot, cnt = {}, {}
        for name, ms in timed:
            tot[name] = tot.get(name, 0.0) + ms
            cnt[name] = cnt.get(name, 0) + 1
        avg = {k: tot[k] / cnt[k] for k in tot}                    # ms per launch
        per_step = {k: cnt[k] / float(steps_cov) for k in tot}       # launches per step (detector: one per chunk)
        acc = {k: tot[k] / steps_cov for k in tot}                   # ms per step, summed over the step's launches
        kp_b, cand_b = counts_per_batch[0], cand_per_batch[0]
        sb = stage_bytes(ctx, B, kp_b, cand_b, 10, a.voc_levels)  # per step (B frames)
        dom = max(acc, key=acc.get)
        fpl = B / per_step[dom]                                    # frames per launch of the dominant stage
        bytes_per_launch = sb[dom] / per_step[dom]
        achieved = bytes_per_launch / (avg[dom] * 1e-3) / 1e9
        roofline = {"kernel": STAGE_KERNEL.get(dom, dom), "stage": dom, "bound": "hbm", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic(dom, 7 if dom == "resize" else 1, fpl),
                    "launches_per_step": per_step[dom], "frames_per_launch": fpl, "avg_ms": round(avg[dom], 4),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "timing": "HIP events on the launching stream around every stage launch of %d of the %d timed steps "
                              "(%d launches of the dominant stage); the resize entry spans its 7 per-level kernels" % (
                                  steps_cov, a.steps, cnt[dom]),
                    "stages_ms_per_launch": {k: round(x, 4) for k, x in avg.items()},
                    "stages_ms_per_step": {k: round(x, 4) for k, x in acc.items()},
                    "stages_gbs": {k: round(sb[k] / (x * 1e-3) / 1e9, 1) for k, x in acc.items() if x > 0 and k in sb}}
        if a.serialized:
            ctx.set_profiling(1)
            ser = {}
            reps = 5
            for i in range(reps):
                step(i)
                for name, ms in ctx.stage_times():
                    ser[name] = ser.get(name, 0.0) + ms / reps
            ctx.set_profiling(0)
            roofline["stages_ms_serialized"] = {k: round(x, 4) for k, x in ser.items()}
        if "match_knn2" in acc:
            # the matcher is not HBM-bound
            pairs = B * (kp_b / B) ** 2
            t_s = avg["match_knn2"] * 1e-3
            roofline["match_kernel"] = STAGE_KERNEL["match_knn2"]
            if os.environ.get("MSLAM_HIP_MATCHER") == "popcount":
                # xor/popcount form: 8 xor + 8 bcnt + 3 top-2 lane-ops per pair on the integer VALU
                tops = pairs * 19 / t_s / 1e12
                roofline["match_valu"] = {"bound": "int-valu", "pairs_per_launch": int(pairs), "ops_per_pair": 19,
                                          "achieved": round(tops, 2), "peak": round(VALU_PEAK_TOPS, 2),
                                          "unit": "Tlane-op/s", "frac": round(tops / VALU_PEAK_TOPS, 3)}
            else:
                # distances on the matrix cores as FP4 +-1 dot products (2*256 flop per pair, dense FP4 peak ~10
                # PFLOP/s); the top-2 selection is one v_med3 + one v_max per pair on the VALU (34 lane-ops per 16
                # pairs), which is the pipe that bounds it
                roofline["match_mfma"] = {"bound": "mfma", "dtype": "fp4 (+-1, exact)", "pairs_per_launch": int(pairs),
                                          "achieved": round(pairs * 512 / t_s / 1e12, 1), "peak": MFMA_FP4_PEAK_TFLOPS,
                                          "unit": "TFLOP/s",
                                          "frac": round(pairs * 512 / t_s / 1e12 / MFMA_FP4_PEAK_TFLOPS, 3)}
                tops = pairs * (34 / 16) / t_s / 1e12
                roofline["match_valu"] = {"bound": "valu-issue", "ops_per_pair": 34 / 16, "achieved": round(tops, 2),
                                          "peak": round(VALU_PEAK_TOPS, 2), "unit": "Tlane-op/s",
                                          "frac": round(tops / VALU_PEAK_TOPS, 3)}
        w, h, _ = ctx.level_geometry()
        P = sum(x * y for x, y in zip(w, h))
        extract_bytes = 3 * a.width * a.height + 2 * P + 48 * (kp_b / B)
        out = {
            "metric": baseline_metric(), "value": kp_total / dt_max,
            "unit": "keypoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "cfg2: synthetic %dx%d stream, 8-level ORB (reference defaults 1.2/20/7/min-area 1000), "
                                   "extract + BF-Hamming knn-2 match (ratio 0.7) vs previous frame%s" % (
                           a.width, a.height, (" + DBoW3 k=10 L=%d loop scoring vs last 64 frames%s" % (
                               a.voc_levels, " + RCCL all-gather of BoW vectors, cross-stream scores" if world > 1 else ""))
                           if a.bow else ""),
                       "warmup_steps_run": w_eff,  # at least one per distinct batch: the warm-up also counts the keypoints
                       "frames_per_step": B, "frames_per_gpu": a.steps * B, "distinct_frames": n_unique,
                       "keypoints_per_frame": round(kp_b / B, 1), "fast_candidates_per_frame": round(cand_b / B, 1),
                       "frames_per_s": a.steps * B * world / dt_max,
                       "extract_algorithmic_GBps_whole_job": extract_bytes * a.steps * B * world / dt_max / 1e9},
            "roofline": roofline,
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, a.cpu_sample)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
