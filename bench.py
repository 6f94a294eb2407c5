#!/usr/bin/env python3
"""bench.py — ORB extract + BF-Hamming match throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of HBM-resident synthetic RGB-D frames:
detect (gray -> pyramid -> FAST cells -> quadtree -> blur -> orientation + rBRIEF), the knn-2 Hamming
match of every frame against its predecessor (chained across batches) — the call order of
RgbdFeatureFrontend (rgbd_feature_frontend.cpp:187,237) — and the depth lookup + back-projection of the
keypoints (rgbd_feature_frontend.cpp:101-138), the step that consumes the D of RGB-D.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  When started WITHOUT a torch.distributed environment, bench.py spawns the N
ranks itself (a child `python -m torch.distributed.run ... bench.py <same arguments>`, before anything
touches the GPU) and exits with the child's code; started by torch.distributed.run it reads RANK /
LOCAL_RANK / WORLD_SIZE.  Every rank runs its own stream (seed 1234 + 100*rank); extract + match need no
data-path collective ("weak" scaling); timing is barrier + synchronize on both sides and the MAX over
ranks.  With N > 1 the loop-candidate exchange (DBoW3 vectors, ONE all-gather per batch over RCCL,
cross-stream scoring) is exercised and timed after the headline region ("exchange"); `--bow` puts BoW
scoring and the exchange inside the step (cfg3 / cfg4).

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launching stream for
the dominant kernel; `cpu_baseline` times the CPU oracle (oracle/, the restatement of the reference's
CPU plugin) on a bounded sample of the same stream on the host cores; `pcie_inclusive` is the same
pipeline fed from pinned host memory (double-buffered H2D) with the results copied back (D2H).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
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
STAGE_KERNEL = {"levels": "void mslam::k_level_chain<false, 8, true, W, C> (gray + blur and every resize + blur level, one launch per chunk)",
                "pnp_gather": "mslam::k_pnp_gather", "pnp_ransac": "mslam::k_pnp_ransac_batch", "gray": "void mslam::k_gray_blur<true, 0>",
                "resize": "void mslam::k_resize_blur<false, N, true, 0> (one launch per level)", "fast": "mslam::k_fast_cells",
                "quadtree": "mslam::k_quadtree", "blur": "mslam::k_blur2", "describe": "void mslam::k_describe<true>",
                "match_knn2": "void mslam::k_match_knn2_fp4<4, false, true>", "ratio_compact": "mslam::k_ratio_compact",
                "backproject": "mslam::k_backproject"}
POPCOUNT_KERNEL = "void mslam::k_match_knn2<8, 1, 8>"
# kernel name (rocprofv3, without the argument list) -> stage of the step; a stage can be several kernels / launches
STAGE_PREFIXES = (("mslam::k_level_chain", "levels"), ("mslam::k_gray", "gray"), ("mslam::k_resize", "resize"),
                  ("mslam::k_blur", "blur"), ("mslam::k_fast", "fast"), ("mslam::k_zero_u32", "fast"),
                  ("mslam::k_quadtree", "quadtree"), ("mslam::k_cv_select", "select"), ("mslam::k_describe", "describe"),
                  ("mslam::k_match_knn2", "match_knn2"), ("mslam::k_ratio_compact", "ratio_compact"),
                  ("mslam::k_backproject", "backproject"), ("mslam::k_pnp_gather", "pnp_gather"),
                  ("mslam::k_pnp_ransac", "pnp_ransac"), ("mslam::k_bow_descend", "bow_descend"),
                  ("mslam::k_bow_flat", "bow_descend"), ("mslam::k_bow_vector", "bow_vector"),
                  ("mslam::k_bow_score", "bow_score"), ("mslam::k_bow_sum", "bow_score"),
                  ("mslam::k_merge_ratio", "ratio_compact"), ("mslam::k_denorm_selfcheck", "setup"))


def stage_of_kernel(name):
    if name.startswith("void "):  # template instances are printed with their return type
        name = name[5:]
    for pre, st in STAGE_PREFIXES:
        if name.startswith(pre):
            return st
    return None
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc_per_step.json")  # tools/summarize_counters.py
K2000_MIN_AREA = 984   # 640x480: 1986 keypoints per frame on the synthetic stream (1000, the reference default: ~1890)
CFG4_MIN_AREA = 6340   # 1280x720, 8 levels: ~2015 keypoints per frame
CLOCK_HZ = 2.4e9  # MI355X max shader clock (MI355X_MICROARCH.md); the vector-ALU issue figures are quoted at this clock


def csrc_sha():
    """fingerprint of the kernel sources: a committed PMC profile is only quoted for the sources it was taken on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "modular-slam_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def pmc_profile():
    """(per-stage counter totals per step, frames per step, note): the committed rocprofv3 PMC passes, or (None, ..)
    with the reason when there is none or it was taken on other kernel sources"""
    try:
        j = json.load(open(PMC_PROFILE))
        meta = j["_meta"]
    except (OSError, KeyError, ValueError):
        return None, 1000, "no PMC profile committed (%s)" % os.path.basename(PMC_PROFILE)
    if meta.get("csrc_sha") != csrc_sha():
        return None, 1000, "stale: %s was collected on kernel sources %s, this build is %s" % (
            os.path.basename(PMC_PROFILE), meta.get("csrc_sha"), csrc_sha())
    return j["stages"], float(meta.get("frames_per_step", 1000)), (
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes), summed over the launches of a step, "
        "(2*FETCH + WRITE) KB, %s" % os.path.basename(PMC_PROFILE))


MEMPATH_PROFILE = os.path.join(ROOT, "profiles", "r06_mempath.json")  # tools/mempath_counters.sh + tools/summarize_mempath.py


def busy_fracs(stage):
    """(TA busy fraction, vector-ALU issue fraction, note) of a stage ALONE on the GPU, from the committed memory-path passes
    (`TA_TA_BUSY_sum` / `SQ_INSTS_VALU`, each with `GRBM_GUI_ACTIVE` in its own rocprofv3 pass over the serialized stage run):
    counter sums over every launch of the stage's kernels, TA busy cycles per CU (and vector instructions x 4 cycles per SIMD)
    over the kernels' own active cycles per XCD — no clock assumption.  (None, None, reason) when there is no profile or it was
    collected on other kernel sources."""
    try:
        j = json.load(open(MEMPATH_PROFILE))
        meta = j["_meta"]
    except (OSError, KeyError, ValueError):
        return None, None, "no memory-path profile committed (%s)" % os.path.basename(MEMPATH_PROFILE)
    if meta.get("csrc_sha") != csrc_sha():
        return None, None, "stale: %s was collected on kernel sources %s, this build is %s" % (
            os.path.basename(MEMPATH_PROFILE), meta.get("csrc_sha"), csrc_sha())
    out = []
    for key in ("ta_busy_frac_sums", "valu_issue_frac_sums"):
        x = g = 0.0
        for name, row in j["kernels"].items():
            if stage_of_kernel(name if name.startswith("mslam::") else "void " + name) == stage and key in row:
                x += row[key]["counter"]
                g += row[key]["GRBM_GUI_ACTIVE"]
        scale = 1.0 / 256.0 if key.startswith("ta") else 4.0 / 1024.0
        out.append(x * scale / (g / 8.0) if g > 0 else None)
    return out[0], out[1], "rocprofv3 --pmc TA_TA_BUSY_sum / SQ_INSTS_VALU with GRBM_GUI_ACTIVE (separate passes), %s" % os.path.basename(
        MEMPATH_PROFILE)


def pmc_traffic(stage, frames_per_step):
    """HBM-side bytes per STEP of a stage from the committed PMC passes (FETCH_SIZE and WRITE_SIZE are collected in
    separate runs, in KB; on gfx950 FETCH_SIZE counts half of the bytes of coalesced streaming reads —
    MI355X_MICROARCH.md §HBM — hence the factor 2, which the gray kernel's known input confirms)."""
    st, fps, note = pmc_profile()
    if st is None or stage not in st or "FETCH_SIZE" not in st[stage]:
        return None, note
    d = st[stage]
    return int((2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024 * frames_per_step / fps), note


def valu_issue_ms(stage, frames_per_step):
    """vector-ALU issue time of a stage per step: SQ_INSTS_VALU x 4 cycles over 1024 SIMDs at 2.4 GHz (None when there is
    no valid profile).  stage = None: the whole step."""
    st, fps, _ = pmc_profile()
    if st is None:
        return None
    names = [stage] if stage else list(st)
    tot = sum(st[n].get("SQ_INSTS_VALU", 0.0) for n in names if n in st)
    return tot * 4 / (1024 * CLOCK_HZ) * 1e3 * frames_per_step / fps


def baseline_metric():
    """the metric string of BASELINE.json (this file sits next to it)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "ORB keypoints extracted+matched /sec, 640×480 RGB-D, 1/2/4/8 GPU; HBM GB/s vs roofline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1000, help="frames per step (one step = the 1000-frame stream of cfg2)")
    ap.add_argument("--unique", type=int, default=1000, help="distinct synthetic frames kept in HBM (cycled)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--min-area", type=int, default=1000, help="quadtree stop area (reference default 1000)")
    ap.add_argument("--detector", default="distributed", choices=["distributed", "cvorb"],
                    help="distributed = DistributedOrbOpenCvDetector (the headline); cvorb = the cv::ORB-based "
                         "OrbOpenCvDetector drop-in (n_features keypoints per frame)")
    ap.add_argument("--n-features", type=int, default=1000, help="cvorb: cv::ORB::create(nfeatures)")
    ap.add_argument("--bow", action="store_true",
                    help="cfg3/cfg4: DBoW3 loop scoring every frame (synthetic k=10 vocabulary) inside the step and, "
                         "with N>1, the RCCL all-gather of BoW vectors + cross-stream scoring")
    ap.add_argument("--voc-levels", type=int, default=6, help="vocabulary depth L (k=10): 6 -> 1e6 words")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real multi-GPU run) or gloo (rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pnp", action="store_true",
                    help="also estimate the frame-to-frame pose of every frame in the step (batched RANSAC PnP on the "
                         "matches + back-projected points: mslam_hip_pnp_batch_dev)")
    ap.add_argument("--exchange-granularity", default="batch", choices=["batch", "frame"],
                    help="loop-candidate exchange: one all-gather per batch (16 KB x frames per rank) or one per frame (16 KB per rank)")
    ap.add_argument("--extras-timeout", type=int, default=300,
                    help="with several ranks: seconds the legs after the timed region may take before rank 0 prints "
                         "the line without them")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the cfg3 / cfg5 / single-frame-latency legs that follow the timed region of the default run")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs after the timed region (popcount matcher, serialized stages, PCIe-inclusive, "
                         "exchange): what the profiling passes use")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU work budget of the cpu_baseline leg")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="single-GPU runs: create a one-rank process group on the nccl (= RCCL) backend and issue the cfg4 leg's "
                         "all-gather as a real RCCL collective instead of the world-1 device copy (what one GPU can exercise of "
                         "the RCCL path; not part of the default run: a box without a usable RCCL would hang it)")
    return ap.parse_args()


def spawn_ranks(a):
    """python bench.py --gpus N without a torch.distributed environment: start the N ranks as a child process
    (never exec: under rocprofv3 this process may already have initialised the GPU) and return its exit code"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def stage_bytes(ctx, B, n_kp, n_cand, voc_k=10, voc_L=6, db_entries=64):
    """Algorithmic HBM bytes per STEP for every stage: each input byte read once, each output byte
    written once, for a batch of B frames with n_kp keypoints / n_cand FAST candidates in total."""
    w, h, _ = ctx.level_geometry()
    px = [a * b for a, b in zip(w, h)]
    P = sum(px)
    return {
        # level kernels (k_level.hip): gray + blur writes level 0 twice (raw, blurred); resize + blur reads level l-1 and
        # writes level l twice
        "gray": B * (3 * px[0] + 2 * px[0]),
        "resize": B * sum(px[l - 1] + 2 * px[l] for l in range(1, len(px))),
        # the one-launch level chain (k_level_chain) = gray + resize
        "levels": B * (3 * px[0] + 2 * px[0]) + B * sum(px[l - 1] + 2 * px[l] for l in range(1, len(px))),
        "fast": B * P + 4 * n_cand,
        "quadtree": 4 * n_cand + 4 * n_kp,
        "blur": B * 2 * P,
        "describe": B * 2 * P + 48 * n_kp,  # reads both planes around each keypoint, writes desc+xy+angle+octave+resp
        "match_knn2": 2 * 32 * n_kp + 16 * n_kp,
        "ratio_compact": 12 * n_kp + 8 * n_kp,
        "backproject": 8 * n_kp + 2 * n_kp + 25 * n_kp,
        # matches (2 x i32) + 3-D point (24) + validity + pixel (8) in, correspondences (20) out; then read per hypothesis
        "pnp_gather": 0.3 * n_kp * (8 + 25 + 8 + 20),
        "pnp_ransac": 0.3 * n_kp * 20 * 100,
        # bow_tree = n*(32 + L*k*32) + n*12 out; vectors are <= n x (u32, f64)
        "bow_descend": n_kp * (32 + voc_L * voc_k * 32) + 12 * n_kp,
        "bow_vector": 12 * n_kp + 12 * n_kp,
        "bow_score": db_entries * 2 * 12 * n_kp,
    }


def host_cpu_share():
    """What this process may actually use of the host: the affinity mask and the cgroup CPU quota (a container on a
    256-thread host is usually given a fraction of it), plus the cgroup's throttling clock for before / after deltas."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    return affinity, quota


def cgroup_throttled_s():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()[:2]
                if k == "throttled_usec":
                    return float(v) * 1e-6
                if k == "throttled_time":
                    return float(v) * 1e-9
        except (OSError, ValueError):
            continue
    return None


def cpu_baseline(frames, params_kw, seconds, cv=False, n_features=1000):
    """Oracle (= port of the reference CPU plugin) detect + match on the host cores, timed by the C harness
    oracle/mslam_cpu_bench.c: frames sharded over pthreads, one mso_detect + mso_match loop per thread, at least 8 frames
    per thread — first one thread, then as many threads as this process may use (affinity mask, cgroup quota)."""
    import __graft_entry__ as graft
    orc = graft.load_oracle()
    orc.lib()
    affinity, quota = host_cpu_share()
    cores = max(1, min(affinity, int(np.ceil(quota)) if quota else affinity))
    if cv:
        p, cvp = None, orc.cvorb_params(n_features=n_features, n_levels=params_kw["n_levels"])
    else:
        p, cvp = orc.params(**params_kw), None
    sample = np.ascontiguousarray(frames[:min(len(frames), 64)])
    # size the two legs from a short probe so that the whole leg stays within its budget on any host
    probe = orc.bench_stream(sample, p, 1, 2, cv_params=cvp)
    per_frame = max(probe["seconds"] / 2, 1e-3)
    n1 = int(max(8, 0.3 * seconds / per_frame))
    r1 = orc.bench_stream(sample, p, 1, n1, cv_params=cvp)
    v1 = r1["keypoints"] / r1["seconds"]
    # all threads: 8 frames per thread first; when that took less than 5 s, again with the block length that fills
    # max(5 s, 0.7 x budget) at the rate the first run measured (a host that gives this process fewer cores than it
    # shows stretches the wall time, not the budget)
    per_thread = 8
    thr0 = cgroup_throttled_s()
    rall = orc.bench_stream(sample, p, cores, per_thread, cv_params=cvp)
    thr1 = cgroup_throttled_s()
    want = max(5.0, 0.7 * seconds)
    for _ in range(3):
        if rall["seconds"] >= 5.0:
            break
        per_thread = int(min(4096, max(per_thread + 1, np.ceil(1.1 * per_thread * want / max(rall["seconds"], 1e-3)))))
        thr0 = cgroup_throttled_s()
        rall = orc.bench_stream(sample, p, cores, per_thread, cv_params=cvp)
        thr1 = cgroup_throttled_s()
    vall = rall["keypoints"] / rall["seconds"]
    eff = vall / (cores * v1)
    out = {"value": vall, "unit": "keypoints/s", "cores": cores, "kind": "port", "value_1core": v1,
           "scaling_efficiency": eff, "host_cpu_count": os.cpu_count(), "affinity_cpus": affinity,
           "cgroup_cpu_quota": quota, "frames_per_thread": per_thread, "seconds_all_cores": rall["seconds"],
           "thread_seconds_min_max": [rall["thread_seconds_min"], rall["thread_seconds_max"]],
           "harness": "oracle/mslam_cpu_bench.c (pthreads, one detect + match loop per thread)",
           "sample": "the same synthetic stream (%d distinct frames, taken cyclically), detect + knn-2 match vs the "
                     "previous frame, oracle/ C restatement (scalar, -O2 -mpopcnt; OpenCV's internal SIMD is not "
                     "reproduced): %d frames on %d threads (%.1f s), %d frames on 1 thread (%.1f s)"
                     % (len(sample), rall["frames"], cores, rall["seconds"], n1, r1["seconds"])}
    if thr0 is not None and thr1 is not None:
        out["cgroup_throttled_s_during_all_cores"] = thr1 - thr0
    if eff < 0.3:
        # a thread's loop takes per_thread x per_frame seconds when it has a core to itself
        slow = rall["thread_seconds_max"] / max(per_thread * (r1["seconds"] / n1), 1e-9)
        why = "the slowest thread's loop took %.1fx the one-thread time per frame" % slow
        if out.get("cgroup_throttled_s_during_all_cores", 0) > 0.05 * rall["seconds"]:
            why += "; the cgroup throttled this process for %.1f s of CPU time during the leg (CPU quota)" % (thr1 - thr0)
        elif quota is None and affinity >= (os.cpu_count() or 1):
            why += "; no cgroup quota or affinity limit is visible, so the threads share cores with other tenants of " \
                   "the host or are bound by its memory system"
        out["binds_scaling"] = why
    return out


def cfg4_multi_rank_leg(a, pkg, synth, torch, dist, rank, world, dev, red_dev, W4=1280, H4=720, B4=250, n_steps=8):
    """cfg4 of BASELINE.json at world > 1: every rank extracts + matches its OWN 1280x720 stream (seed
    1234 + 100 * rank), builds the DBoW3 vectors of the batch, and the exchange — pack, ONE all_gather_into_tensor per batch
    (or one per frame), cross-stream L1 scores — runs INSIDE the timed step on the communication stream.  Barrier +
    synchronize on both sides, MAX over ranks; keypoints summed over ranks.  After the timed region (never inside it) every
    rank checks the HIP cross scores of two frames against the CPU oracle's L1 score on the vectors exactly as they were
    transmitted (the oracle is the checker here, nothing it computes is timed or reported as a rate)."""
    import __graft_entry__ as graft
    from modular_slam_amd.multi_stream import CrossStreamLoopCandidates, set_dwords, unpack_set
    f4 = synth.make_stream(B4, W4, H4, seed=1234 + 100 * rank)
    d4 = torch.from_numpy(f4).cuda()
    dd4 = torch.from_numpy(np.ascontiguousarray(
        np.stack([synth.make_depth(1, W4, H4, seed=1234 + 100 * rank)[0]] * B4)).view(np.int16)).cuda()
    area4 = -(-W4 * H4 // (640 * 480))
    ts4 = torch.cuda.Stream()
    ctx4 = pkg.Context(width=W4, height=H4, max_batch=B4, n_levels=8, min_node_area=CFG4_MIN_AREA,
                       max_keypoints=min(8192, 4096 * area4), max_candidates=16384 * area4, device=dev, stream=ts4.cuda_stream)
    ctx4.bow_load(synth.make_vocabulary(10, a.voc_levels, seed=77))
    ctx4.bow_db_reserve(32 * B4)
    cross4 = CrossStreamLoopCandidates(k_max=2048, granularity=a.exchange_granularity)

    def step4(i):
        ctx4.detect_batch_dev(d4.data_ptr(), B4)
        ctx4.match_batch_dev(0.7, True)
        ctx4.backproject_batch_dev(dd4.data_ptr())
        ctx4.bow_batch_dev(True)
        return cross4.step_gpu(ctx4, ts4, B4)

    def settle4():
        cross4.finish(ts4)
        ctx4.sync()
        torch.cuda.synchronize()
        dist.barrier()

    def reduce(x, op):
        t = torch.tensor([float(x)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    for i in range(2):
        step4(i)
    settle4()
    kp4 = int(pkg.read_device(ctx4, ctx4.batch_view().count, (B4,), np.int32).sum())
    c0 = cross4.collectives
    t0 = time.perf_counter()
    for i in range(n_steps):
        scores = step4(i)
    settle4()
    dt = reduce(time.perf_counter() - t0, dist.ReduceOp.MAX)
    n_coll = cross4.collectives - c0
    kp_all = reduce(kp4, dist.ReduceOp.SUM)
    # the exchange alone (pack + gather + cross scores of the last batch), on the streams the step uses
    n_ex = 5
    t0 = time.perf_counter()
    for i in range(n_ex):
        scores = cross4.step_gpu(ctx4, ts4, B4)
    settle4()
    dt_ex = reduce(time.perf_counter() - t0, dist.ReduceOp.MAX)
    # ---- oracle check, outside every timed region: own frames 0 and B4-1 against frame t of EVERY stream, from the gathered
    # buffer as it arrived on this rank
    orc = graft.load_oracle()
    _, gathered, _ = cross4.last
    s_host = scores.cpu().numpy()
    sets = [unpack_set(gathered[r], B4, cross4.k_max) for r in range(world)]
    mw, mv, mn = sets[rank]
    checked, ok = [], True
    for t in (0, B4 - 1):
        for r in range(world):
            w2, v2, n2 = sets[r]
            exp = orc.bow_score_l1(mw[t, :mn[t]], mv[t, :mn[t]], w2[t, :n2[t]], v2[t, :n2[t]])
            ok = ok and bool(exp == s_host[t, r])
        checked.append(t)
    ok = ok and bool(mn.min() > 0) and bool((s_host[:, rank] > 0.99).all())
    all_ok = reduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) == 1.0
    out = {"workload": "cfg4: %d independent synthetic 1280x720 RGB-D streams (one per rank, seed 1234 + 100 x rank, %d distinct "
                       "frames each), 8 levels, min-area %d, extract + knn-2 match + back-projection + DBoW3 (k=10, L=%d) "
                       "vectors / scores / inverted-file adds + the cross-stream exchange (pack, all_gather_into_tensor, "
                       "cross scores) inside the step" % (world, B4, CFG4_MIN_AREA, a.voc_levels),
           "value": kp_all * n_steps / dt, "unit": "keypoints/s", "n_gpus": world, "scaling": "weak",
           "ms_per_step": dt / n_steps * 1e3, "frames_per_step_per_rank": B4, "steps": n_steps,
           "keypoints_per_frame": round(kp_all / world / B4, 1),
           "timing": "barrier + synchronize on both sides, MAX over ranks; keypoints summed over ranks",
           "exchange": {"backend": dist.get_backend(), "world_size": world, "granularity": cross4.granularity,
                        "k_max": cross4.k_max, "collectives_per_step": n_coll / float(n_steps),
                        "bytes_per_rank_per_collective": cross4.bytes_per_collective,
                        "bytes_per_rank_per_batch": 4 * set_dwords(B4, cross4.k_max),
                        "ms_per_batch": dt_ex / n_ex * 1e3,
                        "what": "mslam_hip_bow_pack_dev -> all_gather_into_tensor -> mslam_hip_bow_cross_score_packed_dev "
                                "alone (MAX over ranks); inside the step it runs on the communication stream beside the next "
                                "batch's extraction"},
           "oracle_check": all_ok,
           "oracle_check_what": "every rank, after the timed region: HIP cross scores of own frames %s against frame t of all %d "
                                "streams == the CPU oracle's L1 score on the vectors as transmitted (f32 values); MIN over "
                                "ranks" % (checked, world)}
    ctx4.close()
    return out


def dry_run(a, rank, world):
    """launcher rehearsal without a GPU (MSLAM_BENCH_DRY=1, used by the CPU test of `--gpus N`): rendezvous, the
    barrier + MAX/SUM reductions of the real run, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    k = torch.tensor([100.0], dtype=torch.float64)
    exchange = None
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        # the loop-candidate exchange of the real run, on host tensors of the real shape: ONE all_gather_into_tensor of
        # the rank's packed BoW set per batch (modular_slam_amd/multi_stream.py), here 3 batches of 8 frames
        import __graft_entry__ as graft
        graft.load_package()
        from modular_slam_amd.multi_stream import CrossStreamLoopCandidates, set_dwords, pack_vectors, unpack_set
        cross = CrossStreamLoopCandidates(k_max=2048)
        n_b, fr = 3, 8
        for b in range(n_b):
            local = torch.full((set_dwords(fr, cross.k_max),), rank * 1000 + b, dtype=torch.int32)
            got = cross.all_gather_sets(local)
            assert got.shape[0] == world and all(int(got[r].view(-1)[0]) == r * 1000 + b for r in range(world))
        exchange = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                    "collectives_per_batch": cross.collectives / float(n_b), "bytes_per_rank_per_batch": 4 * set_dwords(fr, cross.k_max)}
        # the cfg4 leg of the real run at world > 1, rehearsed on host tensors: per-rank BoW vectors (ragged counts) in the
        # wire format, the per-batch collective, cross scores of own frame t against frame t of every stream (a numpy L1
        # here; the HIP kernel on the GPU box), the oracle check on the vectors as transmitted, and the MIN / MAX / SUM
        # reductions of cfg4_multi_rank_leg
        orc = graft.load_oracle()
        rng = np.random.default_rng(4321 + rank)
        B4, cap = 6, 512
        W = torch.zeros((B4, cap), dtype=torch.int32)
        V = torch.zeros((B4, cap), dtype=torch.float64)
        N = torch.zeros(B4, dtype=torch.int32)
        for t_ in range(B4):
            n_ = int(rng.integers(100, cap))
            W[t_, :n_] = torch.from_numpy(np.sort(rng.choice(5000, n_, replace=False)).astype(np.int32))
            v_ = rng.random(n_)
            V[t_, :n_] = torch.from_numpy(v_ / v_.sum())
            N[t_] = n_

        def l1(w1, v1, w2, v2):
            d = dict(zip(w2.tolist(), v2.tolist()))
            acc = 0.0
            for w_, x_ in zip(w1.tolist(), v1.tolist()):  # ascending word order, as the reference's std::map walk
                if w_ in d:
                    acc += abs(x_ - d[w_]) - abs(x_) - abs(d[w_])
            return -acc / 2.0
        c1 = cross.collectives
        got4 = cross.step_with(W, V, N, l1)
        sets4 = cross.all_gather_sets(pack_vectors(W, V, N, cross.k_max), n_frames=B4)
        un4 = [unpack_set(sets4[r], B4, cross.k_max) for r in range(world)]
        ok4 = True
        for t_ in (0, B4 - 1):
            for r in range(world):
                exp = orc.bow_score_l1(un4[rank][0][t_, :N[t_]], un4[rank][1][t_, :N[t_]],
                                       un4[r][0][t_, :un4[r][2][t_]], un4[r][1][t_, :un4[r][2][t_]])
                ok4 = ok4 and abs(exp - got4[t_, r]) < 1e-12
        okt = torch.tensor([1.0 if ok4 else 0.0], dtype=torch.float64)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        cfg4 = {"n_gpus": world, "exchange": {"backend": dist.get_backend(), "world_size": world,
                                              "collectives_per_step": (cross.collectives - c1) / 2.0,
                                              "bytes_per_rank_per_batch": 4 * set_dwords(B4, cross.k_max)},
                "oracle_check": bool(okt.item() == 1.0)}
    if rank == 0:
        line = {"dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "dist": {"backend": "gloo" if world > 1 else None, "world_size": world,
                         "launched_by": os.environ.get("MSLAM_BENCH_LAUNCHER", "external")},
                "t_max": float(t.item()), "units": float(k.item())}
        if exchange is not None:
            line["exchange"] = exchange
            line["cfg4"] = cfg4
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        os.environ["MSLAM_BENCH_LAUNCHER"] = "bench.py"
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (a.gpus, world))
        sys.exit(2)
    if os.environ.get("MSLAM_BENCH_DRY") == "1":
        return dry_run(a, rank, world)
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
    from modular_slam_amd.multi_stream import CrossStreamLoopCandidates, view_as_tensor, set_dwords

    B = a.batch
    n_unique = max(B, (a.unique // B) * B)
    frames = synth.make_stream(n_unique, a.width, a.height, seed=1234 + 100 * rank)
    d_frames = torch.from_numpy(frames).cuda()
    depth = synth.make_depth(1, a.width, a.height, seed=1234 + 100 * rank)[0]
    # depth of frame t = the base depth map shifted like the texture (a cheap stand-in for a moving camera)
    d_depth = torch.from_numpy(np.ascontiguousarray(
        np.stack([np.roll(depth, (-(t % 48), -((2 * t) % 64)), (0, 1)) for t in range(n_unique)])).view(np.int16)).cuda()
    frame_bytes = a.width * a.height * 3
    depth_bytes = a.width * a.height * 2

    # capacities scale with the frame area (4096 keypoints / 16384 FAST candidates per level at 640x480)
    area = max(1, -(-a.width * a.height // (640 * 480)))
    k_scale = area * max(1, 1000 // max(a.min_area, 1))
    ts = torch.cuda.Stream()  # the context's stream is a torch stream: torch copies / collectives order against it
    cv = a.detector == "cvorb"
    if cv:
        STAGE_KERNEL.update({"resize": "void mslam::k_resize_blur<true, N, true, 0> (one launch per level)", "fast": "mslam::k_fast_tiles",
                             "select": "mslam::k_cv_select"})
    ctx = pkg.Context(width=a.width, height=a.height, max_batch=B, n_levels=a.levels, min_node_area=a.min_area,
                      max_keypoints=min(32736, max(4096 * k_scale, 2 * a.n_features if cv else 0)),  # 32736: the matrix-core matcher's train range
                      max_candidates=16384 * area, device=dev, stream=ts.cuda_stream,
                      detector=pkg.DETECTOR_CV_ORB if cv else pkg.DETECTOR_DISTRIBUTED, n_features=a.n_features)
    n_batches = n_unique // B
    cross = None
    need_voc = a.bow or (world > 1 and not a.no_extras)
    if need_voc:
        ctx.bow_load(synth.make_vocabulary(10, a.voc_levels, seed=77))
        if a.bow:
            # every step adds a batch of entries to the inverted file: reserve them, so that no storage doubling (a
            # reallocation + copy of the posting log) falls into the timed steps
            # (at most 2^31 / max_keypoints entries are addressable; beyond the reservation the storage doubles as usual)
            ctx.bow_db_reserve(min((max(a.warmup, n_batches) + 3 * a.steps + 16) * B, (1 << 31) // ctx.params.max_keypoints - 1))
        cross = CrossStreamLoopCandidates(k_max=2048 * k_scale, granularity=a.exchange_granularity)

    def step(i, bow=a.bow):
        off = (i % n_batches) * B
        ctx.detect_batch_dev(d_frames.data_ptr() + off * frame_bytes, B)
        ctx.match_batch_dev(0.7, True)
        ctx.backproject_batch_dev(d_depth.data_ptr() + off * depth_bytes)
        if a.pnp:
            ctx.pnp_batch_dev(seed=i)
        if bow:
            ctx.bow_batch_dev(True)
            if world > 1:
                cross.step_gpu(ctx, ts, B)

    def settle():
        if cross is not None:
            cross.finish(ts)
        ctx.sync()  # also surfaces capacity overflows loudly
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # warm-up; the first pass over each distinct batch also counts its keypoints (the unit of the metric) and
    # FAST candidates, so that no extra launches are needed after the timed region
    counts_per_batch, cand_per_batch = {}, {}

    def count_batch(b):
        ctx.sync()
        v = ctx.batch_view()  # the output set alternates between batches: fetch the view after every detect
        counts_per_batch[b] = int(pkg.read_device(ctx, v.count, (B,), np.int32).sum())
        cand_per_batch[b] = int(ctx.debug_counts(pkg.DBG_CANDIDATES, B).sum())

    w_eff = max(a.warmup, n_batches)
    for i in range(w_eff):
        step(i)
        if i % n_batches not in counts_per_batch:
            count_batch(i % n_batches)
    settle()

    def timed_run(first, profile):
        if profile:
            ctx.set_profiling(2)  # HIP events around every stage launch, in place on its stream
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(first + i)
        settle()
        dt = time.perf_counter() - t0
        timed = ctx.stage_times(cap=8192) if profile else None  # (stage, ms) of every launch of the timed steps
        if profile:
            ctx.set_profiling(0)
        t_max = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item()), timed

    dt_max, timed_all = timed_run(w_eff, True)

    # units processed: keypoints extracted (and matched against the previous frame) in the timed steps
    n_kp = sum(counts_per_batch[(w_eff + i) % n_batches] for i in range(a.steps))
    kp_sum = torch.tensor([float(n_kp)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(kp_sum, op=dist.ReduceOp.SUM)
    kp_total = float(kp_sum.item())

    def make_line(extras, with_cpu=True):
        timed = timed_all
        # per-stage launch durations measured inside the timed region (HIP events on the launching stream);
        # a step's entries end with its last stage; a very long run stops recording at 8192 entries: keep whole steps
        last = "bow_score" if a.bow else "pnp_ransac" if a.pnp else "backproject"
        ends = [i for i, (name, _) in enumerate(timed) if name == last]
        steps_cov = max(len(ends), 1)
        timed = timed[:ends[-1] + 1] if ends else timed
        tot, cnt = {}, {}
        for name, ms in timed:
            tot[name] = tot.get(name, 0.0) + ms
            cnt[name] = cnt.get(name, 0) + 1
        avg = {k: tot[k] / cnt[k] for k in tot}                    # ms per launch
        per_step = {k: cnt[k] / float(steps_cov) for k in tot}       # launches per step (detector: one per chunk)
        acc = {k: tot[k] / steps_cov for k in tot}                   # ms per step, summed over the step's launches
        kp_b, cand_b = counts_per_batch[0], cand_per_batch[0]
        sb = stage_bytes(ctx, B, kp_b, cand_b, 10, a.voc_levels)   # per step (B frames)
        if cv:  # the selection stage of the cv::ORB mode stands where the quadtree is (same order of bytes)
            sb["select"] = sb["quadtree"]
        # the kernel that binds: the stage with the largest duration ALONE on the GPU (the serialized pass; an in-place
        # interval on one of the two chunk streams is inflated unevenly by the sibling chunk's kernels), falling back to
        # the in-place sums when the serialized pass did not run (--no-extras)
        ser = extras.get("stages_ms_serialized", {})
        ser_ok = {k: v for k, v in ser.items() if k in sb and k in acc}
        dom = max(ser_ok, key=ser_ok.get) if ser_ok else max((k for k in acc if k in sb), key=acc.get)
        fpl = B / per_step[dom]                                    # frames per launch of the dominant stage
        bytes_per_launch = sb[dom] / per_step[dom]
        achieved = bytes_per_launch / (avg[dom] * 1e-3) / 1e9
        kern = STAGE_KERNEL.get(dom, dom)
        traffic_step, traffic_note = pmc_traffic(dom, B)
        traffic = int(traffic_step / per_step[dom]) if traffic_step is not None else None
        # the kernel's own duration: alone on the GPU when the serialized pass ran (in place it shares the GPU with
        # the sibling chunk's kernels, which would halve the fraction)
        alone_step_ms = ser.get(dom)                                # ms per step with every launch of the stage alone
        frac_alone = (sb[dom] / (alone_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if alone_step_ms else None
        v_ms = valu_issue_ms(dom, B)
        valu_frac_alone = (v_ms / alone_step_ms) if (v_ms is not None and alone_step_ms) else None
        step_valu = valu_issue_ms(None, B)
        step_ms = dt_max / a.steps * 1e3
        # `bound` names the roofline that achieved / peak / unit / frac are quoted against (HBM: BASELINE.json's metric);
        # `binding_limit` says what actually binds the dominant kernel: the larger of its HBM fraction and its vector-ALU
        # issue fraction (both alone)
        bound = "hbm"
        # the unit that binds the dominant kernel: the largest of its three busy fractions alone on the GPU — HBM (algorithmic
        # bytes over the peak), vector-ALU issue and the texture-address path (TA busy).  The last two come from the committed
        # memory-path passes over the kernel's own active cycles when they match these sources (no clock assumption), else
        # the vector figure falls back to SQ_INSTS_VALU x 4 cycles at the nominal 2.4 GHz
        ta_busy, valu_busy, busy_note = busy_fracs(dom)
        cands = {"hbm": frac_alone, "valu-issue": valu_busy if valu_busy is not None else valu_frac_alone, "texture-address (TA)": ta_busy}
        cands = {k: v for k, v in cands.items() if v is not None}
        binding = max(cands, key=cands.get) if cands else "hbm"
        limiter = "not HBM (see DESIGN.md §4: every stage but gray is issue-, TA- or latency-bound)"
        if cands:
            limiter = "alone on the GPU: " + ", ".join("%s %.0f %%" % (k, 100 * v) for k, v in sorted(cands.items(), key=lambda kv: -kv[1]))
            limiter += " (%s)" % (busy_note if ta_busy is not None else
                                  "vector issue = SQ_INSTS_VALU x 4 cycles at 2.4 GHz, %s; TA busy unavailable: %s" % (
                                      os.path.basename(PMC_PROFILE), busy_note))
            if dom == "describe":
                limiter += "; 4.25 LDS-DMA window fetches per keypoint through the texture addresser (DESIGN.md §4.6)"
        # per-stage table: every stage alone on the GPU against both roofs, and its counter traffic over its algorithmic bytes
        stage_table = {}
        for k, ms_alone in ser.items():
            if k not in sb or not ms_alone:
                continue
            row = {"ms_alone": round(ms_alone, 4), "hbm_frac_alone": round(sb[k] / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            vk = valu_issue_ms(k, B)
            row["valu_issue_frac_alone"] = round(vk / ms_alone, 3) if vk is not None else None
            tk, _ = pmc_traffic(k, B)
            row["traffic_over_algorithmic"] = round(tk / sb[k], 2) if tk is not None and sb[k] else None
            ta_k, va_k, _ = busy_fracs(k)
            row["ta_busy_frac_alone"] = round(ta_k, 3) if ta_k is not None else None
            row["valu_issue_frac_alone_counter_cycles"] = round(va_k, 3) if va_k is not None else None
            stage_table[k] = row
        roofline = {"kernel": kern, "stage": dom, "bound": bound, "binding_limit": binding, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "frac_alone": round(frac_alone, 4) if frac_alone is not None else None,
                    "valu_issue_frac_alone": round(valu_frac_alone, 3) if valu_frac_alone is not None else None,
                    # from the memory-path passes, over the kernel's own active cycles (GRBM_GUI_ACTIVE): TA busy and vector issue
                    "ta_busy_frac_alone": round(ta_busy, 3) if ta_busy is not None else None,
                    "valu_issue_frac_alone_counter_cycles": round(valu_busy, 3) if valu_busy is not None else None,
                    "busy_fracs_source": busy_note,
                    # the roofline that binds the STEP: vector-ALU issue time of all its kernels / the step time
                    "step_valu_issue": round(step_valu / step_ms, 3) if step_valu is not None else None,
                    "step_valu_issue_ms": round(step_valu, 3) if step_valu is not None else None,
                    "traffic": traffic, "traffic_source": traffic_note,
                    "limiter": limiter,
                    "stage_chosen_by": "largest duration alone on the GPU (stages_ms_serialized)" if ser_ok else
                                       "largest in-place time per step (no serialized pass in this run)",
                    "stages": stage_table,
                    "launches_per_step": per_step[dom], "frames_per_launch": fpl, "avg_ms": round(avg[dom], 4),
                    # a "launch" of the resize stage is one kernel per level > 0: rocprofv3's per-kernel average is avg_ms / this
                    "kernels_per_launch": (a.levels - 1) if dom == "resize" else 1,
                    "avg_ms_per_kernel": round(avg[dom] / ((a.levels - 1) if dom == "resize" else 1), 4),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "timing": "HIP events on the launching stream around every stage launch of %d of the %d timed steps "
                              "(%d launches of the dominant stage); chunks of a step run concurrently on two streams, so "
                              "an in-place duration includes the sibling chunk's share of the GPU (`frac` is in place, "
                              "`frac_alone` from the serialized pass); the resize entry spans its per-level kernels" % (
                                  steps_cov, a.steps, cnt[dom]),
                    "stages_ms_per_launch": {k: round(x, 4) for k, x in avg.items()},
                    "stages_ms_per_step": {k: round(x, 4) for k, x in acc.items()},
                    "stages_gbs": {k: round(sb[k] / (x * 1e-3) / 1e9, 1) for k, x in acc.items() if x > 0 and k in sb}}
        w, h, _ = ctx.level_geometry()
        P = sum(x * y for x, y in zip(w, h))
        extract_bytes = 3 * a.width * a.height + 2 * P + 48 * (kp_b / B)
        st_prof, fps_prof, _ = pmc_profile()
        if st_prof is not None:
            roofline["traffic_MB_per_frame"] = round(sum(
                (2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024 for d in st_prof.values()) / fps_prof / 1e6, 2)
            roofline["stages_valu_issue_ms"] = {k: round(valu_issue_ms(k, B), 3) for k in acc if k in st_prof}
        # the whole step against the fused-ideal figure: how much of the memory-side traffic is re-reads / re-writes of planes
        # between the kernels of the chain (counter traffic from the committed PMC passes, when they match these sources)
        ws = {"algorithmic_MB_per_frame": round(extract_bytes / 1e6, 3),
              "counter_MB_per_frame": roofline.get("traffic_MB_per_frame"),
              "what": "SURVEY.md §8d fused-ideal bytes (3WH + 2P + 48K) vs (2 x FETCH_SIZE + WRITE_SIZE) of every kernel of the step"}
        ws["ratio"] = round(ws["counter_MB_per_frame"] / ws["algorithmic_MB_per_frame"], 2) if ws["counter_MB_per_frame"] else None
        if ws["counter_MB_per_frame"] is None:
            ws["counter_note"] = traffic_note
        roofline["whole_step"] = ws
        if "stages_ms_serialized" in extras:
            roofline["stages_ms_serialized"] = extras.pop("stages_ms_serialized")
        if "match_knn2" in acc:
            # the matcher is not HBM-bound
            pairs = B * (kp_b / B) ** 2
            t_s = avg["match_knn2"] * 1e-3
            roofline["match_kernel"] = STAGE_KERNEL["match_knn2"]
            # distances on the matrix cores as FP4 +-1 dot products (2*256 flop per pair, dense FP4 peak ~10
            # PFLOP/s); the top-2 selection is 5 vector instructions per 4 pairs beside them (since round 6 in the MFMAs' shadow:
            # the matrix cores bound the kernel)
            roofline["match_mfma"] = {"bound": "mfma", "dtype": "fp4 (+-1, exact)", "pairs_per_launch": int(pairs),
                                      "achieved": round(pairs * 512 / t_s / 1e12, 1), "peak": MFMA_FP4_PEAK_TFLOPS,
                                      "unit": "TFLOP/s",
                                      "frac": round(pairs * 512 / t_s / 1e12 / MFMA_FP4_PEAK_TFLOPS, 3)}
            m_alone = ser.get("match_knn2") or (roofline.get("stages_ms_serialized") or {}).get("match_knn2")
            if m_alone:
                # alone on the GPU (in place the matcher runs on its own stream beside the next batch's detector kernels, which
                # stretches its launch): the figure to hold against the matrix cores' measured ceiling (6.15 PFLOP/s)
                roofline["match_mfma"]["achieved_alone"] = round(pairs * 512 / (m_alone * 1e-3) / 1e12, 1)
                roofline["match_mfma"]["frac_alone"] = round(pairs * 512 / (m_alone * 1e-3) / 1e12 / MFMA_FP4_PEAK_TFLOPS, 3)
            tops = pairs * (34 / 16) / t_s / 1e12
            roofline["match_valu"] = {"bound": "valu-issue", "ops_per_pair": 34 / 16, "achieved": round(tops, 2),
                                      "peak": round(VALU_PEAK_TOPS, 2), "unit": "Tlane-op/s",
                                      "frac": round(tops / VALU_PEAK_TOPS, 3)}
        extract_gbs = extract_bytes * a.steps * B * world / dt_max / 1e9
        roofline["extract_fused_ideal"] = {
            "bytes_per_frame": int(extract_bytes), "achieved_GBps_per_gpu": round(extract_gbs / world, 1),
            "frac": round(extract_gbs / world / HBM_PEAK_GBS, 4),
            "what": "SURVEY.md §8d whole-extract figure 3WH + 2P + 48K per frame x frames/s of the whole step"}
        cfg = "cfg2" if (a.width, a.height, a.levels, cv) == (640, 480, 8, False) else "%dx%d, %d levels%s" % (
            a.width, a.height, a.levels, ", cv::ORB detector mode (n_features %d)" % a.n_features if cv else "")
        out = {
            "metric": baseline_metric(), "value": kp_total / dt_max,
            "unit": "keypoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: synthetic %dx%d RGB-D stream, device-resident frames (RGB + depth in HBM when the timed "
                                   "region starts; the PCIe-inclusive rate is `value_pcie_inclusive`), %d-level ORB (reference "
                                   "defaults 1.2/20/7, min-area %d), "
                                   "extract + BF-Hamming knn-2 match (ratio 0.7) vs previous frame + depth back-projection%s%s" % (
                           cfg, a.width, a.height, a.levels, a.min_area,
                           " + RANSAC PnP (100 hypotheses, 5 px) of every frame against its predecessor" if a.pnp else "",
                           (" + DBoW3 k=10 L=%d loop scoring vs last 64 frames%s" % (
                               a.voc_levels, " + all-gather of BoW vectors, cross-stream scores" if world > 1 else ""))
                           if a.bow else ""),
                       "warmup_steps_run": w_eff,  # at least one per distinct batch: the warm-up also counts the keypoints
                       "frames_per_step": B, "frames_per_gpu": a.steps * B, "distinct_frames": n_unique,
                       "keypoints_per_frame": round(kp_b / B, 1), "fast_candidates_per_frame": round(cand_b / B, 1),
                       "frames_per_s": a.steps * B * world / dt_max,
                       "extract_algorithmic_GBps_whole_job": extract_gbs},
            "match_kernel": STAGE_KERNEL["match_knn2"],
            "dist": {"backend": dist.get_backend() if world > 1 else None,
                     "world_size": dist.get_world_size() if world > 1 else 1,
                     "launched_by": os.environ.get("MSLAM_BENCH_LAUNCHER", "external")},
            "roofline": roofline,
        }
        if "value_popcount_matcher" in extras:
            extras["popcount_match_kernel"] = POPCOUNT_KERNEL
        out.update(extras)
        if "pcie_inclusive" in out:
            # the same step fed from page-locked host memory and with its results copied back: never `value`
            out["value_pcie_inclusive"] = out["pcie_inclusive"]["keypoints_per_s"]
        if with_cpu and not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, dict(n_levels=a.levels, min_size=a.min_area), a.cpu_seconds, cv,
                                               a.n_features)
        return out

    # ---- after the timed region ---------------------------------------------------------------------------
    extras = {}
    # Nothing after the timed region may cost the headline line: a failure in an extra leg is recorded in the line, and
    # with several ranks a watchdog prints the line without the extras if a collective of the extras never completes
    # (rank 0) / ends the rank quietly (the others).
    watchdog = None
    line_lock = threading.Lock()
    printed = [False]

    def emit(line):
        """exactly one JSON line leaves the process, whoever gets here first (main thread or watchdog)"""
        with line_lock:
            if printed[0]:
                return False
            printed[0] = True
            print(json.dumps(line), flush=True)
            return True

    # the fallback line is a snapshot of the headline taken NOW, on the main thread, before any extra leg runs: the
    # watchdog thread only copies it, it never calls into the context or reads `extras` while the main thread works on them
    # (also armed for --rccl-world1: its collective is a real RCCL call, and a hung one must end the run non-zero)
    armed = world > 1 or a.rccl_world1
    headline = make_line({}, with_cpu=False) if (armed and rank == 0) else None
    if armed:
        def give_up():
            # a collective / GPU step after the timed region never completed: that is a hang, not a pass.  Rank 0 still
            # prints the headline (the timed region was complete) with the reason, and EVERY rank exits non-zero.
            if rank == 0 and not printed[0]:
                line = dict(headline)
                line["extras_error"] = ("the legs after the timed region did not finish within %d s (hung collective or GPU "
                                        "step): exit code 3" % a.extras_timeout)
                emit(line)
            os._exit(3)
        watchdog = threading.Timer(a.extras_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
    try:
        if not a.no_extras:
            # the same run with the xor/popcount matcher (the form north_star names): both values are always quoted
            ctx.set_matcher(pkg.MATCHER_POPCOUNT)
            for i in range(2):
                step(i)
            settle()
            dt_pop, _ = timed_run(w_eff, False)
            ctx.set_matcher(pkg.MATCHER_AUTO)
            extras["value_popcount_matcher"] = kp_total / dt_pop
            extras["ms_per_step_popcount_matcher"] = dt_pop / a.steps * 1e3

            # every stage alone on the GPU (everything serialised on one stream)
            ctx.set_profiling(1)
            ser, reps = {}, 5
            for i in range(reps):
                step(i)
                for name, ms in ctx.stage_times():
                    ser[name] = ser.get(name, 0.0) + ms / reps
            ctx.set_profiling(0)
            settle()
            extras["stages_ms_serialized"] = {k: round(x, 4) for k, x in ser.items()}

            if world > 1:
                # the loop-candidate exchange: BoW vectors of the batch -> ONE all-gather -> cross-stream scores
                n_ex = max(3, min(10, a.steps))
                for i in range(2):
                    step(i, bow=False)
                    ctx.bow_batch_dev(True)
                    cross.step_gpu(ctx, ts, B)
                settle()
                c0 = cross.collectives
                t0 = time.perf_counter()
                for i in range(n_ex):
                    ctx.bow_batch_dev(True)
                    scores = cross.step_gpu(ctx, ts, B)
                settle()
                dt_ex = time.perf_counter() - t0
                t_ex = torch.tensor([dt_ex], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t_ex, op=dist.ReduceOp.MAX)
                s_host = scores.cpu().numpy()
                extras["exchange"] = {
                    "what": "per batch: DBoW3 vectors of %d frames (k=10 L=%d vocabulary) -> pack -> %s -> L1 scores of own "
                            "frame t vs frame t of every stream" % (B, a.voc_levels, "ONE all_gather_into_tensor" if
                                                                      cross.granularity == "batch" else
                                                                      "one all_gather_into_tensor per frame"),
                    "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                    "granularity": cross.granularity,
                    "collectives_per_batch": (cross.collectives - c0) / float(n_ex),
                    "bytes_per_rank_per_batch": 4 * set_dwords(B, cross.k_max),
                    "bytes_per_rank_per_collective": cross.bytes_per_collective,
                    "ms_per_batch_incl_bow": float(t_ex.item()) / n_ex * 1e3,
                    "self_score_min": float(s_host[:, rank].min()),
                    "cross_score_max": float(np.delete(s_host, rank, 1).max())}

            if world > 1 and not a.no_legs and not cv:
                # ---- cfg4 as BASELINE.json states it: every rank its own 1280x720 stream, DBoW3 scoring and the all-gather of
                # the BoW sets + cross-stream scores INSIDE the step (feed point: rgbd_feature_frontend.cpp:176)
                try:
                    extras["cfg4"] = cfg4_multi_rank_leg(a, pkg, synth, torch, dist, rank, world, dev, red_dev)
                except Exception as e:  # noqa: BLE001 - recorded; a rank that fails here leaves its peers to the watchdog
                    extras["cfg4_error"] = "%s: %s" % (type(e).__name__, e)
                    torch.cuda.synchronize()

            if world == 1 and rank == 0:
                # PCIe-inclusive: frames + depth start in pinned host memory, double-buffered H2D on a copy stream while
                # the previous batch is processed; keypoints, descriptors, matches and 3-D points are copied back (D2H)
                PB = min(250, B)
                n_pb = 12
                h_rgb = torch.from_numpy(frames[:PB]).pin_memory()
                h_dep = torch.from_numpy(np.ascontiguousarray(d_depth[:PB].cpu().numpy())).pin_memory()
                dbuf = [(torch.empty_like(h_rgb, device="cuda"), torch.empty_like(h_dep, device="cuda")) for _ in range(2)]
                copy_stream = torch.cuda.Stream()
                ev_in = [torch.cuda.Event() for _ in range(2)]
                ev_free = [torch.cuda.Event() for _ in range(2)]
                # results: ONE packed block per batch (mslam_hip_pack_batch_dev: exactly count[t] keypoint / match_count[t] match
                # records per frame), written by the pack kernel straight into page-locked host memory — nothing but the
                # records that exist crosses PCIe; two blocks alternate
                pcap = ctx.packed_capacity(PB, True)
                host_pk = [torch.empty(pcap, dtype=torch.uint8).pin_memory() for _ in range(2)]
                pk_turn = [0]

                def results_to_host():
                    ctx.pack_batch_dev(host_pk[pk_turn[0] & 1].data_ptr(), pcap, True)
                    pk_turn[0] += 1

                def h2d(i):
                    with torch.cuda.stream(copy_stream):
                        if i >= 2:
                            copy_stream.wait_event(ev_free[i % 2])  # the batch that used this buffer is done with it
                        dbuf[i % 2][0].copy_(h_rgb, non_blocking=True)
                        dbuf[i % 2][1].copy_(h_dep, non_blocking=True)
                        ev_in[i % 2].record(copy_stream)

                def run_pcie(n):
                    h2d(0)
                    for i in range(n):
                        if i + 1 < n:
                            h2d(i + 1)
                        ts.wait_event(ev_in[i % 2])
                        ctx.detect_batch_dev(dbuf[i % 2][0].data_ptr(), PB)
                        ctx.match_batch_dev(0.7, True)
                        ctx.backproject_batch_dev(dbuf[i % 2][1].view(torch.int16).data_ptr())
                        ev_free[i % 2].record(ts)
                        results_to_host()
                    settle()

                run_pcie(2)
                t0 = time.perf_counter()
                run_pcie(n_pb)
                dt_p = time.perf_counter() - t0
                pk = pkg.unpack_batch(host_pk[(pk_turn[0] - 1) & 1].numpy())
                kp_p = int(pk["kp_offset"][-1]) * n_pb
                d2h = pk["bytes"]
                extras["pcie_inclusive"] = {
                    "frames_per_s": n_pb * PB / dt_p, "keypoints_per_s": kp_p / dt_p,
                    "h2d_GBps": n_pb * (h_rgb.numel() + 2 * h_dep.numel()) / dt_p / 1e9, "d2h_GBps": n_pb * d2h / dt_p / 1e9,
                    "how": "%d batches of %d frames: RGB + depth from pinned host memory by double-buffered async H2D on a "
                           "copy stream, overlapped with extract + match + back-projection of the previous batch; keypoints, "
                           "descriptors, matches and 3-D points packed on the device (exactly `count` records per frame, "
                           "mslam_hip_pack_batch_dev) and written by the pack kernel into page-locked host memory; "
                           "the headline `value` is the HBM-resident rate" % (n_pb, PB)}

            if world == 1 and rank == 0 and not a.no_legs and not cv and not a.bow and not a.pnp:
                # ---- driver-visible legs of the other single-GPU configurations of BASELINE.json (bounded: ~10 steps each)
                def leg(tag, lctx, lstep, LB, n_steps, kp_per_step, cand_per_step, voc_L, what):
                    """time n_steps steps (both matchers), stage events in place, and every stage alone"""
                    def run(n, profile):
                        if profile:
                            lctx.set_profiling(2)
                        t0 = time.perf_counter()
                        for i in range(n):
                            lstep(i)
                        lctx.sync()
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        tm = lctx.stage_times(cap=8192) if profile else None
                        if profile:
                            lctx.set_profiling(0)
                        return dt, tm
                    for i in range(2):
                        lstep(i)
                    lctx.sync()
                    dt, tm = run(n_steps, True)
                    tot = {}
                    for name, ms in tm:
                        tot[name] = tot.get(name, 0.0) + ms / n_steps
                    lctx.set_matcher(pkg.MATCHER_POPCOUNT)
                    lstep(0)
                    lctx.sync()
                    dt_pop, _ = run(n_steps, False)
                    lctx.set_matcher(pkg.MATCHER_AUTO)
                    lctx.set_profiling(1)
                    alone = {}
                    for i in range(3):
                        lstep(i)
                        for name, ms in lctx.stage_times():
                            alone[name] = alone.get(name, 0.0) + ms / 3
                    lctx.set_profiling(0)
                    lctx.sync()
                    lsb = stage_bytes(lctx, LB, kp_per_step, cand_per_step, 10, voc_L)
                    # (the dominant stage by its duration ALONE: an in-place interval on one of the two chunk streams also
                    # contains the time its launch waited for the other stream's kernels)
                    ldom = max((k for k in alone if k in lsb), key=alone.get)
                    pairs = LB * (kp_per_step / LB) ** 2
                    out = {"workload": what, "value": kp_per_step * n_steps / dt, "unit": "keypoints/s",
                           "ms_per_step": dt / n_steps * 1e3, "frames_per_step": LB, "steps": n_steps,
                           "keypoints_per_frame": round(kp_per_step / LB, 1),
                           "value_popcount_matcher": kp_per_step * n_steps / dt_pop,
                           "ms_per_step_popcount_matcher": dt_pop / n_steps * 1e3,
                           "dominant_stage": ldom, "dominant_kernel": STAGE_KERNEL.get(ldom, ldom),
                           "dominant_ms_per_step_in_place": round(tot.get(ldom, 0.0), 4),
                           "dominant_ms_per_step_alone": round(alone.get(ldom, 0.0), 4),
                           # algorithmic bytes of the dominant stage over its duration alone on the GPU, against the HBM peak
                           "dominant_hbm_frac_alone": round(lsb[ldom] / (alone[ldom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                           if alone.get(ldom) else None,
                           "stages_ms_per_step_alone": {k: round(x, 4) for k, x in alone.items()}}
                    if alone.get("match_knn2"):
                        out["match_pairs_per_step"] = int(pairs)
                        out["match_mfma_frac_alone"] = round(pairs * 512 / (alone["match_knn2"] * 1e-3) / 1e12 / MFMA_FP4_PEAK_TFLOPS, 3)
                    return out

                def guard(name, fn):
                    """a leg that fails is recorded under <name>_error and the remaining legs still run"""
                    if name in os.environ.get("MSLAM_BENCH_SKIP_LEGS", "").split(","):
                        return
                    try:
                        fn()
                    except Exception as e:  # noqa: BLE001
                        extras[name + "_error"] = "%s: %s" % (type(e).__name__, e)
                        torch.cuda.synchronize()

                voc = synth.make_vocabulary(10, a.voc_levels, seed=77)  # cfg3 and cfg4

                def _leg_cfg3():
                    # cfg3: the same stream + DBoW3 loop scoring against a 1e6-word vocabulary every frame (k = 10, L = voc_levels)
                    ctx.bow_load(voc)
                    ctx.bow_db_reserve(32 * B)  # the leg adds 2 + 10 + 1 + 10 + 3 = 26 batches of entries: no storage doubling inside its timed steps
                    extras["cfg3"] = leg("cfg3", ctx, lambda i: step(i, bow=True), B, 10, counts_per_batch[0], cand_per_batch[0],
                                         a.voc_levels,
                                         "cfg3: the cfg2 step + DBoW3 transform (k=10, L=%d: %d words), tf-idf + L1 vectors, scores vs the "
                                         "last 64 frames, inverted-file adds" % (a.voc_levels, 10 ** a.voc_levels))
                guard('cfg3', _leg_cfg3)

                def _leg_cfg5():
                    # cfg5: 1920x1080, 3 levels, ~10 k keypoints per frame, k = 2 matcher with ratio test (64 M+ distances per frame)
                    W5, H5, B5 = 1920, 1080, 64
                    f5 = synth.make_stream(B5, W5, H5, seed=4321)
                    d5 = torch.from_numpy(f5).cuda()
                    dd5 = torch.from_numpy(np.ascontiguousarray(np.stack([synth.make_depth(1, W5, H5, seed=4321)[0]] * B5)).view(np.int16)).cuda()
                    area5 = -(-W5 * H5 // (640 * 480))
                    ks5 = area5 * max(1, 1000 // 370)
                    ts5 = torch.cuda.Stream()
                    ctx5 = pkg.Context(width=W5, height=H5, max_batch=B5, n_levels=3, min_node_area=370,
                                       max_keypoints=min(32736, 4096 * ks5), max_candidates=16384 * area5, device=dev,
                                       stream=ts5.cuda_stream)

                    def step5(i):
                        ctx5.detect_batch_dev(d5.data_ptr(), B5)
                        ctx5.match_batch_dev(0.7, True)
                        ctx5.backproject_batch_dev(dd5.data_ptr())
                    step5(0)
                    ctx5.sync()
                    kp5 = int(pkg.read_device(ctx5, ctx5.batch_view().count, (B5,), np.int32).sum())
                    cand5 = int(ctx5.debug_counts(pkg.DBG_CANDIDATES, B5).sum())
                    extras["cfg5"] = leg("cfg5", ctx5, step5, B5, 10, kp5, cand5, a.voc_levels,
                                         "cfg5: synthetic 1920x1080 RGB-D stream (%d distinct frames), 3-level pyramid, min-area 370, "
                                         "extract + knn-2 ratio-test matcher vs previous frame + back-projection" % B5)
                    ctx5.close()
                    del d5, dd5, f5

                guard('cfg5', _leg_cfg5)

                def _leg_cfg2_k2000():
                    # cfg2 at BASELINE.json's nominal K: the quadtree's stop area tuned so that a frame keeps 2000 +- 2 % keypoints
                    # (SURVEY.md §8d "min-area tuned so K ~ 2000"; the headline keeps the reference's default 1000 -> ~1880)
                    ts2 = torch.cuda.Stream()
                    ctx2 = pkg.Context(width=a.width, height=a.height, max_batch=B, n_levels=a.levels, min_node_area=K2000_MIN_AREA,
                                       max_keypoints=4096 * k_scale, max_candidates=16384 * area, device=dev, stream=ts2.cuda_stream)

                    def step2(i):
                        off2 = (i % n_batches) * B
                        ctx2.detect_batch_dev(d_frames.data_ptr() + off2 * frame_bytes, B)
                        ctx2.match_batch_dev(0.7, True)
                        ctx2.backproject_batch_dev(d_depth.data_ptr() + off2 * depth_bytes)
                    step2(0)
                    ctx2.sync()
                    kp2 = int(pkg.read_device(ctx2, ctx2.batch_view().count, (B,), np.int32).sum())
                    cand2 = int(ctx2.debug_counts(pkg.DBG_CANDIDATES, B).sum())
                    extras["cfg2_k2000"] = leg("cfg2_k2000", ctx2, step2, B, 10, kp2, cand2, a.voc_levels,
                                               "cfg2 with the quadtree stop area tuned to BASELINE.json's nominal 2000 keypoints per "
                                               "frame (min-area %d instead of the reference default 1000), otherwise the headline step" %
                                               K2000_MIN_AREA)
                    ctx2.close()

                guard('cfg2_k2000', _leg_cfg2_k2000)

                def _leg_cfg4_one_rank():
                    # cfg4, ONE rank of it: a 1280x720 stream, 8 levels, ~2000 keypoints per frame, the cfg3 BoW step and the device
                    # side of the cross-stream exchange (pack -> the world-1 "all-gather" = a device copy -> cross scores)
                    W4, H4, B4 = 1280, 720, 250
                    f4 = synth.make_stream(B4, W4, H4, seed=1234)
                    d4 = torch.from_numpy(f4).cuda()
                    dd4 = torch.from_numpy(np.ascontiguousarray(np.stack([synth.make_depth(1, W4, H4, seed=1234)[0]] * B4)).view(np.int16)).cuda()
                    area4 = -(-W4 * H4 // (640 * 480))
                    ts4 = torch.cuda.Stream()
                    ctx4 = pkg.Context(width=W4, height=H4, max_batch=B4, n_levels=8, min_node_area=CFG4_MIN_AREA,
                                       max_keypoints=min(8192, 4096 * area4),  # (the BoW scoring kernel keeps a frame's vector in LDS: <= 8192 keypoints)
                                       max_candidates=16384 * area4, device=dev, stream=ts4.cuda_stream)
                    ctx4.bow_load(voc)
                    ctx4.bow_db_reserve(32 * B4)
                    if a.rccl_world1 and not dist.is_initialized():
                        s1 = socket.socket()
                        s1.bind(("127.0.0.1", 0))
                        port1 = s1.getsockname()[1]
                        s1.close()
                        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port1, rank=0, world_size=1,
                                                device_id=torch.device("cuda", dev))
                    cross4 = CrossStreamLoopCandidates(k_max=2048, always_collective=a.rccl_world1)

                    def step4(i):
                        ctx4.detect_batch_dev(d4.data_ptr(), B4)
                        ctx4.match_batch_dev(0.7, True)
                        ctx4.backproject_batch_dev(dd4.data_ptr())
                        ctx4.bow_batch_dev(True)
                        cross4.step_gpu(ctx4, ts4, B4)
                    step4(0)
                    cross4.finish(ts4)
                    ctx4.sync()
                    kp4 = int(pkg.read_device(ctx4, ctx4.batch_view().count, (B4,), np.int32).sum())
                    cand4 = int(ctx4.debug_counts(pkg.DBG_CANDIDATES, B4).sum())
                    c4 = cross4.collectives
                    l4 = leg("cfg4_one_rank", ctx4, step4, B4, 10, kp4, cand4, a.voc_levels,
                             "cfg4, one rank: synthetic 1280x720 RGB-D stream (%d distinct frames), 8 levels, min-area %d, extract + match + "
                             "back-projection + DBoW3 (k=10, L=%d) vectors / scores / inverted-file adds + exchange (pack, world-1 gather, "
                             "cross-stream scores)" % (B4, CFG4_MIN_AREA, a.voc_levels))
                    cross4.finish(ts4)
                    n_coll = cross4.collectives - c4
                    # the exchange alone: pack + gather + cross score of one batch, on the streams the step uses
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(10):
                        cross4.step_gpu(ctx4, ts4, B4)
                    cross4.finish(ts4)
                    torch.cuda.synchronize()
                    l4["exchange"] = {"bytes_per_collective": 4 * set_dwords(B4, cross4.k_max), "k_max": cross4.k_max,
                                      "collectives_per_step": 1.0 if n_coll else 0.0,
                                      "ms_per_batch_alone": (time.perf_counter() - t0) / 10 * 1e3,
                                      "collective": ("all_gather_into_tensor on the %s backend, world size 1" % dist.get_backend())
                                      if (a.rccl_world1 and dist.is_initialized()) else "world 1: a device copy of the set",
                                      "what": "mslam_hip_bow_pack_dev -> all_gather_into_tensor -> "
                                              "mslam_hip_bow_cross_score_packed_dev on the communication stream"}
                    # the same exchange at frame granularity (one collective of 2 k_max + 1 dwords per frame): same wire format,
                    # same scores; what a live rig whose batch is one frame pays per frame
                    n_fr = B4  # (the pack kernel writes the whole last batch)
                    cross4f = CrossStreamLoopCandidates(k_max=2048, always_collective=a.rccl_world1, granularity="frame")
                    cross4f.step_gpu(ctx4, ts4, n_fr)
                    cross4f.finish(ts4)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(4):
                        cross4f.step_gpu(ctx4, ts4, n_fr)
                    cross4f.finish(ts4)
                    torch.cuda.synchronize()
                    l4["exchange_per_frame"] = {"bytes_per_collective": cross4f.bytes_per_collective, "frames": n_fr,
                                                "collectives_per_frame": (cross4f.collectives / 5.0) / n_fr,
                                                "us_per_frame_alone": (time.perf_counter() - t0) / 4 / n_fr * 1e6}
                    extras["cfg4_one_rank"] = l4
                    ctx4.close()
                    del d4, dd4, f4
                guard('cfg4_one_rank', _leg_cfg4_one_rank)


                def _leg_latency_us():
                    # ---- what the reference's caller sees: ONE frame per call through the synchronous C-ABI entry points
                    # (rgbd_feature_frontend.cpp:187 detect, :237 match): host frame in, host arrays out
                    import ctypes as C
                    c1 = pkg.Context(width=a.width, height=a.height, max_batch=1, n_levels=a.levels, min_node_area=a.min_area, device=dev)
                    L = c1.L
                    K1 = c1.params.max_keypoints
                    xy = np.empty((K1, 2), np.float32)
                    de = np.empty((2, K1, 32), np.uint8)
                    oc = np.empty(K1, np.int32)
                    an = np.empty(K1, np.float32)
                    rs = np.empty(K1, np.float32)
                    fi = np.empty(K1, np.int32)
                    ti = np.empty(K1, np.int32)
                    n = C.c_int(0)
                    m = C.c_int(0)
                    pp = lambda x: x.ctypes.data_as(C.c_void_p)
                    t_det, t_mat, counts = [], [], [0, 0]
                    for i in range(220):
                        fr = frames[i % 16]
                        t0 = time.perf_counter()
                        rc = L.mslam_hip_detect(c1._h, pp(fr), a.width, a.height, K1, pp(xy), pp(de[i & 1]), pp(oc), pp(an), pp(rs), C.byref(n))
                        t1 = time.perf_counter()
                        counts[i & 1] = n.value
                        if rc == 0 and i > 0:
                            rc = L.mslam_hip_match(c1._h, pp(de[i & 1]), counts[i & 1], pp(de[(i & 1) ^ 1]), counts[(i & 1) ^ 1],
                                                   C.c_double(0.7), pp(fi), pp(ti), C.byref(m))
                        t2 = time.perf_counter()
                        if rc != 0:
                            raise RuntimeError("single-frame call failed: %d" % rc)
                        if i >= 20:
                            t_det.append(t1 - t0)
                            t_mat.append(t2 - t1)
                    c1.close()
                    extras["latency_us"] = {
                        "detect": round(float(np.median(t_det)) * 1e6, 1), "match": round(float(np.median(t_mat)) * 1e6, 1),
                        "detect_p95": round(float(np.percentile(t_det, 95)) * 1e6, 1),
                        "match_p95": round(float(np.percentile(t_mat, 95)) * 1e6, 1),
                        "what": "median over 200 synchronous single-frame calls through the C ABI (mslam_hip_detect / mslam_hip_match: "
                                "host frame in, host keypoints / descriptors / matches out, %d keypoints per frame), what "
                                "IFeatureDetector::detect / IFeatureMatcher::match of the plugin cost per call" % counts[0]}
                guard('latency_us', _leg_latency_us)

    except Exception as e:  # noqa: BLE001 - any failure of an extra leg must not cost the headline
        extras["extras_error"] = "%s: %s" % (type(e).__name__, e)

    # the line leaves the process BEFORE the context is closed and before the closing barrier: a peer that hung or died in
    # its extras can no longer cost it.  The watchdog stays armed over close / barrier / destroy (a rank stuck there exits 3).
    if rank == 0:
        emit(make_line(extras))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif a.rccl_world1 and dist.is_initialized():
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
