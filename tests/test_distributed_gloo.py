"""world_size-2 gloo test of the multi-stream exchange (the only collective on the path): every rank packs its BoW
vectors into the exchange format (k_max x {u32 word, f32 value} + count per frame), ONE all_gather_into_tensor per
batch moves them, and every rank scores its frame t against the other stream's frame t from the gathered buffer.
The same class, format and collective run on the GPU box (step_gpu: HIP pack + HIP scorer on the gathered sets);
the scorer injected here is the oracle (test infrastructure)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import __graft_entry__ as graft
    import synth
    orc = graft.load_oracle()
    pkg = graft.load_package()
    from modular_slam_amd.multi_stream import CrossStreamLoopCandidates
    V = orc.Vocabulary(synth.make_vocabulary(10, 3))
    B, K = 2, 2048
    # both streams look at the same scene (seed 1234); stream 1 starts two frames later
    frames = synth.make_stream(B, 320, 240, seed=1234, t0=2 * rank)
    words = torch.zeros((B, K), dtype=torch.int32)
    values = torch.zeros((B, K), dtype=torch.float64)
    counts = torch.zeros(B, dtype=torch.int32)
    for t in range(B):
        w, v = V.bow_vector(orc.detect(frames[t], orc.params(n_levels=4))["desc"])
        words[t, :len(w)] = torch.from_numpy(w.view(np.int32))
        values[t, :len(w)] = torch.from_numpy(v)
        counts[t] = len(w)
    x = CrossStreamLoopCandidates(k_max=K)
    assert (x.world, x.rank) == (world, rank)
    scores = x.step_with(words, values, counts, orc.bow_score_l1)
    assert x.collectives == 1                     # one fused collective per batch
    # the wire format round-trips: what rank r unpacks of its own set is its vectors with f32 values
    from modular_slam_amd.multi_stream import pack_vectors, unpack_set, set_dwords
    mine = pack_vectors(words, values, counts, K)
    assert mine.numel() == set_dwords(B, K) == B * (2 * K + 1)
    w, v, n = unpack_set(mine, B, K)
    for t in range(B):
        k = int(counts[t])
        assert n[t] == k and np.array_equal(w[t, :k], words[t, :k].numpy().view(np.uint32))
        assert np.array_equal(v[t, :k], values[t, :k].numpy().astype(np.float32).astype(np.float64))
        assert not w[t, k:].any() and not v[t, k:].any()
    q.put((rank, scores, [(int(counts[t]), words[t, :5].tolist()) for t in range(B)]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_and_cross_scores_world2():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, s, meta = q.get(timeout=120)
        res[r] = (s, meta)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s0, s1 = res[0][0], res[1][0]
    assert s0.shape == (2, 2) and s1.shape == (2, 2)
    # self scores are 1 (up to rounding), cross scores are symmetric between the two ranks
    # (values travel as f32: a self score is 1 up to the f32 rounding of the normalised weights)
    assert abs(s0[0, 0] - 1) < 1e-6 and abs(s0[1, 0] - 1) < 1e-6 and abs(s1[1, 1] - 1) < 1e-6
    assert np.array_equal(s0[:, 1], s1[:, 0])
    assert (s0[:, 1] > 0.05).all() and (s0[:, 1] < 1).all()        # overlapping views of one scene


def _vectors(rank, B, K, world):
    """deterministic per-rank BoW vectors with ragged word counts: rank 2's batch has no keypoints at all, every rank has
    one empty frame, rank 1 has a frame that fills k_max; neighbouring ranks share words, so cross scores are non-trivial"""
    rng = np.random.default_rng(100 + rank)
    words = torch.zeros((B, K), dtype=torch.int32)
    values = torch.zeros((B, K), dtype=torch.float64)
    counts = torch.zeros(B, dtype=torch.int32)
    for t in range(B):
        if rank == 2 or t == (rank % B):
            n = 0
        elif rank == 1 and t == 2:
            n = K
        else:
            n = min(K - 1, int(rng.integers(1, K // 2)) + 5 * rank + t)
        pool = np.arange(50 * rank, 50 * rank + 3 * K, dtype=np.uint32)     # overlaps the neighbours' pools
        w = np.sort(rng.choice(pool, size=n, replace=False)).astype(np.uint32)
        v = rng.random(n) + 0.01
        v = v / max(v.sum(), 1e-300)
        words[t, :n] = torch.from_numpy(w.view(np.int32))
        values[t, :n] = torch.from_numpy(v)
        counts[t] = n
    return words, values, counts


def _worker4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    orc = graft.load_oracle()
    graft.load_package()
    from modular_slam_amd.multi_stream import CrossStreamLoopCandidates, set_dwords
    B, K = 4, 64
    words, values, counts = _vectors(rank, B, K, world)
    out = {}
    for gran in ("batch", "frame"):
        x = CrossStreamLoopCandidates(k_max=K, granularity=gran)
        s = x.step_with(words, values, counts, orc.bow_score_l1)
        assert x.collectives == (1 if gran == "batch" else B), (gran, x.collectives)
        assert x.bytes_per_collective == 4 * set_dwords(B if gran == "batch" else 1, K)
        out[gran] = s
    # the raw gathered buffers of the two granularities are the same bytes
    from modular_slam_amd.multi_stream import pack_vectors
    mine = pack_vectors(words, values, counts, K)
    g_batch = CrossStreamLoopCandidates(k_max=K, granularity="batch").all_gather_sets(mine, n_frames=B)
    g_frame = CrossStreamLoopCandidates(k_max=K, granularity="frame").all_gather_sets(mine, n_frames=B)
    assert torch.equal(g_batch, g_frame) and torch.equal(g_batch[rank], mine)
    q.put((rank, out["batch"], out["frame"], counts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_world4_ragged_counts_empty_rank_and_per_frame_exchange():
    """world 4 over gloo: unequal per-frame word counts, one rank whose batch has zero keypoints, a frame that fills
    k_max; the per-frame exchange (one 2 k_max + 1 dword collective per frame) gives the same gathered bytes and the
    same scores as the per-batch collective"""
    world, port = 4, 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, sb, sf, cnt = q.get(timeout=180)
        res[r] = (sb, sf, cnt)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import __graft_entry__ as graft
    orc = graft.load_oracle()
    B, K = 4, 64
    vec = {r: _vectors(r, B, K, world) for r in range(world)}

    def sent(r, t):
        w, v, n = vec[r]
        k = int(n[t])
        return w[t, :k].numpy().view(np.uint32), v[t, :k].numpy().astype(np.float32).astype(np.float64)
    for r in range(world):
        sb, sf, cnt = res[r]
        assert np.array_equal(sb, sf)                       # granularity does not change a single score
        assert np.array_equal(cnt, vec[r][2].numpy())
        for t in range(B):
            for o in range(world):
                assert sb[t, o] == orc.bow_score_l1(*sent(r, t), *sent(o, t)), (r, t, o)
    assert not res[2][0].any() and not np.stack([res[r][0][:, 2] for r in range(world)]).any()   # the empty rank scores 0 everywhere
    assert res[0][0][2, 1] > 0 and res[1][0][0, 1] > 0.99   # neighbours share words; a non-empty frame scores ~1 against itself
