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
