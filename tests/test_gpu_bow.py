"""GPU parity: DBoW3 word assignment, BoW vectors, L1 scores and the database, against the oracle."""
import os
import sys

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def descs(orc, synth_frames):
    return [orc.detect(f, orc.params())["desc"] for f in synth_frames[:4]]


@pytest.mark.parametrize("k,L,weighting", [(10, 3, 0), (7, 4, 1), (20, 2, 2), (3, 5, 3), (16, 2, 0)])
def test_words_and_vectors(pkg, orc, descs, k, L, weighting):
    blob = synth.make_vocabulary(k, L, seed=77 + k, weighting=weighting)
    V = orc.Vocabulary(blob)
    c = pkg.Context(width=640, height=480, max_keypoints=4096)
    c.bow_load(blob)
    info = c.bow_info()
    assert (info["k"], info["L"], info["n_nodes"], info["n_words"]) == (V.k, V.L, V.n_nodes, V.n_words)
    rng = np.random.default_rng(3)
    for d in (descs[0], rng.integers(0, 256, (1500, 32), dtype=np.uint8), descs[1][:1]):
        gw, gwt = c.bow_words(d)
        rw, rwt = V.words(d)
        assert np.array_equal(gw, rw) and np.array_equal(gwt, rwt)
        gv = c.bow_transform(d)
        rv = V.bow_vector(d)
        assert np.array_equal(gv[0], rv[0])
        assert np.array_equal(gv[1], rv[1])  # f64, bit-exact: same summation order
    c.close()


def test_scores_and_database(pkg, orc, descs):
    blob = synth.make_vocabulary(10, 3)
    V = orc.Vocabulary(blob)
    c = pkg.Context(width=640, height=480, max_keypoints=4096)
    c.bow_load(blob)
    vecs = [V.bow_vector(d) for d in descs]
    for i in range(4):
        for j in range(4):
            assert c.bow_score(*vecs[i], *vecs[j]) == orc.bow_score_l1(*vecs[i], *vecs[j])
    assert c.bow_score(vecs[0][0][:0], vecs[0][1][:0], *vecs[1]) == 0.0
    # database: add three frames, query with the fourth and with a copy of the second
    ids = [c.bow_db_add(d) for d in descs[:3]]
    assert ids == [0, 1, 2]
    for q in (descs[3], descs[1]):
        got_ids, got_sc = c.bow_db_query(q, 3)
        qv = V.bow_vector(q)
        exp = sorted(((orc.bow_score_l1(*qv, *vecs[i]), i) for i in range(3)), key=lambda t: (-t[0], t[1]))
        assert list(got_ids) == [i for _, i in exp]
        assert list(got_sc) == [s for s, _ in exp]
    c.bow_db_clear()
    assert len(c.bow_db_query(descs[0], 3)[0]) == 0
    c.close()


def test_no_vocabulary_is_loud(pkg):
    c = pkg.Context(width=640, height=480)
    with pytest.raises(pkg.MslamHipError) as e:
        c.bow_words(np.zeros((1, 32), np.uint8))
    assert e.value.code == pkg.E_NO_VOCABULARY
    with pytest.raises(pkg.MslamHipError) as e:
        c.bow_load(b"not a vocabulary")
    assert e.value.code == pkg.E_FORMAT
    c.close()


def test_bow_batch_device(pkg, orc, synth_frames):
    """detect batch -> BoW vectors -> score against the database (query-then-add per frame)."""
    import torch
    blob = synth.make_vocabulary(10, 3)
    V = orc.Vocabulary(blob)
    frames = synth_frames[:6]
    dev = torch.from_numpy(frames).cuda()
    K = 4096
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=K)
    c.bow_load(blob)
    vecs = [V.bow_vector(orc.detect(f, orc.params())["desc"]) for f in frames]
    for b in range(2):
        c.detect_batch_dev(dev[3 * b:].data_ptr(), 3)
        c.bow_batch_dev(True)
        c.sync()
        v = c.bow_view()
        n = pkg.read_device(c, v.n_words, (3,), np.int32)
        words = pkg.read_device(c, v.words, (3, K), np.uint32)
        vals = pkg.read_device(c, v.values, (3, K), np.float64)
        be = pkg.read_device(c, v.best_entry, (3,), np.int32)
        bs = pkg.read_device(c, v.best_score, (3,), np.float64)
        for i in range(3):
            t = 3 * b + i
            assert n[i] == len(vecs[t][0])
            assert np.array_equal(words[i, :n[i]], vecs[t][0]) and np.array_equal(vals[i, :n[i]], vecs[t][1])
            exp = sorted(((orc.bow_score_l1(*vecs[t], *vecs[e]), e) for e in range(t)), key=lambda x: (-x[0], x[1]))
            exp = [x for x in exp if x[0] > 0]
            if not exp:
                assert be[i] == -1
            else:
                assert be[i] == exp[0][1] and bs[i] == exp[0][0]
    c.close()


def test_rccl_world1_exchange(pkg, orc, synth_frames):
    """what ONE GPU can exercise of the RCCL path (the 8-GPU run is the driver's): a process group on the
    `nccl` backend (= RCCL on ROCm) with a single rank, the exchange step's all_gather_into_tensor issued as a real
    collective on the communication stream (always_collective), self-scores against the oracle — per batch and per frame.
    Runs in a child process with a time limit.  ONLY a torch build without the nccl backend skips; a timeout (the child is
    killed), a fault, an abort or any other non-zero exit FAILS with the child's stderr tail."""
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import os, sys, socket, numpy as np, torch, torch.distributed as dist
        if not dist.is_nccl_available():
            print("RCCL_ABSENT"); sys.exit(0)
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
        import __graft_entry__ as g, synth
        pkg = g.load_package(); orc = g.load_oracle()
        from modular_slam_amd.multi_stream import CrossStreamLoopCandidates
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        torch.cuda.set_device(0)
        print("RCCL_INIT_BEGIN", flush=True)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        print("RCCL_INIT_OK", flush=True)
        blob = synth.make_vocabulary(10, 3)
        V = orc.Vocabulary(blob)
        frames = synth.make_stream(3, 640, 480, seed=1234)
        ts = torch.cuda.Stream()
        c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=4096, stream=ts.cuda_stream)
        c.bow_load(blob)
        dev = torch.from_numpy(frames).cuda()
        for gran, per_batch in (("batch", 1), ("frame", 3)):
            x = CrossStreamLoopCandidates(k_max=2048, always_collective=True, granularity=gran)
            for rep in range(3):
                c.detect_batch_dev(dev.data_ptr(), 3)
                c.bow_batch_dev(False)
                sc = x.step_gpu(c, ts, 3)
            x.finish(ts); c.sync()
            sc = sc.cpu().numpy()
            for t in range(3):
                w, v = V.bow_vector(orc.detect(frames[t], orc.params())["desc"])
                v = v.astype(np.float32).astype(np.float64)
                assert sc[t, 0] == orc.bow_score_l1(w, v, w, v), (gran, t)
            assert x.collectives == 3 * per_batch and dist.get_backend() == "nccl", (gran, x.collectives)
            print("RCCL_GRANULARITY_OK", gran, x.bytes_per_collective, flush=True)
        dist.barrier(); dist.destroy_process_group(); c.close()
        print("RCCL_WORLD1_OK")
    """ % (ROOT, ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired as e:  # subprocess.run has killed the child
        def _txt(b):
            return (b.decode("utf-8", "replace") if isinstance(b, bytes) else (b or ""))[-1500:]
        pytest.fail("the RCCL world-1 exchange did not finish within 240 s (child killed): stdout %r stderr %r" % (
            _txt(e.stdout), _txt(e.stderr)))
    if "RCCL_ABSENT" in r.stdout:
        pytest.skip("this torch build has no nccl (RCCL) backend")
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, "exit code %s\nstdout: %s\nstderr: %s" % (
        r.returncode, r.stdout[-600:], r.stderr[-2000:])


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_cfg4_two_rank_exchange(backend):
    """cfg4 of BASELINE.json between TWO ranks (feed point rgbd_feature_frontend.cpp:176): every rank extracts its
    own 1280x720 stream (seed 1234 + 100 * rank), builds the DBoW3 vectors, packs them, ONE all_gather_into_tensor moves the sets
    and the HIP kernel scores own frame t against frame t of both streams; every rank compares ALL its cross scores with the
    oracle's L1 score on the oracle's own vectors of both streams (f32 values as transmitted).
    `nccl` (= RCCL over xGMI): needs two GPUs, one rank per GPU — skipped on a one-GPU box, this is the test the first multi-GPU
    box activates.  `gloo`: the rehearsal a one-GPU box can run — both ranks on GPU 0, the collective through host memory, the
    same pack / score kernels and the same checks.  The ranks are child processes started before this process touches the GPU
    API for them; a timeout kills them and FAILS."""
    import subprocess
    import textwrap
    import torch
    if backend == "nccl":
        import torch.distributed as dist
        if not dist.is_nccl_available():
            pytest.skip("this torch build has no nccl (RCCL) backend")
        if torch.cuda.device_count() < 2:
            pytest.skip("the RCCL two-rank cfg4 exchange needs >= 2 GPUs (this box has %d)" % torch.cuda.device_count())
    code = textwrap.dedent("""
        import os, sys, numpy as np, torch, torch.distributed as dist
        backend = %r
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dev = rank if backend == "nccl" else 0
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
        import __graft_entry__ as g, synth
        pkg = g.load_package(); orc = g.load_oracle()
        from modular_slam_amd.multi_stream import CrossStreamLoopCandidates
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
        print("INIT_OK", rank, flush=True)
        W, H, B = 1280, 720, 3
        blob = synth.make_vocabulary(10, 3)
        V = orc.Vocabulary(blob)
        P = orc.params(min_size=6340)
        streams = [synth.make_stream(B, W, H, seed=1234 + 100 * r) for r in range(world)]
        sent = []
        for r in range(world):
            row = []
            for t in range(B):
                w, v = V.bow_vector(orc.detect(streams[r][t], P)["desc"])
                row.append((w, v.astype(np.float32).astype(np.float64)))
            sent.append(row)
        ts = torch.cuda.Stream()
        c = pkg.Context(width=W, height=H, max_batch=B, max_keypoints=8192, max_candidates=16384 * 3, min_node_area=6340,
                        device=dev, stream=ts.cuda_stream)
        c.bow_load(blob)
        d = torch.from_numpy(streams[rank]).cuda()
        for gran, per_batch in (("batch", 1), ("frame", B)):
            x = CrossStreamLoopCandidates(k_max=2048, granularity=gran)
            assert x.world == world and x.rank == rank
            for rep in range(3):
                c.detect_batch_dev(d.data_ptr(), B)
                c.bow_batch_dev(False)
                sc = x.step_gpu(c, ts, B)
            x.finish(ts); c.sync()
            sc = sc.cpu().numpy()
            assert sc.shape == (B, world)
            for t in range(B):
                for r in range(world):
                    exp = orc.bow_score_l1(*sent[rank][t], *sent[r][t])
                    assert sc[t, r] == exp, (gran, rank, t, r, sc[t, r], exp)
            assert x.collectives == 3 * per_batch, (gran, x.collectives)
            assert np.delete(sc, rank, 1).max() < sc[:, rank].min()     # another stream never scores like the own frame
            print("GRAN_OK", rank, gran, x.bytes_per_collective, flush=True)
        dist.barrier(); dist.destroy_process_group(); c.close()
        print("CFG4_RANK_OK", rank, flush=True)
    """ % (backend, ROOT, ROOT))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=env))
    outs = []
    import time as _t
    deadline = _t.time() + 420
    for p in procs:
        try:
            o, e = p.communicate(timeout=max(1.0, deadline - _t.time()))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, e = p.communicate()
            pytest.fail("the two-rank cfg4 exchange (%s) did not finish within 420 s (ranks killed): stdout %r stderr %r" % (
                backend, (o or "")[-800:], (e or "")[-1500:]))
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and ("CFG4_RANK_OK %d" % r) in o, "rank %d exit code %s\nstdout: %s\nstderr: %s" % (r, rc, o[-600:], e[-2000:])


def test_cross_stream_scores(pkg, orc, synth_frames):
    """the multi-GPU exchange step on one rank: HIP pack -> all-gather (trivial at world 1) -> HIP scoring of the
    gathered sets on the communication stream, against the oracle on the vectors as transmitted (f32 values);
    then the scorer alone on three hand-built sets; and the f64 form on explicit foreign vectors"""
    import torch
    from modular_slam_amd.multi_stream import CrossStreamLoopCandidates, pack_vectors, set_dwords
    blob = synth.make_vocabulary(10, 3)
    V = orc.Vocabulary(blob)
    frames = synth_frames[:6]
    dev = torch.from_numpy(frames).cuda()
    B, K, k_max = 3, 4096, 2048
    ts = torch.cuda.Stream()
    c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=K, stream=ts.cuda_stream)
    c.bow_load(blob)
    vecs = [V.bow_vector(orc.detect(f, orc.params())["desc"]) for f in frames]
    as_sent = [(w, v.astype(np.float32).astype(np.float64)) for w, v in vecs]
    x = CrossStreamLoopCandidates(k_max=k_max)
    for rep in range(3):                                   # the two buffer sets alternate
        c.detect_batch_dev(dev[3 * (rep & 1):].data_ptr(), B)
        c.bow_batch_dev(False)
        s = x.step_gpu(c, ts, B)
        x.finish(ts)
        c.sync()
        s = s.cpu().numpy()
        assert s.shape == (B, 1)
        for t in range(B):
            v = as_sent[3 * (rep & 1) + t]
            assert s[t, 0] == orc.bow_score_l1(*v, *v)
    assert x.collectives == 3
    # three gathered sets built on the host in the wire format: own = frames 0..2, foreign = 3..5 and 1,2,0
    sets_idx = [[0, 1, 2], [3, 4, 5], [1, 2, 0]]
    packed = []
    for ids in sets_idx:
        W = torch.zeros((B, k_max), dtype=torch.int32)
        Vv = torch.zeros((B, k_max), dtype=torch.float64)
        N = torch.zeros(B, dtype=torch.int32)
        for t, i in enumerate(ids):
            n = len(vecs[i][0])
            W[t, :n] = torch.from_numpy(vecs[i][0].view(np.int32))
            Vv[t, :n] = torch.from_numpy(vecs[i][1])
            N[t] = n
        packed.append(pack_vectors(W, Vv, N, k_max))
    G = torch.stack(packed).cuda()
    assert G.shape == (3, set_dwords(B, k_max))
    for me in range(3):
        out = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        c.bow_cross_score_packed_dev(G.data_ptr(), 3, me, B, k_max, out.data_ptr())
        c.sync()
        out = out.cpu().numpy()
        for r, ids in enumerate(sets_idx):
            for t, i in enumerate(ids):
                assert out[t, r] == orc.bow_score_l1(*as_sent[sets_idx[me][t]], *as_sent[i])
    assert CrossStreamLoopCandidates.candidates(out, rank=-1, min_score=0.0)
    # the f64 form: local vectors of the last BoW batch against explicit foreign vectors
    c.detect_batch_dev(dev.data_ptr(), B)
    c.bow_batch_dev(False)
    sets = [[3, 4, 5], [1, 2, 0]]
    W = torch.zeros((2, B, k_max), dtype=torch.int32)
    Vv = torch.zeros((2, B, k_max), dtype=torch.float64)
    N = torch.zeros((2, B), dtype=torch.int32)
    for r, ids in enumerate(sets):
        for t, i in enumerate(ids):
            n = len(vecs[i][0])
            W[r, t, :n] = torch.from_numpy(vecs[i][0].view(np.int32))
            Vv[r, t, :n] = torch.from_numpy(vecs[i][1])
            N[r, t] = n
    W, Vv, N = W.cuda(), Vv.cuda(), N.cuda()
    out = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    c.bow_cross_score_dev(W.data_ptr(), Vv.data_ptr(), N.data_ptr(), 2, k_max, out.data_ptr())
    c.sync()
    out = out.cpu().numpy()
    for r, ids in enumerate(sets):
        for t, i in enumerate(ids):
            assert out[t, r] == orc.bow_score_l1(*vecs[t], *vecs[i])
    c.close()


def _l1_model_best(orc, vecs, t, window=64):
    """what the database must report for entry t queried before it is added: the best of the last `window`
    entries, ties to the lower entry id, entries without a common word absent"""
    best = None
    for e in range(max(0, t - window), t):
        s = orc.bow_score_l1(*vecs[t], *vecs[e])
        if s > 0 and (best is None or s > best[0]):
            best = (s, e)
    return best


def test_cfg3_million_word_vocabulary(pkg, orc, descs):
    """BASELINE cfg3: the k=10, L=6 vocabulary (1 111 111 nodes, 10^6 words, 35.6 MB of node descriptors: the last
    two tree levels stream from HBM / Infinity Cache) — words, idf weights, BoW vectors (f64 bit patterns) and L1
    scores against the oracle, for detected and random descriptors, through the host and the batched entry points."""
    import torch
    blob = synth.make_vocabulary(10, 6, seed=77)
    V = orc.Vocabulary(blob)
    assert (V.k, V.L, V.n_nodes, V.n_words) == (10, 6, 1111111, 1000000)
    K = 4096
    c = pkg.Context(width=640, height=480, max_batch=4, max_keypoints=K)
    c.bow_load(blob)
    info = c.bow_info()
    assert (info["k"], info["L"], info["n_nodes"], info["n_words"]) == (10, 6, 1111111, 1000000)
    rng = np.random.default_rng(9)
    vecs = []
    for d in (descs[0], descs[1], descs[2], rng.integers(0, 256, (4000, 32), dtype=np.uint8)):
        gw, gwt = c.bow_words(d)
        rw, rwt = V.words(d)
        assert np.array_equal(gw, rw) and np.array_equal(gwt, rwt)
        assert len(np.unique(rw)) > 0.9 * len(rw) and rw.max() > 900000      # the deep levels are really used
        gv, rv = c.bow_transform(d), V.bow_vector(d)
        assert np.array_equal(gv[0], rv[0]) and np.array_equal(gv[1], rv[1])
        vecs.append(rv)
    for i in range(4):
        for j in range(4):
            assert c.bow_score(*vecs[i], *vecs[j]) == orc.bow_score_l1(*vecs[i], *vecs[j])
    # batched device path on the same vocabulary: vectors, best entry and score per frame (query-then-add)
    frames = synth.make_stream(4, 640, 480, seed=1234)
    c.detect_batch_dev(torch.from_numpy(frames).cuda().data_ptr(), 4)
    c.bow_batch_dev(True)
    c.sync()
    v = c.bow_view()
    n = pkg.read_device(c, v.n_words, (4,), np.int32)
    words = pkg.read_device(c, v.words, (4, K), np.uint32)
    vals = pkg.read_device(c, v.values, (4, K), np.float64)
    be = pkg.read_device(c, v.best_entry, (4,), np.int32)
    bs = pkg.read_device(c, v.best_score, (4,), np.float64)
    fv = [V.bow_vector(orc.detect(f, orc.params())["desc"]) for f in frames]
    for t in range(4):
        assert n[t] == len(fv[t][0])
        assert np.array_equal(words[t, :n[t]], fv[t][0]) and np.array_equal(vals[t, :n[t]], fv[t][1])
        exp = _l1_model_best(orc, fv, t)
        assert (be[t], bs[t]) == ((exp[1], exp[0]) if exp else (-1, 0.0))
    c.close()


def test_database_inverted_file_and_batch_window(pkg, orc):
    """host side: the database is an unbounded inverted file (DBoW3::Database(voc, false, 0), orb_relocalizer.cpp:29):
    300 adds (the storage grows past its first reservation of 64), queries against EVERY entry, removals.
    Batched device path: every frame is scored against the 64 entries that precede it (cfg3); 50 batches of 3 wrap
    that window more than twice, and its entries land in the same inverted file."""
    import torch
    blob = synth.make_vocabulary(10, 3)
    V = orc.Vocabulary(blob)
    K = 4096
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=K)
    c.bow_load(blob)
    rng = np.random.default_rng(21)
    pool = rng.integers(0, 256, (40, 300, 32), dtype=np.uint8)      # 40 descriptor sets that share words
    sets, vecs = [], []
    n_add = 300
    for t in range(n_add):
        d = pool[rng.integers(0, 40)].copy()
        d[:rng.integers(1, 200)] = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        sets.append(d)
        vecs.append(V.bow_vector(d))
    c.bow_db_reserve(64)
    removed = set()

    def model(t, live):
        exp = sorted(((orc.bow_score_l1(*vecs[t], *vecs[e]), e) for e in live), key=lambda x: (-x[0], x[1]))
        return [x for x in exp if x[0] > 0]
    for t in range(n_add):
        if t % 13 == 0 or t > n_add - 8:
            ids, sc = c.bow_db_query(sets[t], n_add)
            exp = model(t, [e for e in range(t) if e not in removed])
            assert list(ids) == [e for _, e in exp] and list(sc) == [s for s, _ in exp], t
        assert c.bow_db_add(sets[t]) == t and c.bow_db_size() == t + 1
        if t in (40, 41, 170):
            c.bow_db_remove(t - 7)                                  # IRelocalizer::removeKeyframe
            removed.add(t - 7)
    ids, sc = c.bow_db_query(sets[5], 10)                           # top-10 of 300, an entry of the database itself
    exp = model(5, [e for e in range(n_add) if e not in removed])[:10]
    assert list(ids) == [e for _, e in exp] and list(sc) == [s for s, _ in exp] and ids[0] == 5 and sc[0] > 0.999
    c.bow_db_clear()
    # batched path: frames cycle with period 5 (not a divisor of the batch size), so equal scores occur inside the
    # window and the tie rule (lower entry id) is exercised across the wrap
    frames = synth.make_stream(5, 640, 480, seed=1234)
    fv5 = [V.bow_vector(orc.detect(f, orc.params())["desc"]) for f in frames]
    order = [(7 * i) % 5 if i % 11 else (i // 11) % 5 for i in range(150)]
    fv = [fv5[i] for i in order]
    dev = torch.from_numpy(np.ascontiguousarray(frames[order])).cuda()
    for b in range(50):
        c.detect_batch_dev(dev[3 * b:].data_ptr(), 3)
        c.bow_batch_dev(True)
        c.sync()
        v = c.bow_view()
        be = pkg.read_device(c, v.best_entry, (3,), np.int32)
        bs = pkg.read_device(c, v.best_score, (3,), np.float64)
        for i in range(3):
            exp = _l1_model_best(orc, fv, 3 * b + i)
            assert (be[i], bs[i]) == ((exp[1], exp[0]) if exp else (-1, 0.0)), (b, i)
    # the 150 batch entries are in the inverted file: a host query sees all of them, not just the last 64
    assert c.bow_db_size() == 150
    q = orc.detect(frames[2], orc.params())["desc"]
    ids, sc = c.bow_db_query(q, 150)
    exp = sorted(((orc.bow_score_l1(*fv5[2], *fv[e]), e) for e in range(150)), key=lambda x: (-x[0], x[1]))
    exp = [x for x in exp if x[0] > 0]
    assert list(ids) == [e for _, e in exp] and list(sc) == [s for s, _ in exp]
    assert sum(1 for s in sc if s > 0.999) == order.count(2) > 20      # every revisit of frame 2, old ones included
    c.close()


def test_flat_word_assignment(pkg, orc, descs):
    """a15b: the exhaustive descriptor-vs-vocabulary Hamming search (north_star's "batched descriptor-vs-vocabulary
    Hamming kernel", SURVEY §8d bow_flat) against brute force: least distance over ALL words, lower word id on ties;
    then the whole BoW path (vectors, database) on flat assignments, and the 10^6-word vocabulary."""
    import torch
    blob = synth.make_vocabulary(10, 4, seed=31)                      # 10^4 words
    V = orc.Vocabulary(blob)
    K = 4096
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=K)
    c.bow_load(blob)
    c.bow_set_assignment(pkg.BOW_ASSIGN_FLAT)
    rng = np.random.default_rng(12)
    for d in (descs[0], rng.integers(0, 256, (777, 32), dtype=np.uint8), descs[1][:1]):
        gw, gwt = c.bow_words(d)
        rw, rwt = V.words_flat(d)
        assert np.array_equal(gw, rw) and np.array_equal(gwt, rwt)
    # ties: queries that ARE leaf descriptors, and a vocabulary with duplicated leaves -> the lower word id wins
    tree_w, _ = V.words(descs[0])
    flat_w, _ = V.words_flat(descs[0])
    assert (tree_w != flat_w).mean() > 0.3                            # the descent is only an approximation
    # BoW vectors built on flat words: the oracle's vector code on the flat assignment
    gv = c.bow_transform(descs[0])
    w, wt = V.words_flat(descs[0])
    order = np.argsort(w, kind="stable")
    uw, first = np.unique(w[order], return_index=True)
    # addWeight: a word hit cnt times accumulates its idf weight cnt times, sequentially
    vals = np.array([sum([wt[order][f]] * cnt, 0.0) for f, cnt in zip(first, np.diff(np.append(first, len(w))))])
    norm = 0.0
    for x in vals:
        norm += abs(x)
    assert np.array_equal(gv[0], uw) and np.array_equal(gv[1], vals / norm)
    # batched device path in flat mode
    c.detect_batch_dev(torch.from_numpy(np.ascontiguousarray(synth.make_stream(3, 640, 480, seed=1234))).cuda().data_ptr(), 3)
    c.bow_batch_dev(True)
    c.sync()
    v = c.bow_view()
    n = pkg.read_device(c, v.n_words, (3,), np.int32)
    words = pkg.read_device(c, v.words, (3, K), np.uint32)
    fr = synth.make_stream(3, 640, 480, seed=1234)
    for t in range(3):
        fw, _ = V.words_flat(orc.detect(fr[t], orc.params())["desc"])
        assert np.array_equal(words[t, :n[t]], np.unique(fw))
    c.bow_set_assignment(pkg.BOW_ASSIGN_TREE)
    gw, _ = c.bow_words(descs[0])
    assert np.array_equal(gw, tree_w)
    c.close()
    # cfg3's 10^6-word vocabulary, flat: 200 x 10^6 distances against brute force
    blob = synth.make_vocabulary(10, 6, seed=77)
    V = orc.Vocabulary(blob)
    c = pkg.Context(width=0, height=0, max_keypoints=K)              # BoW-only context
    c.bow_load(blob)
    c.bow_set_assignment(pkg.BOW_ASSIGN_FLAT)
    d = np.concatenate([descs[2][:150], rng.integers(0, 256, (50, 32), dtype=np.uint8)])
    gw, gwt = c.bow_words(d)
    rw, rwt = V.words_flat(d)
    assert np.array_equal(gw, rw) and np.array_equal(gwt, rwt)
    c.close()


def test_bow_edge_cases(pkg, orc):
    """empty inputs, removal / clear bookkeeping and loud failures of the database entry points"""
    blob = synth.make_vocabulary(10, 3)
    V = orc.Vocabulary(blob)
    c = pkg.Context(width=0, height=0, max_keypoints=1024)          # no detector buffers
    with pytest.raises(pkg.MslamHipError) as e:
        c.detect(np.zeros((480, 640, 3), np.uint8))
    assert e.value.code == pkg.E_INVALID and "without a detector" in str(e.value)
    c.bow_load(blob)
    rng = np.random.default_rng(2)
    d = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    assert len(c.bow_db_query(d, 5)[0]) == 0 and c.bow_db_size() == 0   # empty database
    w, v = c.bow_transform(d[:0])                                        # no descriptors: empty vector
    assert len(w) == 0
    assert c.bow_db_add(d) == 0 and c.bow_db_add(d[:0]) == 1              # an entry without words is legal
    ids, sc = c.bow_db_query(d, 5)
    assert list(ids) == [0] and sc[0] == orc.bow_score_l1(*V.bow_vector(d), *V.bow_vector(d))
    with pytest.raises(pkg.MslamHipError):
        c.bow_db_remove(7)                                                # no such entry
    c.bow_db_remove(0)
    assert len(c.bow_db_query(d, 5)[0]) == 0 and c.bow_db_size() == 2     # removed entries keep their ids
    c.bow_db_clear()
    assert c.bow_db_size() == 0 and c.bow_db_add(d) == 0                  # ids restart after clear
    assert list(c.bow_db_query(d, 5)[0]) == [0]
    with pytest.raises(pkg.MslamHipError) as e:
        c.bow_words(rng.integers(0, 256, (2000, 32), dtype=np.uint8))     # more descriptors than max_keypoints
    assert e.value.code == pkg.E_INVALID
    c.close()
    c = pkg.Context(width=0, height=0, max_keypoints=16384)
    with pytest.raises(pkg.MslamHipError) as e:                           # LDS budget of the scoring kernel, at load time
        c.bow_load(blob)
    assert e.value.code == pkg.E_INVALID and "LDS" in str(e.value)
    c.close()
