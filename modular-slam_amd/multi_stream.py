"""Cross-stream loop candidates for multi-camera / multi-GPU runs.

One process per GPU, one camera stream per process (torch.distributed: backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  ORB extraction, matching and the per-stream BoW database
need no communication at all — frames of different streams are independent.  The only exchange step
is this one: after a batch, every rank contributes its per-frame BoW vectors in the wire format of
include/mslam_hip.h (`mslam_hip_bow_pack_dev`): per frame k_max x {u32 word, f32 value} + a count,
k_max = 2048 -> 16 KB per frame, and ONE `all_gather_into_tensor` per batch moves all of them (a batch
is one collective: fewer, larger messages suit the point-to-point xGMI mesh).  Every rank then scores
its frame t against frame t of every other stream from the gathered buffer alone.

`granularity="frame"` is the other end of the trade: one collective PER FRAME, each moving one frame's set
(`set_dwords(1, k_max)` dwords = 16 KB per rank at k_max = 2048, 128 KB gathered on 8 ranks — SURVEY.md §8e's
per-frame exchange).  A frame's set is exactly the batch format for a batch of one frame, so both forms speak
the same wire format and a frame-granular rank interoperates with nothing but itself being slower per byte: it
is the form for a live rig whose "batch" is the frame that just arrived, where the exchange is latency-bound;
the per-batch collective is the throughput form.  `split_set` / `join_sets` convert between the two layouts.

GPU path (`step_gpu`): pack -> all-gather -> score are enqueued on a communication stream behind an
event of the context's stream — no host synchronisation — so the collective of batch i overlaps the
extraction of batch i+1; `finish()` joins the two streams.
Nothing like this exists in the reference (single process, single camera).
"""
import numpy as np
import torch
import torch.distributed as dist


class DeviceArray:
    """Zero-copy view of a raw device pointer for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def view_as_tensor(ptr, shape, dtype):
    typestr = {torch.int32: "<i4", torch.float64: "<f8", torch.uint8: "|u1", torch.float32: "<f4",
               torch.int16: "<i2"}[dtype]
    return torch.as_tensor(DeviceArray(ptr, shape, typestr), device="cuda")


def set_dwords(n_frames, k_max):
    """dwords of one stream's batch in the exchange format"""
    return n_frames * (2 * k_max + 1)


def pack_vectors(words, values, counts, k_max):
    """torch restatement of mslam_hip_bow_pack_dev for host-side callers and CPU tests: words [B,cap] int32 (bit
    pattern of the u32 ids), values [B,cap] float64, counts [B] int32 -> int32 [B*(2*k_max+1)]."""
    B = words.shape[0]
    if int(counts.max()) > k_max:
        raise ValueError("a BoW vector has more than k_max=%d words" % k_max)
    cap = min(words.shape[1], k_max)
    live = torch.arange(cap, device=words.device)[None, :] < counts[:, None]
    vec = torch.zeros((B, k_max, 2), dtype=torch.int32, device=words.device)
    vec[:, :cap, 0] = torch.where(live, words[:, :cap], torch.zeros_like(words[:, :cap]))
    f32 = values[:, :cap].to(torch.float32).view(torch.int32)
    vec[:, :cap, 1] = torch.where(live, f32, torch.zeros_like(f32))
    return torch.cat([vec.reshape(-1), counts.to(torch.int32)])


def unpack_set(buf, n_frames, k_max):
    """one gathered set -> (words [B,k_max] uint32, values [B,k_max] float32 widened to float64, counts [B]) as numpy"""
    a = buf.cpu().numpy()
    vec = a[:n_frames * k_max * 2].reshape(n_frames, k_max, 2)
    return (vec[:, :, 0].view(np.uint32).copy(), vec[:, :, 1].copy().view(np.float32).astype(np.float64),
            a[n_frames * k_max * 2:].copy())


def split_set(batch_set, n_frames, k_max):
    """a batch's set [set_dwords(n_frames)] -> its frames' sets [n_frames, set_dwords(1)] (frame t: its k_max (word, value)
    pairs, then its count): each row is the wire format of a batch of one frame"""
    vec = batch_set[:n_frames * 2 * k_max].view(n_frames, 2 * k_max)
    cnt = batch_set[n_frames * 2 * k_max:].view(n_frames, 1)
    return torch.cat([vec, cnt], dim=1).contiguous()


def join_sets(frame_sets, k_max):
    """gathered per-frame sets [n_frames, world, set_dwords(1)] -> the per-batch layout [world, set_dwords(n_frames)]"""
    n_frames, world = frame_sets.shape[0], frame_sets.shape[1]
    vec = frame_sets[:, :, :2 * k_max].permute(1, 0, 2).reshape(world, n_frames * 2 * k_max)
    cnt = frame_sets[:, :, 2 * k_max].permute(1, 0)
    return torch.cat([vec, cnt], dim=1).contiguous()


class CrossStreamLoopCandidates:
    def __init__(self, k_max=2048, group=None, always_collective=False, granularity="batch"):
        if granularity not in ("batch", "frame"):
            raise ValueError("granularity must be 'batch' (one collective per batch) or 'frame' (one per frame)")
        self.granularity = granularity
        self.k_max = k_max
        self.group = group
        # world 1 normally short-cuts the gather to a device copy; always_collective issues the collective anyway (what a
        # single-GPU box can exercise of the RCCL path: tests/test_gpu_bow.py::test_rccl_world1_exchange)
        self.always_collective = always_collective
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._comm = None      # communication stream (GPU path)
        self._bufs = {}        # (n_frames, slot) -> (local set, gathered sets, scores)
        self._slot = 0
        self._pack_done = None
        self.collectives = 0   # all-gathers issued (one per batch, or one per frame with granularity="frame")
        self.bytes_per_collective = 0  # payload each rank contributed to the last collective
        self.last = None

    # ---- the exchange step -------------------------------------------------------------------
    def all_gather_sets(self, local_set, out=None, n_frames=None):
        """The exchange of one batch: local_set int32 [set_dwords(n_frames)] -> int32 [world, set_dwords(n_frames)].
        granularity "batch": ONE collective; "frame" (needs n_frames): one collective per frame, each moving that
        frame's set_dwords(1) dwords; the result is put back into the per-batch layout, so callers see no difference."""
        if self.granularity == "frame" and n_frames is not None and n_frames > 1:
            mine = split_set(local_set, n_frames, self.k_max)                         # [B, 2 k_max + 1]
            got = torch.empty((n_frames, self.world, mine.shape[1]), dtype=mine.dtype, device=mine.device)
            for t in range(n_frames):
                self._all_gather_one(mine[t], got[t])
            joined = join_sets(got, self.k_max)
            if out is None:
                return joined
            out.copy_(joined)
            return out
        return self._all_gather_one(local_set, out)

    def _all_gather_one(self, local_set, out=None):
        if out is None:
            out = torch.empty((self.world, local_set.numel()), dtype=local_set.dtype, device=local_set.device)
        if self.world > 1 and local_set.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks on one GPU): gloo moves host memory, so bounce through it
            torch.cuda.current_stream().synchronize()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h_out.view(-1), local_set.cpu(), group=self.group)
            out.copy_(h_out)
        elif self.world > 1 or (self.always_collective and dist.is_initialized()):
            dist.all_gather_into_tensor(out.view(-1), local_set, group=self.group)
        else:
            out[0].copy_(local_set)
        self.collectives += 1
        self.bytes_per_collective = local_set.numel() * local_set.element_size()
        return out

    # ---- GPU path: vectors come straight out of the context's BoW view --------------------------
    def _buffers(self, n_frames):
        key = (n_frames, self._slot)
        if key not in self._bufs:
            n = set_dwords(n_frames, self.k_max)
            self._bufs[key] = (torch.empty(n, dtype=torch.int32, device="cuda"),
                               torch.empty((self.world, n), dtype=torch.int32, device="cuda"),
                               torch.zeros((n_frames, self.world), dtype=torch.float64, device="cuda"))
        return self._bufs[key]

    def step_gpu(self, ctx, ctx_stream, n_frames):
        """After ctx.bow_batch_dev() on `ctx_stream` (the torch stream the context was created on): pack the batch's
        vectors on that stream, then — on the communication stream, behind an event — all-gather the sets of all
        streams and score own frame t against every stream's frame t.  Nothing here waits on the host.  Returns
        the scores tensor [n_frames, world] (float64, cuda; column `rank` is the self-score), valid once
        `finish()` (or a wait on the communication stream) has been called; two buffer sets alternate, so a
        result stays intact until the second-next call."""
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        local, gathered, scores = self._buffers(n_frames)
        self._slot ^= 1
        self.last = (local, gathered, scores)  # the buffer set of this call (valid after finish())
        # these buffers were last used two steps ago by the communication stream
        ctx_stream.wait_stream(self._comm)
        ctx.bow_pack_dev(self.k_max, local.data_ptr())            # on the context's stream
        self._comm.wait_stream(ctx_stream)
        with torch.cuda.stream(self._comm):
            self.all_gather_sets(local, gathered, n_frames)
            ctx.bow_cross_score_packed_dev(gathered.data_ptr(), self.world, self.rank, n_frames, self.k_max,
                                           scores.data_ptr(), stream=self._comm.cuda_stream)
        return scores

    def finish(self, ctx_stream=None):
        """join the communication stream (host wait, and optionally make `ctx_stream` wait for it)"""
        if self._comm is not None:
            if ctx_stream is not None:
                ctx_stream.wait_stream(self._comm)
            self._comm.synchronize()

    # ---- generic path (CPU tests): same format, same collective; the scorer is injected ----------
    def step_with(self, words, values, counts, scorer):
        """words/values/counts as in pack_vectors (CPU tensors); scorer(w1, v1, w2, v2) -> float is applied to the
        vectors exactly as transmitted (f32 values widened to f64)."""
        B = words.shape[0]
        sets = self.all_gather_sets(pack_vectors(words, values, counts, self.k_max), n_frames=B)
        un = [unpack_set(sets[r], B, self.k_max) for r in range(self.world)]
        mw, mv, mn = un[self.rank]
        out = np.zeros((B, self.world))
        for t in range(B):
            for r in range(self.world):
                w2, v2, n2 = un[r]
                out[t, r] = scorer(mw[t, :mn[t]], mv[t, :mn[t]], w2[t, :n2[t]], v2[t, :n2[t]])
        return out

    @staticmethod
    def candidates(scores, rank, min_score=0.05):
        """[(frame, other_rank, score)] for cross-stream pairs whose L1 score reaches min_score."""
        s = scores.cpu().numpy() if hasattr(scores, "cpu") else np.asarray(scores)
        out = []
        for t in range(s.shape[0]):
            for r in range(s.shape[1]):
                if r != rank and s[t, r] >= min_score:
                    out.append((t, r, float(s[t, r])))
        return out
