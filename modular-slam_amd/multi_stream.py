"""Cross-stream loop candidates for multi-camera / multi-GPU runs.

One process per GPU, one camera stream per process (torch.distributed: backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  ORB extraction, matching and the per-stream BoW database
need no communication at all — frames of different streams are independent.  The only exchange step
is this one: after a batch, every rank all-gathers its per-frame BoW vectors (padded sparse vectors:
k_max x {u32 word, f64 value} + a count), then scores its own frame t against frame t of every other
stream.  The payload is small (k_max = 2048 -> 24.6 KB per frame), so the all-gather is latency-bound
and a single fused collective per batch is the right shape for the point-to-point xGMI mesh.
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
    typestr = {torch.int32: "<i4", torch.float64: "<f8", torch.uint8: "|u1", torch.float32: "<f4"}[dtype]
    return torch.as_tensor(DeviceArray(ptr, shape, typestr), device="cuda")


class CrossStreamLoopCandidates:
    def __init__(self, k_max=2048, group=None):
        self.k_max = k_max
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    # ---- the exchange step -------------------------------------------------------------------
    def all_gather_vectors(self, words, values, counts):
        """words [B,k] int32 (bit pattern of the u32 ids), values [B,k] float64, counts [B] int32 ->
        (W [world,B,k], V [world,B,k], N [world,B]).  One collective per tensor, same device as the inputs."""
        if int(counts.max()) > self.k_max:
            raise ValueError("a BoW vector has more than k_max=%d words" % self.k_max)
        w = words[:, :self.k_max].contiguous()
        v = values[:, :self.k_max].contiguous()
        n = counts.contiguous()
        W = torch.empty((self.world,) + tuple(w.shape), dtype=w.dtype, device=w.device)
        V = torch.empty((self.world,) + tuple(v.shape), dtype=v.dtype, device=v.device)
        N = torch.empty((self.world,) + tuple(n.shape), dtype=n.dtype, device=n.device)
        if self.world > 1:
            # outputs are the contiguous slices W[r] etc.; works for both nccl (RCCL) and gloo
            dist.all_gather(list(W.unbind(0)), w, group=self.group)
            dist.all_gather(list(V.unbind(0)), v, group=self.group)
            dist.all_gather(list(N.unbind(0)), n, group=self.group)
        else:
            W[0], V[0], N[0] = w, v, n
        return W, V, N

    # ---- GPU path: vectors come straight out of the context's BoW view --------------------------
    def step_gpu(self, ctx):
        """After ctx.bow_batch_dev(): gather all streams' vectors and score own frame t against every
        stream's frame t.  Returns scores [B, world] (float64 cuda tensor; column `rank` is the
        self-score)."""
        v = ctx.bow_view()
        B, cap = ctx.params.max_batch, v.capacity
        words = view_as_tensor(v.words, (B, cap), torch.int32)
        values = view_as_tensor(v.values, (B, cap), torch.float64)
        counts = view_as_tensor(v.n_words, (B,), torch.int32)
        torch.cuda.current_stream().synchronize()
        ctx.sync()  # the context runs on its own stream
        W, V, N = self.all_gather_vectors(words, values, counts)
        scores = torch.zeros((B, self.world), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.bow_cross_score_dev(W.data_ptr(), V.data_ptr(), N.data_ptr(), self.world, self.k_max, scores.data_ptr())
        ctx.sync()
        return scores

    # ---- generic path (CPU tests): the scorer is injected ---------------------------------------
    def step_with(self, words, values, counts, scorer):
        W, V, N = self.all_gather_vectors(words, values, counts)
        B = words.shape[0]
        out = np.zeros((B, self.world))
        for t in range(B):
            n1 = int(counts[t])
            w1 = words[t, :n1].numpy().view(np.uint32)
            v1 = values[t, :n1].numpy()
            for r in range(self.world):
                n2 = int(N[r, t])
                out[t, r] = scorer(w1, v1, W[r, t, :n2].numpy().view(np.uint32), V[r, t, :n2].numpy())
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
