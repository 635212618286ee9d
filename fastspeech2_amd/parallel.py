"""Utterance sharding across the GPUs of one node + the single collective of the path.

The reference has no distributed code at all.  Utterances never interact
(per-utterance semantics), weights are small (~130 MB) and replicated, so the path shards by independent
units: every rank holds the (tiny) id batch, takes a cost-balanced subset of the utterances, runs the
single-GPU path on it, and the only exchange is an all-gather of the final mels over RCCL/xGMI (backend
"nccl" on ROCm; "gloo" in the CPU tests).  The gather and the order restoration are exact; an utterance's values do not
depend on its batch-mates (only, in the last bits, on the size-dependent kernel variants: DESIGN.md section 1).
"""
import torch
import torch.distributed as dist


def path_flops(T, L):
    """Algorithmic FLOPs of the path for one utterance of T phonemes and L frames, default dims, valid positions only
    (SURVEY.md section 8d): T (23,855,616 + 4,096 T) + L (40,383,488 + 6,144 L)."""
    T, L = float(T), float(L)
    return T * (23855616.0 + 4096.0 * T) + L * (40383488.0 + 6144.0 * L)


def utterance_cost(T, frames_per_token=7.87):
    """Relative cost model of one utterance from its phoneme count (SURVEY.md section 8d FLOPs formula with
    L ~= 7.87 T): dominated by the decoder, 40.4 MFLOP/frame + 6144 L^2 attention."""
    L = frames_per_token * float(T)
    return T * (23855616.0 + 4096.0 * T) + L * (40383488.0 + 6144.0 * L)


def shard_indices(ilens, world_size, costs=None):
    """Longest-processing-time-first assignment.  Returns a list (one per rank) of utterance indices,
    each sorted ascending; deterministic, identical on every rank."""
    ilens = [int(t) for t in ilens]
    costs = [utterance_cost(t) for t in ilens] if costs is None else [float(c) for c in costs]
    order = sorted(range(len(ilens)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += costs[i]
    return [sorted(p) for p in parts]


def _segments(lengths_dev, total):
    """For ragged segments of the given lengths (device int64 [n], sum == total, known on the host):
    (segment id, position inside the segment) of every element, computed on the device without a sync."""
    n = lengths_dev.numel()
    seg = torch.repeat_interleave(torch.arange(n, device=lengths_dev.device), lengths_dev, output_size=total)
    start = torch.cumsum(lengths_dev, 0) - lengths_dev
    pos = torch.arange(total, device=lengths_dev.device) - start[seg]
    return seg, pos


def gather_mels(mel_local, olens_local, index_local, total, group=None):
    """All-gather the ragged per-rank mel batches and restore the original utterance order.

    mel_local [b, L_local, odim] (pads zero), olens_local [b] (host or device), index_local: the global
    utterance index of each local row.  Returns (mels [total, Lmax, odim] on mel_local's device, olens
    [total] int64 on the host).

    Only VALID frames travel: each rank packs its utterances back to back ([frames, odim], 2-3x fewer bytes
    than the padded batch), the packs are padded to the largest rank's frame count for one equal-count
    all_gather_into_tensor (RCCL over xGMI), and every rank scatters the received rows into the padded,
    ordered result with device-side index arithmetic.  A small all-gather of (count, olens, index)
    metadata precedes it (its host read-back sizes the buffers)."""
    world = dist.get_world_size(group)
    dev = mel_local.device
    odim = mel_local.shape[-1]
    b, Lloc = mel_local.shape[0], mel_local.shape[1]
    cap = total                                         # upper bound on any rank's utterance count
    ol_loc = torch.as_tensor(olens_local, dtype=torch.int64)
    meta = torch.full((1 + 2 * cap,), -1, dtype=torch.int64)
    meta[0] = b
    meta[1:1 + b] = ol_loc.cpu()
    meta[1 + cap:1 + cap + b] = torch.as_tensor(index_local, dtype=torch.int64)
    meta = meta.to(dev)
    metas = torch.empty(world * meta.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world, -1).cpu()                 # host needs the sizes (one sync per gather)
    counts = metas[:, 0].tolist()
    ol_all = [metas[r, 1:1 + counts[r]] for r in range(world)]
    gidx_all = [metas[r, 1 + cap:1 + cap + counts[r]] for r in range(world)]
    nfr = [int(o.sum()) for o in ol_all]
    nmax, Lmax = max(max(nfr), 1), max(int(o.max()) if o.numel() else 0 for o in ol_all)
    rank = dist.get_rank(group)
    # pack the local valid frames
    send = mel_local.new_zeros(nmax, odim)
    if nfr[rank] > 0:
        seg, pos = _segments(ol_loc.to(dev), nfr[rank])
        send[: nfr[rank]] = mel_local.reshape(b * Lloc, odim).index_select(0, seg * Lloc + pos)
    recv = mel_local.new_empty(world * nmax, odim)
    dist.all_gather_into_tensor(recv, send, group=group)
    # scatter every rank's pack into the ordered padded result
    lens_cat = torch.cat(ol_all).to(dev)
    gidx_cat = torch.cat(gidx_all).to(dev)
    rank_of = torch.cat([torch.full((counts[r],), r, dtype=torch.int64) for r in range(world)]).to(dev)
    nall = sum(nfr)
    out = mel_local.new_zeros(total * Lmax, odim)
    if nall > 0:
        seg, pos = _segments(lens_cat, nall)
        # offset of each utterance inside its rank's pack
        start_all = torch.cumsum(lens_cat, 0) - lens_cat
        rank_base = torch.tensor([sum(nfr[:r]) for r in range(world)], dtype=torch.int64, device=dev)
        src = rank_of[seg] * nmax + (start_all[seg] - rank_base[rank_of[seg]]) + pos
        dst = gidx_cat[seg] * Lmax + pos
        out.index_copy_(0, dst, recv.index_select(0, src))
    olens = torch.zeros(total, dtype=torch.int64)
    olens[torch.cat(gidx_all)] = torch.cat(ol_all)
    return out.view(total, Lmax, odim), olens


def unpack_rows(packed, starts, lens, Lout):
    """packed [n, W] -> [len(lens), Lout, W] zero padded.  HIP kernel (fs2_op_unpack_rows) on the GPU; plain torch on
    the CPU (only the gloo tests take that branch)."""
    B, W = len(lens), packed.shape[1]
    if packed.is_cuda:
        import ctypes as C
        from . import _lib
        out = torch.empty(B, Lout, W, dtype=torch.float32, device=packed.device)
        arr = lambda v: (C.c_int32 * B)(*[int(i) for i in v])
        with torch.cuda.device(packed.device):
            _lib.check(_lib.lib().fs2_op_unpack_rows(C.c_void_p(torch.cuda.current_stream(packed.device).cuda_stream),
                                                     packed.data_ptr(), W, B, arr(starts), arr(lens), Lout, out.data_ptr()))
        return out
    out = packed.new_zeros(B, Lout, W)
    for i, (s0, L) in enumerate(zip(starts, lens)):
        out[i, :L] = packed[s0:s0 + L]
    return out


def gather_packed(packed_local, olens_local, index_local, total, group=None):
    """Same contract as gather_mels for an already packed local batch [sum(olens_local), odim] (what
    ``FeedForwardTransformer.inference_batch(packed=True)`` returns): metadata all-gather (sizes -> host), one
    equal-count all_gather_into_tensor of the packs, one unpack kernel into the ordered padded result."""
    world = dist.get_world_size(group)
    dev = packed_local.device
    odim = packed_local.shape[-1]
    ol_loc = torch.as_tensor(olens_local, dtype=torch.int64).cpu()
    b, cap = ol_loc.numel(), total
    meta = torch.full((1 + 2 * cap,), -1, dtype=torch.int64)
    meta[0] = b
    meta[1:1 + b] = ol_loc
    meta[1 + cap:1 + cap + b] = torch.as_tensor(index_local, dtype=torch.int64)
    meta = meta.to(dev)
    metas = torch.empty(world * meta.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world, -1).cpu()
    counts = metas[:, 0].tolist()
    nfr = [int(metas[r, 1:1 + counts[r]].sum()) for r in range(world)]
    nmax = max(max(nfr), 1)
    if packed_local.shape[0] == nmax:
        send = packed_local.contiguous()
    else:
        send = packed_local.new_zeros(nmax, odim)
        send[: packed_local.shape[0]] = packed_local
    recv = packed_local.new_empty(world * nmax, odim)
    dist.all_gather_into_tensor(recv, send, group=group)
    starts = [0] * total
    lens = [0] * total
    for r in range(world):
        off = r * nmax
        for g, L in zip(metas[r, 1 + cap:1 + cap + counts[r]].tolist(), metas[r, 1:1 + counts[r]].tolist()):
            starts[g], lens[g] = off, L
            off += L
    return unpack_rows(recv, starts, lens, max(lens)), torch.tensor(lens, dtype=torch.int64)


def gather_packed_async(packed_cap, olens_dev, index_dev, total, Lout, group=None):
    """Sync-free form of :func:`gather_packed` for ``inference_batch(sync=False, packed=True)``: nothing is read back to the
    host.  Every rank passes a pack of the SAME capacity ([rows_cap, odim], valid frames first), its frame counts and the
    global indices of its utterances as device int64 tensors of the SAME length b; ``total`` = world * b; ``Lout`` = padded
    length of the result (>= the longest utterance of any rank; e.g. the agreed per-utterance capacity).  Three collectives
    (packs, counts, indices: RCCL over xGMI), device-side offset arithmetic, one unpack kernel.  Returns
    (mels [total, Lout, odim] in global utterance order, olens [total] int64 on the device)."""
    world = dist.get_world_size(group)
    dev = packed_cap.device
    cap, odim = packed_cap.shape
    b = olens_dev.numel()
    recv = packed_cap.new_empty(world * cap, odim)
    dist.all_gather_into_tensor(recv, packed_cap.contiguous(), group=group)
    ol_all = torch.empty(world * b, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ol_all, olens_dev.contiguous(), group=group)
    gi_all = torch.empty(world * b, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gi_all, index_dev.contiguous(), group=group)
    ol = ol_all.view(world, b)
    starts = (torch.cumsum(ol, 1) - ol + torch.arange(world, device=dev).unsqueeze(1) * cap).reshape(-1)
    starts_g = torch.zeros(total, dtype=torch.int32, device=dev).scatter_(0, gi_all, starts.to(torch.int32))
    lens_g = torch.zeros(total, dtype=torch.int32, device=dev).scatter_(0, gi_all, ol_all.to(torch.int32))
    if packed_cap.is_cuda:
        import ctypes as C
        from . import _lib
        out = torch.empty(total, Lout, odim, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().fs2_op_unpack_rows_dev(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), recv.data_ptr(), odim, total,
                                                         starts_g.data_ptr(), lens_g.data_ptr(), Lout, out.data_ptr()))
    else:       # CPU (gloo tests): same arithmetic in plain torch
        pos = torch.arange(Lout).unsqueeze(0)
        idx = (starts_g.long().unsqueeze(1) + pos).clamp(max=recv.shape[0] - 1)
        out = torch.where((pos < lens_g.long().unsqueeze(1)).unsqueeze(-1), recv[idx], torch.zeros((), dtype=recv.dtype))
    return out, lens_g.to(torch.int64)


class ShardedSynthesizer:
    """Free-running batched synthesis over all ranks of the default process group.

    Every rank calls ``synth(xs, ilens)`` with the same full batch; each computes its shard with
    ``run_local(xs_shard, ilens_shard) -> (mels, olens)`` (normally ``model.inference_batch``) and all
    ranks return the complete, ordered result."""

    def __init__(self, run_local, group=None):
        self.run_local = run_local
        self.group = group

    def __call__(self, xs, ilens, **kw):
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        il = torch.as_tensor(ilens).to("cpu", torch.int64)
        parts = shard_indices(il.tolist(), world)
        mine = parts[rank]
        sel = torch.as_tensor(mine, dtype=torch.int64)
        il_loc = il[sel]
        xs_loc = xs[sel.to(xs.device)][:, : int(il_loc.max())] if len(mine) else xs[:0]
        kw_loc = {k: (v[sel.to(v.device)][:, : xs_loc.shape[1]] if torch.is_tensor(v) else v) for k, v in kw.items()}
        mel, olens = self.run_local(xs_loc, il_loc, **kw_loc)
        if world == 1:
            return mel, torch.as_tensor(olens)
        return gather_mels(mel, olens, mine, xs.shape[0], self.group)
