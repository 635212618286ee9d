"""Utterance sharding across the GPUs of one node + the single collective of the path.

The reference has no distributed code at all.  Utterances never interact
(per-utterance semantics), weights are small (~130 MB) and replicated, so the path shards by independent
units: every rank holds the (tiny) id batch, takes a cost-balanced subset of the utterances, runs the
single-GPU path on it, and the only exchange is an all-gather of the final mels over RCCL/xGMI (backend
"nccl" on ROCm; "gloo" in the CPU tests).  The gather and the order restoration are exact; an utterance's values do not
depend on its batch-mates, and -- every rank naming the whole batch as the basis of its kernel-variant choice (``regime``) -- not on
how the batch was split either: sharded == unsharded bit for bit (DESIGN.md sections 1 and 5).
"""
import contextlib

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


def meta_rows(bmax, odim):
    """Rows of a [*, odim] float32 pack that hold ``bmax`` int64 frame counts (bit-cast).  The tail must be a whole number of
    8-byte words for the int64 view, so with an odd ``odim`` the row count is made even."""
    rows = -(-(8 * int(bmax)) // (4 * int(odim)))
    return rows + ((rows * int(odim)) & 1)


def row_capacity(n_utt, total_frames):
    """fs2_row_capacity (include/fs2.h) for a shard of ``n_utt`` utterances: rows of a capacity pack."""
    import ctypes as C
    from . import _lib
    return int(_lib.lib().fs2_row_capacity(C.byref(_lib.Batch(max(int(n_utt), 1), 1, None, 0, 0)), int(total_frames)))


_TABLE_CACHE = {}


def _scatter_table(parts, bmax, total, dev):
    """Host-known scatter table of :func:`gather_shards`: slot (r, j) -> global utterance index, holes -> a dummy slot
    ``total``; on the device, int64 [world * bmax].  Built in pinned memory and copied without blocking the host (a pageable
    copy would wait for the stream, i.e. for the forward just enqueued); the last table is cached (a server re-uses bucketed
    batch shapes, the benchmark repeats one batch)."""
    key = (tuple(tuple(p) for p in parts), bmax, str(dev))
    hit = _TABLE_CACHE.get("last")
    if hit is not None and hit[0] == key:
        return hit[1]
    gi = torch.full((len(parts), bmax), total, dtype=torch.int64)
    for r, p in enumerate(parts):
        if p:
            gi[r, : len(p)] = torch.as_tensor(p, dtype=torch.int64)
    gi = gi.reshape(-1)
    if dev.type == "cuda":
        gi = gi.pin_memory().to(dev, non_blocking=True)
    _TABLE_CACHE["last"] = (key, gi)
    return gi


def gather_shards(packed_cap, olens_dev, parts, Lout, cap=None, group=None, send_buf=None, unpack=True):
    """The ONE collective of the sharded path, sync-free (for ``inference_batch(sync=False, packed=True)``): nothing is read
    back to the host.

    ``parts`` = ``shard_indices(...)`` for the whole batch (a list of global utterance indices per rank, any sizes, possibly
    empty; identical on every rank, which all hold the full id batch).  Every rank passes its pack ([rows, odim], valid frames
    first; at most ``cap`` rows, ``cap`` being the same number on every rank - the largest shard's row capacity; a rank
    without utterances may pass 0 rows) and its own frame counts as a device int64 tensor [len(parts[rank])].  The
    frame counts ride in the tail rows of the pack (bit-cast int64 -> float32), so ONE equal-count all_gather_into_tensor
    (RCCL over xGMI) moves everything; the global indices are already known to every rank.  Offsets are computed on the
    device, one unpack kernel scatters the received rows into the ordered, zero-padded result.  ``Lout`` = padded length of
    the result (>= the longest utterance of any rank, e.g. the agreed per-utterance capacity).
    ``send_buf``: the [cap + meta_rows(bmax, odim), odim] buffer whose leading rows ARE ``packed_cap`` (the model wrote its pack
    straight into it: no staging copy); otherwise one is allocated and the pack copied in.
    Returns (mels [total, Lout, odim] in global utterance order, olens [total] int64 on the device); with ``unpack=False`` the
    padded result is not built: returns ``(recv, starts, olens)`` = the gathered packs [world * (cap + meta rows), odim], the first row
    of every utterance inside them (device int32 [total], global utterance order) and the frame counts -- for consumers that read
    frames in place (a vocoder that takes packed frames) and skip the [total, Lout, odim] tensor (c5: 0.2 GB valid in 0.6 GB)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = packed_cap.device
    rows, odim = packed_cap.shape
    cap = rows if cap is None else int(cap)
    if rows > cap:
        raise ValueError("pack of %d rows exceeds the agreed capacity %d" % (rows, cap))
    counts = [len(p) for p in parts]
    total, bmax = sum(counts), max(max(counts), 1)
    b = counts[rank]
    if olens_dev.numel() != b:
        raise ValueError("rank %d holds %d utterances but passed %d frame counts" % (rank, b, olens_dev.numel()))
    mrows = meta_rows(bmax, odim)
    if (cap * odim) % 2:
        raise ValueError("the frame counts ride behind cap * odim floats as int64: cap * odim = %d * %d must be even" % (cap, odim))
    if send_buf is not None and send_buf.shape[0] == cap + mrows and send_buf.data_ptr() == packed_cap.data_ptr():
        send = send_buf                                      # rows beyond the valid frames are never read (lengths are clamped)
        send[cap:].zero_()                                   # the buffer came from torch.empty: unused count slots = 0
    else:
        send = packed_cap.new_zeros(cap + mrows, odim)
        send[:rows].copy_(packed_cap)
    if b:
        send[cap:].view(-1).view(torch.int64)[:b].copy_(olens_dev)
    recv = packed_cap.new_empty(world * (cap + mrows), odim)
    dist.all_gather_into_tensor(recv, send, group=group)
    ol = recv.view(world, cap + mrows, odim)[:, cap:].reshape(world, -1).view(torch.int64)[:, :bmax]       # [world, bmax], holes = 0
    base = torch.arange(world, device=dev).unsqueeze(1) * (cap + mrows)
    starts = torch.cumsum(ol, 1) - ol + base
    # a rank whose capacities overflowed ships a NaN-filled pack with its true frame counts: never index past its region
    ol = torch.minimum(ol, (base + cap - starts).clamp(min=0))
    gi = _scatter_table(parts, bmax, total, dev)
    starts_g = torch.zeros(total + 1, dtype=torch.int32, device=dev).scatter_(0, gi, starts.reshape(-1).to(torch.int32))[:total]
    lens_g = torch.zeros(total + 1, dtype=torch.int32, device=dev).scatter_(0, gi, ol.reshape(-1).to(torch.int32))[:total]
    if not unpack:
        return recv, starts_g, lens_g.to(torch.int64)
    if packed_cap.is_cuda:
        import ctypes as C
        from . import _lib
        out = torch.empty(total, Lout, odim, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().fs2_op_unpack_rows_dev(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), recv.data_ptr(), odim, total,
                                                         starts_g.contiguous().data_ptr(), lens_g.contiguous().data_ptr(), Lout, out.data_ptr()))
    else:       # CPU (gloo tests): same arithmetic in plain torch
        pos = torch.arange(Lout).unsqueeze(0)
        idx = (starts_g.long().unsqueeze(1) + pos).clamp(max=recv.shape[0] - 1)
        out = torch.where((pos < lens_g.long().unsqueeze(1)).unsqueeze(-1), recv[idx], torch.zeros((), dtype=recv.dtype))
    return out, lens_g.to(torch.int64)


class ShardedSynthesizer:
    """Free-running batched synthesis over all ranks of a process group (BASELINE config c5: "batch sharded 8 x MI355X, RCCL
    all-gather of the mels over xGMI").

    Every rank calls ``synth(xs, ilens)`` with the same full batch; the utterances are dealt to the ranks longest-cost-first
    (``shard_indices``: shards of unequal size, possibly empty), each rank runs the single-GPU path on its shard and all
    ranks return the complete, ordered result.

    * ``ShardedSynthesizer(model)``: the production form.  The first call is synchronous (it teaches the capacity predictor
      the frames-per-phoneme ratio and agrees on it between the ranks with one all-reduce); later calls never wait for the
      GPU: device-driven layout inside capacities every rank derives from the SAME global numbers, one collective
      (:func:`gather_shards`).  Returns (mels [B, Lcap, odim], olens [B] device int64); ``ok()`` afterwards tells whether the
      capacities sufficed on every rank (one small all-reduce + host sync; on overflow the mels are NaN-filled).
    * ``ShardedSynthesizer(run_local)`` with a callable ``run_local(xs_shard, ilens_shard) -> (mels, olens)``: generic form
      (host-driven gather of ragged padded batches, :func:`gather_mels`)."""

    def __init__(self, model_or_fn, group=None, overlap=False, global_regime=True):
        """``overlap=True`` (throughput mode): the collective and the unpack of a sync-free call run on a side stream, so the
        all-gather of batch i over xGMI overlaps the forward of batch i+1; the returned tensors then belong to that stream:
        call ``wait()`` (or synchronize the device) before touching them on the current stream.

        ``global_regime=True`` (default): every rank names the WHOLE batch's size as the basis of its kernel-variant choice
        (``inference_batch(regime=(phonemes, utterances))``), so each shard is computed by exactly the kernels the one-GPU run of
        the whole batch uses and the gathered result is bit-identical to that run (SURVEY.md section 8e's criterion).  ``False``:
        each shard picks the variants of its own size -- results then agree with the one-GPU run to ~2e-5, not bit for bit."""
        self.model = model_or_fn if hasattr(model_or_fn, "inference_batch") else None
        self.run_local = None if self.model is not None else model_or_fn
        self.group = group
        self.overlap = bool(overlap)
        self.global_regime = bool(global_regime)
        self._comm = None
        self._ratio = None          # (mean, max) frames per phoneme agreed between the ranks
        self._last = None           # the most recent sync-free call's AsyncMels (introspection; ok() is cumulative)

    def _world(self):
        """(world size, rank, collective in use).  With an initialised process group the collective runs even at world size
        1 (what `FS2_FORCE_DIST=1 python bench.py` and the single-GPU nccl test exercise); without one the same data path
        runs minus the all-gather."""
        on = dist.is_available() and dist.is_initialized()
        return (dist.get_world_size(self.group), dist.get_rank(self.group), True) if on else (1, 0, False)

    def capacities(self, ilens, parts, alpha=1.0):
        """(total frames, per-utterance frames) reserved on every rank for this batch: the largest shard's prediction, the
        frames-per-phoneme ratios scaled by the duration scale ``alpha`` exactly as ``model.predict_capacity`` does."""
        a = float(alpha)
        extra = 0.5 if a != 1.0 else 0.0      # round(d * alpha) can add half a frame per phoneme
        mean_r, max_r = self._ratio[0] * a + extra, self._ratio[1] * a + extra
        il = torch.as_tensor(ilens).to("cpu", torch.int64)
        tok = max(int(il[torch.as_tensor(p, dtype=torch.int64)].sum()) if p else 0 for p in parts)
        nmax = max(max(len(p) for p in parts), 1)
        total = int(tok * mean_r * 1.15) + 64 * nmax
        Lcap = -(-int(float(il.max()) * max_r * 1.25 + 64) // 32) * 32
        return max(total, Lcap), Lcap

    def __call__(self, xs, ilens, sync=False, packed=False, **kw):
        """``packed=True`` (sync-free calls with a collective only): skip the padded [B, Lcap, odim] result and return
        ``(recv, starts, olens)`` as :func:`gather_shards` does with ``unpack=False``."""
        if "alpha" in kw and not float(kw["alpha"]) > 0.0:
            raise ValueError("alpha must be > 0 (reference length_regulator.py:57), got %r" % (kw["alpha"],))
        world, rank, coll = self._world()
        skip_unpack = bool(packed)
        self._dev = xs.device
        il = torch.as_tensor(ilens).to("cpu", torch.int64)
        parts = shard_indices(il.tolist(), world)
        mine = parts[rank]
        sel = torch.as_tensor(mine, dtype=torch.int64)
        il_loc = il[sel]
        # (with model.overlap_encoder the shard's ids are cut out on the encoder's side stream: on the current stream that gather would queue behind
        #  the previous call's frame-level kernels, and the side stream does not wait for the current one)
        in_stream = self.model.input_stream(xs.device) if (self.model is not None and xs.is_cuda and self._ratio is not None and not sync) else None
        with (torch.cuda.stream(in_stream) if in_stream is not None else contextlib.nullcontext()):
            xs_loc = xs[sel.to(xs.device)][:, : int(il_loc.max())] if len(mine) else xs[:0]
            kw_loc = {k: (v[sel.to(v.device)][:, : xs_loc.shape[1]] if torch.is_tensor(v) else v) for k, v in kw.items()}
        if self.model is None:
            if len(mine):
                mel, olens = self.run_local(xs_loc, il_loc, **kw_loc)
            else:       # a rank without utterances (fewer utterances than ranks) still takes part in the collective
                mel, olens = xs.new_zeros((0, 1, 1), dtype=torch.float32), torch.zeros(0, dtype=torch.int64)
            if not coll:
                return mel, torch.as_tensor(olens)
            # the feature width is only known to ranks that ran something: agree on it (all ranks take part)
            odim = torch.tensor([mel.shape[-1] if len(mine) else 0], dtype=torch.int64, device=xs.device)
            dist.all_reduce(odim, op=dist.ReduceOp.MAX, group=self.group)
            if not len(mine):
                mel = mel.new_zeros(0, 1, int(odim))
            return gather_mels(mel, olens, mine, xs.shape[0], self.group)
        model = self.model
        if self.global_regime:
            kw_loc = dict(kw_loc, regime=(int(il.sum()), int(il.numel())))
        if self._ratio is None or sync:
            # synchronous pass: exact sizes on every rank (host-driven layout), learn and agree on the ratio
            if len(mine):
                packed, olens = model.inference_batch(xs_loc, il_loc, packed=True, **kw_loc)
                r = model._frames_per_token
                local = torch.tensor([r[0], r[1]], dtype=torch.float64, device=xs.device)
            else:
                packed, olens = xs.new_zeros((0, model.odim), dtype=torch.float32), torch.zeros(0, dtype=torch.int64)
                local = torch.zeros(2, dtype=torch.float64, device=xs.device)
            if coll:
                dist.all_reduce(local, op=dist.ReduceOp.MAX, group=self.group)
            self._ratio = (float(local[0]), float(local[1]))
            if not coll:
                L = int(olens.max()) if olens.numel() else 1
                st = torch.cumsum(olens, 0) - olens
                return unpack_rows(packed, st.tolist(), olens.tolist(), L), olens.to(xs.device)
            mel, ol = gather_packed(packed, olens, mine, xs.shape[0], self.group)
            return mel, ol.to(xs.device)
        total, Lcap = self.capacities(il, parts, kw.get("alpha", 1.0))
        # (capacities count decoder frames; the pack and the padded result hold mel frames: reduction_factor per decoder frame)
        rf = int(getattr(model, "reduction_factor", 1))
        cap = max(row_capacity(len(p), total) for p in parts) * rf
        send_buf = None
        if len(mine):
            if coll and xs.is_cuda:      # the model writes its pack straight into the send buffer of the all-gather
                bmax = max(max(len(p) for p in parts), 1)
                send_buf = torch.empty(cap + meta_rows(bmax, model.odim), model.odim, dtype=torch.float32, device=xs.device)
                kw_loc = dict(kw_loc, packed_out=send_buf)
            res = model.inference_batch(xs_loc, il_loc, packed=True, sync=False, capacity=(total, Lcap), **kw_loc)
            packed, olens_dev = res
            self._last = res
        else:       # no utterance for this rank: it only takes part in the collective
            packed, olens_dev = xs.new_zeros((0, model.odim), dtype=torch.float32), torch.zeros(0, dtype=torch.int64, device=xs.device)
            self._last = None
        if not coll:
            # same data path without the collective: offsets on the device, one unpack kernel
            return _unpack_local(packed, olens_dev, Lcap * rf), olens_dev
        if self.overlap and packed.is_cuda:
            cur = torch.cuda.current_stream(xs.device)
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=xs.device)
            self._comm.wait_stream(cur)                       # the pack of this batch is complete
            with torch.cuda.stream(self._comm):
                out = gather_shards(packed, olens_dev, parts, Lcap * rf, cap=cap, group=self.group, send_buf=send_buf, unpack=not skip_unpack)
            packed.record_stream(self._comm)                  # allocated on the compute stream, read by the side stream
            olens_dev.record_stream(self._comm)
            return out
        return gather_shards(packed, olens_dev, parts, Lcap * rf, cap=cap, group=self.group, send_buf=send_buf, unpack=not skip_unpack)

    def wait(self):
        """Make the current stream wait for the gathers issued in overlap mode (no host synchronisation)."""
        if self._comm is not None:
            torch.cuda.current_stream(self._dev).wait_stream(self._comm)

    def ok(self):
        """True if the capacities of EVERY sync-free call since the previous ``ok()`` sufficed on EVERY rank (overlap mode keeps
        several calls in flight: an overflow of any of them returned NaN-filled mels).  Waits for the GPU; collective."""
        _, _, coll = self._world()
        good = 1 if (self.model is None or self.model.async_ok()) else 0      # cumulative: drains every call in flight
        if coll:
            t = torch.tensor([good], dtype=torch.int32, device=self._dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            good = int(t)
        return bool(good)


def _unpack_local(packed_cap, olens_dev, Lout):
    """[rows_cap, odim] pack with device frame counts -> [b, Lout, odim] zero padded, no host read-back."""
    import ctypes as C
    from . import _lib
    b, odim, dev = olens_dev.numel(), packed_cap.shape[1], packed_cap.device
    st64 = torch.cumsum(olens_dev, 0) - olens_dev
    starts = st64.to(torch.int32).contiguous()
    lens = torch.minimum(olens_dev, (packed_cap.shape[0] - st64).clamp(min=0)).to(torch.int32).contiguous()     # (overflowed call: stay inside the pack)
    out = torch.empty(b, Lout, odim, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fs2_op_unpack_rows_dev(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), packed_cap.data_ptr(), odim, b,
                                                     starts.data_ptr(), lens.data_ptr(), Lout, out.data_ptr()))
    return out
