"""A stand-in for `torch.distributed` that plays the W ranks of one node IN TURN inside one process, for the sync-free path of
`fastspeech2_amd.parallel.ShardedSynthesizer` (one `all_gather_into_tensor` per call and rank): rank r's call deposits its send buffer, and the
LAST rank's call finds all W of them in its receive buffer -- exactly what the collective hands every rank on a node.  Lets a one-GPU box run
the whole sharded data path (LPT shards, capacities, packs written straight into the send buffers, `gather_shards`' offset arithmetic,
`fs2_op_unpack_rows_dev`) on real kernels; the collective itself is covered over gloo (tests/test_parallel_gloo.py) and nccl (tests/test_gpu_zmulti.py)."""
import torch.distributed as _dist


def make(world):
    state = {"rank": 0, "sends": {}}

    class FakeDist:
        ReduceOp = _dist.ReduceOp
        is_available = staticmethod(lambda: True)
        is_initialized = staticmethod(lambda: True)
        get_world_size = staticmethod(lambda group=None: world)
        get_rank = staticmethod(lambda group=None: state["rank"])

        @staticmethod
        def all_reduce(t, op=None, group=None):
            return None

        @staticmethod
        def all_gather_into_tensor(recv, send, group=None):
            state["sends"][state["rank"]] = send.clone()
            recv.zero_()
            if len(state["sends"]) == world:
                rv = recv.view(world, send.shape[0], send.shape[1])
                for q in range(world):
                    rv[q].copy_(state["sends"][q])

    return FakeDist, state


def run_all_ranks(P, monkeypatch, model, world, xs, il, ratio, **kw):
    """Every rank's sync-free `ShardedSynthesizer` call in turn (frames-per-phoneme ratio given, as agreed by an earlier synchronous call);
    returns the LAST rank's result -- (mels [B, Lcap, odim] in global order, olens device int64) -- which saw all W packs."""
    FakeDist, state = make(world)
    monkeypatch.setattr(P, "dist", FakeDist)
    try:
        synth = P.ShardedSynthesizer(model)
        synth._ratio = ratio
        out = None
        for r in range(world):
            state["rank"] = r
            out = synth(xs, il, packed=(r < world - 1), **kw)      # the last "rank" also scatters into the padded result
    finally:
        monkeypatch.undo()
    return out
