"""CPU, world_size 2, gloo: the N>1 path (cost-balanced utterance sharding + all-gather of ragged mels +
order restoration) with a stand-in per-utterance compute function.  The exact same code runs over
RCCL/xGMI on the GPU box (backend "nccl"); sharded == unsharded must hold bit-for-bit."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_run_local(xs, ilens, **kw):
    """Deterministic per-utterance 'mel': depends only on that utterance's ids (like the real path)."""
    B = xs.shape[0]
    ol = torch.tensor([int(xs[b, : int(ilens[b])].sum() % 37) + 5 for b in range(B)], dtype=torch.int64)
    Lmax = int(ol.max()) if B else 1
    mel = torch.zeros(B, Lmax, 8)
    for b in range(B):
        L = int(ol[b])
        base = xs[b, : int(ilens[b])].float().mean()
        mel[b, :L] = base + torch.arange(L * 8, dtype=torch.float32).view(L, 8) * 0.01
    return mel, ol


def _make_inputs():
    g = torch.Generator().manual_seed(7)
    B = 11
    il = torch.randint(3, 40, (B,), generator=g)
    xs = torch.zeros(B, int(il.max()), dtype=torch.int64)
    for b in range(B):
        xs[b, : il[b]] = torch.randint(1, 68, (int(il[b]),), generator=g)
    return xs, il


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastspeech2_amd.parallel import ShardedSynthesizer
    xs, il = _make_inputs()
    mel, ol = ShardedSynthesizer(_fake_run_local)(xs, il)
    # packed form (what bench.py ships over RCCL): same result
    from fastspeech2_amd.parallel import shard_indices, gather_packed
    mine = shard_indices(il.tolist(), world)[rank]
    sel = torch.as_tensor(mine)
    m_loc, ol_loc = _fake_run_local(xs[sel][:, : int(il[sel].max())], il[sel])
    packed = torch.cat([m_loc[i, : int(ol_loc[i])] for i in range(len(mine))])
    mel2, ol2 = gather_packed(packed, ol_loc, mine, xs.shape[0])
    assert torch.equal(ol2, ol) and torch.equal(mel2, mel[:, : mel2.shape[1]])
    # sync-free form (capacity packs of equal size, counts and indices as "device" tensors, equal utterance count per rank):
    # utterances 0..9 split evenly, rank r takes r, r+2, ...
    from fastspeech2_amd.parallel import gather_packed_async
    ev = list(range(rank, 10, world))
    sel = torch.as_tensor(ev)
    m_loc, ol_loc = _fake_run_local(xs[sel][:, : int(il[sel].max())], il[sel])
    cap, Lout = 400, 48
    pk = torch.full((cap, 8), float("nan"))                   # rows beyond the valid frames are never read
    valid = torch.cat([m_loc[i, : int(ol_loc[i])] for i in range(len(ev))])
    pk[: valid.shape[0]] = valid
    mel3, ol3 = gather_packed_async(pk, ol_loc, sel, 10, Lout)
    assert torch.equal(ol3, ol[:10]) and torch.equal(mel3[:, : mel.shape[1]], mel[:10]) and float(mel3[:, mel.shape[1]:].abs().sum()) == 0.0
    q.put((rank, mel, ol))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded_gloo():
    from fastspeech2_amd.parallel import shard_indices
    xs, il = _make_inputs()
    want_mel, want_ol = _fake_run_local(xs, il)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mel, ol in got:
        assert torch.equal(ol, want_ol)
        assert mel.shape[0] == xs.shape[0]
        L = want_mel.shape[1]
        assert torch.equal(mel[:, :L], want_mel) and float(mel[:, L:].abs().sum()) == 0.0
    parts = shard_indices(il.tolist(), 2)
    assert sorted(parts[0] + parts[1]) == list(range(xs.shape[0])) and parts[0] and parts[1]


def test_shard_balance():
    from fastspeech2_amd.parallel import shard_indices, utterance_cost
    il = [16 + (i * 37) % 160 for i in range(128)]
    parts = shard_indices(il, 8)
    loads = [sum(utterance_cost(il[i]) for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) < 1.05      # LPT keeps the heaviest rank within 5 % of the mean
