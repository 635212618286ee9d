"""One process per GPU over RCCL (backend "nccl"): the sharded path of SURVEY.md section 8e on real devices.

world_size 1 runs on the one-GPU box (it exercises exactly this worker code and the collective at world size 1); world_size 2 and
8 skip unless that many GPUs are visible, so the first `pytest -m gpu` on a multi-GPU node also is the first time `gather_shards`
sees RCCL with more than one rank.  Criterion: the gathered result is bit-identical to the 1-GPU result in the same
precision mode -- every rank also runs the whole batch unsharded and compares.  That holds because every rank names the WHOLE batch as the
basis of its kernel-variant choice (`ShardedSynthesizer(global_regime=True)`, include/fs2.h: fs2_batch.regime_*): these very shapes (B = 19
over 2 ranks, B = 67 over 8) put the batch and its shards on different sides of the kernel thresholds, and
tests/test_gpu_parity.py::test_shards_named_after_the_whole_batch_equal_the_one_gpu_call proves the identity for them on ONE GPU (ranks played in
turn), so that a red result here can only come from the collective.  (The file sorts behind the operator and parity tests on purpose: under
`pytest -x` nothing multi-GPU can hide them.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
        from fastspeech2_amd.parallel import ShardedSynthesizer, shard_indices
        from fastspeech2_amd.synthetic import portable_state_dict, make_batch
        dev = torch.device("cuda:%d" % rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=dev)
        hp = default_hparams()
        model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
        model.load_state_dict(portable_state_dict(model.state_dict(), seed=0))
        model = model.to(dev)
        model.precision = "bf16x3"
        b = make_batch("c5", B=8 * world + 3)                      # not a multiple of the world size: unequal shards
        xs, il, ds = b["xs"].to(dev), b["ilens"], b["ds"].to(dev)
        parts = shard_indices(il.tolist(), world)
        with torch.no_grad():
            ref, ol = model.inference_batch(xs, il, d_override=ds)             # the whole batch on this GPU alone
            L = ref.shape[1]
            synth = ShardedSynthesizer(model)
            m1, o1 = synth(xs, il, d_override=ds)                  # synchronous first call (host-driven gather)
            m2, o2 = synth(xs, il, d_override=ds)                  # sync-free: device layout + ONE all_gather_into_tensor
            assert synth.ok()
            assert torch.equal(o1.cpu(), ol) and torch.equal(o2.cpu(), ol)
            assert torch.equal(m1[:, :L], ref), "rank %d: synchronous sharded != unsharded" % rank
            assert torch.equal(m2[:, :L], ref) and float(m2[:, L:].abs().sum()) == 0.0, "rank %d: sync-free sharded != unsharded" % rank
            recv, starts, o3 = synth(xs, il, d_override=ds, packed=True)       # packed return form: frames read in place
            assert synth.ok() and torch.equal(o3.cpu(), ol)
            for g in (0, xs.shape[0] // 2, xs.shape[0] - 1):
                s0, n = int(starts[g]), int(ol[g])
                assert torch.equal(recv[s0:s0 + n], ref[g, :n])
            over = ShardedSynthesizer(model, overlap=True)         # throughput mode: gather + unpack on a side stream
            over._ratio = synth._ratio
            outs = [over(xs, il, d_override=ds) for _ in range(3)]
            over.wait()
            assert over.ok()
            for m, o in outs:
                assert torch.equal(o.cpu(), ol) and torch.equal(m[:, :L], ref)
            # capacities far too small (the same on every rank: the all-gather is equal-count): reported by the cumulative ok()
            # even after a later call, and the overflowed packs are NaN-filled
            over._ratio = (0.05, 0.05)
            m4, _ = over(xs, il, d_override=ds)
            over(xs, il, d_override=ds)
            over.wait()
            assert not over.ok()
            assert torch.isnan(m4).any()
        q.put((rank, "ok", [len(p) for p in parts]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        q.put((rank, "error: %s\n%s" % (e, traceback.format_exc()), None))


@pytest.mark.parametrize("world", [1, 2, 8])
def test_sharded_equals_unsharded_over_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, torch.cuda.device_count()))
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(world):
            got.append(q.get(timeout=240))
            if got[-1][1] != "ok":
                break                       # the other ranks may be stuck in a collective with the failed one: do not wait for them
    finally:
        for p in procs:
            p.join(timeout=60 if len(got) == world and all(g[1] == "ok" for g in got) else 1)
            if p.is_alive():
                p.terminate()
    bad = [g for g in got if g[1] != "ok"]
    assert not bad and len(got) == world, bad
    sizes = got[0][2]
    assert sum(sizes) == 8 * world + 3 and (world == 1 or len(set(sizes)) > 1)
    assert all(p.exitcode == 0 for p in procs)
