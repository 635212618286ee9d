#!/usr/bin/env python3
"""Benchmark of the MI355X FastSpeech2 mel-generation path (BASELINE.json metric: mel-frames/sec).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...,
   or as the plain command above: without WORLD_SIZE in the environment the script re-launches itself under torch.distributed.run
   with N ranks on 127.0.0.1 and a free port; rank 0 prints the one JSON line either way)

One "step" = one full free-running forward of the hot path over one synthetic batch already resident in
HBM: phoneme ids -> encoder -> duration predictor -> length regulator -> pitch/energy -> decoder -> mel
projection -> Postnet -> padded mels on one GPU; for N > 1 the valid frames leave as a pack and ONE RCCL all-gather over xGMI returns
every rank the gathered packs + offsets (`config.gather` = "packed", the default since round 4) or, with --padded, the ordered padded
[B, Lcap, odim] tensor ("padded", what rounds 1-3 measured): lines of different `gather` forms do not compare.

Workloads (BASELINE.json configs, fastspeech2_amd/synthetic.py):
  N = 1 : c3, "batch=64 LJSpeech-shape" -- the config the >= 50x-CPU target is quoted on.
  N > 1 : c5, "batch=1024 sharded 8 x MI355X, RCCL all-gather of the mels": ONE global batch of 128 * N utterances (N = 8: the
          whole of c5), dealt to the ranks by the cost model (LPT, `shard_indices`: shards of unequal size), every rank
          synthesises its shard without ever waiting for the host and ONE all-gather returns all mels, in order, to every
          rank (`ShardedSynthesizer`).  Weak scaling: the work per GPU is constant in N.
Random-init default model (portable generator, seed 0) with the duration bias set so that predicted durations are
LJSpeech-like; there is no network for real checkpoints or datasets.
value = valid mel frames of the global batch / max-over-ranks wall time of exactly K steps.
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "mix_f16x2": 2500.0, "mix_f16x1": 2500.0, "mix_mx": 2500.0, "mix_mx4": 2500.0}   # MI355X_MICROARCH.md dense MFMA peaks
DTYPE_NAME = {"fp32": "f32", "bf16x3": "bf16x3", "bf16": "bf16", "mix_f16x2": "bf16x3 (FFN conv: f16x2)", "mix_f16x1": "bf16x3 (FFN conv: f16)",
              "mix_mx": "bf16x3 (FFN conv: f16 x f16 + block-scaled e4m3 cross terms)",
              "mix_mx4": "bf16x3 (FFN conv: f16 x f16 + block-scaled e4m3 cross terms; the decoder's: block-scaled e2m1 cross terms, one E8M0 scale per 16-channel block on both operands)"}
MFMA_PER_PRODUCT = {"bf16x3": 3, "mix_f16x2": 2, "mix_f16x1": 1, "mix_mx": 2.0, "mix_mx4": 1.5}       # MFMAs issued per algorithmic product in the dominant kernel (the FFN conv)
HBM_PEAK_GBPS = 8000.0
WORKLOAD_TEXT = {
    "c1": "c1: 1 utterance, 80 phonemes",
    "c2": "c2: batch=16 synthetic phoneme seqs len 64-128",
    "c3": "c3: batch=64 LJSpeech-shape",
    "c4": "c4: batch=256 mixed-length (32-512 phonemes) with Postnet, length-regulator stress",
    "c5": "c5: batch=1024 LJSpeech-shape sharded over 8 GPUs (128 per GPU)",
}


def csrc_sha16():
    """Fingerprint of the kernel sources (what profiles/r06_traffic.json was measured on)."""
    import hashlib
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "fastspeech2_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def cpu_baseline(sd, cfg, batch, gpu, budget_s=14.0):
    """Times the CPU oracle (the validated port of the reference's fp32 PyTorch path) on the host cores, on a bounded
    sample of the same batch: (i) per-utterance loop of B=1 calls, (ii) one padded batch of 16 utterances; returns the faster
    in mel-frames/s, the measured mel max-abs diff GPU vs oracle (durations forced to the GPU's, so that frames align), and the
    free-running decision statistics (SURVEY.md hard part 2): how often the data-dependent integer decisions of the GPU path
    -- durations clamp(round(exp(x) - 1)), frame counts, pitch / energy bucket indices -- equal the CPU path's."""
    from oracle import fs2_oracle as O
    xs, il, ds = batch["xs"], batch["ilens"], gpu["d_int"]
    B = xs.shape[0]
    # Pick the thread count that is fastest for this op mix (small GEMM/conv calls: all logical cores of a
    # 2-socket host oversubscribe badly), so that the CPU side is not handicapped.
    # (picked from the MEDIAN of three repeats on the batch's MEDIAN-length utterance -- one timing of the first, short utterance
    #  gave 32 threads on one box and 16 on the next, 11.7 k against 21.5 k frames/s)
    im = int(torch.argsort(il)[B // 2])
    Tm_ = int(il[im])
    one = lambda: O.padded_forward(sd, cfg, xs[im:im + 1, :Tm_], il[im:im + 1], is_inference=True, d_override=ds[im:im + 1, :Tm_])
    ncpu = os.cpu_count() or 1
    best_t, best_dt, tuning = 1, float("inf"), {}
    for nt in sorted({t for t in (4, 8, 16, 32, 64, ncpu // 2, ncpu) if 1 <= t <= ncpu}):
        torch.set_num_threads(nt)
        one()
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            one()
            reps.append(time.perf_counter() - t0)
        dt_ = statistics.median(reps)
        tuning[nt] = round(1e3 * dt_, 1)
        if dt_ < best_dt:
            best_t, best_dt = nt, dt_
        if dt_ > 4 * best_dt:
            break
    # ... and the two best candidates are then held against each other on what is actually timed below -- a slice of the per-utterance loop (8 utterances of mixed
    # length) -- because the ranking of one utterance does not always carry over (one box of round 6: 32 threads won the single utterance and ran the loop at half
    # the rate of 16)
    cands = sorted(tuning, key=lambda t_: tuning[t_])[:2]
    if len(cands) == 2:
        slice_ms = {}
        for nt in cands:
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            for b_ in range(0, B, max(B // 8, 1)):
                T_ = int(il[b_])
                O.padded_forward(sd, cfg, xs[b_:b_ + 1, :T_], il[b_:b_ + 1], is_inference=True, d_override=ds[b_:b_ + 1, :T_])
            slice_ms[nt] = round(1e3 * (time.perf_counter() - t0), 1)
        best_t = min(slice_ms, key=lambda t_: slice_ms[t_])
        tuning = dict(single_utterance_ms=tuning, loop_slice_ms=slice_ms)
    torch.set_num_threads(best_t)
    frames, t0, n, worst = 0, time.perf_counter(), 0, 0.0
    for b in range(B):
        T = int(il[b])
        o = O.padded_forward(sd, cfg, xs[b:b + 1, :T], il[b:b + 1], is_inference=True, d_override=ds[b:b + 1, :T])
        L = int(o["olens"][0])
        frames += L
        n += 1
        worst = max(worst, float((o["after"][0] - gpu["after"][b, :L]).abs().max()))
        if time.perf_counter() - t0 > budget_s:
            break
    per_utt = frames / (time.perf_counter() - t0)
    nb = min(16, B)
    Tm = int(il[:nb].max())
    t1 = time.perf_counter()
    o = O.padded_forward(sd, cfg, xs[:nb, :Tm], il[:nb], is_inference=True, d_override=ds[:nb, :Tm])
    padded = int(o["olens"].sum()) / (time.perf_counter() - t1)
    best = max(per_utt, padded)
    # free-running: nothing forced; the oracle makes its own decisions
    t2 = time.perf_counter()
    tok = tok_eq = utt = utt_eq = fr = qe_eq = qp_eq = 0
    for b in range(B):
        T = int(il[b])
        o = O.padded_forward(sd, cfg, xs[b:b + 1, :T], il[b:b + 1], is_inference=True)
        same = o["d_outs"][0, :T] == ds[b, :T]
        tok += T
        tok_eq += int(same.sum())
        utt += 1
        utt_eq += int(int(o["olens"][0]) == int(gpu["olens"][b]))
        if bool(same.all()):          # frames align one to one: compare the bucket indices actually embedded
            L = int(o["olens"][0])
            fr += L
            qe_eq += int((o["qe"][0, :L] == gpu["qe"][b, :L]).sum())
            qp_eq += int((o["qp"][0, :L] == gpu["qp"][b, :L]).sum())
        if time.perf_counter() - t2 > 8.0:
            break
    flips = dict(utterances_compared=utt, dur_exact_rate=round(tok_eq / max(tok, 1), 6), olens_exact_rate=round(utt_eq / max(utt, 1), 6),
                 qe_agree_rate=round(qe_eq / max(fr, 1), 6), qp_agree_rate=round(qp_eq / max(fr, 1), 6), frames_compared=fr,
                 note="free-running GPU vs free-running CPU oracle; bucket indices compared on utterances whose durations all agree")
    cpu_model, phys, logical = host_cpu()
    return dict(value=round(best, 1), unit="mel-frames/s", cores=torch.get_num_threads(), threads=torch.get_num_threads(), host_physical_cores=phys,
                host_logical_cores=logical, cpu_model=cpu_model, kind="port",
                thread_tuning_ms=dict(utterance_phonemes=Tm_, median_of=3, ms_by_threads=tuning, chosen=best_t),
                cores_note="cores = threads the oracle ran on (the fastest of 4 .. all logical cores for this op mix: median of 3 repeats on the "
                           "median-length utterance), not the host's core count",
                sample="oracle (validated fp32 PyTorch port of the reference path; measured equal to the real reference within noise, "
                       "BASELINE.md section 3) on the c3 batch: per-utterance loop over the first %d utterances (%d frames) = %.0f fr/s; "
                       "one padded batch of %d = %.0f fr/s; faster quoted" % (n, frames, per_utt, nb, padded)), worst, flips


def host_cpu():
    """(model string, physical cores, logical cores) of the host the CPU baseline runs on (north_star: "core count stated")."""
    model, phys = None, set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None and cid is not None:
                phys.add((pid, cid))
                pid = cid = None
        if pid is not None and cid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or None), os.cpu_count()


def sclk_reader(local):
    """Reader of the current shader clock (MHz) from the amdgpu sysfs node of GPU `local` (pp_dpm_sclk: the level marked '*'), or None."""
    import glob
    import re
    nodes = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"), key=lambda q: int(re.search(r"card(\d+)", q).group(1)))
    if not nodes:
        return None
    path = nodes[min(local, len(nodes) - 1)]

    def read():
        try:
            for ln in open(path):
                if ln.rstrip().endswith("*"):
                    return float(re.search(r"(\d+)\s*[Mm][Hh]z", ln).group(1))
        except (OSError, AttributeError, ValueError):
            pass
        return None
    return read if read() is not None else None


def smi_sclk_start():
    """When the sysfs node reads an implausible shader clock (< 500 MHz at normal step times: the node's, not the chip's -- the driver's box in
    round 5 read 95 MHz), ask the SMI tool once WHILE the queue is full: returns a started subprocess (or None) for smi_sclk_finish()."""
    import shutil
    import subprocess
    for tool, argv in (("amd-smi", ["metric", "--clock"]), ("rocm-smi", ["--showclocks"])):
        exe = shutil.which(tool) or (os.path.join("/opt/rocm/bin", tool) if os.path.exists(os.path.join("/opt/rocm/bin", tool)) else None)
        if exe:
            try:
                return tool, subprocess.Popen([exe] + argv, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            except OSError:
                continue
    return None


def smi_sclk_finish(started, local):
    """-> (MHz of GPU `local`'s shader clock as the SMI tool printed it, tool name) or (None, None)."""
    import re
    if started is None:
        return None, None
    tool, proc = started
    try:
        out, _ = proc.communicate(timeout=20)
    except Exception:      # noqa: BLE001 -- a hung or missing tool only costs the field
        proc.kill()
        return None, None
    vals, in_gfx = [], False
    for ln in out.splitlines():
        if tool == "amd-smi":      # "GFX_3:" opens a block (one per XCD) whose "CLK: 2103 MHz" line is the current clock (MIN_CLK / MAX_CLK are limits)
            if re.match(r"\s*GFX_\d+:", ln):
                in_gfx = True
            elif in_gfx and re.match(r"\s*CLK:", ln):
                m = re.search(r"(\d+)\s*M[Hh]z", ln)
                if m:
                    vals.append(float(m.group(1)))
                in_gfx = False
        elif re.search(r"sclk", ln, re.I):      # rocm-smi: "GPU[0] : sclk clock level: 7: (2100Mhz)"
            m = re.search(r"\((\d+)\s*M[Hh]z\)", ln)
            if m:
                vals.append(float(m.group(1)))
    vals = [v for v in vals if v >= 500.0]          # (below the minimum DPM level: a sleeping XCD / the node's idle clock)
    if not vals:
        return None, None
    return (vals[min(local, len(vals) - 1)] if tool == "rocm-smi" else statistics.median(vals)), tool


def mfma_peak_record():
    """Whole-chip bf16 MFMA issue rates measured on this chip family by a committed probe (profiles/mfma_rate_probe.json <- r04_mfma_rate_probe.txt)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "mfma_rate_probe.json")))
    except (OSError, ValueError):
        return None


def dist_diagnostics(model, synth, xs, il, parts, rank, world, dev, k, ms_per_step, step_streams=None, sched_steps=0):
    """What an N-GPU line needs to be read (run AFTER the timed region, by every rank): the cost model's imbalance, each
    rank's forward alone (min / max over ranks), and the same step with the collective serialised behind the forward -- so that a scaling
    efficiency below target can be attributed to compute imbalance, to the all-gather or to neither."""
    from fastspeech2_amd.parallel import ShardedSynthesizer, utterance_cost
    costs = [sum(utterance_cost(int(il[i])) for i in p) for p in parts]
    mine = parts[rank]
    fwd = 0.0
    if mine:
        sel = torch.as_tensor(mine, dtype=torch.int64)
        il_loc = il[sel]
        xs_loc = xs[sel.to(dev)][:, : int(il_loc.max())]
        cap = synth.capacities(il, parts)
        regime = (int(il.sum()), int(il.numel())) if synth.global_regime else None      # (the kernel variants the rank's shard really runs: the whole batch's)
        run = lambda: model.inference_batch(xs_loc, il_loc, packed=True, sync=False, capacity=cap, regime=regime)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            run()
        e1.record()
        torch.cuda.synchronize()
        fwd = e0.elapsed_time(e1) / k
        # the same forwards in the TIMED region's schedule (issued on the step streams in turn, several in flight), still without a collective: the number
        # `ms_per_step` of the line is comparable with (round-5 review: against the one-stream figure the exposed collective time could come out negative)
        fwd_sched = fwd
        if step_streams is not None:
            kk = max(k, sched_steps, 4 * len(step_streams.streams))      # (as many steps as the timed region: the same share of pipeline fill and drain)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(kk):
                with step_streams.next():
                    run()
            torch.cuda.synchronize()
            fwd_sched = 1e3 * (time.perf_counter() - t0) / kk
    else:
        fwd_sched = 0.0
    t = torch.tensor([fwd, fwd_sched], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    fwd_all = [round(float(x[0]), 3) for x in every]
    fwd_sched_all = [round(float(x[1]), 3) for x in every]
    ser = ShardedSynthesizer(model, overlap=False)
    ser._ratio = synth._ratio
    ser(xs, il, packed=True)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(k):
        ser(xs, il, packed=True)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([(time.perf_counter() - t0) / k * 1e3], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    serial = float(t)
    model.async_ok()
    busiest = max(fwd_all)
    return dict(backend=dist.get_backend(), world_size=dist.get_world_size(), steps_measured=k,
                lpt_cost_imbalance_max_over_mean=round(max(costs) / (sum(costs) / len(costs)), 4),
                forward_ms_per_rank=fwd_all, forward_ms_min=min(fwd_all), forward_ms_max=busiest,
                forward_ms_per_rank_timed_schedule=fwd_sched_all, forward_ms_max_timed_schedule=max(fwd_sched_all),
                ms_per_step_serial_collective=round(serial, 3), collective_ms_exposed_serial=round(serial - busiest, 3),
                collective_ms_exposed_overlapped=round(ms_per_step - max(fwd_sched_all), 3),
                note="forward alone = this rank's shard through the sync-free single-GPU path (no collective), one step at a time on one stream; "
                     "`..._timed_schedule` = the same forwards issued as the timed region issues its steps (on the step streams in turn), still without a collective; "
                     "serial = one step at a time with the all-gather on the compute stream (FS2_DIST_SERIAL=1's form); exposed_serial = serial step - the busiest rank's "
                     "forward alone; exposed_overlapped = the line's ms_per_step - the busiest rank's forward in the SAME schedule (comparable figures)")


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def relaunch_command(argv, gpus):
    """`python bench.py --gpus N ...` without a launcher: the same command under torch.distributed.run, one rank per GPU of this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def fake_main(args, json_fd):
    """TEST HARNESS (tests/test_parallel_gloo.py, FS2_BENCH_FAKE=1): the launch / sharding / collective / timing skeleton of this script
    on CPU ranks over gloo with the stand-in model of the gloo tests -- what lets a machine without GPUs check that
    `python bench.py --gpus 2` spawns its ranks and prints ONE line.  Measures nothing."""
    from fastspeech2_amd.parallel import ShardedSynthesizer
    from tests.test_parallel_gloo import FakeModel, _make_inputs
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(free_port())
    dist.init_process_group("gloo", rank=rank, world_size=world)      # (also at world size 1: the collective path is the one under test)
    xs, il = _make_inputs(B=4 * max(world, 1) + 3)
    synth = ShardedSynthesizer(FakeModel())
    out = synth(xs, il, sync=True)
    for _ in range(args.warmup):
        out = synth(xs, il, sync=False, packed=not args.padded)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = synth(xs, il, sync=False, packed=not args.padded)
    dist.barrier()
    dt = time.perf_counter() - t0
    frames = int(torch.as_tensor(out[-1]).sum())
    # the fields of dist_diagnostics() that exist without a GPU: the same collectives in the same order, stand-in numbers
    from fastspeech2_amd.parallel import shard_indices, utterance_cost
    parts = shard_indices(il.tolist(), world)
    costs = [sum(utterance_cost(int(il[i])) for i in p) for p in parts]
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    fwd_all = [float(x) for x in every]
    diag = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), lpt_cost_imbalance_max_over_mean=round(max(costs) / (sum(costs) / len(costs)), 4),
                forward_ms_per_rank=fwd_all, forward_ms_min=min(fwd_all), forward_ms_max=max(fwd_all))
    if rank == 0:
        os.write(json_fd, (json.dumps({"metric": "mel-frames/sec", "value": round(frames * args.steps / dt, 1), "unit": "mel-frames/s", "n_gpus": world,
                                       "steps": args.steps, "warmup": args.warmup, "data": "FAKE (FS2_BENCH_FAKE test harness: gloo, stand-in model, measures nothing)",
                                       "config": {"workload": "fake", "gather": "padded" if args.padded else "packed", "collective_world_size": dist.get_world_size()},
                                       "multi_gpu": diag}) + "\n").encode())
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, help="c1..c5 (default: c3 on one GPU, c5 = 128 utterances per GPU on several)")
    ap.add_argument("--precision", default=os.environ.get("FS2_PRECISION", "mix_mx4"), help="arithmetic mode (default since round 6: mix_mx4 -- mix_mx with the decoder FFN "
                    "conv's cross terms in block-scaled fp4 where the planes-only regime holds; rounds 3-5: mix_mx; rounds 1-2: bf16x3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-kernels", action="store_true", help="print the per-kernel hipEvent table to stderr")
    ap.add_argument("--graph", action="store_true", help="replay the forward as one captured HIP graph (single GPU; the launch-bound small configs)")
    ap.add_argument("--no-overlap-encoder", action="store_true", help="keep every launch of a step on one stream (rounds 1-4; the A/B of `overlap_encoder`)")
    ap.add_argument("--streams", type=int, default=None, help="streams the steps are issued on in turn (default: 3 per GPU -- that many whole "
                    "steps in flight, the dispatcher fills one step's tail rounds and HBM bursts with another's workgroups; 1 = the schedule of rounds 1-4, "
                    "and the only one with --graph / --profile-kernels)")
    ap.add_argument("--overlap-encoder", action="store_true", help="model.overlap_encoder also with --streams > 1 (default: only with one stream)")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds the step keeps running after the timed region for `sustained_ms_per_step` (0: off)")
    ap.add_argument("--regime-utterances", type=int, default=0, help="one GPU: run the batch as a SHARD of a batch of this many utterances (its phoneme count scaled "
                    "alike): the kernel variants of the larger batch on this one, what every rank of the sharded path does (fs2_batch.regime_*); measures the cost of that choice")
    ap.add_argument("--padded", action="store_true", help="N > 1: every rank also unpacks the gathered mels into the padded [B, Lcap, odim] tensor "
                                                           "(default: the packed form, gathered packs + offsets, no unpack launch)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched as a plain command: become `torch.distributed.run ... bench.py <same arguments>` (one rank per GPU, 127.0.0.1, a free port)
        cmd = relaunch_command(sys.argv[1:], args.gpus)
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    # stdout carries exactly ONE line, the JSON record: whatever a library prints there (RCCL's version banner, ...) goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("FS2_BENCH_FAKE") == "1":
        return fake_main(args, json_fd)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("FS2_FORCE_DIST") == "1"    # FS2_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # (only the one-rank FS2_FORCE_DIST form gets here without a launcher)
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    workload = args.workload or ("c5" if use_dist else "c3")

    from fastspeech2_amd import FeedForwardTransformer, StepStreams, default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.parallel import ShardedSynthesizer, shard_indices, path_flops
    from fastspeech2_amd.synthetic import portable_state_dict, ljspeech_durations, make_batch

    hp = default_hparams()
    odim = hp.audio.num_mels
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, odim, hp).eval()
    sd = ljspeech_durations(portable_state_dict(model.state_dict(), seed=0))
    model.load_state_dict(sd)
    model = model.to(dev)
    model.precision = args.precision
    # throughput mode: every step's token-level half (encoder + duration predictor, ~30 launches that leave most of the chip idle) runs on a side
    # stream and overlaps the previous step's frame-level kernels; the batch is resident and complete, which is what the mode asks of its caller
    # (with several steps in flight -- see --streams below -- the other steps' kernels fill the same holes and the extra streams only compete for
    #  the hardware queues: profiles/r05_ab_stream_schedules.txt)
    eager = not (args.profile_kernels or args.graph)
    n_streams = args.streams if args.streams else (3 if eager else 1)
    if n_streams > 1 and not eager:
        raise SystemExit("--streams > 1 serves the eager path")
    model.overlap_encoder = eager and not args.no_overlap_encoder and (n_streams == 1 or args.overlap_encoder)

    # ONE global batch, identical on every rank (numpy RandomState, seed = config number).  c5 is cut to 128 utterances per
    # GPU when fewer than 8 GPUs run it, so the work per GPU does not depend on N (weak scaling); 8 GPUs run all 1024.
    if workload == "c5":
        batch = make_batch("c5", B=128 * world)
    else:
        batch = make_batch(workload)
    xs, il = batch["xs"].to(dev), batch["ilens"]
    B = xs.shape[0]
    parts = shard_indices(il.tolist(), world)
    mine = parts[rank]
    # throughput mode: the all-gather of step i (side stream) overlaps the forward of step i + 1; FS2_DIST_SERIAL=1 keeps everything on one stream
    synth = ShardedSynthesizer(model, overlap=os.environ.get("FS2_DIST_SERIAL") != "1") if use_dist else None

    regime = None
    if args.regime_utterances:
        if use_dist or args.graph or args.regime_utterances < B:
            raise SystemExit("--regime-utterances N >= the batch's %d utterances serves the one-GPU eager path" % B)
        regime = (int(round(float(il.sum()) * args.regime_utterances / B)), int(args.regime_utterances))

    graph_run = None
    # throughput mode: consecutive steps go to `n_streams` streams in turn, so that many whole steps are in flight and every kernel's tail round,
    # HBM burst and launch gap is filled by another step's workgroups (profiles/r05_ab_stream_schedules.txt).  Nothing in the library knows
    # about it: a call runs on the caller's current stream.  Every step still does all its work; the timed region still covers exactly K steps
    # between two full-device synchronisations.  (N > 1: the all-gathers stay on the synthesizer's one side stream, in program order on every rank.)
    step_streams = StepStreams(n_streams, device=dev) if n_streams > 1 else None
    step_no = [0]
    one_stream = [False]        # (the A/B region behind the timed one)

    def step(done=None):
        """one pass of the hot path over the batch; `done` (an event) is recorded behind it on the stream it was issued on"""
        if step_streams is not None and step_no[0] > 0 and not one_stream[0]:
            step_no[0] += 1
            with step_streams.next():
                r = step_here()
                if done is not None:
                    done.record()
            return r
        step_no[0] += 1
        r = step_here()
        if done is not None:
            done.record()
        return r

    def step_here():
        if graph_run is not None:
            mel_, ol_, _ = graph_run(xs)
            return mel_, ol_
        if synth is not None:
            # LPT shard -> sync-free single-GPU path -> one all-gather (packs + frame counts) -> on every rank the gathered packs +
            # offsets (default) or, with --padded, one more launch that scatters them into the ordered padded [B, Lcap, odim] tensor
            if args.padded or args.profile_kernels:
                return synth(xs, il, sync=args.profile_kernels)
            r = synth(xs, il, packed=True)        # (recv, starts, olens); the very first call is synchronous and returns (mels, olens)
            return r[0], r[-1]
        # nothing on the host waits for the GPU: the frame layout is built on the device (fs2_decode's device-driven
        # mode); the very first call is synchronous and teaches the capacity predictor the frames-per-phoneme ratio
        # (--profile-kernels uses the host-driven layout so that the per-site FLOP counts are those of the rows in use)
        return model.inference_batch(xs, il, sync=args.profile_kernels, regime=regime)

    def all_ok():
        if graph_run is not None:
            return int(graph_run(xs)[2].cpu()[2]) == 0
        if args.profile_kernels:
            return True
        return synth.ok() if synth is not None else model.async_ok()

    with torch.no_grad():
        mel, olens_all = step()                      # first call: synchronous, builds the handle, learns the frame ratio
        if args.graph and not use_dist and not args.profile_kernels:
            graph_run = model.capture_graph(xs, il)
        for _ in range(max(args.warmup, 1)):
            mel, olens_all = step()
        if synth is not None:
            synth.wait()
        assert all_ok(), "capacities of the asynchronous path were exceeded during warm-up"
        total_frames = int(olens_all.sum())
        local_frames = int(olens_all.cpu()[torch.as_tensor(mine, dtype=torch.int64)].sum()) if mine else 0
        local_tokens = int(il[torch.as_tensor(mine, dtype=torch.int64)].sum()) if mine else 0
        # find the dominant launch site with one fully bracketed (untimed) step, then bracket only that site
        # inside the timed region so that hipEvent records do not perturb the measurement
        model.set_profiling(True)
        if graph_run is not None:
            model.inference_batch(xs, il)        # a graph replay has no per-launch events: scout with an eager step
        else:
            step()
        torch.cuda.synchronize()
        scout, scout_n = {}, {}
        scout_prof = model.get_profile()
        for name, ms, fl, by in scout_prof:
            scout[name] = scout.get(name, 0.0) + ms
            scout_n[name] = scout_n.get(name, 0) + 1
        dom_site = max(scout.items(), key=lambda kv: kv[1])[0]
        kernel_ms_per_step = sum(scout.values())
        model.set_profiling(True, only=None if args.profile_kernels else dom_site)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.steps):
            mel, olens_all = step(ev[i + 1])
        if synth is not None:
            synth.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        assert all_ok(), "capacities of the asynchronous path were exceeded in the timed region"
        mel_timed = mel                      # the LAST timed step's result, as its schedule produced it (compared with the checked forward below)
        # spacing of the steps' completions (with S streams: over windows of S steps, the steps of one stream)
        step_ms = [ev[i + 1 - n_streams].elapsed_time(ev[i + 1]) / n_streams for i in range(n_streams - 1, args.steps)]
        prof = model.get_profile() if graph_run is None else scout_prof      # (graph mode: the roofline comes from the eager scouting step)
        model.set_profiling(False)
        # ---- steady state: the timed region above is K steps (0.1 s at c3) after ~0.05 s of warm-up, and sustained MFMA load
        # lowers the shader clock within tens of ms.  The same step runs on for >= args.sustain seconds (not part of `value`), timed the same way,
        # while the host samples the shader clock from sysfs between enqueues.
        sustained = None
        if args.sustain > 0:
            n_sus = max(args.steps, int(args.sustain / max(dt / args.steps, 1e-6)) + 1)
            read_clk = sclk_reader(local)
            clk = []
            smi = None
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(n_sus):
                mel, olens_all = step()
                if read_clk is not None and i % 4 == 0:
                    v = read_clk()
                    if v:
                        clk.append(v)
                if i == n_sus // 4 and rank == 0 and sum(1 for v in clk if v >= 500.0) < max(3, len(clk) // 2):
                    smi = smi_sclk_start()          # sysfs is absent or reads the node's clock: one SMI query while the queue is full
            if synth is not None:
                synth.wait()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            dts = time.perf_counter() - ts
            assert all_ok(), "capacities of the asynchronous path were exceeded in the sustained region"
            if use_dist:
                t = torch.tensor([dts], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dts = float(t.item())
            smi_mhz, smi_tool = smi_sclk_finish(smi, local)
            good = [v for v in clk if v >= 500.0]          # (readings below the lowest DPM level, 94-112 MHz, are a sleeping clock domain's, not this load's: discarded and counted)
            if len(good) >= max(3, len(clk) // 2):
                sclk = dict(min=min(good), median=statistics.median(good), max=max(good), samples=len(good), discarded_below_500mhz=len(clk) - len(good),
                            source="amdgpu sysfs pp_dpm_sclk, sampled by the host while the queue is full")
            elif smi_mhz is not None:
                sclk = dict(median=smi_mhz, samples=1, source="%s, queried once while the queue was full (median over the XCDs; %d of %d sysfs samples read below 500 MHz)"
                                                              % (smi_tool, len(clk) - len(good), len(clk)))
            else:
                sclk = None                          # no trustworthy reading on this box: null, not the node's 95 MHz
            sustained = dict(steps=n_sus, seconds=round(dts, 3), ms_per_step=round(1e3 * dts / n_sus, 3), sclk_mhz=sclk)
        # ---- the same steps on ONE stream (not part of `value`): the A/B of the schedule, and the dominant kernel alone on the chip -- in the
        # timed region above its launches share the CUs with the other step's kernels, so their bracketed duration there is not the kernel's own
        alone = None
        if step_streams is not None:
            n_al = max(4, args.steps)               # the same K as the timed region (10 steps behind a 2-s burn were not comparable)
            one_stream[0] = True
            for _ in range(2):
                step()
            if synth is not None:
                synth.wait()
            torch.cuda.synchronize()
            model.set_profiling(True, only=dom_site)
            if use_dist:
                dist.barrier()
            ta = time.perf_counter()
            for _ in range(n_al):
                mel, olens_all = step()
            if synth is not None:
                synth.wait()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            dta = time.perf_counter() - ta
            if use_dist:
                t = torch.tensor([dta], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dta = float(t.item())
            pa = [ms for name, ms, fl, by in model.get_profile() if name == dom_site]
            model.set_profiling(False)
            one_stream[0] = False
            assert all_ok(), "capacities of the asynchronous path were exceeded in the one-stream region"
            alone = dict(steps=n_al, ms_per_step=round(1e3 * dta / n_al, 3), avg_launch_ms=sum(pa) / max(len(pa), 1))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # which binary ran: the ISA-audit state of libfs2_hip.so and whether the hand-scheduled kernels were allowed to run (a library without a clean audit
    # record of its own hash runs attn_bf16 / gemm_row8_bf16 instead -- slower, and until round 5 invisible in this line), and how many waves of attn_w32
    # left the fast path since the handle was created (0 on this synthetic model's flat attention; a trained model's peaked rows would show here)
    from fastspeech2_amd import _lib as _fs2lib
    kernels = _fs2lib.kernel_state()
    slow_waves = model.counter("attn_slow_path_waves")
    multi_gpu = None
    if use_dist and synth is not None and graph_run is None:
        with torch.no_grad():
            multi_gpu = dist_diagnostics(model, synth, xs, il, parts, rank, world, dev, max(2, min(args.steps, 5)), 1e3 * dt / args.steps, step_streams, args.steps)

    # ---- per-kernel table (hipEvents on the launch stream, accumulated over the timed steps) ----
    agg = {}
    for name, ms, fl, by in prof:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += fl
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    dom_name, (dom_n, dom_ms, _) = dom
    c = model._cfg
    # algorithmic FLOPs per launch of the dominant kernel, valid frames/tokens of THIS rank only
    per_row = {"dec.ffn1": 2.0 * c["ffn_kernel"] * c["ddim"] * c["dunits"], "enc.ffn1": 2.0 * c["ffn_kernel"] * c["adim"] * c["eunits"],
               "dec.ffn2_ln": 2.0 * c["dunits"] * c["ddim"], "dec.qkv": 6.0 * c["ddim"] ** 2, "dec.out_ln": 2.0 * c["ddim"] ** 2}
    rows = local_tokens if dom_name.startswith("enc") or dom_name.startswith("dur") else local_frames
    mine_t = torch.as_tensor(mine, dtype=torch.int64)
    ol_mine = olens_all.cpu()[mine_t].double() if mine else torch.zeros(0, dtype=torch.float64)
    il_mine = il[mine_t].double() if mine else torch.zeros(0, dtype=torch.float64)
    per_launch = {"dec.attn": 4.0 * c["ddim"] * float((ol_mine * ol_mine).sum()),       # 4 D L^2 per utterance and layer
                  "enc.attn": 4.0 * c["adim"] * float((il_mine * il_mine).sum())}
    if dom_name in per_row:
        algo = per_row[dom_name] * rows
    elif dom_name in per_launch:
        algo = per_launch[dom_name]
    else:   # fall back to the launch's own count (includes the ~1.5 % gap rows)
        algo = sum(fl for n_, ms, fl, by in prof if n_ == dom_name) / dom_n
    avg_ms = dom_ms / dom_n
    bracket_in_flight = None
    if alone is not None and alone["avg_launch_ms"] > 0:
        # With several steps in flight a hipEvent bracket is not the kernel's duration: the second event waits for the kernel, the kernel waits in its
        # hardware queue for CUs that other streams' kernels hold (c3, 3 streams: bracket 0.57 ms, rocprofv3 dispatch timestamps of the same region
        # 0.46 ms, the kernel alone 0.42 ms: profiles/r06_kernel_trace_by_region.txt).  The roofline of the kernel is therefore taken where the
        # bracket IS the duration: the same launch site in the one-stream region behind the timed one (it agrees with rocprofv3 to 1 %).
        bracket_in_flight = avg_ms
        avg_ms = alone["avg_launch_ms"]
    achieved = algo / (avg_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.precision]
    roofline = dict(bound="mfma", kernel=dom_name, achieved=round(achieved, 2), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=None, avg_launch_ms=round(avg_ms, 4),
                    launches_per_step=dom_n // max(args.steps, 1),
                    # this site's launches (events around it alone, inside the timed region) as a share of the step's wall time: cannot exceed 1
                    share_of_step_time=round(dom_ms / (1e3 * dt), 3))
    if bracket_in_flight is not None:
        lps = dom_n // max(args.steps, 1)
        roofline["share_of_step_time"] = round(lps * avg_ms / (1e3 * dt / args.steps), 3)      # (kernel time / step interval: steps overlap)
        roofline["measured_in"] = ("the one-stream region behind the timed one (%d steps, hipEvents around this site on the launch stream): with %d steps "
                                   "in flight an event bracket also holds the launch's wait in its hardware queue" % (alone["steps"], n_streams))
        roofline["timed_region_bracket"] = dict(steps_in_flight=n_streams, avg_bracket_ms=round(bracket_in_flight, 4),
                                                frac_if_read_as_duration=round(algo / (bracket_in_flight * 1e-3) / 1e12 / peak, 4),
                                                note="rocprofv3 dispatch timestamps of this region: profiles/r06_kernel_trace_by_region.txt")
    if args.precision != "fp32":
        # what the chip sustains on random bf16 operands with nothing but MFMAs in flight (tools/probes/mfma_shape_probe.hip,
        # profiles/r02_mfma_shape_power_probe.txt: 1.8-2.1 PFLOP/s at 1.8-2.1 GHz, power-limited); `peak` stays the nominal figure
        # -- and the chip's clock gives way under that load: the figure is a band, quoted from the committed probe record
        # (profiles/mfma_rate_probe.json: random operands 1,710, all-ones 2,470 TFLOP/s at 32.0 cycles per MFMA in both cases)
        rec = mfma_peak_record()
        if rec is not None:
            roofline["measured_mfma_peak"] = rec["random_operands_tflops"]
            roofline["measured_mfma_peak_all_ones"] = rec["all_ones_operands_tflops"]
            roofline["measured_mfma_peak_source"] = rec["source"]
            roofline["issued_frac_of_measured_peak"] = round(achieved * {"bf16x3": 3, "mix_f16x2": 2, "mix_f16x1": 1, "mix_mx": 2, "mix_mx4": 1.5}.get(args.precision, 1)
                                                             / rec["random_operands_tflops"], 4) \
                if dom_name.endswith("ffn1") or args.precision == "bf16x3" else None
    # Algorithmic HBM bytes of one launch of the dominant kernel, from the launch's own shapes (valid rows of this rank): every operand read
    # once, every result written once.  bf16 modes: an activation travels as planes of 4 bytes per element (hi + lo bf16, or the mx image
    # of the same size), a weight image has 4 bytes per weight; the attention kernel reads the Q | K | V^T planes and writes the context planes.
    eb = 4
    shapes = {"dec.ffn1": (c["ddim"], c["dunits"], c["ffn_kernel"]), "enc.ffn1": (c["adim"], c["eunits"], c["ffn_kernel"]),
              "dec.ffn2_ln": (c["dunits"], c["ddim"], 1), "dec.qkv": (c["ddim"], 3 * c["ddim"], 1), "dec.out_ln": (c["ddim"], c["ddim"], 1)}
    algo_bytes = None
    if dom_name in shapes:
        Cc, Nn, kk = shapes[dom_name]
        # (mix_mx4, decoder FFN conv: 9 of the 12 units of a plane row are read -- fp16 + the fp4 cross units -- and the weight image has those 9 only: 3 bytes per element;
        #  the LayerNorm-fused launches write planes only since round 6: 4 bytes per element, not 8)
        ab = 3 if (args.precision == "mix_mx4" and dom_name == "dec.ffn1") else eb
        algo_bytes = dict(a_planes_read=rows * Cc * ab, weight_image_read=Nn * kk * Cc * ab, result_written=rows * Nn * eb)
    elif dom_name in per_launch:
        Dd = c["ddim"] if dom_name.startswith("dec") else c["adim"]
        algo_bytes = dict(qkv_planes_read=rows * 3 * Dd * eb, context_planes_written=rows * Dd * eb)
    if algo_bytes is not None:
        roofline["algorithmic_bytes"] = int(sum(algo_bytes.values()))
        roofline["algorithmic_bytes_parts"] = {k: int(v) for k, v in algo_bytes.items()}
    # the whole step against the same peak: algorithmic FLOP of the step (valid tokens / frames of the global batch)
    # / wall time per step / (peak x GPUs) -- `frac` above is the best kernel's, this is the path's
    step_flop = sum(path_flops(int(t), int(l)) for t, l in zip(il, olens_all.cpu()))
    roofline["step_frac"] = round(step_flop / (dt / args.steps) / 1e12 / (peak * world), 4)
    try:    # HBM-side bytes per launch + the matrix-pipe occupancy of the same kernel: rocprofv3 PMC passes of this same command (tools/profile_round.sh
        # -> tools/pmc_summary.py -> profiles/).  The record names the kernel sources it was measured on: after any change to them it is stale and
        # the fields stay null instead of quoting an old kernel
        tr = json.load(open(os.path.join(ROOT, "profiles", "r06_traffic.json")))
        if (tr["workload"] == workload and tr["precision"] == args.precision and tr["kernel_site"] == dom_name and world == 1
                and tr.get("csrc_sha16") == csrc_sha16()):
            roofline["traffic"] = tr["traffic_bytes"]
            roofline["traffic_note"] = ("rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) per launch, profiles/r06_traffic.json (separate --pmc passes of this "
                                        "command on these kernel sources)")
            if roofline.get("algorithmic_bytes"):
                roofline["traffic_over_algorithmic"] = round(tr["traffic_bytes"] / roofline["algorithmic_bytes"], 2)
            if tr.get("mfma_busy") is not None:
                roofline["mfma_busy"] = tr["mfma_busy"]          # SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) of the dominant kernel
    except (OSError, KeyError, ValueError):
        pass
    if args.precision in MFMA_PER_PRODUCT and dom_name.endswith("ffn1"):   # split operands: several MFMAs are issued per algorithmic product
        m = MFMA_PER_PRODUCT[args.precision]
        roofline["issued_tflops"] = round(m * achieved, 1)
        roofline["issued_frac"] = round(m * achieved / peak, 4)
    # HBM-bound kernels of the path against the 8 TB/s roofline: algorithmic bytes of the valid rows / time
    ad, dd, od = c["adim"], c["ddim"], c["odim"]
    pl = 2 if args.precision != "fp32" else 1          # fp32 tensor + split-bf16 planes of the same size
    hbm_bytes = {"lr.expand": local_frames * ad * 4 * pl + local_tokens * ad * 4 + local_frames * 4,
                 "var.embed": local_frames * ad * 4 * (1 + pl) + local_frames * 16,
                 "enc.embed": local_tokens * ad * 4 * (1 + pl),
                 "unpack": local_frames * od * 4 + len(mine) * int(mel.shape[1]) * od * 4}
    roofline_hbm = []
    for name, nbytes in hbm_bytes.items():
        if name in scout and scout[name] > 0:
            ms = scout[name] / scout_n[name]
            gbps = nbytes / (ms * 1e-3) / 1e9
            roofline_hbm.append(dict(kernel=name, bytes=int(nbytes), avg_launch_ms=round(ms, 4), achieved=round(gbps, 1), peak=HBM_PEAK_GBPS,
                                     unit="GB/s", frac=round(gbps / HBM_PEAK_GBPS, 4)))
    # the weak end of the step: the three largest launch sites that run below 0.10 of the peak, from the fully
    # bracketed (untimed) scouting step: FLOPs as the launches count them (rows in use incl. ~1.5 % gap rows), time by hipEvents
    scout_fl = {}
    for name, ms, fl, by in scout_prof:
        scout_fl[name] = scout_fl.get(name, 0.0) + fl
    weak = []
    for name, ms in scout.items():
        if scout_fl.get(name, 0.0) > 0 and ms > 0:
            tf = scout_fl[name] / (ms * 1e-3) / 1e12
            if tf / peak < 0.10:
                weak.append(dict(site=name, launches_per_step=scout_n[name], ms_per_step=round(ms, 4), tflops=round(tf, 1), frac=round(tf / peak, 4),
                                 share_of_step_kernel_time=round(ms / max(kernel_ms_per_step, 1e-9), 4)))
    weak.sort(key=lambda w: -w["ms_per_step"])
    below = sum(w["ms_per_step"] for w in weak)
    roofline_worst = dict(threshold_frac=0.10, sites=weak[:3], sites_below_threshold=len(weak),
                          share_of_step_kernel_time_below_threshold=round(below / max(kernel_ms_per_step, 1e-9), 4))
    if args.profile_kernels and rank == 0:
        tot = sum(v[1] for v in agg.values())
        for name, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-14s launches %4d  avg %8.3f ms  %5.1f %%  %7.1f TFLOP/s" % (name, n, ms / n, 100 * ms / tot, fl / max(ms, 1e-9) / 1e9),
                  file=sys.stderr)

    if rank == 0:
        ol_host = olens_all.cpu()
        if graph_run is not None:
            launch = "HIP graph replay"
        elif args.profile_kernels:
            launch = "eager, host-driven layout"
        else:
            launch = "eager, device-driven layout (no host sync)"
        line = {
            "metric": "mel-frames/sec", "value": round(total_frames * args.steps / dt, 1), "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "ms_per_step_median": round(statistics.median(step_ms), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[args.precision],
            "data": "synthetic",
            "config": {"workload": "%s%s; default.yaml dims, free-running, random-init weights (seed 0), duration bias calibrated to 7.87 frames/phoneme"
                                   % (WORKLOAD_TEXT[workload], " -- here %d utterances on %d GPU(s)" % (B, world) if workload == "c5" else ""),
                       "precision": args.precision,      # the arithmetic mode the line was measured in (rounds 1-2: bf16x3; rounds 3-5: mix_mx; round 6: mix_mx4)
                       "utterances": B, "utterances_per_gpu": [len(p) for p in parts], "valid_frames_per_step": total_frames,
                       "phonemes": int(il.sum()),
                       "algorithmic_gflop_per_step": round(sum(path_flops(int(t), int(l)) for t, l in zip(il, ol_host)) / 1e9, 1),
                       "parallelism": ("LPT utterance-sharded x%d (unequal shards), one all-gather(packed mels + frame counts) over RCCL%s"
                                       % (world, ", overlapped with the next step's forward (side stream)" if synth.overlap else ""))
                                      if use_dist else "single GPU",
                       "launch": launch, "overlap_encoder": bool(model.overlap_encoder), "streams": n_streams},
            "roofline": roofline,
            "roofline_worst": roofline_worst,
            "roofline_hbm": roofline_hbm,
        }
        if sustained is not None:
            line["sustained_ms_per_step"] = sustained["ms_per_step"]
            # (sustained.sclk_mhz: boxes of one pool differ -- one ran capped at 1,412 MHz with every kernel 40 % slower; the sysfs node is not always
            #  trustworthy the other way round: one run read 158 .. 202 MHz at unchanged step times.  Read it beside `one_stream` and `roofline`.)
            line["sustained"] = dict(sustained, ratio_to_timed=round(sustained["ms_per_step"] / (1e3 * dt / args.steps), 4),
                                     value=round(total_frames * 1e3 / sustained["ms_per_step"], 1))
        line["kernels"] = kernels
        line["attn_slow_path_waves"] = slow_waves
        if regime is not None:
            line["config"]["regime"] = dict(phonemes=regime[0], utterances=regime[1], note="kernel variants chosen as for a batch of this size (this batch run as its shard)")
        elif use_dist:
            line["config"]["regime"] = "the whole batch on every rank (sharded == unsharded bit for bit)" if synth.global_regime else "each shard's own"
        if alone is not None:
            line["value_one_forward"] = round(total_frames * 1e3 / alone["ms_per_step"], 1)
            line["ms_per_forward"] = alone["ms_per_step"]
            line["value_note"] = ("value / ms_per_step: %d whole forwards in flight on %d streams (a throughput schedule; each step does all its work); value_one_forward / "
                                  "ms_per_forward: the same K steps one after another on one stream -- SURVEY 8(d)'s 'wall time of one full forward', the figure of rounds 1-4"
                                  % (n_streams, n_streams))
            line["one_stream"] = dict(steps=alone["steps"], ms_per_step=alone["ms_per_step"], value=round(total_frames * 1e3 / alone["ms_per_step"], 1),
                                      note="the same steps issued on one stream (rounds 1-4's schedule%s), measured behind the timed region"
                                           % (" + overlap_encoder" if model.overlap_encoder else ""))
        elif not model.overlap_encoder:      # --streams 1 --no-overlap-encoder (or --graph / --profile-kernels): the timed schedule IS one forward after another
            line["value_one_forward"], line["ms_per_forward"] = line["value"], line["ms_per_step"]
        if use_dist:
            line["config"]["gather"] = "padded" if (args.padded or args.profile_kernels) else "packed"
            line["config"]["collective_world_size"] = dist.get_world_size()
            line["multi_gpu"] = multi_gpu
        if world == 1 and not use_dist and not args.no_cpu_baseline:      # the only leg of this script that touches oracle/ (as the measured CPU baseline and the checker)
            from oracle import fs2_oracle as O
            cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, odim)
            with torch.no_grad():
                r = model._run(xs, il, is_inference=True, want=("after", "qe", "qp"))
            gpu = dict(after=r["after"].cpu(), d_int=r["d_int"].cpu(), olens=r["olens"], qe=r["qe"].cpu().long(), qp=r["qp"].cpu().long())
            cb, worst, flips = cpu_baseline(sd, cfg, dict(xs=batch["xs"], ilens=il), gpu)
            line["cpu_baseline"] = cb
            line["vs_cpu"] = round(line["value"] / cb["value"], 1)
            line["mel_max_abs_diff"] = worst
            # the output that was checked is a separate synchronous forward; what the timed steps produced (sync-free, several in flight) must be that, bit for bit
            if torch.is_tensor(mel_timed) and mel_timed.dim() == 3 and mel_timed.shape[1] >= r["after"].shape[1]:
                lm = r["after"].shape[1]
                line["timed_output_identical_to_checked"] = bool(torch.equal(mel_timed[:, :lm], r["after"]) and float(mel_timed[:, lm:].abs().sum()) == 0.0)
            line["decision_agreement"] = flips
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
