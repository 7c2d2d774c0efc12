#!/usr/bin/env python
"""Throughput of the `--com disco` hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one pass of the hot path over one batch of synthetic scenes already
resident in HBM: batched dense rebuild of the sparse voxel lists (K1/a2) ->
MotionNet encoder -> DiscoGraph fusion (warp, pairwise attention, agent
softmax, weighted sum) -> decoder -> cls/reg heads.  Workload = BASELINE.json
configs[1]: 5 agents, batch 4, 256x256x13 BEV, no KD, eval forward.

Multi-GPU: scene-parallel -- every rank runs its own batch of scenes, no
collective on the data path -> "scaling": "weak".

Steps are replayed from one captured hipGraph on ONE stream, strictly one at a
time (--in-flight 1, the default; DESIGN.md 3.6 explains why nothing runs side
by side).  The run checks that a graph replay equals the eager step bit for bit
(`graph_equals_eager`).  --task seg measures BASELINE configs[3], --mode agent
configs[4].

Prints ONE JSON line (rank 0) with the driver's contract fields plus
`roofline` (dominant kernel = the split-f16 MFMA implicit-GEMM conv on
split-planar activations, HIP-event timed per launch on the launch stream, and
the committed rocprofv3 figure beside it) and `cpu_baseline` (the CPU oracle,
i.e. a port: the reference itself is not in the mount; batch 1 and batch 4).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AGENTS, BATCH, MAP_HW = 5, 4, 256
# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks
MFMA_PEAK_TFLOPS = {"f32": 157.3,      # v_mfma_f32_32x32x2_f32
                    "f16x3": 2500.0,   # v_mfma_f32_32x32x16_f16 (the split-f16 paths run 3 per product)
                    "sp": 2500.0}
EXECUTED_FLOP_FACTOR = {"f32": 1, "f16x3": 3, "sp": 3}
CONV_KERNELS = ("conv_mfma_kernel", "conv_sp_kernel")
DTYPE_NAME = {"f32": "f32", "f16x3": "f32 storage; conv products as split-f16 x3 MFMA, f32 accumulate",
              "sp": "f16 hi+lo pair per value (32 bits; 22-bit significand for |x| >= 0.125, f16 exponent range, "
                    "lo subnormal below: absolute floor 2^-25); conv products as split-f16 x3 MFMA, f32 accumulate"}
MATH_LABEL = {"f32": "exact-fp32", "f16x3": "split-f16x3 (fp32 NHWC activations)",
              "sp": "split-f16x3, split-planar activations staged by LDS-DMA"}


def _sp_bevs(ops, indices, offsets, n_img, dims):
    """the voxel batch in the form the split-planar engine's first layer reads: one occupancy word per pixel
    (DN_BEV_FORM=hi: round 3's hi-only planes, for A/B runs -- the results are bit-identical)"""
    if os.environ.get("DN_BEV_FORM", "bits") == "hi":
        return ops.scatter_dense_sp(indices, offsets, n_img, dims, hi_only=True)
    return ops.scatter_dense_bits(indices, offsets, n_img, dims)



def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pre-roll", type=int, default=60,
                    help="untimed replays after the W warm-up steps and before the timed region (clock settling)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record per-launch HIP events in the timed region")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table (stderr)")
    ap.add_argument("--math", choices=["f32", "f16x3", "sp"], default=os.environ.get("DISCONET_CONV_MATH", "sp"),
                    help="conv engine: sp = split-f16 (3 f16 MFMAs / product) on split-planar activations "
                         "(default); f16x3 = the same arithmetic on fp32 NHWC activations; f32 = exact-fp32 MFMA")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the other math mode's timed region")
    ap.add_argument("--mode", choices=["scene", "agent"], default="scene",
                    help="multi-GPU partitioning: 'scene' = every rank its own scenes (default, weak "
                         "scaling, no data-path collective); 'agent' = BASELINE configs[4]: 8-agent "
                         "scenes, agents sharded across ranks, one RCCL all-gather of the layer-3 maps "
                         "per step (strong scaling of a fixed batch of scenes)")
    ap.add_argument("--task", choices=["det", "seg"], default="det",
                    help="det = BASELINE configs[1] (the headline metric); seg = configs[3]: the segmentation "
                         "variant (UNet + DiscoGraph fusion at the bottleneck), eval forward + cross entropy")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel from Python instead of replaying a captured hipGraph")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="steps in flight.  1 (default) = strictly one step at a time on one stream.  N > 1: "
                         "consecutive steps (independent batches) are replayed on N alternating HIP streams, each "
                         "from its own captured graph with its own buffers (+4 %% scenes/s) -- NOT the default "
                         "because a kernel that shares a SIMD with another stream's f16-MFMA waves has been "
                         "observed to compute with corrupted VGPR lanes (DESIGN.md 3.6, profiles/r02_hazard_repro.txt). "
                         "With N > 1 EVERY replay of the timed region is checksummed on its stream and compared "
                         "with the serial replay's checksum; one mismatch and the serial figure is reported")
    ap.add_argument("--no-voxelize", action="store_true", help="skip the K1 (raw points) timing extra")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=240.0)
    ap.add_argument("--train-steps", type=int, default=6,
                    help="also time this many training steps (forward + loss + backward + Adam) of the "
                         "same batch; 0 = skip (reported next to the headline value, never in it)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="--mode agent on ONE GPU: also time rank 0's share of a W-rank agent-sharded run (8 / W "
                         "agents, peers' maps by device copy) and print projected_speedup = T(unsharded) / T(share)")
    ap.add_argument("--agent-check", type=int, default=0,
                    help="--mode agent: replay this many extra steps and compare every step's output checksum with "
                         "the first (all-gather kernel between the two compute graphs)")
    ap.add_argument("--agent-batch", type=int, default=BATCH,
                    help="scenes per step of the agent-sharded leg (configs[4] does not fix it; 4 = the det batch)")
    ap.add_argument("--no-pg", action="store_true",
                    help="--mode agent on one GPU: no process group at all (the exchange is a local copy): isolates "
                         "the compute graphs from the collective in --agent-check")
    ap.add_argument("--no-agent-leg", action="store_true",
                    help="skip the agent-sharded leg (BASELINE configs[4]) that the default line carries as "
                         "`agent_sharded` when the GPU count divides 8")
    ap.add_argument("--force-process-group", action="store_true",
                    help="create the RCCL process group even for one rank (exercises the N>1 code path)")
    ap.add_argument("--via-launcher", action="store_true",
                    help="go through the self-launcher (torch.distributed.run --standalone) even for --gpus 1: the N > 1 launch "
                         "path -- launcher, rank environment, RCCL process group, relay of rank 0's line -- on a one-GPU box")
    ap.add_argument("--dry-launch", action="store_true",
                    help="--gpus N > 1 without a launcher: print the torch.distributed.run command that would be "
                         "exec'd (one JSON object on stdout) and exit 0")
    ap.add_argument("--cpu-baseline-child", nargs=3, metavar=("STATE", "OUT", "THREADS"),
                    help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the
    cgroup CPU quota (os.cpu_count() reports the whole machine in a container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(state_dict, threads):
    """Times the CPU oracle (kind 'port') on ONE scene (5 agents, 256x256x13,
    batch 1), 1 warm-up + best of 3.  Bounded sample of the same workload."""
    from oracle.disconet_ref import RefConfig, build_ref_model
    from disconet_amd.synthetic import make_scene_batch
    torch.set_num_threads(threads)
    ref = build_ref_model(RefConfig(MAP_HW), kd_flag=0, num_agent=AGENTS)
    ref.load_state_dict(state_dict, strict=False)
    bevs, trans, na = make_scene_batch(1, AGENTS, MAP_HW)
    best, reps = float("inf"), 0
    with torch.no_grad():
        t0 = time.perf_counter()
        out = ref(bevs, trans, na, 1)                     # warm-up (also the parity sample)
        warm = time.perf_counter() - t0
        budget = 25.0                                     # seconds of CPU work for the repeats
        while reps < 3 and (reps == 0 or budget > 0):
            t0 = time.perf_counter()
            ref(bevs, trans, na, 1)
            dt = time.perf_counter() - t0
            best, reps, budget = min(best, dt), reps + 1, budget - dt
            if warm > 60:
                break
    base = {"value": round(1.0 / best, 4), "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": "1 scene (5 agents, 256x256x13, batch 1), eval fwd, fp32, best of %d after 1 "
                      "warm-up; torch-CPU oracle (reference source not in the mount)" % reps}
    # the bench's own batch: 4 scenes in one forward, once (still a bounded sample of the workload)
    if best * BATCH * 1.5 < 60:
        b4 = make_scene_batch(BATCH, AGENTS, MAP_HW)
        with torch.no_grad():
            t0 = time.perf_counter()
            ref(b4[0], b4[1], b4[2], BATCH)
            dt4 = time.perf_counter() - t0
        # like for like with the GPU step: the headline CPU figure is the bench's own batch; batch 1 beside it
        base["batch_1"] = {"value": base["value"], "unit": "scenes/s", "sample": base["sample"]}
        base["value"] = round(BATCH / dt4, 4)
        base["sample"] = ("one forward of %d scenes (5 agents, 256x256x13: the bench's batch), eval fwd, fp32, after the "
                          "batch-1 runs as warm-up; torch-CPU oracle (reference source not in the mount)" % BATCH)
    # The timed model carries torch's default init: its logits are ~1e-2, so an error against them
    # says nothing about the arithmetic.  The parity sample runs a kaiming-initialised copy (activations and logits of
    # O(1): what the -m gpu parity tests use) on the TIMED step's own inputs -- the bench's batch of sparse voxel lists
    # (as the dense grid they describe) and rank 0's poses; the parent runs that state through the HIP path from the
    # lists themselves (dn_scatter_dense_bits -> stem pair -> ... -> K-sliced conv5_1: the launches `value` times).
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices
    pb = BATCH if best * BATCH * 1.5 < 60 else 1
    _, _, pbevs = make_sparse_scene_batch(pb, AGENTS, MAP_HW)
    ptrans = make_trans_matrices(pb, AGENTS, jitter_seed=0)
    pna = torch.full((pb, AGENTS), AGENTS, dtype=torch.int64)
    refk = build_ref_model(RefConfig(MAP_HW), init="kaiming", kd_flag=0, num_agent=AGENTS)
    with torch.no_grad():
        outk = refk(pbevs, ptrans, pna, pb)
    parity = {"state": refk.state_dict(), "cls": outk["cls"], "loc": outk["loc"], "batch": pb}
    return base, (bevs, trans, na, out), parity


def cpu_baseline_child(state_path, out_path, threads):
    base, (_, _, _, out), parity = cpu_baseline(torch.load(state_path), int(threads))
    torch.save({"base": base, "cls": out["cls"], "loc": out["loc"], "parity": parity}, out_path)


def cpu_baseline_bounded(state_dict, threads, timeout_s):
    """Runs the CPU oracle in a child process so a slow host can never take the
    bench line down with it; on timeout the baseline is reported as missing."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dn_bench_")
    sp, op = os.path.join(tmp, "state.pt"), os.path.join(tmp, "out.pt")
    torch.save(state_dict, sp)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", sp, op, str(threads)]
    try:
        subprocess.run(cmd, timeout=timeout_s, check=True, env=env, stdout=subprocess.DEVNULL)
        r = torch.load(op)
        return r["base"], {"cls": r["cls"], "loc": r["loc"], "parity": r.get("parity")}
    except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
        return {"value": None, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": "1 scene (5 agents, 256x256x13)", "error": type(e).__name__}, None


def agent_sharded_leg(args, world, rank, dist, emulate_world=0, check_steps=0):
    """BASELINE configs[4] (the north star's multi-GPU split): 8-agent scenes, the agents sharded across the
    ranks, ONE RCCL all-gather of the layer-3 maps per step as the V2X exchange, everything else rank-local.
    Fixed work (--agent-batch, default 4, 8-agent scenes) for every N -> strong scaling.  One step = dense rebuild of this
    rank's voxel lists + encoder (hipGraph A) -> all-gather (eager, in stream order) -> fusion of this rank's
    egos + decoder + heads (hipGraph B): disconet_amd.sharded.GraphedAgentStep.

    emulate_world = W on ONE GPU: time rank 0's share of a W-rank run (8 / W agents; the exchange is a device
    copy of the peers' maps) next to the unsharded 8-agent step -> projected_speedup = T(8 agents, 1 GPU) /
    T(share): an upper bound for the W-GPU curve (no link latency, no rank skew).
    check_steps: replay that many steps and compare a checksum of every step's outputs with the first one's
    (the collective's kernel runs between our two graphs; outputs must stay bit-identical).
    Returns the result dict on rank 0, None elsewhere.  Needs an initialised process group unless emulating."""
    from disconet_amd import Config, DiscoNet, ops, sharded
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats
    agents = 8
    batch = args.agent_batch
    torch.manual_seed(0)
    model = DiscoNet(Config(map_hw=MAP_HW), kd_flag=0, num_agent=agents)
    randomize_bn_stats(model)
    model.conv_math = args.math
    model.eval().cuda()
    engine = sharded.HipEngine(model)
    indices, offsets, _ = make_sparse_scene_batch(batch, agents, MAP_HW)
    trans = make_trans_matrices(batch, agents).cuda()
    na = torch.full((batch, agents), agents, dtype=torch.int64).cuda()
    dims = (MAP_HW, MAP_HW, 13)

    def rank_inputs(first, count):
        # this rank's agents: images [first*B, (first+count)*B) of the agent-major stack
        lo, hi = int(offsets[first * batch]), int(offsets[(first + count) * batch])
        idx = indices[lo:hi].contiguous().cuda()
        off = (offsets[first * batch:(first + count) * batch + 1] - lo).to(torch.int32).cuda()

        def make_bevs():
            if model.conv_math == "sp":
                return _sp_bevs(ops, idx, off, count * batch, dims)
            return ops.scatter_dense(idx, off, count * batch, dims)
        return make_bevs

    def checksum(out):
        """three order-independent checksums: cls, loc, the fused layer-3 map"""
        res, fused = out
        f = fused.data if isinstance(fused, ops.SpTensor) else fused
        return torch.stack([res["cls"].view(torch.int64).sum(), res["loc"].view(torch.int64).sum(),
                            f.contiguous().view(torch.int32).sum().to(torch.int64)])

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    def time_steps(stepper, steps, warmup):
        for _ in range(warmup):
            stepper()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper()
        fence()
        dt = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def phase_us(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(1e3 * e0.elapsed_time(e1) / reps, 2)

    first, count = sharded.agent_range(agents, world, rank)
    stepper = sharded.GraphedAgentStep(engine, rank_inputs(first, count), trans, na, batch, first, count, range_guard=False)
    elapsed = time_steps(stepper, args.steps, args.warmup)
    # per-phase times on EVERY rank (the exchange is a collective: all ranks must call it the same number of times)
    phases = {"graph_a_encode": phase_us(stepper.graph_a), "allgather": phase_us(stepper.exchange),
              "graph_b_fuse_decode_heads": phase_us(stepper.graph_b)}
    fence()
    res = None
    if rank == 0:
        res = {
            "metric": "scenes/sec (8-agent 256x256 BEV, agents sharded across GPUs)",
            "value": round(batch * args.steps / elapsed, 3), "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NAME[args.math], "data": "synthetic",
            "config": {"workload": "DiscoNet det eval forward (--com disco), 8-agent scenes, batch %d, 256x256x13 BEV, "
                                   "%d agent(s) per GPU, one RCCL all-gather of the 256x32x32 layer-3 maps per "
                                   "step (BASELINE configs[4])" % (batch, count),
                       "agents": agents, "batch": batch, "conv_math": args.math,
                       "launch": "hipGraph A (rebuild + encoder) -> all-gather -> hipGraph B (fusion + decoder + heads)",
                       "parallelism": "agent-parallel x%d" % world},
            "phases_us": phases,
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 0,
            "collective_backend": str(dist.get_backend()) if dist.is_initialized() else "none",
            "exchanged_bytes_per_rank_per_step": int(stepper.feat_all.numel() * stepper.feat_all.element_size()
                                                     * (world - 1) // max(world, 1)),
        }
    if check_steps > 0:
        want = checksum(stepper()).clone()
        sums = torch.zeros((check_steps, 3), dtype=torch.int64, device="cuda")
        enc_sums = torch.zeros(check_steps, dtype=torch.int64, device="cuda")
        x3 = stepper.enc[engine.layer]
        enc_want = None
        for i in range(check_steps):
            sums[i] = checksum(stepper())
            enc_sums[i] = (x3.data if isinstance(x3, ops.SpTensor) else x3).contiguous().view(torch.int32).sum()
        torch.cuda.synchronize()
        per_part = (sums != want).sum(0).tolist()
        enc_diff = int((enc_sums != enc_sums[0]).sum().item())
        differing = int((sums != want).any(1).sum().item())
        if dist.is_initialized():
            t = torch.tensor([differing], device="cuda", dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            differing = int(t.item())
        if rank == 0:
            res["replay_check"] = {"steps": check_steps, "differing": differing,
                                   "differing_by_output": {"cls": per_part[0], "loc": per_part[1], "fused": per_part[2],
                                                           "encoder_layer3_map": enc_diff},
                                   "exchange": "local copy (no process group)" if not dist.is_initialized() else "RCCL all-gather",
                                   "how": "checksum of cls + loc + fused map after every step (graph A, all-gather, "
                                          "graph B back to back) vs the first step's"}
    if emulate_world and world == 1 and rank == 0:
        W = emulate_world
        if agents % W:
            raise SystemExit("--emulate-world must divide 8 agents")
        full_feat = stepper.feat_all.clone()              # every agent's layer-3 map, from the unsharded step
        full_out = stepper()
        full = {"cls": full_out[0]["cls"].clone(), "loc": full_out[0]["loc"].clone()}
        cnt = agents // W
        # the collective's launch + kernel latency belongs in the share: a one-rank RCCL group runs the real
        # all_gather_into_tensor between the two graphs (what it cannot show is link time and rank skew)
        have_pg = dist.is_initialized()
        if not have_pg and not args.no_pg:
            try:
                import datetime
                dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29530 + os.getpid() % 400), rank=0,
                                        world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()),
                                        timeout=datetime.timedelta(seconds=60))
                have_pg = True
            except Exception as e:      # noqa: BLE001 -- the projection then carries no collective (said in the note)
                print("bench: no one-rank process group for the emulated share (%r)" % (e,), file=sys.stderr)
        share_plain = sharded.GraphedAgentStep(engine, rank_inputs(0, cnt), trans, na, batch, 0, cnt, range_guard=False,
                                               emulate_feat_all=full_feat)
        t_plain = time_steps(share_plain, args.steps, args.warmup)
        share = share_plain
        t_share = t_plain
        if have_pg:
            share = sharded.GraphedAgentStep(engine, rank_inputs(0, cnt), trans, na, batch, 0, cnt, range_guard=False,
                                             emulate_feat_all=full_feat, emulate_collective=True)
            t_share = time_steps(share, args.steps, args.warmup)
        one_graph = None
        if os.environ.get("DN_AGENT_ONE_GRAPH") == "1":      # opt-in: A + exchange (collective included) + B as ONE captured graph
            err = share.capture_one_graph()
            if err is None:
                t_one = time_steps(share, args.steps, args.warmup)
                one_graph = {"ms_per_step": round(1e3 * t_one / args.steps, 4), "projected_speedup": round(elapsed / t_one, 3)}
            else:
                one_graph = {"error": err}
        got = share()
        torch.cuda.synchronize()
        rows = cnt * batch
        same = bool(torch.equal(got[0]["cls"], full["cls"][:rows]) and torch.equal(got[0]["loc"], full["loc"][:rows]))
        res["emulated_share"] = {
            "world": W, "agents_per_rank": cnt, "ms_per_step": round(1e3 * t_share / args.steps, 4),
            "phases_us": {"graph_a_encode": phase_us(share.graph_a), "exchange_copy": phase_us(share.exchange),
                          "graph_b_fuse_decode_heads": phase_us(share.graph_b)},
            "projected_speedup": round(elapsed / t_share, 3),
            "collective_in_share": ("RCCL all_gather_into_tensor of this rank's maps on a one-rank group (launch + kernel "
                                    "latency; no link time)") if have_pg else "none (no process group)",
            "ms_per_step_without_collective": round(1e3 * t_plain / args.steps, 4),
            "projected_speedup_without_collective": round(elapsed / t_plain, 3),
            "outputs_equal_unsharded_rows": same,
            **({"one_graph": one_graph} if one_graph is not None else {}),
            "note": "rank 0's share of a %d-rank run timed on one GPU: the peers' maps arrive by a device copy and, when a "
                    "process group exists, the real RCCL all-gather kernel runs between the two graphs (its launch + "
                    "kernel latency; no link time, no rank skew); projected_speedup = T(8 agents, 1 GPU) / T(share) "
                    "bounds the measured curve from above" % W}
    return res


def agent_sharded_bench(args, world, rank, dist):
    if 8 % world:
        raise SystemExit("--mode agent needs a GPU count that divides 8 agents")
    if world == 1 and not dist.is_initialized() and not args.no_pg:
        import datetime
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()),
                                timeout=datetime.timedelta(seconds=120))
    res = agent_sharded_leg(args, world, rank, dist, emulate_world=args.emulate_world, check_steps=args.agent_check)
    if rank == 0:
        emit(res)
    if dist.is_initialized():
        dist.destroy_process_group()


def _time_train_steps(mod, data, batch, steps, warm=4, losses=False):
    """-> (first, last, mean seconds per step, per-step milliseconds).  `warm` untimed steps first: the calibration pass (the
    first backward measures the gradient maps' lifts in fp32), then the split-f16 path's own warm-up (allocator, LDS
    attributes, clocks after the idle CPU-baseline phase).  A step ends in a host read of its losses, so the per-step wall
    times cost nothing extra; the reported mean is over the whole timed region, synchronised on both sides."""
    first = mod.step(data, batch)
    traj = [first["loss"]]
    for _ in range(max(warm - 1, 0)):
        traj.append(mod.step(data, batch)["loss"])
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        last = mod.step(data, batch)
        per.append(round(1e3 * (time.perf_counter() - t1), 3))
        traj.append(last["loss"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if losses:      # the loss of every step from the first (untimed warm-up steps included)
        return first, last, dt, per, [round(float(x), 4) for x in traj]
    return first, last, dt, per


def seg_bench(args, world, rank, dist, use_pg):
    """BASELINE configs[3]: DiscoNet seg, 5-agent, 256x256 BEV; scene-parallel like the det bench.
    One step = dense rebuild of the voxel lists (into the conv engine's layout) + the UNet forward with
    the DiscoGraph fusion at the 512-channel bottleneck + the per-pixel 8-class cross entropy."""
    from disconet_amd import SegDiscoNet, SegModule, ops
    from disconet_amd.graph import GraphedStep
    from disconet_amd.profiling import KernelTimer, timing
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats
    torch.manual_seed(0)
    model = SegDiscoNet(num_agent=AGENTS)
    randomize_bn_stats(model)
    model.eval().cuda()
    mod = SegModule(model)
    indices, offsets, _ = make_sparse_scene_batch(BATCH, AGENTS, MAP_HW)
    indices, offsets = indices.cuda(), offsets.cuda()
    trans = make_trans_matrices(BATCH, AGENTS, jitter_seed=rank).cuda()
    na = torch.full((BATCH, AGENTS), AGENTS, dtype=torch.int64).cuda()
    labels = torch.randint(0, 8, (AGENTS * BATCH, MAP_HW, MAP_HW), device="cuda", dtype=torch.int32)
    dims, n_img = (MAP_HW, MAP_HW, 13), AGENTS * BATCH

    def step():
        bevs = ops.scatter_dense_sp(indices, offsets, n_img, dims)
        with torch.no_grad():
            logits = model(bevs, trans, na, BATCH)
            z = logits.permute(0, 2, 3, 1)
            # (check_labels=False: the label-range check is a host read, illegal inside the capture; the labels are randint(0, 8))
            return ops.seg_ce_loss(z, labels, want_grad=False, check_labels=False)[0], logits

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
            torch.cuda.synchronize()

    graphed = None
    if not args.no_graph:
        try:
            graphed = GraphedStep(step, range_guard=False)
        except Exception as e:      # noqa: BLE001
            print("bench: hipGraph capture failed (%r); launching eagerly" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
    run = graphed if graphed is not None else step
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, logits = run()
    fence()
    elapsed = time.perf_counter() - t0
    if use_pg:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    result = {
        "metric": "scenes/sec (5-agent 256x256 BEV, seg)", "value": round(world * BATCH * args.steps / elapsed, 3),
        "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16 hi+lo pair per value (32 bits, 22-bit significand); conv products as split-f16 x3 MFMA, "
                 "f32 accumulate",
        "data": "synthetic",
        "config": {"workload": "DiscoNet seg eval forward (UNet 13->64..512, DiscoGraph fusion at the 512-channel "
                               "bottleneck, 8 classes) + per-pixel cross entropy, 5-agent, batch 4 per GPU, "
                               "256x256x13 BEV (BASELINE configs[3])",
                   "agents": AGENTS, "batch_per_gpu": BATCH, "bev": [MAP_HW, MAP_HW, 13], "conv_math": "sp",
                   "launch": "hipGraph replay" if graphed is not None else "eager",
                   "parallelism": "scene-parallel x%d (no data-path collective)" % world},
        "loss": round(float(loss), 6)}
    if not args.no_kernel_events:
        timer = KernelTimer()
        torch.cuda.synchronize()
        with timing(timer):
            for _ in range(args.steps):
                step()
        summ = timer.summary()
        conv = {k: v for k, v in summ.items() if v["kernel"] in CONV_KERNELS}
        flops = sum(v["flops"] for v in conv.values())
        ms = sum(v["ms_total"] for v in conv.values())
        ach = flops / (ms * 1e-3) / 1e12 if ms else 0.0
        result["roofline"] = {
            "kernel": "conv_sp_kernel (%s, all %d launches/step)" % (MATH_LABEL["sp"], sum(
                v["calls"] for v in conv.values()) // args.steps),
            "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_PEAK_TFLOPS["sp"], "unit": "TFLOP/s",
            "frac": round(ach / MFMA_PEAK_TFLOPS["sp"], 4), "executed_flop_factor": 3,
            "frac_executed": round(3 * ach / MFMA_PEAK_TFLOPS["sp"], 4), "traffic": None,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": None,
            "flop_per_step": flops / args.steps, "kernel_ms_per_step": round(ms / args.steps, 4),
            "note": "achieved = algorithmic FLOP (true channel counts) / HIP-event kernel time"}
        try:   # HBM bytes per conv launch from the committed rocprofv3 --pmc passes over this command (eager)
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic_seg.json"))[-1]
            result["roofline"]["traffic"] = json.load(open(os.path.join(ROOT, "profiles", prof)))["hbm_bytes_per_launch"]
            result["roofline"]["traffic_source"] = "profiles/" + prof
        except (OSError, IndexError, KeyError, ValueError):
            pass
        if args.layers:
            for k, v in summ.items():
                print("[seg] %-12s %8.4f ms/step %9.2f TFLOP/s" % (
                    k, v["ms_total"] / args.steps, v["flops"] / (v["ms_total"] * 1e-3) / 1e12 if v["ms_total"] else 0),
                    file=sys.stderr)
    if world == 1 and not args.no_cpu_baseline:
        # the CPU oracle of the same model on ONE scene (bounded sample), all usable cores
        from oracle.seg_ref import SegDiscoNetRef, seg_loss
        from disconet_amd.synthetic import make_scene_batch
        threads = usable_cores()
        torch.set_num_threads(threads)
        ref = SegDiscoNetRef(num_agent=AGENTS).eval()
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        bevs1, trans1, na1 = make_scene_batch(1, AGENTS, MAP_HW)
        x1 = bevs1[:, 0].permute(0, 3, 1, 2).contiguous()
        best = float("inf")
        with torch.no_grad():
            want = ref(x1, trans1, na1, 1)
            for _ in range(2):
                t0 = time.perf_counter()
                ref(x1, trans1, na1, 1)
                best = min(best, time.perf_counter() - t0)
            got = model(x1.cuda(), trans1.cuda(), na1.cuda(), 1)
        result["cpu_baseline"] = {
            "value": round(1.0 / best, 4), "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": "1 scene (5 agents, 256x256x13), eval fwd, fp32, best of 2 after 1 warm-up; torch-CPU "
                      "oracle (reference source not in the mount)",
            "parity_max_abs_err": float((got.cpu() - want).abs().max())}
    if world == 1 and args.train_steps > 0:
        # SegModule.step on the same batch (train() mode: batch statistics, cross entropy, explicit HIP reverse
        # pass through the UNet and the fusion, Adam) -- reported beside the headline value, never in it
        try:
            tmodel = SegDiscoNet(num_agent=AGENTS).cuda()
            tmod = SegModule(tmodel, lr=1e-3)
            x = ops.scatter_dense(indices, offsets, n_img, dims).view(n_img, MAP_HW, MAP_HW, 13)
            tdata = {"bev_seq": x, "trans_matrices": trans, "num_agent": na, "labels": labels}
            first, last, dt, per = _time_train_steps(tmod, tdata, BATCH, args.train_steps)
            result["train_step"] = {"ms_per_step": round(1e3 * dt, 3), "scenes_per_s": round(BATCH / dt, 2),
                                    "steps": args.train_steps, "batch_per_gpu": BATCH,
                                    "loss_first": round(first["loss"], 5), "loss_last": round(last["loss"], 5),
                                    "step_ms": per, "dgrad_math": tmod.engine.dgrad_math, "wgrad_math": tmod.engine.wgrad_math,
                                    "layers_with_a_measured_gradient_lift": len(tmod.engine._dz_lift),
                                    "note": "SegModule.step: train() forward + cross entropy + explicit HIP backward + Adam, "
                                            "eager launches, wall clock"}
        except Exception as e:
            result["train_step"] = {"error": repr(e)}
    emit(result)


_RESULT_FD = None


def claim_stdout():
    """stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on the process's stdout (at
    process-group teardown, i.e. possibly AFTER our line), so file descriptor 1 is pointed at stderr for everything
    that is not the result and the line goes out through a saved duplicate of the original stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result):
    line = (json.dumps(result) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def line_summary(result):
    """The line's figures without its notes, as the LAST key: the driver keeps the parsed contract keys and the last
    2000 characters of stdout, so everything a reader needs beside `roofline` / `cpu_baseline` is repeated here, short."""
    def pick(d, *keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    out = {"value": result.get("value"), "n_gpus": result.get("n_gpus"), "ms_per_step": result.get("ms_per_step")}
    roof = result.get("roofline")
    if roof:
        out["roofline"] = pick(roof, "frac", "frac_executed", "kernel_ms_per_step", "achieved")
        out["other_kernels_ms"] = roof.get("other_kernels_ms_per_step")
    alt = result.get("alt_math")
    if alt:
        out["alt_math"] = pick(alt, "conv_math", "value", "ms_per_step")
        if "roofline" in alt:
            out["alt_math"]["roofline"] = pick(alt["roofline"], "frac", "peak", "achieved")
    for key in ("agent_sharded", "agent_sharded_batch16"):
        a = result.get(key)
        if isinstance(a, dict):
            o = pick(a, "value", "n_gpus", "ms_per_step", "rccl_ranks", "exchanged_bytes_per_rank_per_step", "phases_us", "error")
            if "emulated_share" in a:
                o["emulated_share"] = pick(a["emulated_share"], "world", "ms_per_step", "projected_speedup",
                                           "outputs_equal_unsharded_rows")
            out[key] = o
    t = result.get("train_step")
    if isinstance(t, dict):
        o = pick(t, "ms_per_step", "losses", "f32_fallback_steps", "sp_vs_f32_loss_rel_diff_max", "error")
        if isinstance(t.get("all_gradients_f32"), dict):
            o["all_gradients_f32"] = pick(t["all_gradients_f32"], "ms_per_step", "losses", "error")
        if isinstance(t.get("with_kd"), dict):
            o["with_kd"] = pick(t["with_kd"], "ms_per_step", "kd_loss")
        out["train_step"] = o
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = pick(cb, "value", "cores", "parity_max_abs_err", "map_vs_oracle_detections")
        if isinstance(out["cpu_baseline"].get("map_vs_oracle_detections"), dict):
            out["cpu_baseline"]["map_vs_oracle_detections"].pop("note", None)
    for k in ("graph_equals_eager", "range_flags_after_run", "invalid"):
        if k in result:
            out[k] = result[k]
    return out


def launcher_command(gpus, argv):
    """`python bench.py --gpus N` launched bare (no WORLD_SIZE): the command that runs it as N ranks of one node, one
    per GPU -- the form the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    ... bench.py --gpus N ...`), --standalone so no port has to be agreed on, rendezvous on 127.0.0.1 (the container's
    host name may not resolve)."""
    rest = [a for a in argv if a not in ("--dry-launch", "--via-launcher")]
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
            "--nproc-per-node", str(gpus), os.path.abspath(__file__)] + rest


def self_launch(args, argv):
    """Re-runs this command under torch.distributed.run and relays rank 0's ONE JSON line (the ranks point their fd 1 at
    stderr for everything else, claim_stdout) and the launcher's return code."""
    import subprocess
    cmd = launcher_command(args.gpus, argv)
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "gpus": args.gpus, "cmd": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.read().decode(errors="replace").splitlines() if ln.strip()]
    rc = proc.wait()
    result = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in result[-1:]:
            print(ln, file=sys.stderr)      # anything else a rank or the launcher wrote to stdout
    if result:
        print(result[-1], flush=True)
    if rc == 0 and not result:
        print("bench: the %d-rank run printed no result line" % args.gpus, file=sys.stderr)
        rc = 1
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.cpu_baseline_child:
        return cpu_baseline_child(*args.cpu_baseline_child)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch or args.via_launcher):
        raise SystemExit(self_launch(args, argv))
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and args.gpus > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count must equal --gpus" % (args.gpus, world))
    import torch.distributed as dist
    # tests only: DN_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and uses gloo (on CUDA tensors) for the collectives --
    # RCCL refuses two ranks on one device -- so that the N > 1 control flow of this file (barriers, max over ranks, the
    # agent-sharded leg's exchange) runs end to end on a one-GPU box; the line then says `collective_backend: "gloo"`
    share_device = os.environ.get("DN_BENCH_SHARE_DEVICE") == "1"
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    launched = "TORCHELASTIC_RUN_ID" in os.environ or "LOCAL_WORLD_SIZE" in os.environ      # under torch.distributed.run
    if world == 1 and launched and args.mode != "agent":
        args.force_process_group = True      # a one-rank launch still runs the RCCL code path (barriers, max-over-ranks, agent leg)
    use_pg = world > 1 or args.force_process_group
    if world > 1 and share_device:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.force_process_group and args.mode != "agent":
        import datetime
        if "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ:
            # launched by torch.distributed.run with one rank: its agent hosts the store (an explicit tcp://
            # address would wait 10 minutes for a server nobody started)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29518", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=120))

    if args.mode == "agent":
        return agent_sharded_bench(args, world, rank, dist)
    if args.task == "seg":
        return seg_bench(args, world, rank, dist, use_pg)

    from disconet_amd import Config, DiscoNet, ops
    from disconet_amd.profiling import KernelTimer, timing
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats

    torch.manual_seed(0)
    model = DiscoNet(Config(map_hw=MAP_HW), kd_flag=0, num_agent=AGENTS)
    randomize_bn_stats(model)
    model.conv_math = args.math
    if args.in_flight > 1:      # measurement option: every replay is checksummed below
        os.environ["DISCONET_UNSAFE_OVERLAP"] = "1"
    # (DISCONET_OVERLAP=1 + DISCONET_UNSAFE_OVERLAP=1: the intra-step side branch alone, a measurement option -- the line says so)
    intra_overlap = os.environ.get("DISCONET_OVERLAP", "0") == "1" and os.environ.get("DISCONET_UNSAFE_OVERLAP", "0") == "1"
    model.overlap_streams = args.in_flight > 1 or intra_overlap        # concurrency features together, guarded below
    model.eval().cuda()
    state_dict_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    # synthetic scenes, resident in HBM before the timed region
    indices, offsets, _ = make_sparse_scene_batch(BATCH, AGENTS, MAP_HW)
    indices, offsets = indices.cuda(), offsets.cuda()
    trans = make_trans_matrices(BATCH, AGENTS, jitter_seed=rank).cuda()
    na = torch.full((BATCH, AGENTS), AGENTS, dtype=torch.int64).cuda()
    dims = (MAP_HW, MAP_HW, 13)
    n_img = AGENTS * BATCH
    na_live = na[:, 0].to(torch.int32).contiguous()     # the kernels' [B] live-agent counts, cast once (plan time)

    def step():
        # dense rebuild of the batch (K1/a2) in the layout the conv engine of the chosen mode reads
        if model.conv_math == "sp":      # occupancy words (1/32 of the float32 grid), expanded inside conv_pre_1
            bevs = _sp_bevs(ops, indices, offsets, n_img, dims)
        else:
            bevs = ops.scatter_dense(indices, offsets, n_img, dims)
        with torch.no_grad():
            return model(bevs, trans, na_live, BATCH)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
            torch.cuda.synchronize()

    graphed = {}
    launch_mode = {"mode": "eager" if args.no_graph else "hipGraph replay"}

    def run_step(i=0, depth=None):
        """One step for the un-instrumented region: a captured hipGraph of step() (the
        same launches, replayed without per-launch host work) unless --no-graph.  With
        --in-flight N, step i is replayed from graph i % N on stream i % N."""
        if args.no_graph:
            return step()
        depth = max(1, args.in_flight) if depth is None else depth
        key = model.conv_math
        if key not in graphed:
            from disconet_amd.graph import GraphedStep
            try:
                graphed[key] = [(GraphedStep(step, range_guard=False), torch.cuda.Stream()) for _ in range(max(1, args.in_flight))]      # the range flags are read once, blocking, after the timed regions (`range_flags_after_run`)
            except Exception as e:   # capture refused (driver / collective library state): run eagerly
                print("bench: hipGraph capture failed (%r); launching eagerly" % (e,), file=sys.stderr)
                torch.cuda.synchronize()
                graphed[key] = None
                launch_mode["mode"] = "eager (graph capture failed)"
        slots = graphed[key]
        if slots is None:
            return step()
        if depth == 1:
            return slots[0][0]()
        g, st = slots[i % depth]
        with torch.cuda.stream(st):
            out = g()
            if checks is not None:      # every replay in flight leaves a checksum of everything it returned
                checks[i].copy_(checksum(out))
            return out

    def checksum(out):
        """order-independent 64-bit checksum of the step's outputs (wrapping integer sum of the raw words)"""
        return out["cls"].view(torch.int64).sum() + 3 * out["loc"].view(torch.int64).sum()

    checks = None

    def timed(events, depth=None):
        nonlocal checks
        """K steps of the hot path.  events=False: nothing but the steps (-> value).
        events=True: a HIP-event pair around every launch, on the launch stream (->
        per-kernel durations); kept apart because each event record drains the queue
        (~30 us per launch), which would understate `value` by ~15 %."""
        timer = KernelTimer() if events else None
        d = max(1, args.in_flight) if depth is None else depth
        checks = torch.zeros(args.steps, dtype=torch.int64, device="cuda") if (d > 1 and not events and not args.no_graph) else None
        fence() if not events else torch.cuda.synchronize()
        t0 = time.perf_counter()
        if events:
            with timing(timer):
                for _ in range(args.steps):
                    step()
        else:
            for i in range(args.steps):
                run_step(i, depth)
        fence() if not events else torch.cuda.synchronize()
        return time.perf_counter() - t0, timer

    def roofline_of(timer, elapsed_events, math):
        summ = timer.summary()
        conv = {k: v for k, v in summ.items() if v["kernel"] in CONV_KERNELS}
        flops = sum(v["flops"] for v in conv.values())
        ms = sum(v["ms_total"] for v in conv.values())
        launches = sum(v["calls"] for v in conv.values())
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak, factor = MFMA_PEAK_TFLOPS[math], EXECUTED_FLOP_FACTOR[math]
        exec_flops = sum(v["exec_flops"] if v.get("exec_known") else factor * v["flops"] for v in conv.values())
        traffic, traffic_src = None, None
        try:   # HBM bytes per conv launch from the committed rocprofv3 --pmc passes
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                          if f.endswith("_pmc_traffic_%s.json" % math))[-1]
            traffic = json.load(open(os.path.join(ROOT, "profiles", prof)))["hbm_bytes_per_launch"]
            traffic_src = "profiles/" + prof
        except (OSError, IndexError, KeyError, ValueError):
            pass
        alg_bytes = sum(v["bytes"] for v in conv.values())
        rocprof = None
        try:   # the committed rocprofv3 --kernel-trace summary of the default command (graph replay)
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                          if f.endswith("_rocprof_conv_%s.json" % math))[-1]
            rp = json.load(open(os.path.join(ROOT, "profiles", prof)))
            rp_ms = rp["conv_ms_per_step"]
            from disconet_amd import _lib as _dl
            live_launches = launches / args.steps
            fresh = (rp.get("dn_version") == _dl.load().dn_version() and abs(rp["launches_per_step"] - live_launches) < 0.5)
            if not fresh:       # a kernel changed since the profile was committed: do not quote its durations
                raise KeyError("stale")
            rocprof = {"conv_ms_per_step": round(rp_ms, 4), "avg_launch_us": round(rp["avg_launch_us"], 2),
                       "achieved": round(flops / args.steps / (rp_ms * 1e-3) / 1e12, 3),
                       "frac": round(flops / args.steps / (rp_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[math], 4),
                       "source": "profiles/" + prof,
                       "note": "kernel durations from rocprofv3 --kernel-trace over `python bench.py` (hipGraph replay); "
                               "must agree with kernel_ms_per_step (HIP events, eager) measured live"}
        except KeyError as e:
            if e.args and e.args[0] == "stale":
                rocprof = {"stale": True, "source": "profiles/" + prof,
                           "note": "the committed rocprofv3 summary was taken with another build of the library "
                                   "(dn_version / launches per step differ from this run): not quoted"}
        except (OSError, IndexError, ValueError):
            pass
        roof = {
            "kernel": "%s (%s MFMA implicit-GEMM conv, all %d launches/step)"
                      % ("conv_sp_kernel" if math == "sp" else "conv_mfma_kernel", MATH_LABEL[math],
                         launches // args.steps),
            "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "note": "achieved = algorithmic FLOP (true channel counts) / HIP-event kernel time",
            "roofline_serial_basis": True,   # every launch timed alone on its stream (eager, event pair around it)
            # executed MFMA work per LAYER (model.py regions): 3 per product, 2 on conv_pre_1's hi-only operand, 4 of 9 taps
            # on the upsampled chunks of the tap-merged decoder layers -- not a uniform 3 x
            "executed_flop_factor": round(exec_flops / flops, 3) if flops else factor,
            "frac_executed": round(exec_flops / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 else 0.0,
            "hbm_algorithmic_TBps": round(alg_bytes / (ms * 1e-3) / 1e12, 3) if ms > 0 else None,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": round(alg_bytes / max(launches, 1)),
            "flop_per_step": flops / args.steps, "kernel_ms_per_step": round(ms / args.steps, 4),
            "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
            "ms_per_step_with_events": round(1e3 * elapsed_events / args.steps, 4),
            "other_kernels_ms_per_step": {k: round(v["ms_total"] / args.steps, 4)
                                          for k, v in summ.items() if v["kernel"] not in CONV_KERNELS},
            "rocprof": rocprof,
        }
        if args.layers:
            print("[%s] %-12s %8s %10s %9s %8s" % (math, "layer", "ms/step", "GFLOP/step", "TFLOP/s", "GB/s"),
                  file=sys.stderr)
            for k, v in summ.items():
                m = v["ms_total"] / args.steps
                print("[%s] %-12s %8.4f %10.3f %9.2f %8.1f" % (
                    math, k, m, v["flops"] / args.steps / 1e9,
                    v["flops"] / (v["ms_total"] * 1e-3) / 1e12 if v["ms_total"] else 0,
                    v["bytes"] / (v["ms_total"] * 1e-3) / 1e9 if v["ms_total"] else 0), file=sys.stderr)
        return roof

    for i in range(args.warmup):
        run_step(i)
    # The timed region is K = 20 steps = 36 ms: shorter than the clock governor's settling time after the capture /
    # warm-up phase, and the first region measured 2-3 % below the five that follow it (`repeat`).  A pre-roll of
    # untimed replays brings the part to its steady clock before the barrier; the timed region itself is unchanged
    # (exactly K steps between two fences).  Stated in the line (`pre_roll_steps`).
    for i in range(args.pre_roll):
        run_step(i)
    elapsed, _ = timed(False)                      # timed region #1 -> value
    replay_checks = checks
    # lease noise made visible inside one line: the same K steps, five more times (never part of `value`)
    repeats = None
    if args.in_flight == 1 and args.steps > 0:
        reps = sorted(timed(False)[0] for _ in range(5))
        per = [round(BATCH * world * args.steps / r, 1) for r in reps[::-1]]       # scenes/s, ascending
        repeats = {"runs": 5, "steps_each": args.steps, "scenes_per_s": {"min": per[0], "median": per[2], "max": per[4]},
                   "ms_per_step": {"min": round(1e3 * reps[0] / args.steps, 4), "median": round(1e3 * reps[2] / args.steps, 4),
                                   "max": round(1e3 * reps[4] / args.steps, 4)},
                   "note": "five more timed regions of the same K steps right after the one `value` is taken from; "
                           "box-to-box spread is larger (profiles/)"}
    elapsed_serial = timed(False, depth=1)[0] if (args.in_flight > 1 and not args.no_graph) else None
    timer, elapsed_events = None, 0.0
    if rank == 0 and not args.no_kernel_events:
        elapsed_events, timer = timed(True)        # timed region #2 -> roofline
    # every rank checks that EVERY replay it had in flight wrote the same bits as the serial replay
    outputs_same, replays_differing = None, None
    if elapsed_serial is not None and graphed.get(model.conv_math) and replay_checks is not None:
        serial_out = graphed[model.conv_math][0][0]()        # one more replay, alone on the device
        torch.cuda.synchronize()
        want = checksum(serial_out)
        replays_differing = int((replay_checks != want).sum().item())
        outputs_same = replays_differing == 0
    # the captured graph must reproduce the eager step bit for bit (profiles/r02_hazard_repro.txt part A)
    graph_ok = None
    if graphed.get(model.conv_math):
        eager = step()
        eager = {k: eager[k].clone() for k in ("cls", "loc")}
        replay = graphed[model.conv_math][0][0]()
        torch.cuda.synchronize()
        graph_ok = bool(torch.equal(replay["cls"], eager["cls"]) and torch.equal(replay["loc"], eager["loc"]))
        if not graph_ok:
            print("bench: the hipGraph replay differs from the eager step -- the figure is INVALID", file=sys.stderr)
    if use_pg:
        t = torch.tensor([elapsed, elapsed_serial or 0.0, -float(outputs_same is not False)], device="cuda",
                         dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # max time over ranks; "any rank differed" as max of -same
        elapsed = float(t[0].item())
        if elapsed_serial is not None:
            elapsed_serial = float(t[1].item())
        if outputs_same is not None:
            outputs_same = bool(t[2].item() <= -1.0)

    dtype_name = DTYPE_NAME
    # the north star's multi-GPU split rides in the same line: BASELINE configs[4], agents sharded across the ranks,
    # one RCCL all-gather per step (every rank runs it; at N = 1 with rank 0's share of an 8-rank run emulated)
    agent_res = agent_res16 = None
    if not args.no_agent_leg and args.math == "sp" and 8 % world == 0 and args.in_flight == 1:
        try:
            agent_res = agent_sharded_leg(args, world, rank, dist, emulate_world=8 if world == 1 else 0)
        except Exception as e:      # never lose the headline line to the extra leg
            agent_res = {"error": repr(e)}
        torch.cuda.synchronize()
        # configs[4] does not fix the scenes per step: at the det batch (4) a rank of an 8-rank run holds 4 images and its
        # launches are latency-bound; the same leg at 16 scenes per step shows the throughput regime
        if args.agent_batch == BATCH and not (isinstance(agent_res, dict) and "error" in agent_res):
            import copy
            args16 = copy.copy(args)
            args16.agent_batch = 16
            args16.steps, args16.warmup = max(5, args.steps // 2), min(args.warmup, 2)
            try:
                agent_res16 = agent_sharded_leg(args16, world, rank, dist, emulate_world=8 if world == 1 else 0)
            except Exception as e:      # noqa: BLE001
                agent_res16 = {"error": repr(e)}
            torch.cuda.synchronize()
        else:
            agent_res16 = None
    scenes = world * BATCH * args.steps
    result = {
        "metric": "scenes/sec (5-agent 256x256 BEV)",
        "value": round(scenes / elapsed, 3),
        "unit": "scenes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_name[args.math],
        "data": "synthetic",
        "config": {"workload": "DiscoNet det eval forward (--com disco), 5-agent, batch 4 per GPU, "
                               "256x256x13 BEV, no KD, sparse voxel lists -> dense -> enc -> "
                               "DiscoGraph fusion -> dec -> cls/reg heads",
                   "agents": AGENTS, "batch_per_gpu": BATCH, "bev": [MAP_HW, MAP_HW, 13],
                   "conv_math": args.math,
                   "launch": launch_mode["mode"] + (
                       ", %d steps in flight (alternating streams, one captured graph + buffers each)"
                       % args.in_flight if (args.in_flight > 1 and launch_mode["mode"] == "hipGraph replay") else "") + (
                       ", conv4_x as a parallel graph branch beside the fusion block (DISCONET_OVERLAP=1: a measurement option, "
                       "not the product's one-stream contract)" if (intra_overlap and args.in_flight == 1) else ""),
                   "parallelism": "scene-parallel x%d (no data-path collective)" % world},
    }

    if rank == 0:
        result["pre_roll_steps"] = args.pre_roll
        # split-f16 range guard of the timed replays: the captured step carries no poll (GraphedStep(range_guard=False): no
        # launch of the timed region is spent on it), so the sticky device flags are read here, once, blocking
        flags = ops.sp_range_flags(reset=True) if args.math == "sp" else 0
        result["range_flags_after_run"] = flags
        if flags & 5:
            result["invalid"] = "split-f16 range guard: a value was clamped / a NaN was split during the timed replays"
        if repeats is not None:
            result["repeat"] = repeats
        if graph_ok is not None:
            result["graph_equals_eager"] = graph_ok
            if not graph_ok:
                result["invalid"] = "hipGraph replay of the step differs from the eager step"
        if elapsed_serial is not None:
            if outputs_same is not None:
                result["in_flight_outputs_identical"] = outputs_same
                result["in_flight_replays_checked"] = {"replays": args.steps, "differing": replays_differing,
                                                       "how": "64-bit checksum of cls+loc after every replay, on its stream, "
                                                              "inside the timed region, vs the checksum of a replay run alone"}
                if not outputs_same:  # never report a throughput whose results are not the serial ones
                    result["value"] = round(scenes / elapsed_serial, 3)
                    result["ms_per_step"] = round(1e3 * elapsed_serial / args.steps, 4)
                    result["config"]["launch"] = "hipGraph replay (in-flight outputs differed: serial figure)"
            result["one_step_at_a_time"] = {
                "value": round(scenes / elapsed_serial, 3),
                "ms_per_step": round(1e3 * elapsed_serial / args.steps, 4),
                "note": "same K steps replayed back to back on ONE stream per rank (max over ranks): the "
                        "latency of a step; `value` overlaps consecutive, independent batches"}
        else:
            result["one_step_at_a_time"] = {"value": result["value"], "ms_per_step": result["ms_per_step"],
                                            "note": "the headline itself: one captured step replayed back to back "
                                                    "on one stream per rank (--in-flight 1, the default)"}
        if agent_res is not None:
            result["agent_sharded"] = agent_res
            if agent_res16 is not None:
                result["agent_sharded_batch16"] = agent_res16
        if timer is not None:
            result["roofline"] = roofline_of(timer, elapsed_events, args.math)
        if world == 1 and not args.no_alt_math:
            # the other conv arithmetic on the same workload, K steps each way
            alt = "f32" if args.math != "f32" else "sp"
            model.conv_math = alt
            for i in range(args.warmup):
                run_step(i)
            alt_elapsed, _ = timed(False)
            alt_res = {"conv_math": alt, "dtype": dtype_name[alt],
                       "value": round(BATCH * args.steps / alt_elapsed, 3),
                       "ms_per_step": round(1e3 * alt_elapsed / args.steps, 4)}
            if not args.no_kernel_events:
                alt_ev, alt_timer = timed(True)
                alt_res["roofline"] = roofline_of(alt_timer, alt_ev, alt)
            result["alt_math"] = alt_res
            model.conv_math = args.math
        if world == 1 and not args.no_voxelize:
            # K1 from raw points (not part of the step: the reference voxelizes offline and
            # ships sparse lists): one 60k-point cloud -> dense grid -> sorted index list
            try:
                from disconet_amd.synthetic import make_point_cloud
                pts = torch.from_numpy(make_point_cloud(60000, seed=1)).cuda()
                cfgv = Config(map_hw=MAP_HW)
                for _ in range(3):
                    dense = ops.voxelize_occupy(pts, cfgv.voxel_size, cfgv.area_extents, dims)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    dense = ops.voxelize_occupy(pts, cfgv.voxel_size, cfgv.area_extents, dims)
                e1.record()
                torch.cuda.synchronize()
                result["voxelize"] = {"points": int(pts.shape[0]), "us_per_cloud": round(50 * e0.elapsed_time(e1), 2),
                                      "occupied": int(dense.sum().item()),
                                      "note": "dn_voxelize_occupy (memset + scatter), HIP events, not in `value`"}
            except Exception as e:
                result["voxelize"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            threads = usable_cores()
            base, ref_out = cpu_baseline_bounded(state_dict_cpu, threads, args.cpu_baseline_timeout)
            if ref_out is not None:
                # the same scene through the HIP path: parity of the measured configuration
                from disconet_amd.synthetic import make_scene_batch
                bevs1, trans1, na1 = make_scene_batch(1, AGENTS, MAP_HW)
                with torch.no_grad():
                    got = model(bevs1.cuda(), trans1.cuda(), na1.cuda(), 1)
                base["default_init_max_abs_err"] = {k: float((got[k].cpu() - ref_out[k]).abs().max())
                                                    for k in ("cls", "loc")}
                par = ref_out.get("parity")
                if par is not None:
                    # parity of the measured configuration on O(1) activations, through the measured INPUT FORM at the
                    # measured batch: the child's kaiming-initialised oracle on the step's own scenes vs the HIP path
                    # loaded with the same state and fed the sparse voxel lists exactly as the timed step is
                    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices
                    pb = int(par.get("batch", 1))
                    pidx, poff, _ = make_sparse_scene_batch(pb, AGENTS, MAP_HW)
                    ptrans = make_trans_matrices(pb, AGENTS, jitter_seed=0).cuda()
                    pna = torch.full((pb, AGENTS), AGENTS, dtype=torch.int64).cuda()
                    pmodel = DiscoNet(Config(map_hw=MAP_HW), kd_flag=0, num_agent=AGENTS).eval()
                    pmodel.conv_math = args.math
                    pmodel.load_state_dict(par["state"])
                    pmodel.cuda()
                    pdims = (MAP_HW, MAP_HW, 13)
                    with torch.no_grad():
                        if args.math == "sp":
                            px = _sp_bevs(ops, pidx.cuda(), poff.cuda(), AGENTS * pb, pdims)
                            form = "sparse voxel lists -> dn_scatter_dense_bits (occupancy words)"
                        else:
                            px = ops.scatter_dense(pidx.cuda(), poff.cuda(), AGENTS * pb, pdims)
                            form = "sparse voxel lists -> dn_scatter_dense (float32 grid)"
                        gotk = pmodel(px, ptrans, pna, pb)
                    base["parity_max_abs_err"] = {k: float((gotk[k].cpu() - par[k]).abs().max()) for k in ("cls", "loc")}
                    base["parity_ref_max_abs"] = {k: float(par[k].abs().max()) for k in ("cls", "loc")}
                    base["parity_batch"] = pb
                    base["parity_input_form"] = form
                    base["parity_note"] = ("kaiming-initialised weights (logits of O(1)) for this comparison only, on the timed "
                                           "step's own scenes, batch and input form; the timed model keeps torch's default "
                                           "init, whose ~1e-2 logits make an error figure vacuous (default_init_max_abs_err)")
                    del pmodel, gotk, px
                # the metric's accuracy half: mAP of the HIP path's detections scored against
                # the oracle's detections on the same scene (decode + rotated NMS both ways)
                try:
                    from oracle import postprocess_ref as R
                    from disconet_amd import postprocess
                    cfg = Config(map_hw=MAP_HW)
                    anchors_np = R.make_anchors(cfg)
                    gts = [R.detections_from_logits(ref_out["cls"][i].numpy(), ref_out["loc"][i].numpy(),
                                                    anchors_np, pre_nms_top_k=200)[0] for i in range(AGENTS)]
                    dets = postprocess.predict_all(model, postprocess.make_anchors(cfg), bevs1.cuda(),
                                                   trans1.cuda(), na1.cuda(), 1, pre_nms_top_k=200)
                    base["map_vs_oracle_detections"] = {
                        "mAP@0.5": round(R.average_precision([d[0] for d in dets], [d[1] for d in dets], gts, 0.5), 4),
                        "mAP@0.7": round(R.average_precision([d[0] for d in dets], [d[1] for d in dets], gts, 0.7), 4),
                        "note": "random-init weights: the oracle's own detections are the ground truth"}
                except Exception as e:   # never lose the bench line to the accuracy add-on
                    base["map_vs_oracle_detections"] = {"error": repr(e)}
            result["cpu_baseline"] = base
        if world == 1 and args.train_steps > 0:
            # CoDetModule.step on the same batch (train() mode, batch statistics)
            try:
                from disconet_amd import CoDetModule
                from disconet_amd.synthetic import make_train_targets
                TRAIN_SEED = 1234      # both arithmetic modes start from the SAME initial weights
                torch.manual_seed(TRAIN_SEED)
                tmodel = DiscoNet(Config(map_hw=MAP_HW), kd_flag=0, num_agent=AGENTS)
                tmodel.conv_math = args.math
                tmodel.cuda()
                labels, targets, mask = make_train_targets(n_img, MAP_HW)
                data = {"bev_seq": ops.scatter_dense(indices, offsets, n_img, dims),
                        "trans_matrices": trans, "num_agent": na, "labels": labels.cuda(),
                        "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
                mod = CoDetModule(tmodel, lr=1e-3)
                first, last, dt, per, traj = _time_train_steps(mod, data, BATCH, args.train_steps, losses=True)
                result["train_step"] = {
                    "ms_per_step": round(1e3 * dt, 3), "scenes_per_s": round(BATCH / dt, 2),
                    "steps": args.train_steps, "batch_per_gpu": BATCH, "loss_first": round(first["loss"], 4),
                    "loss_last": round(last["loss"], 4), "losses": traj, "init_seed": TRAIN_SEED,
                    "f32_fallback_steps": getattr(mod.engine, "f32_fallback_steps", 0),
                    "step_ms": per, "warmup_steps": 4, "dgrad_math": mod.engine.dgrad_math,
                    "wgrad_math": mod.engine.wgrad_math,
                    "layers_with_a_measured_gradient_lift": len(mod.engine._dz_lift),
                    "note": "train() forward (batch-stat BN) + focal/smooth-L1 loss + explicit HIP backward (3x3 stride-1 data "
                            "gradients: split-f16 LDS-DMA engine on a lifted, pre-split dz; stride-2 data gradients: one launch of "
                            "that engine over the four parity classes, read in place by the next BatchNorm backward; 3x3 weight "
                            "gradients, stride 1 and 2: f16 MFMA on operands lifted / split / transposed while staged, "
                            "dn_conv_wgrad_sp; 1x1 gradients: fp32 MFMA) + Adam, eager launches, wall clock"}
                del mod
                # the same step with every data and weight gradient on the exact-fp32 MFMA (rounds 2-4's step)
                try:
                    torch.manual_seed(TRAIN_SEED)
                    fmodel = DiscoNet(Config(map_hw=MAP_HW), kd_flag=0, num_agent=AGENTS)
                    fmodel.conv_math = args.math
                    fmodel.cuda()
                    fmod = CoDetModule(fmodel, lr=1e-3, dgrad_math="f32", wgrad_math="f32")
                    ffirst, flast, dtf, perf32, ftraj = _time_train_steps(fmod, data, BATCH, args.train_steps, losses=True)
                    result["train_step"]["all_gradients_f32"] = {"ms_per_step": round(1e3 * dtf, 3), "scenes_per_s": round(BATCH / dtf, 2),
                                                         "loss_first": round(ffirst["loss"], 4),
                                                         "loss_last": round(flast["loss"], 4), "losses": ftraj, "step_ms": perf32,
                                                         "note": "same initial weights (init_seed) and batch as train_step: the two "
                                                                 "loss trajectories are comparable step for step"}
                    result["train_step"]["sp_vs_f32_loss_rel_diff_max"] = round(max(
                        abs(a - b) / max(abs(b), 1e-30) for a, b in zip(traj, ftraj)), 6)
                    del fmod, fmodel
                except Exception as e:
                    result["train_step"]["all_gradients_f32"] = {"error": repr(e)}
                # BASELINE configs[2]'s per-GPU step: + frozen teacher forward and the KD KL terms
                from disconet_amd import TeacherNet
                from disconet_amd.synthetic import make_bevs
                teacher = TeacherNet(Config(map_hw=MAP_HW)).cuda().eval()
                teacher.conv_math = args.math
                kmodel = DiscoNet(Config(map_hw=MAP_HW), kd_flag=1, num_agent=AGENTS)
                kmodel.conv_math = args.math
                kmodel.cuda()
                data["bev_seq_teacher"] = make_bevs(BATCH, AGENTS, MAP_HW, p=0.05).cuda()
                data["kd_weight"] = 1e5
                kmod = CoDetModule(kmodel, teacher, None, None, kd_flag=1, lr=1e-3)
                _, klast, dtk, perkd = _time_train_steps(kmod, data, BATCH, args.train_steps)
                result["train_step"]["with_kd"] = {"ms_per_step": round(1e3 * dtk, 3),
                                                   "scenes_per_s": round(BATCH / dtk, 2),
                                                   "kd_loss": round(klast["kd_loss"], 4), "step_ms": perkd}
            except Exception as e:
                result.setdefault("train_step", {})["error"] = repr(e)
        result["summary"] = line_summary(result)
        emit(result)

    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
