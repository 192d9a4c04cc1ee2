#!/usr/bin/env python3
"""bench.py -- Msamples/s of the book-one random-spheres scene, 1200x800x500 spp,
depth 100 (BASELINE.json configs[1]) on N MI355X GPUs of one node.

A "step" is one whole render of the image: every rank renders the 8x8 tiles with
tile_id % N == rank into a packed device buffer (hand-written HIP kernel through
the C ABI), one RCCL gather brings the shards to rank 0, one small kernel
un-permutes them and rank 0 copies the framebuffer to the host.  Total work is
fixed as N grows (strong scaling).  Inputs (the committed scene) are resident in
HBM before the timed region.  Consecutive steps are pipelined: a step's render kernel runs
on the render stream, everything behind it (the sums of its sample records, gather, unpack,
copy) on a second stream, beside the next step's render kernel (--no-defer switches that off);
every step's image is complete on the host inside the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks: a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The JSON line carries
  roofline      what bounds render_kernel: VALU issue.  The scene (< 1 MB) lives in LDS / L1 / L2, so HBM is not the
                roof (0.27 TB/s of real traffic = 3 % of peak; the algorithmic bytes of SURVEY.md 8(d) are kept as the
                secondary "hbm_equivalent" object).  achieved = sum over instruction classes of (wave instructions per
                launch, rocprofv3 SQ_INSTS_VALU_* of profiles/rNN_<scene>/summary.json) x (issue cycles per instruction
                per SIMD measured on the MI355X, profiles/r02_valu_issue.json) / (kernel duration measured live here
                with HIP events); peak = 1024 SIMDs x 2.4 GHz.  Beside frac: its envelope with the unclassified half of the
                instructions at 2 and at 4 cycles, f64_math_frac (binary64 arithmetic alone) and valu_busy_frac_pmc (the
                hardware's own VALU-busy share of the profiled launch: the figure that bounds).  The counts are only used
                when the profile was taken with the loaded KERNELS (rt_version kernel hash) on this workload; otherwise null.
  cpu_baseline  the oracle in reference form on the host cores (count and CPU model stated), on a bounded sample of the same
                workload, plus BASELINE configs[0] (400x225x50, depth 50) in full.
With N > 1 every rank also reports render / gather / unpack / copy times of one extra, untimed step.  For every N the line vouches
for itself: rank 0 compares the image the last step left on the host with one plain render of the whole image on its GPU
(image_matches_single_render; for N > 1 also image_matches_single_gpu) and, in the CPU-baseline leg, eight of its pixels with the
oracle at the full sample count (oracle_pixels); a mismatch is in the line AND a non-zero exit code (3 / 4).
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
N_SIMD = 256 * 4        # 256 CUs x 4 SIMDs
MAX_CLOCK_GHZ = 2.4     # same table: max clock 2400 MHz
BASELINE_CONFIGS = {("book_one", 1200, 800, 500): "configs[1]", ("cornell", 600, 600, 1000): "configs[2]",
                    ("cover", 800, 800, 1000): "configs[3]", ("book_one", 3840, 2160, 2000): "configs[4]"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--scene", default="book_one", choices=["book_one", "cornell", "cover"],
                    help="book_one is the headline workload; the others are for measuring BASELINE configs[2], [3]")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs (shards staged through host memory)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--collective-at-one", action="store_true",
                    help="rehearsal on a one-GPU box: with ONE rank, still bring the communicator up (RCCL for --backend nccl) and send "
                         "the tiles through dist.gather, as the N > 1 path does")
    ap.add_argument("--check", action="store_true", help="(the comparison with a plain render of the whole image is the default for every N since round 5; kept for old command lines)")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison of the last step's image with a plain render of the whole image")
    ap.add_argument("--ascending-tiles", action="store_true",
                    help="N > 1: hand every shard's tiles out in ascending order instead of the learnt deepest-first order (A/B)")
    ap.add_argument("--no-defer", action="store_true",
                    help="sum each step's sample records on the render stream instead of behind it (RT_FLAG_DEFERRED_OUTPUT off): "
                         "reduce_kernel (HBM-bound) then no longer overlaps the next step's render_kernel (VALU-bound)")
    ap.add_argument("--pipeline", action="store_true",
                    help="experiment: consecutive steps on two alternating streams (measured slower, see docs/experiments.md section 2)")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(desc, name, W, H, spp_full, depth, seed, target_s, scenes=None, image=None):
    """The oracle in reference form (recursive unpruned BVH, 4x4 per sprite, uv on every hit), threaded like the example
    drivers, on a bounded sample of the same workload; plus (book-one) BASELINE configs[0] -- 400x225, 50 spp, depth 50, the
    reference's own CPU-runnable case -- in full.
    image: the last image of the timed steps (host memory).  Eight of its pixels are rendered by the oracle at the FULL sample
    count, in the kernels' iterative evaluation order, and must equal it bit for bit (`oracle_pixels`): the oracle as the checker
    of the run that was just timed, after the timed region."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import oracle_binding
    o = oracle_binding.build_oracle(desc)
    cores = os.cpu_count() or 1
    checked = None
    if image is not None:
        rng = np.random.default_rng(12345)
        pix = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(8)]
        differing = []
        for x, y in pix:
            ref = o.render(W, H, spp_full, depth, seed, region=(x, y, x + 1, y + 1), iterative=True, nthreads=1)[y, x]
            if not np.array_equal(ref, image[y, x]):
                differing.append({"x": x, "y": y, "oracle": ref.tolist(), "gpu": image[y, x].tolist()})
        checked = {"checked": len(pix), "differing": len(differing), "spp": spp_full, "pixels": pix, "mismatches": differing}
    t0 = time.perf_counter()
    o.render(W, H, 1, depth, seed, nthreads=cores)
    t1 = time.perf_counter() - t0
    spp = max(1, min(64, int(target_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    o.render(W, H, spp, depth, seed, nthreads=cores)
    dt = time.perf_counter() - t0
    out = {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port", "oracle_pixels": checked,
           "sample": f"{name} {W}x{H} at {spp} spp (of {spp_full}), depth {depth}, whole image, {dt:.1f} s; "
                     f"oracle reference form, {cores} threads dealt rows y % n like examples/book-one.rs:56-65"}
    if scenes is not None and desc.name == "book-one":
        w0, h0, s0, d0 = 400, 225, 50, 50
        o0 = oracle_binding.build_oracle(scenes.book_one(1, w0 / h0))
        t0 = time.perf_counter()
        o0.render(w0, h0, s0, d0, seed, nthreads=cores)
        dt0 = time.perf_counter() - t0
        out["configs0_full"] = {"value": w0 * h0 * s0 / dt0 / 1e6, "unit": "Msamples/s", "seconds": dt0,
                                "workload": f"book-one random-spheres {w0}x{h0}, {s0} spp, depth {d0} (BASELINE.json configs[0]), the whole render"}
    return out


def issue_roofline(rt, scene, W, H, spp, depth, world, kernel_ms, lane_util):
    """VALU-issue roofline of render_kernel from the committed rocprofv3 summary of this workload, if it belongs to this build."""
    out = {"bound": "valu_issue", "achieved": None, "peak": N_SIMD * MAX_CLOCK_GHZ, "unit": "Gcycle/s (VALU issue cycles, all 1024 SIMDs)",
           "frac": None, "useful_frac": None, "traffic": None, "source": None}
    # the newest committed profile of this scene (profiles/rNN_<scene>/summary.json)
    cands = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{scene}/summary.json"), reverse=True)
    prof = cands[0] if cands else ROOT / "profiles" / f"r04_{scene}" / "summary.json"
    if world != 1:
        out["reason"] = "instruction counts are profiled on the whole image (N = 1) only"
        return out
    if not prof.exists():
        out["reason"] = f"no committed profile {prof.relative_to(ROOT)}"
        return out
    s = json.load(open(prof))
    want = f"--scene {scene} --width {W} --height {H} --spp {spp}"
    if want not in s.get("command", "") or depth != 100:
        out["reason"] = f"{prof.relative_to(ROOT)} was taken on another workload ({s.get('command')})"
        return out
    # the profile belongs to this build when the DEVICE code is the same (kernel hash of rt_version); host-side changes do not matter
    prof_kernels = s["library"].split("kernels ", 1)[1].split(" ", 1)[0] if "kernels " in (s.get("library") or "") else None
    if s.get("library") != rt.version() and prof_kernels != rt.kernel_hash():
        out["reason"] = f"{prof.relative_to(ROOT)} was taken with other kernels ({s.get('library')}) than the loaded {rt.version()}"
        return out
    v = s.get("valu_issue_roofline")
    if not v:
        out["reason"] = "profile has no instruction-class counters"
        return out
    issue = v["issue_cycles_total"]                      # issue cycles per launch (class counts x measured prices)
    achieved = issue / (kernel_ms * 1e-3) / 1e9          # with the kernel duration measured live in this run
    per_s = 1.0 / (kernel_ms * 1e-3) / 1e9 / out["peak"]  # issue cycles per launch -> fraction of the chip's issue slots
    ic = v["issue_cycles_per_launch"]
    n_other = v["wave_instructions_per_launch"]["other"]
    rest = issue - ic["other"]
    pmc_util = s.get("valu_lane_utilisation")  # SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU of the profiled launch: every VALU instruction
    pm = s.get("pmc") or {}
    busy = (pm.get("SQ_ACTIVE_INST_VALU") or {}).get("mean_per_dispatch")
    out.update({"achieved": achieved, "frac": achieved / out["peak"], "useful_frac": achieved / out["peak"] * (pmc_util or lane_util),
                # the hardware exposes no counter for moves / selects / compares / lane operations (the SQ_INSTS_VALU_* classes are
                # ADD / MUL / FMA / TRANS per type, INT32, INT64, CVT), so the unclassified half of the instructions is priced by a
                # static mix; the envelope prices it at the cheapest (2) and the dearest (4 cycles) such instruction instead
                "frac_envelope_other_at_2_and_4_cycles": [(rest + 2.0 * n_other) * per_s, (rest + 4.0 * n_other) * per_s],
                # binary64 arithmetic alone (ADD / MUL / FMA / TRANS f64 issue cycles over the chip's issue slots): the part of
                # `frac` that is the reference's own arithmetic; it rises when bookkeeping instructions are removed
                "f64_math_frac": (ic["f64"] + ic["trans_f64"]) * per_s,
                # the hardware's own word: quad-cycles a SIMD had a VALU instruction in flight x 4 / (1024 SIMDs x elapsed cycles) of
                # the profiled launch.  It bounds: ~1 means no VALU cycle is left, whatever the instruction prices say; the gap to
                # `frac` is what the nominal prices do not see (dependent-issue stalls, 4.2- instead of 4-cycle forms)
                "valu_busy_frac_pmc": (busy * 4.0 / (N_SIMD * v["elapsed_shader_cycles"])) if busy and v.get("elapsed_shader_cycles") else None,
                "useful_frac_vote_blocks_only": achieved / out["peak"] * lane_util,
                "traffic": (s.get("hbm_traffic_bytes_per_launch") or {}).get("total"),
                "source": f"{prof.relative_to(ROOT)} (rocprofv3 --pmc passes of `{s['command']}`, build {rt.build_hash()}); "
                          f"prices profiles/r02_valu_issue.json; kernel_ms live (HIP events)",
                "issue_cycles_per_launch": issue, "wave_instructions_per_launch": v["wave_instructions_per_launch"],
                "issue_cycles_per_instruction": v["issue_cycles_per_instruction"],
                "profiled_kernel_ms": s.get("render_kernel_avg_ms"), "profiled_clock_ghz": v.get("clock_ghz"),
                "profiled_frac_at_measured_clock": v.get("frac"), "valu_lane_utilisation_pmc": s.get("valu_lane_utilisation"),
                "lds_conflict_share": s.get("lds_conflict_share_of_lds_active"), "wave_cycle_shares": s.get("wave_cycle_shares"),
                "note": "frac = VALU issue slots of the whole chip filled during the launch; useful_frac weights it with the VALU lane "
                        "utilisation of the profiled launch (PMC, every instruction: divergence inside a block included); "
                        "useful_frac_vote_blocks_only with the occupancy of the wave-vote blocks alone (live counters)"})
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (this process has
    not touched the GPU and never execs), relay the one JSON line of rank 0 and the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free port for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")  # (what RCCL needs is set by every rank itself: rank_environment)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # stderr passes through
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in r.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if r.returncode == 0 and not lines:
        print("[bench] the ranks exited 0 but rank 0 printed no result line", file=sys.stderr)
        return 1
    return r.returncode


def rank_environment():
    """What a rank needs in its environment, set by the rank ITSELF before torch (and with it the HIP runtime) is imported -- the
    same for ranks started by the driver's `python -m torch.distributed.run ... bench.py` and for ranks started by spawn_ranks:
    one environment for both launch paths.  HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf
    IPC; without it sharing device memory between processes (RCCL's peer-to-peer transport) fails with
    `hipIpcGetMemHandle: invalid argument`.  An operator's own setting wins."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (what the rank really ran with goes into the line: config.rank_environment.  The failure without the variable is the pool's
    # documented behaviour -- the task environment exports it on every box -- not something this repository has a log of.)
    return {k: os.environ[k] for k in sorted(os.environ) if k.startswith(("HSA_", "NCCL_", "RCCL_", "HIP_VISIBLE", "ROCR_VISIBLE"))}


def predicted_strong_scaling(n, ascending=False):
    """profiles/r05_shard_scaling*.log (tools/shard_scaling.py): speed-up over N = 1 predicted from per-shard times measured on ONE
    MI355X (1/N shard of the image, render + sums, before the gather), in the learnt tile order or in ascending order: a prediction
    printed beside the measurement, never instead of it"""
    return ({1: 1.0, 2: 1.96, 4: 3.80, 8: 7.09} if ascending else {1: 1.0, 2: 1.97, 4: 3.84, 8: 7.23}).get(n)


PREDICTION_KERNELS = "114bb391c73d10b8"  # the kernel hash (rt_version) profiles/r05_shard_scaling*.log were measured with


def main():
    a = parse()
    rank_env = rank_environment()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            raise SystemExit(spawn_ranks(a.gpus))
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    if a.same_device:
        local_rank = 0
    # a launcher that shows every rank only its own GPU (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per process) leaves one device,
    # index 0, whatever LOCAL_RANK says
    n_visible = torch.cuda.device_count()
    if n_visible < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if local_rank >= n_visible:
        if n_visible != 1:  # 1 < visible < ranks: folding would put several ranks on some GPUs silently
            raise SystemExit(f"LOCAL_RANK {local_rank} but only {n_visible} devices are visible: one rank per GPU (or --same-device for a rehearsal)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or a.collective_at_one
    if collective:
        if "MASTER_PORT" not in os.environ:  # (only without a launcher: --collective-at-one)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        # the communicator comes up -- and is exercised once -- BEFORE any render, so that an RCCL failure cannot be
        # mistaken for a kernel failure (and the other way round)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if a.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                probe = torch.ones(1)
            dist.all_reduce(probe)
            if a.backend == "nccl":
                torch.cuda.synchronize()
            assert int(probe.item()) == world
        except Exception as e:  # noqa: BLE001
            print(f"[bench rank {rank}] communicator ({a.backend}) failed before any render: {e!r}", file=sys.stderr, flush=True)
            raise

    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    W, H, spp, depth = a.width, a.height, a.spp, a.depth
    if a.scene == "book_one":
        desc = scenes.book_one(a.scene_seed, W / H)
    elif a.scene == "cornell":
        desc = scenes.cornell(W / H)
    else:
        desc = scenes.cover(a.scene_seed, W / H)
    sc, cam = scenes.build_product(desc, device=local_rank)
    info = sc.info()

    n_tiles = [rt.shard_tile_count(W, H, r, world) for r in range(world)]
    pad_tiles = max(n_tiles)
    # --pipeline: two of everything a step writes and consecutive steps on alternating streams, so that the render of
    # step k+1 could fill the CUs while step k's last few 100-segment paths, its reduce, gather and unpack finish (the
    # library keeps two render slots per scene).  Measured SLOWER than one stream on the MI355X, so it is off by default
    # default (deferred output): ONE render stream -- render kernels never overlap each other -- and everything behind a render
    # kernel (the sums of its sample records, gather, unpack, copy) on a second stream, overlapping the next step's render
    defer = not a.no_defer and not a.pipeline
    # N > 1: a shard's tiles go out deepest first, in an order the library learns from the first render of the view (the warm-up)
    order_flag = rt.RT_FLAG_ASCENDING_TILES if a.ascending_tiles else 0
    n_buf = 2 if (a.pipeline or defer) else 1
    mines = [torch.zeros(pad_tiles * 64 * 3, dtype=torch.float64, device=dev) for _ in range(n_buf)]
    gathereds = [torch.zeros(world * pad_tiles * 64 * 3, dtype=torch.float64, device=dev) if rank == 0 else None for _ in range(n_buf)]
    # default: the current stream, as ever; only the --pipeline experiment creates side streams
    render_streams = [torch.cuda.Stream(device=dev) for _ in range(n_buf)] if a.pipeline else [torch.cuda.current_stream()]
    mine = mines[0]
    # rank 0: double-buffered image + a copy stream, so the device->host copy of step k overlaps the render of
    # step k+1 (every copy still completes inside the timed region: sync() waits for all streams)
    images = [torch.zeros(H * W * 3, dtype=torch.float64, device=dev) for _ in range(2)] if rank == 0 else None
    host_images = [torch.zeros(H * W * 3, dtype=torch.float64).pin_memory() for _ in range(2)] if rank == 0 else None
    copy_stream = torch.cuda.Stream(device=dev) if rank == 0 else None
    copy_done = [torch.cuda.Event() for _ in range(2)] if rank == 0 else None
    post_stream = torch.cuda.Stream(device=dev) if defer else None
    post_done = [torch.cuda.Event() for _ in range(2)] if defer else None
    step_no = [0]

    def step(marks=None):
        """one render of the whole image; marks (optional) collects CUDA events after render / gather / unpack / copy"""
        k = step_no[0]
        step_no[0] += 1
        rs = render_streams[k % len(render_streams)]
        mine_k, gathered_k = mines[k % n_buf], gathereds[k % n_buf]

        def mark(name, stream_obj):
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream_obj)
                marks.append((name, e))

        if defer:
            rs.wait_event(post_done[k % 2])  # step k - 2 has finished reading mine_k / gathered_k
            mark("start", rs)
            sc.render_tiles_device(cam, W, H, spp, depth, a.seed, (rank, world), mine_k.data_ptr(), None, rs.cuda_stream,
                                   flags=rt.RT_FLAG_DEFERRED_OUTPUT | order_flag)
            mark("render", rs)
            sc.wait_output(post_stream.cuda_stream)  # the post stream waits for this step's sums
            rs = post_stream
        with torch.cuda.stream(rs):
            stream = rs.cuda_stream
            if not defer:
                mark("start", rs)
                sc.render_tiles_device(cam, W, H, spp, depth, a.seed, (rank, world), mine_k.data_ptr(), None, stream, flags=order_flag)
                mark("render", rs)
            if collective and a.backend == "nccl":
                glist = list(gathered_k.chunk(world)) if rank == 0 else None
                dist.gather(mine_k, glist, dst=0)  # the one collective of the path (RCCL over xGMI)
            elif collective:
                hm = mine_k.cpu()
                hl = [torch.zeros_like(hm) for _ in range(world)] if rank == 0 else None
                dist.gather(hm, hl, dst=0)
                if rank == 0:
                    gathered_k.copy_(torch.cat(hl))
            mark("gather", rs)
            if rank == 0:
                b = k & 1
                src = gathered_k if collective else mine_k
                rs.wait_event(copy_done[b])  # the copy that last read images[b] has finished
                rt.unpack_tiles_device(src.data_ptr(), pad_tiles, world, W, H, images[b].data_ptr(), stream)
                mark("unpack", rs)
                unpacked = torch.cuda.Event()
                unpacked.record(rs)
                copy_stream.wait_event(unpacked)
                with torch.cuda.stream(copy_stream):
                    host_images[b].copy_(images[b], non_blocking=True)
                    copy_done[b].record(copy_stream)
                    mark("d2h", copy_stream)
            if defer:
                post_done[k % 2].record(rs)

    def sync():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    last_kernel_ms = sc.last_kernel_ms()
    tile_order_mode = sc.last_launch_config()["tile_order"]  # of the last timed step

    # one more, UNTIMED step with events after every stage (per-rank anatomy of a step)
    marks = []
    step(marks)
    sync()
    anatomy = {}
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        anatomy[n1 + "_ms"] = e0.elapsed_time(e1)
    anatomy["render_kernel_ms"] = sc.last_kernel_ms()
    anatomy["tile_order"] = tile_order_mode

    # per-launch kernel duration measured with HIP events on the launch stream: time each launch separately, untimed loop
    # ... and ONE render on its own, nothing beside it: render_kernel, then the sums of its sample records on the same stream
    # (no following render to hide them behind): what a caller who renders a single image gets
    per_launch, single = [], []
    for _ in range(max(1, min(a.steps, 3))):
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        sc.render_tiles_device(cam, W, H, spp, depth, a.seed, (rank, world), mine.data_ptr(), None, cur.cuda_stream, flags=order_flag)
        e1.record(cur)
        torch.cuda.synchronize()
        per_launch.append(sc.last_kernel_ms())
        single.append(e0.elapsed_time(e1))
    kernel_avg_ms = float(np.mean(per_launch))
    single_render_ms = float(np.mean(single))
    launch_cfg = sc.last_launch_config()

    tmax = torch.tensor([dt, kernel_avg_ms, single_render_ms], dtype=torch.float64, device=dev)
    per_rank = None
    if collective:
        if a.backend == "gloo":
            tmax = tmax.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_rank = [None] * world if rank == 0 else None
        dist.gather_object(dict(anatomy, rank=rank, device=local_rank, tiles=n_tiles[rank]), per_rank, dst=0)
    dt = float(tmax[0])
    kernel_avg_ms = float(tmax[1])
    single_render_ms = float(tmax[2])

    if rank == 0:
        total_samples = W * H * spp
        value = a.steps * total_samples / dt / 1e6
        # algorithmic bytes per sample from the kernel's own traversal counters,
        # measured on the same image at reduced spp with the counting build of the same kernel
        cspp = min(spp, 64)
        _, cnt = sc.render(cam, W, H, cspp, depth, a.seed, counters=True)
        ns = max(1, cnt["samples"])
        bytes_per_sample = (cnt["nodes_visited"] * info["node_bytes"] + cnt["prims_tested"] * info["prim_bytes"]
                            + cnt["segments"] * info["material_bytes"]) / ns + 24.0 / spp
        launch_samples = total_samples / world
        hbm_equiv = bytes_per_sample * launch_samples / (kernel_avg_ms * 1e-3) / 1e9
        util = {b: cnt[b + "_lane"] / max(1, 64 * cnt[b + "_wave"]) for b in ("node", "leaf", "shade")}
        cyc = {b: cnt[b + "_cycles"] / max(1, cnt["node_cycles"] + cnt["leaf_cycles"] + cnt["shade_cycles"])
               for b in ("node", "leaf", "shade", "finish", "refill", "begin", "swap")}
        lane_util = sum(util[b] * cyc[b] for b in ("node", "leaf", "shade"))  # cycle-weighted lane utilisation of the vote blocks
        roof = issue_roofline(rt, a.scene, W, H, spp, depth, world, kernel_avg_ms, lane_util)
        roof.update({
            "kernel": "render_kernel", "kernel_ms": kernel_avg_ms, "launch": launch_cfg,
            "hbm_equivalent": {"algorithmic_bytes_per_sample": bytes_per_sample, "achieved_GBs": hbm_equiv, "peak_GBs": HBM_PEAK_GBS,
                               "frac": hbm_equiv / HBM_PEAK_GBS,
                               "note": "SURVEY.md 8(d) algorithmic bytes (node steps x 64 B + prim tests x 32 B + segments x 48 B + 24 B / spp); "
                                       "they are served by LDS / L1 / L2, so this is NOT a bound (it exceeds the HBM peak)"},
            "segments_per_sample": cnt["segments"] / ns, "nodes_per_sample": cnt["nodes_visited"] / ns,
            "prims_per_sample": cnt["prims_tested"] / ns, "simd_utilisation": util, "lane_utilisation_cycle_weighted": lane_util,
            "block_executions_per_sample": {b: cnt[b + "_wave"] * 64 / ns for b in ("node", "leaf", "shade")},
            "block_cycle_share": cyc,
            "counters_note": "simd_utilisation, block_executions, block_cycle_share and swap_at_shade come from the COUNTING instantiation "
                             "of the same kernel on this image at min(spp, 64) spp: wall-clock shares by s_memtime, and that build keeps "
                             "76-176 bytes of scratch per lane at the same VGPR budget (ray-tracer_amd/csrc/_obj/resource_usage.txt); "
                             "instruction-level figures (issue roofline, lane utilisation by PMC) are rocprofv3's on the timed build",
            "swap_at_shade": ({k[5:]: cnt[k] for k in cnt if k.startswith("swap_")} if cnt.get("swap_scattered") else None)})
        cfg_name = BASELINE_CONFIGS.get((a.scene, W, H, spp))
        scene_words = {"book_one": "book-one random-spheres", "cornell": "cornell-box", "cover": "book-two cover (main.rs)"}[a.scene]
        res = {
            "metric": f"Msamples/sec (pixels x spp), book-one {W}x{H}x{spp}spp" if a.scene == "book_one" else f"Msamples/sec (pixels x spp), {a.scene} {W}x{H}x{spp}spp",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{scene_words} {W}x{H}, {spp} spp, depth {depth}" + (f" (BASELINE.json {cfg_name})" if cfg_name else ""),
                       "scene_seed": a.scene_seed, "render_seed": a.seed, "n_prims": info["n_prims"],
                       "bvh_nodes": info["n_nodes"], "sharding": f"8x8 tiles, tile_id % {world}, one RCCL gather",
                       "collective": (f"dist.gather over {a.backend}" + (" (RCCL)" if a.backend == "nccl" else "") + f", {world} rank(s)") if collective else None,
                       "library": rt.version(), "rank_environment": rank_env, "steps_pipelined_on_two_streams": bool(a.pipeline),
                       "sums_behind_the_next_render": bool(defer),
                       "tile_order": {0: "ascending" + (" (a whole image always is)" if world == 1 else ""),
                                      1: "learnt from the warm-up's path lengths: deepest tiles first",
                                      2: "ascending (still learning: fewer than two warm-up steps)"}[tile_order_mode]},
            # `value` is the pipelined figure (a step's sums run beside the next step's render); one image rendered alone:
            "single_render_ms": single_render_ms, "single_render_msamples_per_s": total_samples / (single_render_ms * 1e-3) / 1e6,
            "single_render_note": "one rt_render_tiles_device on its own (render_kernel + reduce_kernel on one stream, slowest rank), "
                                  "without the gather / unpack / copy that `value` includes and without a following render to hide the sums",
            "roofline": roof, "wall_s": dt, "last_kernel_ms": last_kernel_ms,
            "step_anatomy_ms": per_rank if per_rank is not None else [dict(anatomy, rank=0, device=local_rank, tiles=n_tiles[0])],
        }
        # The line vouches for itself: the image the LAST step left on the host (the untimed anatomy step: the same code path as the
        # timed ones) against one plain rt_render of the whole image on this GPU -- for N > 1 that is the sharded image against a
        # single-GPU one -- and, with the CPU baseline, eight of its pixels against the oracle at the full sample count.
        image_ok = oracle_ok = True
        last = host_images[(step_no[0] - 1) & 1].numpy().reshape(H, W, 3)
        if not a.no_check:
            whole = sc.render(cam, W, H, spp, depth, a.seed)
            image_ok = bool(np.array_equal(last, whole))
            res["image_matches_single_render"] = image_ok
            if collective or a.check:
                res["image_matches_single_gpu"] = image_ok
        if per_rank is not None:
            # where the time of a step goes on each rank, and how even the shards are (the slowest rank is the step)
            rms = [r["render_ms"] for r in per_rank]
            res["ranks"] = {"render_ms": {"min": min(rms), "max": max(rms), "mean": sum(rms) / len(rms), "spread": max(rms) - min(rms)},
                            "gather_ms_rank0": per_rank[0].get("gather_ms"), "unpack_ms_rank0": per_rank[0].get("unpack_ms"),
                            "d2h_ms_rank0": per_rank[0].get("d2h_ms"),
                            # (a prediction made with other kernels than the loaded ones is not printed beside a measurement: ADVICE r4)
                            "predicted_speedup_over_1_gpu": predicted_strong_scaling(world, a.ascending_tiles) if rt.kernel_hash() == PREDICTION_KERNELS else None,
                            "prediction_source": f"tools/shard_scaling.py: per-shard kernel times on one MI355X, before the gather, kernels {PREDICTION_KERNELS}"
                                                 + ("" if rt.kernel_hash() == PREDICTION_KERNELS else f" (loaded: {rt.kernel_hash()}: prediction withheld)")}
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(desc, scene_words, W, H, spp, depth, a.seed, a.cpu_seconds, scenes, image=last)
            res["oracle_pixels"] = {k: res["cpu_baseline"]["oracle_pixels"][k] for k in ("checked", "differing", "spp")}
            oracle_ok = res["oracle_pixels"]["differing"] == 0
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    else:
        image_ok = oracle_ok = True
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if not image_ok:  # the line above says so; the exit code does too (the launcher relays it)
        print("[bench] the image of the last step differs from a plain single-GPU render of the whole image", file=sys.stderr, flush=True)
        raise SystemExit(3)
    if not oracle_ok:
        print("[bench] pixels of the timed image differ from the oracle at the full sample count (oracle_pixels in the line)", file=sys.stderr, flush=True)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
