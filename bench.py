#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on its headline configuration.

  metric : Mpixels/s encode+decode (4K sRGB, 5-3 lossless)
  step   : --inflight (default 3) independent 3840x2160 RGB8 frames, each on its own context / HIP stream so that the
           latency-bound kernels of one frame (VLC walk, small DWT levels, scans) overlap the bandwidth-bound kernels
           of another; per frame: 512x512 tiles (40 tile-triples), DC shift + RCT + 5-level 5-3
           DWT + HT block coding of every code-block (64x64, reference job order) + stream compaction,
           then HT decode of every block + inverse DWT/RCT/DC shift back to pixels.  Inputs are resident
           in HBM before the timed region.  N>1: weak scaling, every rank codes its own frame per step
           and the compressed streams are gathered to rank 0 over RCCL (dist.gather_streams).
  prints : ONE JSON line (rank 0) with the roofline of the dominant kernel (level-0 5-3 DWT, timed with
           HIP events on the library's own stream) and a CPU baseline (the C oracle, 1 thread).

  python bench.py --gpus N --steps K --warmup W
      N > 1 and no WORLD_SIZE in the environment: this process -- before it has touched the GPU -- starts
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process
      (one rank per GPU over RCCL), passes rank 0's JSON line through and exits with the child's status.  Under
      torch.distributed.run (WORLD_SIZE set) it is a rank; WORLD_SIZE != --gpus is an error.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd"))
os.environ.setdefault("J2K_TUNING", "1")     # bench.py is also the A/B harness: the library honours its J2K_* switches only with this set

sys.path.insert(0, ROOT)
import bench_extra  # noqa: E402  (workload definitions, CPU baseline, the C3 / C5 runner, the tile-sharded mode)
import bench_host   # noqa: E402  (the pinned-host-memory boundary, the closed-loop codec)

W, H, C, TILE, NRES, CB, PREC = 3840, 2160, 3, 512, 6, 64, 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SEED = 0x4A324B30              # "J2K0" 
MIN_TIMED_S = 1.0              # floor of the timed region, whatever --steps says (round 4: 0.1 s -- box-to-box spread of 5 % on 0.09 s of samples)
# sha256 of the int32 `decoded` buffer (every code-block of frame 0 as HTDecoder.Decode returns it, job order) -- what the
# block decoder must have produced in the timed region; tests/test_bench_digest.py recomputes it with the oracle on CPU
DECODED_SHA256 = "76eab55b1f63e2eb3644d138c6d655d4b16975c46310378f3f9f1bc509de7d5e"
# HBM traffic of one level-0 launch from the rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB -> bytes);
# cannot be collected from inside this process, so it is the committed measurement (see profiles/)
def _traffic(fetch_kib, write_kib, src):
    """(bytes, text) from the two per-launch PMC averages of the committed summary: ONE pair of numbers, the text is made from them
    (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> doubled; MI355X_MICROARCH.md)"""
    return int(round((2 * fetch_kib + write_kib) * 1024)), "%s: (2 x %.1f + %.1f) KiB" % (src, fetch_kib, write_kib)


TRAFFIC = {   # frame_io -> (bytes per level-0 forward launch, source); tests/test_bench_traffic_constant.py re-reads the summary
    "planes": _traffic(52495.8, 97200.0, "profiles/r01_bench_v3_inflight1_pmc_{FETCH,WRITE}_SIZE.csv"),
    "rgba8": _traffic(16433.0, 97281.6, "profiles/r05_bench_inflight1_pmc_summary.txt"),
}
# HBM bytes ALL kernels of one C2 frame move (same PMC passes, one frame in flight): sum over the nine kernels of a step of
# 2 x FETCH_SIZE + WRITE_SIZE; tests/test_bench_traffic_constant.py re-adds them from the committed summary
PIPELINE_KERNELS = ("dwt53_fwd_rgba8_wg_kernel", "dwt53_deep_fwd_kernel", "ht_encode_kernel", "gather_scan_kernel", "ht_vlcprep_kernel", "ht_walk_kernel",
                    "ht_decode_kernel", "dwt53_deep_inv_kernel", "dwt53_inv_rgba8_wg_kernel")
PIPELINE_TRAFFIC = (int(round(485558.8 * 1024)), "profiles/r05_bench_inflight1_pmc_summary.txt: sum of (2 x FETCH_SIZE + WRITE_SIZE) over %d kernels" % len(PIPELINE_KERNELS))
COPY_PEAK_GUIDE_GBS = 6290.0   # MI355X_MICROARCH.md: the float4 device-to-device copy the guide measured (what a pure copy reaches)


def synth_frame(np, index):
    """smooth gradient + seeded uniform noise in [-16,16], clamped to 0..255"""
    rng = np.random.default_rng(SEED + index)
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // max(W, H)])
    return np.clip(base + rng.integers(-16, 17, size=(C, H, W)), 0, 255).astype(np.int32)


def cpu_baseline(budget_s=10.0):
    """The C oracle on one host thread and on all host cores, same per-tile pipeline as the GPU step (bench_extra)."""
    return bench_extra.cpu_baseline("c2", index=0, budget_s=budget_s)


def launch_ranks(n, argv):
    """--gpus N without a launcher: start the N ranks as a fresh child process tree and relay.  Called before anything in this
    process has initialised the GPU (no torch.cuda / HIP call yet), and it never replaces this process (no exec): the child
    inherits stdout, so rank 0's JSON line goes straight through."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", str(port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush(); sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def other_configs_summary(budget_s=200.0):
    """One driver-visible record for the configurations the headline line does not cover: C3, C5 and C1-on-the-GPU
    each run as `bench.py --config X` in a FRESH CHILD PROCESS started before this process touches the GPU (C3 / c1gpu need their
    own GPU_MAX_HW_QUEUES, which the runtime reads when it initialises), short step counts, the CPU baseline's one-thread leg
    only.  A child that fails or runs out of time is reported as such; the headline line does not depend on any of them."""
    import subprocess
    out = {}
    # On a fresh box the first `import torch` pages the installation in and can take a minute or two; done HERE once (importing torch does
    # not initialise the GPU), so that the children's budget is spent on their runs and not on the first child's import
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    t_all = time.perf_counter()
    plan = [("c3", ["--config", "c3", "--steps", "5", "--warmup", "1"]), ("c5", ["--config", "c5", "--steps", "20", "--warmup", "3"]),
            ("c1gpu", ["--config", "c1gpu", "--steps", "3", "--warmup", "1"]),
            ("c4", ["--config", "c4", "--shard", "tiles", "--steps", "30", "--warmup", "3"]),
            ("closed_loop", ["--config", "cl", "--steps", "3", "--warmup", "1"]),
            ("closed_loop_ht", ["--config", "clht", "--steps", "50", "--warmup", "5"]),
            ("host_boundary", ["--io", "host", "--steps", "50", "--warmup", "3"])]
    for name, extra in plan:
        left = budget_s - (time.perf_counter() - t_all)
        if left < 10:
            out[name] = {"error": "skipped: the %.0f s budget of other_configs was used up" % budget_s}
            continue
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-s", "2.5", "--cpu-baseline-1t"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=min(left, 60.0))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": "exit %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
                continue
            d = json.loads(lines[-1])
            if name == "host_boundary":
                out[name] = dict(d["host_boundary"], wall_s=round(time.perf_counter() - t0, 1))
                continue
            if name in ("c4", "closed_loop", "closed_loop_ht"):
                out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                             "n_gpus": d["n_gpus"], "scaling": d["scaling"], "wall_s": round(time.perf_counter() - t0, 1)}
                for k in ("value_with_d2h", "codestream_bytes", "tiles", "parallelism", "frames_in_flight", "contexts", "frames_per_context", "codestream_bytes_per_frame", "tiles_parsed_packet_parallel", "single_frame_ms", "round_trip"):
                    if k in d["config"]:
                        out[name][k] = d["config"][k]
                continue
            out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                         "frames_in_flight": d["config"].get("frames_in_flight"), "contexts": d["config"].get("contexts"),
                         "frames_per_context": d["config"].get("frames_per_context"), "hw_queues": d["config"].get("hw_queues"),
                         "dtype": d["dtype"], "self_check": d["config"].get("self_check"),
                         "roofline_kernel": d["roofline"]["kernel"], "roofline_frac": d["roofline"]["frac"],
                         "roofline_avg_launch_us": d["roofline"]["avg_launch_us"],
                         "cpu_1_thread": (d.get("cpu_baseline") or {}).get("value_1_thread"), "cpu_unit": "Mpixels/s",
                         "wall_s": round(time.perf_counter() - t0, 1)}
            for k in ("single_frame_ms", "kernel_ms"):
                if k in d["config"]:
                    out[name][k] = d["config"][k]
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out after %.0f s" % min(left, 60.0)}
        except Exception as exc:                            # noqa: BLE001
            out[name] = {"error": repr(exc)[:300]}
    return out


def run(state):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; > 1 without a launcher: bench.py starts them itself")
    ap.add_argument("--no-other-configs", action="store_true", help="default C2 run at N = 1: do not append the short C3 / C5 / c1gpu records")
    ap.add_argument("--cpu-baseline-s", type=float, default=0.0, help="seconds per leg of the CPU baseline (0: the configuration's default)")
    ap.add_argument("--cpu-baseline-1t", action="store_true", help="CPU baseline: the one-thread leg only")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=-1, help="--config c3|c5: replay each frame's plan calls as one HIP graph (1), launch them one by one (0); -1: the configuration's default")
    ap.add_argument("--io", choices=["planes", "rgba8", "host"], default=os.environ.get("J2K_BENCH_IO", "rgba8"),
                    help="frame format at the boundary: packed 8-bit RGBA pixels (image.RGBA.Pix) read / written directly by "
                         "the level-0 kernels (extractImageData / createImage fused, SURVEY 8f rank 2; the default), or "
                         "int32 component planes (e.componentData, the boundary of SURVEY 8a-e); 'host': the same step with pixels and "
                         "tile-parts in PINNED HOST memory, copies overlapped with the kernels (bench_host.run_host_boundary)")
    ap.add_argument("--decode-rows", choices=["coded", "all"], default=os.environ.get("J2K_BENCH_DECODE_ROWS", "coded"),
                    help="HT block decode: 'coded' = j2k_plan_set_decode_coded_rows_only -- the rows the reference's decoder never "
                         "writes are left alone in a buffer zeroed once before the run (the pooled HTDecoder, ht.go:1393-1429); "
                         "'all' = every call writes all w x h samples of every block (a fresh NewHTDecoder).  Same buffer contents, "
                         "same digest, either way")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("J2K_BENCH_INFLIGHT", "3")),
                    help="independent frames coded concurrently per step, each on its own context/stream")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5", "c1gpu", "cl", "clht"], default="c2",
                    help="BASELINE.json configuration: c2 (default, the headline: 4K RGB8 5-3 + HT), c3 (4K RGB 12-bit, 9-7 lossy + MQ), "
                         "c5 (2048x2048 gray16 frames, 5-3 + HT); c4 only with --shard tiles; cl = the closed-loop codec (4K RGB8, MQ coder, "
                         "pixels -> tile-parts of packets -> pixels, bit-exact)")
    ap.add_argument("--batch", type=int, default=0, help="--config c1gpu: frames per context and call, coded as the tiles of one plan (0: the configuration's default)")
    ap.add_argument("--d2h", action="store_true", help="--shard tiles: copy the finished tile-parts to pinned host memory inside the timed step")
    ap.add_argument("--shard", choices=["frames", "tiles"], default="frames",
                    help="N > 1: every rank codes its own frames (weak scaling, the default) or ONE frame's tiles are sharded over "
                         "the ranks, gathered and assembled into tile-parts on rank 0 (strong scaling, C4 geometry)")
    state["skip_teardown"] = True                       # until this process is a rank: nothing to tear down before that
    args = ap.parse_args()
    ws = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(ws) if ws else 1
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if ws is not None and int(ws) != args.gpus:
        print("bench.py: WORLD_SIZE=%s but --gpus %d: launch with --nproc-per-node %d, or drop the launcher and let --gpus start the ranks"
              % (ws, args.gpus, args.gpus), file=sys.stderr)
        raise SystemExit(2)
    if ws is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    state["skip_teardown"] = False
    # The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 unless set): with more frames in
    # flight than queues, frames share a queue and their kernels run one after the other.  The MQ coder's kernels run at the
    # latency of their longest chain on a tenth of the device and want every frame on a queue of its own (C3, six frames: 181 ms
    # per step with 4 queues, 95 with 8 or more; twenty frames want more than 16).  The bandwidth-bound kernels of the HT
    # configurations want the opposite: C5's eight frames in flight run at 72-74 Gpixel/s on 4 or 8 queues and at 41 on 16 or 32
    # (eight frames' transforms truly at once evict each other's lines).  Read when the runtime initialises, i.e. before the first
    # torch.cuda call.
    if args.config in ("c3", "c1gpu"):
        if "GPU_MAX_HW_QUEUES" not in os.environ and any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
            # a profiler's preloaded library has initialised the runtime before this line: the variable set here is not read
            print("bench.py: GPU_MAX_HW_QUEUES was not in the environment when the HIP runtime initialised (profiler preload): this run "
                  "uses the runtime's 4 hardware queues -- export GPU_MAX_HW_QUEUES=32 in the shell instead", file=sys.stderr)
            os.environ["J2K_BENCH_HWQ_LATE"] = "1"
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    import faulthandler
    faulthandler.enable(all_threads=True)   # a native crash in a rank prints every thread's Python stack
    if args.shard == "tiles":
        if args.config == "c2" and "--config" not in sys.argv:
            args.config = "c4"
        args.inflight = 1
        return bench_extra.run_shard_tiles(args)
    if args.config in ("cl", "clht"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
        if "--inflight" not in " ".join(sys.argv):
            args.inflight = 0
        if "--steps" not in " ".join(sys.argv):
            args.steps, args.warmup = (50, 5) if args.config == "clht" else (3, 1)
        return bench_host.run_closed_loop(args)
    if args.io == "host":
        # two copy streams + the lanes' library streams, each on a hardware queue of its own (a copy stream that shares a queue with the
        # other direction's, or with a lane's kernels, takes turns with it: 28.6 instead of 48.6 GB/s each way on these boxes)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        # ROCr's "recommended engine" choice per copy now and then puts the H2D and the D2H copy on ONE SDMA engine, and the two then share
        # its 57 GB/s (28.6 each way) instead of running on an engine each (48.6 each way): with it off every segment of the run is in the
        # fast mode (5.0-5.6 -> 6.0-6.15 Gpixel/s over a second).  Read when the HSA runtime initialises: set before torch is imported.
        os.environ.setdefault("HSA_ENABLE_SDMA_RECOMMENDED_ENG", "0")
        if "--steps" not in " ".join(sys.argv):
            args.steps, args.warmup = 50, 3
        return bench_host.run_host_boundary(args)
    if args.config != "c2":
        if "--inflight" not in " ".join(sys.argv):
            args.inflight = 0           # the configuration's own default
        if args.config == "c1gpu" and "--steps" not in " ".join(sys.argv):
            args.steps, args.warmup = 5, min(args.warmup, 1)       # a step is 22 x 16 frames of 21 serial chains: 0.7 s
        return bench_extra.run_config(args, args.config)
    # CPU baseline first (N = 1 only): nothing has touched the GPU yet, so the worker processes are plain forks / spawns
    cpu_base = None
    others = None
    solo = int(os.environ.get("WORLD_SIZE", "1")) == 1
    plain = solo and not any(os.environ.get(k, "0") not in ("0", "") for k in ("J2K_BENCH_PEER_REHEARSAL", "J2K_BENCH_ROOT_REHEARSAL"))
    if plain and not args.no_other_configs and not args.no_cpu_baseline and args.io == "rgba8":
        others = other_configs_summary()
    if solo and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(args.cpu_baseline_s or 10.0)

    import numpy as np
    import torch
    import torch.distributed as dist
    from j2kgfx import CODER_HT, Context
    from j2kgfx.codec import FramePlan
    from j2kgfx import dist as jdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if os.environ.get("J2K_BENCH_BACKEND", "nccl") != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    # J2K_BENCH_PEER_REHEARSAL=1 (dev, one GPU): run the N > 1 step as a PEER would -- pack, wait for the encodes only,
    # size exchange -- with a one-rank group and nothing to transfer, to see what the host side of that step costs
    pr_mode = os.environ.get("J2K_BENCH_PEER_REHEARSAL", "0") if world == 1 else "0"   # 1: all of it; 2: no pack; 3: no exchange
    peer_rehearsal = pr_mode in ("1", "2", "3", "4", "5")   # 4: pack, but the sizes are not read back; 5: as 1, and the lone
    # rank sends its packs to itself through RCCL and rebuilds them (the transfer calls, on one GPU)
    # J2K_BENCH_ROOT_REHEARSAL=N (dev, one GPU): the GPU work of rank 0 in an N-GPU run -- its own frames plus the rebuild of
    # N - 1 peers' streams per frame slot (its own pack stands in for theirs) -- with nothing transferred
    root_n = int(os.environ.get("J2K_BENCH_ROOT_REHEARSAL", "0")) if world == 1 else 0
    multi = world > 1 or peer_rehearsal
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # J2K_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share
        # devices; the streams are staged through host memory for the gather) -- not a measurement mode
        backend = os.environ.get("J2K_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            # the per-step size exchange goes through a gloo group: host integers, no GPU kernel in the host's way
            size_group = dist.new_group(backend="gloo") if os.environ.get("J2K_BENCH_SIZE_GROUP", "gloo") == "gloo" else None
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            size_group = None

    F = max(1, args.inflight)
    frame_h = synth_frame(np, rank)

    NSETS = 3     # N > 1: stream / pack buffer sets in rotation (see step())

    class Lane:   # one frame in flight: its own context (= HIP stream), plan and buffers
        def __init__(self):
            self.ctx = Context(local)
            self.plan = FramePlan(W, H, C, precision=PREC, lossless=True, num_resolutions=NRES, cb=(CB, CB), tile=(TILE, TILE),
                                  coder=CODER_HT, ctx=self.ctx, track_streams=False)   # buffers live for the whole run
            p, i = self.plan, self.plan.info
            self.n = int(i.blocks)
            self.frame = torch.from_numpy(frame_h).to(p.device)
            if args.io == "rgba8":     # image.RGBA Pix of the same frame (alpha 255)
                rgba = np.concatenate([frame_h.transpose(1, 2, 0), np.full((H, W, 1), 255, np.int32)], axis=2).astype(np.uint8)
                self.pix = torch.from_numpy(np.ascontiguousarray(rgba.reshape(H, W * 4))).to(p.device)
                self.back_pix = torch.empty_like(self.pix)
            self.coeff = p.alloc_coeff()
            # two sets of stream buffers at N > 1: step k + 1 encodes into one set while RCCL still reads step k's
            nb = NSETS if multi else 1
            self.streams = [p.empty(i.bytes_cap, torch.uint8) for _ in range(nb)]
            self.lenss = [p.empty(self.n, torch.int32) for _ in range(nb)]
            self.numbpss = [p.empty(self.n, torch.uint8) for _ in range(nb)]
            self.offss = [p.empty(self.n + 1, torch.int64) for _ in range(nb)]
            self.stream, self.lens, self.numbps, self.offs = self.streams[0], self.lenss[0], self.numbpss[0], self.offss[0]
            self.decoded = torch.zeros(max(int(i.decoded_elems), 4), dtype=torch.int32, device=p.device)   # (padding between blocks stays 0: digest)
            if args.decode_rows == "coded":
                p.set_decode_coded_rows_only(True)     # `decoded` was zeroed just above, once: the uncoded rows stay zero
            self.back = p.alloc_frame()
            self.gather_bufs = [None] * nb
            # N > 1: what travels is the transport form of the stream (j2k_plan_pack_stream: the blocks without the
            # reference's MEL zero runs, a third of the bytes); rank 0 rebuilds every peer's dense stream from it
            self.packs = [p.empty(p.pack_bound(), torch.uint8) for _ in range(nb)] if multi else []
            self.assembled = None
            if pr_mode == "5":
                self.assembled = [(p.empty(i.bytes_cap, torch.uint8), p.empty(self.n + 1, torch.int64), p.empty(self.n, torch.int32),
                                   p.empty(self.n, torch.uint8))]
            if root_n > 1:
                self.packs = [p.empty(p.pack_bound(), torch.uint8)]
                self.assembled = [(p.empty(i.bytes_cap, torch.uint8), p.empty(self.n + 1, torch.int64), p.empty(self.n, torch.int32),
                                   p.empty(self.n, torch.uint8)) for _ in range(root_n - 1)]
            if world > 1 and rank == 0:
                self.assembled = [(p.empty(i.bytes_cap, torch.uint8), p.empty(self.n + 1, torch.int64), p.empty(self.n, torch.int32),
                                   p.empty(self.n, torch.uint8)) for _ in range(world - 1)]

        def code_set(self, S):
            """the same step on another set of buffers (the rotating-footprint roofline pass, rank 0 after the timed region)"""
            p = self.plan
            p.forward_rgba8(S["pix"], S["coeff"])
            p.encode_stream(S["coeff"], S["stream"], S["offs"], S["lens"], S["numbps"])
            p.decode_blocks(S["stream"], S["offs"], S["lens"], S["numbps"], S["decoded"])
            p.inverse_rgba8(S["coeff"], S["back_pix"])

        def buffer_set(self):
            p, i = self.plan, self.plan.info
            return dict(pix=self.pix.clone(), coeff=p.alloc_coeff(), stream=p.empty(i.bytes_cap, torch.uint8), offs=p.empty(self.n + 1, torch.int64),
                        lens=p.empty(self.n, torch.int32), numbps=p.empty(self.n, torch.uint8),
                        decoded=torch.zeros(max(int(i.decoded_elems), 4), dtype=torch.int32, device=p.device), back_pix=torch.empty_like(self.pix))

        def encode_side(self, b=0):
            p = self.plan
            if args.io == "rgba8":
                p.forward_rgba8(self.pix, self.coeff)
            else:
                p.forward(self.frame, self.coeff)
            p.encode_stream(self.coeff, self.streams[b], self.offss[b], self.lenss[b], self.numbpss[b])   # block coding + compaction
            if multi and (rank != 0 or pr_mode in ("1", "3", "4", "5")):
                p.pack_stream(self.streams[b], self.offss[b], self.lenss[b], self.numbpss[b], self.packs[b])

        def decode_side(self, b=0):
            p = self.plan
            p.decode_blocks(self.streams[b], self.offss[b], self.lenss[b], self.numbpss[b], self.decoded)
            if args.io == "rgba8":
                p.inverse_rgba8(self.coeff, self.back_pix)
            else:
                p.inverse(self.coeff, self.back)

        def code(self):
            self.encode_side()
            self.decode_side()

    lanes = [Lane() for _ in range(F)]
    state["lanes"], state["multi"] = lanes, multi
    # Rank 0 also rebuilds the streams of the other N - 1 ranks (F per rank and step).  From five GPUs on that is more work
    # than coding a frame (measured with J2K_BENCH_ROOT_REHEARSAL: rank 0 at 75 % of the others' rate at N = 8), so rank 0
    # then codes F - 1 frames per step and keeps its last slot for receiving only; `value` counts the frames actually coded.
    root_idle = int(os.environ.get("J2K_BENCH_ROOT_IDLE_SLOTS", "1" if world >= 5 else "0")) if (world > 1 and rank == 0 and F > 1) else 0
    root_idle_all = int(os.environ.get("J2K_BENCH_ROOT_IDLE_SLOTS", "1" if world >= 5 else "0")) if (world > 1 and F > 1) else 0
    for ln in lanes:
        ln.codes = True
    for ln in lanes[F - root_idle:]:
        ln.codes = False
    ctx, plan = lanes[0].ctx, lanes[0].plan
    info, n = plan.info, lanes[0].n
    plan_pack_bound = plan.pack_bound()
    ext = torch.cuda.ExternalStream(ctx.stream)

    exts = [torch.cuda.ExternalStream(ln.ctx.stream) for ln in lanes]

    enc_done = [[torch.cuda.Event() for _ in lanes] for _ in range(NSETS)]
    size_pin = [torch.zeros(len(lanes), dtype=torch.int64).pin_memory() for _ in range(NSETS)]
    size_done = [torch.cuda.Event() for _ in range(NSETS)]
    copy_stream = torch.cuda.Stream()                   # torch-owned: pinned memory stays away from the library's streams
    unpacked = [[torch.cuda.Event() for _ in lanes] for _ in range(NSETS)]
    keep = [None] * NSETS
    pending = [None] * NSETS        # the gather that still reads buffer set b
    coded = [False] * NSETS         # set b holds a step's streams whose exchange has not been started yet
    stepno = [0]

    # ---- the transfers themselves: j2k_gather_streams at the C ABI (RCCL send / recv on the library's own stream; what a Go
    #      host would call), or torch.distributed's batched isend / irecv (J2K_BENCH_GATHER=torch; always for the gloo rehearsal)
    backend_name = os.environ.get("J2K_BENCH_BACKEND", "nccl")
    use_cabi = multi and backend_name == "nccl" and os.environ.get("J2K_BENCH_GATHER", "cabi") == "cabi" and (world > 1 or pr_mode == "5")
    comm, recv_bufs = None, [None] * NSETS
    if use_cabi:
        try:
            comm = jdist.Comm(lanes[0].ctx, rank, world, group=size_group)
            state["comm"] = comm
            if rank == 0:
                cap = max(world - 1, 1) * F * (plan_pack_bound + 16) + 64
                recv_bufs = [torch.empty(cap, dtype=torch.uint8, device=lanes[0].plan.device) for _ in range(NSETS)]
        except Exception as exc:                        # (a box without a usable RCCL for the library: say so, use torch's)
            print("bench.py: j2k_comm_create failed (%s); falling back to torch.distributed transfers" % exc, file=sys.stderr)
            use_cabi, comm = False, None
    if use_cabi:
        # One small exchange through the new communicator before anything is timed, with a time limit, and the ranks agree on
        # the outcome: the driver's N > 1 runs are the first time this path meets a second device, and a transfer that fails
        # or never completes must cost the C-ABI path, not the whole record.
        import threading
        probe = {"ok": False, "err": None}
        def _probe():
            try:
                n = 4096
                mine = torch.full((n,), rank + 1, dtype=torch.uint8, device=lanes[0].plan.device)
                rb = torch.zeros(world * n + 64, dtype=torch.uint8, device=lanes[0].plan.device) if rank == 0 else None
                torch.cuda.synchronize()
                counts = np.full((world, 1), n, np.uint64)
                offs_ = comm.gather([mine], [n], recv=rb, producers=(), all_bytes=counts, self_loop=(pr_mode == "5"))
                comm.wait(None)
                good = True
                if rank == 0:
                    for r in ([0] if pr_mode == "5" else range(1, world)):
                        o = int(offs_[r])
                        good = good and bool((rb[o:o + n] == r + 1).all().item())
                probe["ok"] = good
            except Exception as exc:                    # noqa: BLE001
                probe["err"] = exc
        th = threading.Thread(target=_probe, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("J2K_BENCH_PROBE_S", "60")))
        flag = torch.tensor([1 if (not th.is_alive() and probe["ok"]) else 0], dtype=torch.int32)
        if world > 1:
            if size_group is not None:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=size_group)
            else:
                flag = flag.to(lanes[0].plan.device); dist.all_reduce(flag, op=dist.ReduceOp.MIN); flag = flag.cpu()
        if os.environ.get("J2K_BENCH_PROBE_FAIL"):      # (test hook: take the fall-back)
            flag[0] = 0
        if int(flag.item()) == 0:
            why = "timed out" if th.is_alive() else ("failed: %s" % probe["err"] if probe["err"] else "failed on some rank")
            print("bench.py: rank %d: the probe exchange through j2k_gather_streams %s; falling back to torch.distributed transfers"
                  % (rank, why), file=sys.stderr)
            state["probe_abandoned"] = th.is_alive()
            if not th.is_alive():
                try:
                    comm.close()
                except Exception:                       # noqa: BLE001
                    pass
            use_cabi, comm = False, None
            state["comm"] = None
            recv_bufs = [None] * NSETS
    gather_path = "j2k_gather_streams (C ABI: ncclSend/ncclRecv peer->root on the library's stream)" if use_cabi else (
        "torch.distributed batch_isend_irecv" if multi else None)

    class CabiGather:
        def __init__(self, offs, totals):
            self.offs, self.totals = offs, totals
        def wait(self):
            out = []
            peers = [0] if pr_mode == "5" else list(range(1, world))
            for f in range(F):
                pks = []
                if rank == 0:
                    for r in peers:
                        o, nb_ = int(self.offs[r * F + f]), int(self.totals[r][f])
                        pks.append(self.buf[o:o + nb_])
                out.append(pks)
            return out

    def finish_gather(b):
        """The transfers that read buffer set b are complete (and the library streams know it)."""
        xdone[b].wait()                                 # the helper has started (or never had) this set's exchange
        if xerr:
            raise xerr[0]
        g = pending[b]
        if g is None:
            return
        res = g.wait()
        if isinstance(g, CabiGather):
            for ln in lanes:                            # every library stream waits, on the device, for the transfers
                comm.wait(ln.ctx)
            for ln, pks in zip(lanes, res):
                if pks:
                    ln.plan.unpack_streams(pks, ln.assembled)
            keep[b] = res
            if rank == 0:
                for ev, e in zip(unpacked[b], exts):
                    ev.record(e)
            pending[b] = None
            return
        staged = []
        for ln, (buf, offsets) in zip(lanes, res):
            ln.gather_bufs[b] = buf
            if pr_mode == "5":
                pks = [buf[:int(offsets[1])]]
                staged.append(pks)
            elif rank == 0:
                pks = [buf[int(offsets[r]):int(offsets[r + 1])] for r in range(1, world)]
                if not buf.is_cuda:                     # gloo rehearsal: the packs arrive in host memory
                    pks = [pk.to(ln.plan.device) for pk in pks]
                staged.append(pks)
        cur = torch.cuda.current_stream()
        for e in exts:                                  # the library streams wait for the transfers (and any staging copy)
            e.wait_stream(cur)
        for ln, pks in zip(lanes, staged):              # rank 0: rebuild every peer's dense stream (+ offsets, lengths, bit planes)
            if pks:
                ln.plan.unpack_streams(pks, ln.assembled)   # one launch for the N - 1 packs of this frame slot
        keep[b] = staged                                # alive until this set's next turn
        if rank == 0:                                   # the next gather into this receive buffer waits for these unpacks
            for ev, e in zip(unpacked[b], exts):
                ev.record(e)
        pending[b] = None

    def exchange(b):
        """Exchange of the step whose streams are in set b: its encodes are awaited (an event per library stream), the
        sizes exchanged, and all frames' packs handed to RCCL in one batch of peer->root transfers."""
        for ev in enc_done[b]:
            ev.synchronize()                            # the bytes are complete before RCCL reads them
        if rank != 0 or pr_mode in ("1", "5"):
            # the packs' lengths (their first words) -> pinned host memory by ASYNCHRONOUS copies on a torch-owned stream and
            # an event wait: a blocking read-back here (hipMemcpy) holds up the kernel launches of the main thread
            with torch.cuda.stream(copy_stream):
                for f, ln in enumerate(lanes):
                    size_pin[b][f:f + 1].copy_(ln.packs[b][:8].view(torch.int64), non_blocking=True)
                size_done[b].record(copy_stream)
            size_done[b].synchronize()
        if pr_mode == "3":
            return
        # rank 0's own frames stay where they are (it sends nothing); a peer sends the pack, whose first word is its length
        if rank == 0 and pr_mode not in ("1", "5"):
            sizes = [0] * len(lanes)                    # (also the dev modes 2 and 4)
        else:                                           # written by the asynchronous copies queued before enc_done
            sizes = size_pin[b].tolist()
        if use_cabi:
            # every rank's byte counts through the gloo group (host integers), then ONE j2k_gather_streams for the F packs
            mine = torch.tensor([int(v) for v in sizes], dtype=torch.int64)
            totals = torch.empty(world * F, dtype=torch.int64)
            if world > 1 and size_group is not None:
                dist.all_gather_into_tensor(totals, mine, group=size_group)
            elif world > 1:
                td = torch.empty(world * F, dtype=torch.int64, device=lanes[0].plan.device)
                dist.all_gather_into_tensor(td, mine.to(td.device))
                totals = td.cpu()
            else:
                totals = mine
            if rank == 0 and keep[b] is not None:
                for ev in unpacked[b]:                  # the unpacks that read this receive buffer three steps ago (long done)
                    ev.synchronize()
            offs_ = comm.gather([ln.packs[b] for ln in lanes], [int(v) for v in sizes], recv=recv_bufs[b] if rank == 0 else None,
                                all_bytes=totals.numpy(), self_loop=pr_mode == "5")
            g = CabiGather(offs_, totals.numpy().reshape(world, F))
            g.buf = recv_bufs[b]
            pending[b] = g
            return
        items = [(ln.packs[b], int(sz)) for ln, sz in zip(lanes, sizes)]
        if os.environ.get("J2K_BENCH_BACKEND", "nccl") != "nccl":
            items = [(t[:n_].cpu(), n_) for t, n_ in items]
        if rank == 0 and keep[b] is not None:
            cur = torch.cuda.current_stream()
            for ev in unpacked[b]:
                cur.wait_event(ev)
        pending[b] = jdist.gather_streams_start(items, outs=[ln.gather_bufs[b] for ln in lanes], size_group=size_group,
                                                self_loop=pr_mode == "5")

    # The exchange runs on a helper thread: its waits (encode events, the size exchange) would otherwise sit between two
    # batches of kernel launches of the main thread, and at ~0.5 ms of GPU work per step the host has no slack for that
    # (measured on one GPU with J2K_BENCH_PEER_REHEARSAL: 52.2 -> 46.0 Gpx/s inline).  Every rank's helper handles the
    # steps in the same order, so the collectives line up.
    import queue
    import threading
    xq = queue.Queue()
    xdone = [threading.Event() for _ in range(NSETS)]
    xerr = []
    for ev_ in xdone:
        ev_.set()

    def helper():
        torch.cuda.set_device(local)
        while True:
            b = xq.get()
            if b is None:
                return
            try:
                exchange(b)
            except BaseException as exc:                # surfaced by the main thread at its next wait
                xerr.append(exc)
            finally:
                xdone[b].set()

    xthread = threading.Thread(target=helper, daemon=False)   # joined by the teardown, also on the error path
    state["helper"], state["stop_helper"] = xthread, (lambda: xq.put(None))
    if multi:
        xthread.start()

    def start_gather(b):
        if not coded[b]:
            return
        coded[b] = False
        xdone[b].clear()
        xq.put(b)

    def step():
        if root_n > 1:
            for ln in lanes:
                ln.encode_side()
                ln.plan.pack_stream(ln.stream, ln.offs, ln.lens, ln.numbps, ln.packs[0])
                ln.plan.unpack_streams([ln.packs[0]] * (root_n - 1), ln.assembled)
                ln.decode_side()
            return
        if not multi:
            for ln in lanes:
                ln.code()
            return
        # N > 1, software-pipelined over three sets of stream buffers.  Step k queues its encodes (+ packs) and decodes on
        # the library streams into set k % 3 and only THEN runs the exchange of step k - 1 (whose encodes finished while
        # the host was queueing), so the host never waits for work it has just launched; the transfers of step k - 1 run
        # under the kernels of steps k and k + 1, and set (k - 1) % 3 is rewritten at step k + 2, after its gather finished.
        k = stepno[0]
        stepno[0] += 1
        b = k % NSETS
        finish_gather(b)
        for ln, ev, e in zip(lanes, enc_done[b], exts):
            if ln.codes:
                ln.encode_side(b)
                ev.record(e)
        for ln in lanes:
            if ln.codes:
                ln.decode_side(b)
        coded[b] = True
        start_gather((k - 1) % NSETS)

    def barrier():
        for b in range(NSETS):
            start_gather(b)
        for b in range(NSETS):
            finish_gather(b)
        for ln in lanes:
            ln.ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # The timed region lasts at least MIN_TIMED_S whatever --steps says (a 20-step C2 run is 8 ms: thinner than a record
    # should be): a few calibration steps, every rank takes the largest step count any rank needs, `steps` in the JSON line
    # is the number actually timed and `steps_requested` the flag.
    steps_requested = args.steps
    tc = time.perf_counter()
    for _ in range(5):
        step()
    barrier()
    need = int(np.ceil(MIN_TIMED_S / max((time.perf_counter() - tc) / 5, 1e-6)))
    if world > 1:
        tn = torch.tensor([need], dtype=torch.int64, device=plan.device if os.environ.get("J2K_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
        dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        need = int(tn.item())
    args.steps = max(args.steps, min(need, 20000))
    timed_profile = os.environ.get("J2K_BENCH_TIMED_PROFILE", "1") != "0"
    if timed_profile:
        ctx.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt_issue = time.perf_counter() - t0                  # host time to queue the steps (well below dt: the device is the bound)
    e1.record(ext)
    barrier()
    dt = time.perf_counter() - t0
    launches, k_ms = ctx.profile_read() if timed_profile else (0, 0.0)
    ctx.profile_enable(False)
    # ---- roofline pass: the same step with ONE frame in flight, so the dominant kernel's duration is its own (with
    #      several frames in flight it shares the chip with the other frames' kernels) ----
    iso_launches, iso_ms = 0, 0.0
    tags = [(0, 0.0)] * 4
    if rank == 0:
        for _ in range(3):
            lanes[0].code()
        ctx.sync()
        ctx.profile_enable(2)                     # every transform dispatch stamps its own begin / end, tagged by level group
        for _ in range(30):
            lanes[0].code()
        ctx.sync()
        tags = [ctx.profile_read_tag(t) for t in range(4)]   # forward level 0 / deeper levels, inverse level 0 / deeper levels
        iso_launches, iso_ms = ctx.profile_read()
        ctx.profile_enable(False)
        # ---- the same pass over FOUR buffer sets in rotation (> 1 GB touched between two uses of a line): one lane's ~250 MB fit
        #      the 256 MB Infinity Cache, and FETCH / WRITE_SIZE count fabric requests whether or not the MALL serves them
        #      (MI355X_MICROARCH.md) -- does the solo fraction hold at a footprint that rules the cache out?
        rot_tags, rot_n = None, 0
        if args.io == "rgba8" and world == 1 and os.environ.get("J2K_BENCH_ROTATE", "1") != "0":
            try:
                sets = [lanes[0].buffer_set() for _ in range(4)]
                torch.cuda.synchronize()
                for S_ in sets:
                    lanes[0].code_set(S_)
                ctx.sync()
                ctx.profile_enable(2)
                for k_ in range(32):
                    lanes[0].code_set(sets[k_ % 4])
                ctx.sync()
                rot_tags = [ctx.profile_read_tag(t) for t in range(4)]
                rot_n, _ = ctx.profile_read()
                ctx.profile_enable(False)
                for S_ in sets:
                    assert torch.equal(S_["back_pix"], lanes[0].pix), "rotating-set round trip failed"
                del sets
            except torch.cuda.OutOfMemoryError:
                rot_tags = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=plan.device if os.environ.get("J2K_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness of what was timed (outside the timed region) ----
    nocheck = bool(os.environ.get("J2K_DEV_SKIP"))       # dev experiments that leave kernels out: the line they print is not a result
    for ln in lanes:
        if not ln.codes or nocheck:
            continue
        if args.io == "rgba8":
            assert torch.equal(ln.back_pix, ln.pix), "lossless round trip failed"
        else:
            assert torch.equal(ln.back, ln.frame), "lossless round trip failed"
    total_bytes = int(lanes[0].offs[n].item())
    # the block decoder's output of the timed region (it is not what the inverse transform consumed, see config.workload)
    decoded_sha = None
    if rank == 0 and world == 1 and lanes[0].codes:
        import hashlib
        decoded_sha = hashlib.sha256(lanes[0].decoded[:int(info.decoded_elems)].cpu().numpy().tobytes()).hexdigest()
        if DECODED_SHA256 and not DECODED_SHA256.startswith("%") and not nocheck:
            assert decoded_sha == DECODED_SHA256, "decoded blocks differ from the committed digest (tests/test_bench_digest.py)"
    if pr_mode == "5":                                   # the packs came back through RCCL and were rebuilt: same bytes
        ln = lanes[0]
        a_s, a_o, a_l, a_n = ln.assembled[0]
        assert torch.equal(a_o[:n + 1], ln.offs[:n + 1]) and torch.equal(a_l[:n], ln.lens[:n]) and torch.equal(a_n[:n], ln.numbps[:n])
        assert torch.equal(a_s[:total_bytes], ln.stream[:total_bytes]), "self-loop stream differs"
    if world > 1 and rank == 0:
        # what rank 0 assembled for rank 1's frame == that frame coded here, byte for byte (stream, offsets, lengths, bit planes)
        ln = lanes[0]
        fr1 = torch.from_numpy(synth_frame(np, 1)).to(plan.device)     # kept alive until the sync: the calls are asynchronous
        torch.cuda.synchronize()
        co = plan.forward(fr1)
        s_, o_, l_, n_ = plan.encode_stream(co)
        ctx.sync()
        a_s, a_o, a_l, a_n = ln.assembled[0]
        tot1 = int(o_[n].item())
        assert torch.equal(a_o[:n + 1], o_[:n + 1]) and torch.equal(a_l[:n], l_[:n]) and torch.equal(a_n[:n], n_[:n]), "gathered arrays differ"
        assert torch.equal(a_s[:tot1], s_[:tot1]), "gathered stream differs"

    if rank == 0:
        px = W * H
        ms_step = dt / args.steps * 1e3
        k_conc_s = (k_ms / max(launches, 1)) * 1e-3            # inside the timed region (F frames in flight)
        k_avg_s = (iso_ms / max(iso_launches, 1)) * 1e-3        # roofline pass (one frame in flight)
        # level 0 moves 3 int32 planes in + 3 out (24 B/px) or, with packed pixels, one RGBA8 dword in + 3 int32 out (16 B/px)
        alg_bytes = int(info.dwt_level0_bytes) if args.io == "planes" else int(info.dwt_level0_bytes) * 16 // 24
        achieved = alg_bytes / k_avg_s / 1e9 if iso_launches else 0.0
        # the WHOLE transform, each way: sum of the algorithmic bytes of every level (2 s w_l h_l per level and
        # plane; level 0 with packed pixels 16 B/px) over the sum of the kernel durations of all its dispatches per frame
        tr_bytes = int(info.dwt_bytes) - int(info.dwt_level0_bytes) + alg_bytes
        nfr = max(iso_launches, 1)
        us = [t[1] * 1e3 / nfr for t in tags]                 # per frame: fwd level 0, fwd deeper, inv level 0, inv deeper
        def frac(b, t_us):
            return round(b / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if t_us > 0 else None
        rot = None
        if rot_tags and rot_n:
            rus = [t[1] * 1e3 / rot_n for t in rot_tags]
            rot = {"buffer_sets": 4, "frames": int(rot_n), "frac": frac(alg_bytes, rus[0]), "inv_level0_frac": frac(alg_bytes, rus[2]),
                   "transform_fwd_frac": frac(tr_bytes, rus[0] + rus[1]), "transform_inv_frac": frac(tr_bytes, rus[2] + rus[3]),
                   "us": [round(v, 2) for v in rus],
                   "note": "the solo roofline pass repeated over four buffer sets in rotation (> 1 GB touched between two uses of any line): "
                           "what the fractions are when the 256 MB Infinity Cache cannot hold a frame's buffers from one step to the next"}
        frames_per_step = world * F - root_idle_all
        out = {
            "metric": "Mpixels/s encode+decode (4K sRGB, 5-3 lossless)",
            "value": round((world * F - root_idle_all) * px / (dt / args.steps) / 1e6, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "steps_requested": steps_requested, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "host_issue_ms_per_step": round(dt_issue / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "3840x2160 sRGB 8-bit, 512x512 tiles, 5-3 lossless + HT block coder, 64x64 code-blocks, "
                                   "6 resolutions (BASELINE configs[1]); frames_in_flight independent frames per rank per step, "
                                   "each on its own HIP stream; a step = forward transform + HT block coding + stream compaction, then HT "
                                   "block decode of that stream (checked against a committed sha256) + inverse transform of the ENCODER's "
                                   "coefficients back to pixels (checked equal to the input): two halves, not a round trip through the "
                                   "decoded blocks -- the reference's HT coder codes one row in four (SURVEY fact 3) and its decoder has no "
                                   "packet->plane placement to mirror; N>1 gathers the compressed streams to rank 0 over RCCL (sent without "
                                   "the reference's MEL zero runs, rebuilt byte for byte at rank 0 inside the timed region)",
                       "decoded_sha256": decoded_sha,
                       "tiles": int(info.tiles), "code_blocks": n, "compressed_bytes_per_frame": total_bytes,
                       "frames_in_flight": F, "frames_in_flight_rank0": F - root_idle_all, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "frame_io": args.io,
                       "decode_rows": args.decode_rows + (" (the rows the reference's HT decoder never writes are not re-zeroed on every call: the "
                                                          "buffer was zeroed once, as a pooled HTDecoder's slice is; digest unchanged)" if args.decode_rows == "coded" else ""),
                       "parallelism": "frames/rank" if world > 1 else "single GPU", "gather": gather_path,
                       "ranks_seen": world,
                       "gather_check": ("rank 0's rebuilt stream, offsets, lengths and bit-plane counts of rank 1's frame == that frame coded on rank 0, "
                                        "byte for byte (asserted after the timed region)") if world > 1 else None},
            "roofline": {"bound": "hbm",
                         "kernel": ("dwt53_fwd_rgba8_wg_kernel (level 0: RGBA8 unpack + DC shift + RCT + 5-3 lifting, fused)" if args.io == "rgba8"
                                    else "dwt53_fwd_kernel<8,3,true,false,false> (level 0: DC shift + RCT + 5-3 lifting, fused)"),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": TRAFFIC[args.io][0],
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "inv_level0_frac": frac(alg_bytes, us[2]), "inv_level0_us": round(us[2], 2),
                         "transform_fwd_frac": frac(tr_bytes, us[0] + us[1]), "transform_inv_frac": frac(tr_bytes, us[2] + us[3]),
                         "transform_algorithmic_bytes": tr_bytes,
                         "transform_fwd_us": [round(us[0], 2), round(us[1], 2)], "transform_inv_us": [round(us[2], 2), round(us[3], 2)],
                         "transform_dispatches_per_frame": [round(t[0] / nfr, 2) for t in tags],
                         "avg_launch_us": round(k_avg_s * 1e6, 2), "launches_timed": int(iso_launches),
                         "measured": "HIP start/stop events stamped by every transform dispatch on the library stream "
                                     "(hipExtLaunchKernelGGL), in a pass of 30 frames with one frame in flight run right after the "
                                     "timed region (same process, same buffers, the whole pipeline running); frac = the level-0 "
                                     "forward kernel, inv_level0_frac its inverse, transform_*_frac = all levels of one direction "
                                     "(sum of algorithmic bytes / sum of kernel durations)",
                         "avg_launch_us_in_timed_region": round(k_conc_s * 1e6, 2), "launches_in_timed_region": int(launches),
                         "traffic_source": TRAFFIC[args.io][1],
                         "copy_peak_guide_gbs": COPY_PEAK_GUIDE_GBS,
                         "frac_of_guide_copy_peak": round(achieved / COPY_PEAK_GUIDE_GBS, 4),
                         "rotating_buffers": rot},
            # every kernel of the step, inside the timed region: the HBM bytes one frame's nine kernels move (PMC counters, committed)
            # x the frames coded per second
            "pipeline": {"hbm_bytes_per_frame": PIPELINE_TRAFFIC[0], "source": PIPELINE_TRAFFIC[1],
                         "in_region_tbs": round(frames_per_step / max(world, 1) * PIPELINE_TRAFFIC[0] / (dt / args.steps) / 1e12, 3),
                         "frac_of_hbm_peak": round(frames_per_step / max(world, 1) * PIPELINE_TRAFFIC[0] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "per GPU; all frames in flight, inside the timed region"} if args.io == "rgba8" else None,
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if others is not None:
            out["other_configs"] = others
        state["line"] = out
    # ---- N > 1: the same command also carries the tile-sharded C4 record and the frames-per-rank C5 record (a
    #      SCALE file then answers "do tiles shard near-linearly?" without flags).  The headline's lanes are torn down first (the
    #      process group stays), each record runs inside the same ranks, and a watchdog prints the headline alone if one of them hangs.
    if world > 1 and not peer_rehearsal and not args.no_other_configs and os.environ.get("J2K_BENCH_SCALE_RECORDS", "1") != "0":
        import copy
        import threading
        bench_extra.teardown(state.get("lanes", []), False, state.get("helper"), state.get("stop_helper"), ok=True, comm=state.get("comm"))
        state["lanes"], state["helper"], state["comm"] = [], None, None
        lanes.clear()
        torch.cuda.empty_cache()
        scale = {}

        def bark():
            if rank == 0 and state.get("line") is not None:
                state["line"]["scale_records"] = dict(scale, error="timed out after %s s: the headline line is printed without the missing records" % os.environ.get("J2K_BENCH_SCALE_TIMEOUT", "240"))
                print(json.dumps(state["line"]), flush=True)
            os._exit(0)
        dog = threading.Timer(float(os.environ.get("J2K_BENCH_SCALE_TIMEOUT", "240")), bark)
        dog.daemon = True
        dog.start()
        for name, fn, kw in (("c4_tiles_sharded", bench_extra.run_shard_tiles, dict(config="c4", steps=20, warmup=3, inflight=1, d2h=False)),
                             ("c5_frames_per_rank", bench_extra.run_config, dict(config="c5", steps=20, warmup=3, inflight=0))):
            a2 = copy.copy(args)
            for k_, v_ in kw.items():
                setattr(a2, k_, v_)
            a2.no_cpu_baseline = True
            try:
                rec = fn(a2, embedded=True) if fn is bench_extra.run_shard_tiles else fn(a2, "c5", embedded=True)
                if rank == 0 and rec is not None:
                    scale[name] = {k_: rec[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling")}
                    scale[name].update({k_: rec["config"][k_] for k_ in ("tiles", "codestream_bytes", "value_with_d2h", "parallelism", "gather", "assembly_check",
                                                                         "frames_in_flight", "contexts", "frames_per_context") if k_ in rec["config"]})
                    scale[name]["ranks_seen"] = world
            except Exception as exc:                        # noqa: BLE001  (the headline does not depend on these)
                scale[name] = {"error": repr(exc)[:300]}
                try:
                    dist.barrier()
                except Exception:                           # noqa: BLE001
                    pass
        dog.cancel()
        if rank == 0:
            state["line"]["scale_records"] = scale
    if rank == 0 and state.get("line") is not None:
        print(json.dumps(state["line"]))


def main():
    """Explicit teardown whatever happens in the step loop: the exchange helper is stopped and joined,
    every library stream drained, plans then contexts destroyed, and only then the process group -- see bench_extra.teardown."""
    state = {}
    ok = False
    try:
        run(state)
        ok = True
    finally:
        if not state.get("skip_teardown"):              # (the launcher parent never imported torch: nothing to tear down)
            bench_extra.teardown(state.get("lanes", []), state.get("multi", False), state.get("helper"), state.get("stop_helper"), ok=ok, comm=state.get("comm"))
        if state.get("probe_abandoned"):                # a probe thread still inside the runtime: do not wait for it at interpreter exit
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0 if ok else 1)


if __name__ == "__main__":
    main()
