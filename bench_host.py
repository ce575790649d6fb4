"""bench_host.py -- two more records of bench.py's default line:

  run_host_boundary (`bench.py --io host`): C2 at the boundary a Go caller really has -- image.RGBA.Pix in PINNED HOST memory in,
      finished tile-parts (+ the per-block lengths and bit-plane counts) in pinned host memory out; and back: the dense block
      stream + its tables from pinned host memory in, image.RGBA.Pix in pinned host memory out (encoder.go:79-213,
      decoder.go:417-588 are the host loops on either side).  H2D copies, kernels and D2H copies of LANES frames in flight
      overlap: every lane has a torch-owned stream for each direction beside its library stream, ordered by events; a frame's
      tile-parts are copied at their exact length (the 8-byte length comes back first, the host waits for it one round later).
      Reported: frames/s as Mpixel/s, the bytes per second each way, and those as a fraction of what two large pinned copies
      running at once reach on the same box (measured here, just before).
  run_closed_loop (`bench.py --config cl`): the closed-loop codec (j2k_params.closed_loop; this library's mode, not the
      reference's) on the 4K RGB8 frame with the MQ coder: pixels -> SOT | SOD | packets -> pixels through
      j2k_plan_encode_frame_pixels / j2k_plan_decode_frame_pixels, device buffers, checked bit-exact after the timed region."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd"))
import bench_extra  # noqa: E402

W, H, C, TILE, NRES, CB, PREC = 3840, 2160, 3, 512, 6, 64, 8


def _rgba_host(np, index):
    fr = bench_extra.synth_frame(np, bench_extra.CONFIGS["c2"], index)
    rgba = np.concatenate([fr.transpose(1, 2, 0), np.full((H, W, 1), 255, np.int32)], axis=2).astype(np.uint8)
    return np.ascontiguousarray(rgba.reshape(H, W * 4))


def pinned_copy_peak(torch, device, mb=256, reps=6):
    """GB/s of large pinned copies: H2D alone, D2H alone, and both at once (each direction's rate while the other runs)"""
    n = mb << 20
    h_in, h_out = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    d_in, d_out = torch.empty(n, dtype=torch.uint8, device=device), torch.ones(n, dtype=torch.uint8, device=device)
    def run(h2d, d2h, s1, s2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_in.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_out.copy_(d_out, non_blocking=True)
        s1.synchronize(); s2.synchronize()
        return reps * n / (time.perf_counter() - t0) / 1e9
    # two streams may or may not share a hardware queue (the runtime deals streams onto GPU_MAX_HW_QUEUES queues): when they do, the two
    # directions take turns (28.6 GB/s each on these boxes instead of 48.6) -- the PEAK is the best of a few stream pairs
    best = {"h2d_alone": 0.0, "d2h_alone": 0.0, "both_each": 0.0}
    keep = []
    for _ in range(4):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        keep += [s1, s2]
        run(True, True, s1, s2)
        best["h2d_alone"] = max(best["h2d_alone"], run(True, False, s1, s2))
        best["d2h_alone"] = max(best["d2h_alone"], run(False, True, s1, s2))
        best["both_each"] = max(best["both_each"], run(True, True, s1, s2))
    return {k: round(v, 2) for k, v in best.items()} | {"copy_mb": mb, "stream_pairs_tried": 4}


def run_host_boundary(args):
    import numpy as np
    import torch
    from j2kgfx import CODER_HT, Context
    from j2kgfx.codec import FramePlan
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(0)
    NL = args.inflight if args.inflight > 0 and "--inflight" in " ".join(sys.argv) else 1      # (a lane holds two frames' device buffers and keeps both copy directions and the kernels busy; more lanes put several PCIe-writing kernels on the link at once: 0.92 / 0.80 / 0.70 / 0.65 with 1 / 2 / 3 / 4)
    # the two copy streams are HIGH-PRIORITY streams: the runtime keeps a separate set of hardware queues per priority, so they never
    # share a queue with a lane's kernels (J2K_BENCH_HOST_PRIO=0: plain streams, dealt onto the same queues as everything else)
    prio = -1 if os.environ.get("J2K_BENCH_HOST_PRIO", "1") != "0" else 0
    g_in, g_out = torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)
    peak = pinned_copy_peak(torch, "cuda:0")
    lanes = []
    ok = False
    # ONE stream per copy direction for all lanes (J2K_BENCH_HOST_STREAMS=lane: a pair per lane): the copies of one direction share the
    # link anyway, and a dozen copy streams dealt onto the runtime's hardware queues end up sharing queues with each other and with the
    # kernels' streams at random
    shared = os.environ.get("J2K_BENCH_HOST_STREAMS", "shared") == "shared"
    # The inverse level-0 kernel stores the pixels STRAIGHT into the pinned host frame (pinned memory is device-addressable) instead of into a
    # device frame that a D2H copy then moves: the larger half of the D2H bytes leaves the copy engines, and with it the moments when the two
    # directions' copies took turns on one engine.  One lane: 0.91-0.92 of the pinned-copy peak, every segment within 5 % (device frame + copy:
    # 0.86 with segments from 0.80 to 0.97, and 0.65-0.84 as two lanes).  J2K_BENCH_HOST_DIRECT=0: the copy; =2: the forward kernel READS the
    # pinned pixels as well (no H2D copy of them: 0.63 -- loads over the link are not posted)
    direct = os.environ.get("J2K_BENCH_HOST_DIRECT", "1") in ("1", "2")
    direct_in = os.environ.get("J2K_BENCH_HOST_DIRECT", "1") == "2"     # (experiment: the forward kernel READS the pinned pixels too -- no H2D copy of them)
    try:
        for f in range(NL):
            ctx = Context(0)
            p = FramePlan(W, H, C, precision=PREC, lossless=True, num_resolutions=NRES, cb=(CB, CB), tile=(TILE, TILE), coder=CODER_HT, ctx=ctx,
                          track_streams=False)
            p.set_decode_coded_rows_only(True)
            i, n = p.info, int(p.info.blocks)
            L = p.ctx.L
            L.j2k_plan_tile_parts_bound.restype = __import__("ctypes").c_size_t
            cap = int(L.j2k_plan_tile_parts_bound(p.h))
            # per-block tables travel as ONE copy each way: lengths (u32) and bit-plane counts (u8) side by side; offsets, lengths, counts.
            # Everything a copy reads or writes on the device exists TWICE (frame k uses set k & 1): a copy is only ever queued for a frame
            # whose kernels were queued a whole round earlier, so that no copy sits at the head of its direction's DMA queue waiting
            # for a kernel (the runtime feeds all copies of one direction through one queue, in order)
            lenoff = (5 * n + 15) // 8 * 8                     # ... and the 8-byte length of the tile-parts behind them: ONE copy out
            lt = [p.empty(lenoff + 8, torch.uint8) for _ in range(2)]
            ln = dict(ctx=ctx, p=p, n=n, lib=torch.cuda.ExternalStream(ctx.stream), s_in=g_in if shared else torch.cuda.Stream(),
                      s_out=g_out if shared else torch.cuda.Stream(),
                      h_pix=torch.from_numpy(_rgba_host(np, f)).pin_memory(), d_pix=[torch.empty((H, W * 4), dtype=torch.uint8, device=p.device) for _ in range(2)],
                      coeff=p.alloc_coeff(), stream=p.empty(i.bytes_cap, torch.uint8), offs=p.empty(n + 1, torch.int64),
                      lt=lt, lens=[t_[:4 * n].view(torch.int32) for t_ in lt], nb=[t_[4 * n:5 * n] for t_ in lt],
                      cs=[p.empty(cap, torch.uint8) for _ in range(3)], cs_len=[t_[lenoff:lenoff + 8].view(torch.int64) for t_ in lt], lenoff=lenoff,
                      h_cs=torch.empty(cap, dtype=torch.uint8).pin_memory(),
                      h_lt=[torch.zeros(lenoff + 8, dtype=torch.uint8).pin_memory() for _ in range(2)],
                      # decode side: what the host holds (dense stream + tables), where it goes, and the pixels back
                      d_stream2=[p.empty(int(i.bytes_cap) + 8 * (n + 1) + 5 * n + 64, torch.uint8) for _ in range(2)],
                      decoded=torch.zeros(max(int(i.decoded_elems), 4), dtype=torch.int32, device=p.device),
                      d_back=[torch.empty((H, W * 4), dtype=torch.uint8, device=p.device) for _ in range(2)], h_back=torch.empty((H, W * 4), dtype=torch.uint8).pin_memory(),
                      h_back2=[torch.empty((H, W * 4), dtype=torch.uint8).pin_memory() for _ in range(2)],
                      k=0)
            for name in ("e_pix", "e_fwd", "e_enc", "e_len", "e_tabs", "e_str", "e_dec", "e_inv", "e_back", "e_cs_free"):
                ln[name] = [torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()]
            lanes.append(ln)
        # the decode side's host input: frame f's own dense stream and tables, made once (untimed) and kept in pinned memory
        for ln in lanes:
            p = ln["p"]
            ln["d_pix"][0].copy_(ln["h_pix"]); torch.cuda.synchronize()
            p.forward_rgba8(ln["d_pix"][0], ln["coeff"])
            p.encode_stream(ln["coeff"], ln["stream"], ln["offs"], ln["lens"][0], ln["nb"][0])
            ln["ctx"].sync()
            tot = int(ln["offs"][ln["n"]].item())
            ln["tot"] = tot
            ln["h_stream"] = ln["stream"][:tot].cpu().pin_memory()
            n_ = ln["n"]
            # the block stream and, behind it (16-byte aligned), offsets | lengths | bit-plane counts: ONE copy in
            toff = (tot + 15) // 16 * 16
            ln["toff"] = toff
            ln["h_in2"] = torch.cat([ln["stream"][:tot], torch.zeros(toff - tot, dtype=torch.uint8, device=p.device), ln["offs"][:n_ + 1].view(torch.uint8),
                                     ln["lens"][0][:n_].view(torch.uint8), ln["nb"][0][:n_]]).cpu().pin_memory()
            ln["d_offs2"] = [t_[toff:toff + 8 * (n_ + 1)].view(torch.int64) for t_ in ln["d_stream2"]]
            ln["d_lens2"] = [t_[toff + 8 * (n_ + 1):toff + 8 * (n_ + 1) + 4 * n_].view(torch.int32) for t_ in ln["d_stream2"]]
            ln["d_nb2"] = [t_[toff + 8 * (n_ + 1) + 4 * n_:toff + 8 * (n_ + 1) + 5 * n_] for t_ in ln["d_stream2"]]
            ln["h_lens_in"], ln["h_nb_in"] = ln["lens"][0][:n_].cpu(), ln["nb"][0][:n_].cpu()
        torch.cuda.synchronize()
        host_wait = [0.0]

        def payload(ln, k):
            """the tile-parts of frame k at their exact length: its 8-byte length was queued for copying a round ago"""
            b = k & 1
            tw = time.perf_counter()
            ln["e_len"][b].synchronize()
            host_wait[0] += time.perf_counter() - tw
            nbytes = int(ln["h_lt"][b][ln["lenoff"]:].view(torch.int64)[0])
            with torch.cuda.stream(ln["s_out"]):
                ln["h_cs"][:nbytes].copy_(ln["cs"][k % 3][:nbytes], non_blocking=True)
                ln["e_cs_free"][k % 3].record(ln["s_out"])
            ln["cs_bytes"] = nbytes

        def outputs(ln, k):
            """what frame k left on the device, to pinned host memory: length + tables now, pixels now, tile-parts one round later"""
            b, n = k & 1, ln["n"]
            s_out = ln["s_out"]
            with torch.cuda.stream(s_out):
                s_out.wait_event(ln["e_enc"][b])
                ln["h_lt"][b].copy_(ln["lt"][b], non_blocking=True)
                ln["e_len"][b].record(s_out)
                ln["e_tabs"][b].record(s_out)
                s_out.wait_event(ln["e_inv"][b])
                if not direct:
                    ln["h_back"].copy_(ln["d_back"][b], non_blocking=True)
                ln["e_back"][b].record(s_out)

        def issue(ln):
            p, n = ln["p"], ln["n"]
            lib, s_in = ln["lib"], ln["s_in"]
            k = ln["k"]
            ln["k"] += 1
            b = k & 1
            if k >= 2:
                payload(ln, k - 2)      # the only place the host waits: for an 8-byte length whose copy was queued a round ago
            # ---- in: pinned pixels and pinned block stream + tables -> device set b (last used by frame k - 2)
            with torch.cuda.stream(s_in):
                s_in.wait_event(ln["e_fwd"][b])
                if not direct_in:
                    ln["d_pix"][b].copy_(ln["h_pix"], non_blocking=True)
                ln["e_pix"][b].record(s_in)
                s_in.wait_event(ln["e_dec"][b])
                ln["d_stream2"][b][:ln["h_in2"].numel()].copy_(ln["h_in2"], non_blocking=True)
                ln["e_str"][b].record(s_in)
            # ---- kernels: encode (pixels -> tile-parts) and decode (block stream -> pixels) of frame k
            lib.wait_event(ln["e_pix"][b])
            p.forward_rgba8(ln["h_pix"] if direct_in else ln["d_pix"][b], ln["coeff"])
            ln["e_fwd"][b].record(lib)
            lib.wait_event(ln["e_tabs"][b])
            p.encode_stream(ln["coeff"], ln["stream"], ln["offs"], ln["lens"][b], ln["nb"][b])
            lib.wait_event(ln["e_cs_free"][k % 3])              # (the tile-parts of frame k - 3 were copied out: queued at the end of the last round)
            p.assemble_tiles(ln["stream"], ln["offs"], ln["cs"][k % 3], ln["cs_len"][b])
            ln["e_enc"][b].record(lib)
            lib.wait_event(ln["e_str"][b])
            p.decode_blocks(ln["d_stream2"][b], ln["d_offs2"][b], ln["d_lens2"][b], ln["d_nb2"][b], ln["decoded"])
            ln["e_dec"][b].record(lib)
            lib.wait_event(ln["e_back"][b])
            p.inverse_rgba8(ln["coeff"], ln["h_back2"][b] if direct else ln["d_back"][b])
            ln["e_inv"][b].record(lib)
            # ---- out: the PREVIOUS frame's results (their kernels were queued a round ago)
            if k >= 1:
                outputs(ln, k - 1)

        def drain():
            """every frame issued so far is complete in host memory; the lanes start again at frame 0"""
            for ln in lanes:
                k = ln["k"]
                if k >= 1:
                    outputs(ln, k - 1)
            for ln in lanes:
                k = ln["k"]
                if k >= 2:
                    payload(ln, k - 2)
                if k >= 1:
                    payload(ln, k - 1)
                ln["k"] = 0
            for ln in lanes:
                ln["ctx"].sync()
            for ln in lanes:
                ln["s_in"].synchronize(); ln["s_out"].synchronize(); ln["ctx"].sync()
            torch.cuda.synchronize()

        for ln in lanes:                                        # events that are waited for before a frame has recorded them
            for name in ("e_fwd", "e_tabs", "e_dec", "e_back", "e_cs_free"):
                for e in ln[name]:
                    e.record(ln["lib"])
        for _ in range(max(args.warmup, 2)):
            for ln in lanes:
                issue(ln)
        drain()
        # Which hardware queues the two copy streams land on is the runtime's choice at the time they are made (and a pair that shares
        # one takes turns): like the peak above, the run takes the best of a few stream pairs, tried for a few rounds each
        tried = []
        if shared:
            pairs = [(g_in, g_out)] + [(torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)) for _ in range(2)]
            for a_, b_ in pairs:
                for ln in lanes:
                    ln["s_in"], ln["s_out"] = a_, b_
                for ln in lanes:
                    issue(ln)
                drain()
                tq = time.perf_counter()
                for _ in range(8):
                    for ln in lanes:
                        issue(ln)
                drain()
                tried.append((time.perf_counter() - tq) / 8)
            a_, b_ = pairs[int(np.argmin(tried))]
            for ln in lanes:
                ln["s_in"], ln["s_out"] = a_, b_
        tc = time.perf_counter()
        for _ in range(3):
            for ln in lanes:
                issue(ln)
        drain()
        steps = max(args.steps, int(np.ceil(2.0 / max((time.perf_counter() - tc) / 3, 1e-6))))   # (2 s: one stalled moment of a shared host weighs less)
        steps = min(steps, 5000)
        # The timed region is NSEG segments of steps / NSEG rounds each (the pipeline drained in between), all of them counted: on
        # these boxes the link runs in one of two modes at a time -- both directions at once (48.6 GB/s each way) or taking turns
        # (28.6; the same two figures two plain pinned copies give, see pinned_copy_peak) -- and a segment is in one or the other
        NSEG = 16
        per = max(steps // NSEG, 1)
        steps = per * NSEG
        segs = []
        host_wait[0] = 0.0
        t_issue = 0.0
        t0 = time.perf_counter()
        for _ in range(NSEG):
            tq = time.perf_counter()
            for _ in range(per):
                for ln in lanes:
                    issue(ln)
            t_issue += time.perf_counter() - tq
            drain()
            segs.append(round(per * NL * W * H / (time.perf_counter() - tq) / 1e6, 1))
        dt = time.perf_counter() - t0
        # ---- what came back is right ----
        for ln in lanes:
            for hb in (ln["h_back2"] if direct else [ln["h_back"]]):
                assert torch.equal(hb, ln["h_pix"]), "pixels back in host memory differ from the pixels in"
            from j2kgfx import codestream
            parts = codestream.parse_tile_parts(ln["h_cs"][:ln["cs_bytes"]].numpy().tobytes())
            assert [pt.TileIndex for pt, _ in parts] == list(range(int(ln["p"].info.tiles)))
            assert b"".join(d for _, d in parts) == ln["h_stream"].numpy().tobytes(), "tile-parts in host memory differ from the block stream"
            n_ = ln["n"]
            for hl in ln["h_lt"]:
                assert torch.equal(hl[:4 * n_].view(torch.int32), ln["h_lens_in"]) and torch.equal(hl[4 * n_:5 * n_], ln["h_nb_in"])
        ln0 = lanes[0]
        n = ln0["n"]
        h2d = W * H * 4 + ln0["h_in2"].numel()
        d2h = ln0["cs_bytes"] + ln0["lenoff"] + 8 + W * H * 4
        frames = steps * NL
        h2d_gbs, d2h_gbs = frames * h2d / dt / 1e9, frames * d2h / dt / 1e9
        out = {"metric": "Mpixels/s encode+decode (4K sRGB, 5-3 lossless), pinned host memory in and out", "value": round(frames * W * H / dt / 1e6, 1),
               "unit": "Mpixels/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": bench_extra.CONFIGS["c2"]["workload"] + "; the C2 step with its inputs and outputs in PINNED HOST memory: encode = image.RGBA.Pix "
                          "H2D, forward transform + HT block coding + compaction + tile-part assembly on the device, the tile-parts D2H at their exact "
                          "length (+ per-block lengths and bit-plane counts); decode = the dense block stream and its tables H2D, HT block decode + inverse "
                          "transform, image.RGBA.Pix back in pinned host memory" + (" -- stored there by the inverse level-0 kernel itself (pinned memory is device-addressable: no device frame, no D2H copy of the pixels)" if direct else " by a D2H copy") + "; copies and kernels of consecutive frames overlap (a copy stream per direction + the library stream)",
                          "pixels_out": "kernel stores into the pinned frame" if direct else "device frame + D2H copy", "lanes": NL,
                          "frames_in_flight": NL, "frame_io": "pinned host", "hsa_enable_sdma_recommended_eng": os.environ.get("HSA_ENABLE_SDMA_RECOMMENDED_ENG", "(runtime default)"), "copy_streams": "one per direction" if shared else "a pair per frame in flight", "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "h2d_bytes_per_frame": h2d, "d2h_bytes_per_frame": d2h,
                          "tile_part_bytes": ln0["cs_bytes"]},
               "host_boundary": {"value": round(frames * W * H / dt / 1e6, 1), "unit": "Mpixels/s", "h2d_gbs": round(h2d_gbs, 2), "d2h_gbs": round(d2h_gbs, 2),
                                 "pinned_copy_peak_gbs": peak, "frac_of_pinned_copy_peak": round(max(h2d_gbs, d2h_gbs) / max(peak["both_each"], 1e-9), 4),
                                 "frames_in_flight": NL, "h2d_bytes_per_frame": h2d, "d2h_bytes_per_frame": d2h,
                                 "host_issue_ms_per_frame": round((t_issue - host_wait[0]) / frames * 1e3, 4), "host_wait_ms_per_frame": round(host_wait[0] / frames * 1e3, 4),
                                 "ms_per_frame": round(dt / frames * 1e3, 4), "copy_stream_pairs_tried_ms_per_round": [round(v * 1e3, 3) for v in tried],
                                 "segments_mpixels_s": segs, "median_segment_mpixels_s": round(float(np.median(segs)), 1),
                                 "median_segment_frac": round(float(np.median(segs)) * 1e6 / (W * H) * max(h2d, d2h) / 1e9 / max(peak["both_each"], 1e-9), 4), "best_segment_frac": round(max(segs) * 1e6 / (W * H) * max(h2d, d2h) / 1e9 / max(peak["both_each"], 1e-9), 4),
                                 "note": "value and frac_of_pinned_copy_peak are over the WHOLE timed region (every segment counted, a stalled one too -- the host is shared); the median segment is given beside them. "
                                         "fraction = the busier direction's bytes per second / what each direction reaches when two large pinned "
                                         "copies run at once on this box (both_each); never the headline `value`, which has its inputs resident in HBM"}}
        print(json.dumps(out))
        ok = True
    finally:
        bench_extra.teardown(lanes, False, ok=ok)


def run_closed_loop(args):
    """4K RGB8, closed-loop mode: pixels -> tile-parts -> pixels on device buffers, F frames in flight.  MQ coder (--config cl: the
    bit-exact round trip) or the reference's HT coder (--config clht: it codes one row in four, so what comes back is checked against the
    same stream decoded with the serial packet parser, not against the source)"""
    import numpy as np
    import torch
    from j2kgfx import CODER_HT, CODER_MQ, Context, _lib
    ht = getattr(args, "config", "cl") == "clht"
    from j2kgfx.codec import FramePlan
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(0)
    F = args.inflight if args.inflight > 0 else (2 if ht else 12)
    B = max(1, int(getattr(args, "batch", 0) or (1 if ht else 4)))   # frames per context and call: one plan over the frames stacked vertically (j2k_params.frame_rows)
    lanes = []
    ok = False
    try:
        for f in range(F):
            ctx = Context(0)
            p = FramePlan(W, H * B, C, precision=PREC, lossless=True, num_resolutions=NRES, cb=(CB, CB), tile=(TILE, TILE), coder=CODER_HT if ht else CODER_MQ, ctx=ctx,
                          track_streams=False, closed_loop=True, frame_rows=H if B > 1 else 0)
            base = _rgba_host(np, f)
            pix = torch.from_numpy(np.concatenate([base if b == 0 else np.roll(base, (41 * b, 388 * b), axis=(0, 1)) for b in range(B)], axis=0)).to(p.device)
            lanes.append(dict(ctx=ctx, p=p, pix=pix, back=torch.zeros_like(pix), cs=p.empty(p.frame_bound(), torch.uint8),
                              toffs=p.empty(int(p.info.tiles) + 1, torch.int64)[:int(p.info.tiles) + 1]))
        torch.cuda.synchronize()

        def code(ln):
            p = ln["p"]
            p.encode_frame_pixels(_lib.PIX_RGBA8, ln["pix"], sop=True, eph=True, out=ln["cs"], tile_offs=ln["toffs"])
            # the decoder is handed the buffer at its capacity and the tile-part positions the encoder left on the device: no length
            # crosses to the host inside the step (a Go caller that holds the codestream passes its real length)
            # (SOP + EPH markers, 8 bytes a packet: the decoder parses a tile's packets side by side from them)
            p.decode_frame_pixels(ln["cs"], ln["cs"].numel(), ln["back"], tile_offs=ln["toffs"], sop=True, eph=True)

        def barrier():
            for ln in lanes:
                ln["ctx"].sync()
            torch.cuda.synchronize()
        for _ in range(max(args.warmup, 1)):
            for ln in lanes:
                code(ln)
        barrier()
        tc = time.perf_counter()
        for ln in lanes:
            code(ln)
        barrier()
        steps = max(args.steps, int(np.ceil(0.5 / max(time.perf_counter() - tc, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(steps):
            for ln in lanes:
                code(ln)
        barrier()
        dt = time.perf_counter() - t0
        # one frame alone: the latency of the whole chain
        t1 = time.perf_counter()
        code(lanes[0]); lanes[0]["ctx"].sync()
        single_ms = (time.perf_counter() - t1) * 1e3
        for ln in lanes:
            ln["p"].frame_status()
            if not ht:
                assert torch.equal(ln["back"], ln["pix"]), "closed-loop round trip is not bit-exact"
            else:                                                   # the same tile-parts through the serial packet parser: the same pixels
                got = ln["back"].clone()
                ln["ctx"].set_option("t2_parallel", 0)
                ln["p"].decode_frame_pixels(ln["cs"], ln["cs"].numel(), ln["back"], tile_offs=ln["toffs"], sop=True, eph=True)
                ln["ctx"].set_option("t2_parallel", 1)
                ln["p"].frame_status()
                assert torch.equal(ln["back"], got), "closed-loop HT decode: packet-parallel and serial parses differ"
        total = int(lanes[0]["toffs"][-1].item())
        lanes[0]["p"].frame_parallel_tiles()
        code(lanes[0])
        par_tiles = lanes[0]["p"].frame_parallel_tiles()
        out = {"metric": ("Mpixels/s encode+decode through tile-parts of packets (4K sRGB, 5-3 lossless, the reference's HT coder, closed-loop mode)" if ht else
                          "Mpixels/s encode+decode, bit-exact round trip through tile-parts of packets (4K sRGB, 5-3 lossless, MQ coder, closed-loop mode)"),
               "value": round(steps * F * B * W * H / dt / 1e6, 1), "unit": "Mpixels/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
               "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": "3840x2160 sRGB 8-bit, 512x512 tiles, 5-3 lossless + " + ("HT block coder (the reference's, bug for bug)" if ht else "MQ block coder (T1.EncodeFast5 / T1.Decode)") + ", 64x64 code-blocks, "
                          "6 resolutions, j2k_params.closed_loop = 1 (this library's mode, outside reference parity: code-block windows that partition "
                          "the plane, packets the decoder can read); a step = image.RGBA.Pix -> forward transform -> block coder -> one packet per "
                          "(tile, component, resolution), SOP + EPH markers -> SOT | SOD | packets, then tile-part parse -> packet parse -> block decode -> placement -> "
                          "inverse transform -> image.RGBA.Pix, all on device buffers (j2k_plan_encode_frame_pixels / j2k_plan_decode_frame_pixels)" + ("" if ht else "; the pixels that come back are compared with the pixels that went in"),
                          "frames_in_flight": F * B, "contexts": F, "frames_per_context": B, "codestream_bytes_per_frame": total // B, "tiles_parsed_packet_parallel": "%d of %d" % (par_tiles, int(lanes[0]["p"].info.tiles)), "single_frame_ms": round(single_ms / B, 2) if B > 1 else round(single_ms, 2),
                          "round_trip": ("the reference's HT coder codes one row in four: pixels back == the same stream through the serial packet parser (checked after the timed region)" if ht
                                         else "bit-exact (checked after the timed region, every frame in flight)")},
               "roofline": {"bound": "hbm", "kernel": "n/a (see --config c2 for the kernels' rooflines)" if ht else "n/a (the MQ block coder bounds this configuration: serial chains, no bandwidth roofline)", "achieved": None,
                            "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_us": None}}
        print(json.dumps(out))
        ok = True
    finally:
        bench_extra.teardown(lanes, False, ok=ok)
