"""GPU: a frame's plan calls recorded into a HIP graph (j2k_ctx_capture_begin / _end, j2k_graph_launch) and replayed on new
contents of the same buffers give what the direct calls give."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _buffers(plan, torch):
    i = plan.info
    n = int(i.blocks)
    return dict(n=n, coeff=plan.alloc_coeff(), stream=plan.empty(i.bytes_cap, torch.uint8), lens=plan.empty(n, torch.int32),
                nb=plan.empty(n, torch.uint8), offs=plan.empty(n + 1, torch.int64),
                decoded=torch.zeros(int(i.decoded_elems), dtype=torch.int32, device=plan.device), back=plan.alloc_frame())


def _run(plan, frame, b):
    plan.forward(frame, b["coeff"])
    plan.encode_stream(b["coeff"], b["stream"], b["offs"], b["lens"], b["nb"])
    plan.decode_blocks(b["stream"], b["offs"], b["lens"], b["nb"], b["decoded"])
    plan.inverse(b["coeff"], b["back"])


@pytest.mark.parametrize("W,H,Cn,tile,coder,lossless", [(640, 368, 3, (512, 512), 1, True), (256, 256, 1, (0, 0), 1, True), (320, 200, 3, (128, 128), 0, False)])
def test_graph_replay_equals_direct_calls(W, H, Cn, tile, coder, lossless):
    import torch
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H + coder)
    fa = rng.integers(0, 256, size=(Cn, H, W)).astype(np.int32)
    fb = rng.integers(0, 256, size=(Cn, H, W)).astype(np.int32)
    fb[:, : H // 2] = 7                                       # different block lengths / bit-plane counts than frame a
    kw = dict(precision=8, lossless=lossless, quality=60, num_resolutions=4, cb=(64, 64), tile=tile, coder=coder)
    # direct calls on frame b
    ctx0 = Context(0)
    p0 = FramePlan(W, H, Cn, ctx=ctx0, **kw)
    b0 = _buffers(p0, torch)
    _run(p0, torch.from_numpy(fb).to(p0.device), b0)
    ctx0.sync()
    # graph: warm up on frame a, capture, put frame b into the same buffer, replay twice
    ctx = Context(0)
    p = FramePlan(W, H, Cn, ctx=ctx, **kw)
    b = _buffers(p, torch)
    frame = torch.from_numpy(fa).to(p.device)
    _run(p, frame, b)
    ctx.sync()
    with ctx.capture() as g:
        _run(p, frame, b)
    frame.copy_(torch.from_numpy(fb).to(p.device)); b["decoded"].zero_()
    torch.cuda.synchronize()
    for _ in range(2):
        g.launch()
    ctx.sync()
    n = b["n"]
    tot = int(b0["offs"][n].item())
    assert int(b["offs"][n].item()) == tot
    assert torch.equal(b["coeff"], b0["coeff"]) and torch.equal(b["lens"][:n], b0["lens"][:n]) and torch.equal(b["nb"][:n], b0["nb"][:n])
    assert torch.equal(b["stream"][:tot], b0["stream"][:tot])
    assert torch.equal(b["decoded"], b0["decoded"]) and torch.equal(b["back"], b0["back"])
    g.close()


def test_capture_refuses_calls_that_would_allocate():
    import torch
    from j2kgfx import Context, J2KError
    from j2kgfx.codec import FramePlan
    ctx = Context(0)
    p = FramePlan(128, 128, 1, ctx=ctx, precision=8, lossless=True, num_resolutions=3, cb=(64, 64), coder=1)
    b = _buffers(p, torch)
    frame = torch.zeros((1, 128, 128), dtype=torch.int32, device=p.device)
    torch.cuda.synchronize()
    with pytest.raises(J2KError):
        with ctx.capture():
            _run(p, frame, b)                                 # never run before: the encoder's slot buffer does not exist yet
    _run(p, frame, b)                                         # the context is usable afterwards
    ctx.sync()


def test_capture_refuses_the_one_kernel_ht_path():
    """with J2K_FUSE_COMPACT=1 the look-back epoch is a kernel argument, so a replay would reuse the previous
    replay's status words -- the capture is refused (and the direct calls keep working)."""
    import os
    import torch
    from j2kgfx import Context, J2KError
    from j2kgfx.codec import FramePlan
    old = os.environ.get("J2K_FUSE_COMPACT")
    os.environ["J2K_FUSE_COMPACT"] = "1"
    try:
        ctx = Context(0)
    finally:
        if old is None:
            os.environ.pop("J2K_FUSE_COMPACT", None)
        else:
            os.environ["J2K_FUSE_COMPACT"] = old
    p = FramePlan(256, 256, 3, ctx=ctx, precision=8, lossless=True, num_resolutions=4, cb=(64, 64), coder=1)
    b = _buffers(p, torch)
    rng = np.random.default_rng(3)
    frame = torch.from_numpy(rng.integers(0, 256, (3, 256, 256)).astype(np.int32)).to(p.device)
    _run(p, frame, b)
    ctx.sync()
    ref = b["stream"].clone()
    with pytest.raises(J2KError):
        with ctx.capture():
            _run(p, frame, b)
    _run(p, frame, b)
    ctx.sync()
    n = b["n"]
    assert torch.equal(b["stream"][:int(b["offs"][n].item())], ref[:int(b["offs"][n].item())])


def test_graph_launch_is_ordered_behind_the_current_stream_and_refuses_a_closed_context():
    """g.launch() right after frame.copy_(new) WITHOUT a device synchronisation must see the new frame (the launch
    orders the library stream behind torch's current stream); after ctx.close() a launch raises instead of touching a freed
    context."""
    import torch
    from j2kgfx import Context, J2KError
    from j2kgfx.codec import FramePlan
    W = H = 512
    rng = np.random.default_rng(9)
    ctx = Context(0)
    p = FramePlan(W, H, 3, ctx=ctx, precision=8, lossless=True, num_resolutions=5, cb=(64, 64), coder=1)
    b = _buffers(p, torch)
    frames = [torch.from_numpy(rng.integers(0, 256, (3, H, W)).astype(np.int32)).to(p.device) for _ in range(4)]
    frame = frames[0].clone()
    _run(p, frame, b)
    ctx.sync()
    with ctx.capture() as g:
        _run(p, frame, b)
    for f in frames[1:]:
        frame.copy_(f)                                        # queued on torch's current stream; no synchronize here
        g.launch()
        ctx.sync()
        assert torch.equal(b["back"], f)                      # lossless: the replay saw the new contents
    ctx.close()
    with pytest.raises(J2KError):
        g.launch()


def test_interpreter_exit_with_live_plans_is_clean():
    """a process that exits with plans / contexts / queued work alive (tests/helpers_exit_order.py: SystemExit(3), nothing
    closed) must leave with that exit code and no abort text -- the package's atexit hook closes graphs, plans, contexts in
    that order while torch and the HIP runtime are still up."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers_exit_order.py")], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    for bad in ("terminate called", "bad_variant_access", "Segmentation", "core dumped", "Aborted"):
        assert bad not in out.stderr, out.stderr[-2000:]
