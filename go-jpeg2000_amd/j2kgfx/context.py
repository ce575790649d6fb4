"""j2k_ctx wrapper."""
import atexit
import ctypes as C
import sys
import weakref

from . import _lib

# Live contexts and plans.  They are closed by an atexit hook -- plans first, then contexts -- i.e. while the interpreter, torch
# and the HIP runtime are all still up.  Left to the garbage collector at interpreter shutdown the order is arbitrary: a
# context destroyed after the HIP runtime's own teardown aborts the process from a destructor ("terminate called after
# throwing an instance of 'std::bad_variant_access'", seen whenever a failing test kept a plan alive in its traceback), so
# __del__ does nothing once the interpreter is finalizing.
_live_contexts = weakref.WeakSet()
_live_plans = weakref.WeakSet()
_live_graphs = weakref.WeakSet()
_live_comms = weakref.WeakSet()


def register_plan(plan):
    _live_plans.add(plan)


@atexit.register
def _close_all():
    for g in list(_live_comms) + list(_live_graphs):          # communicators and graphs first: they hold a pointer to their context
        try:
            g.close()
        except Exception:
            pass
    for p in list(_live_plans):
        try:
            p.close()
        except Exception:
            pass
    for c in list(_live_contexts):
        try:
            c.close()
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        L = _lib.lib()
        h = C.c_void_p()
        st = L.j2k_ctx_create(int(device), C.byref(h))
        if st != _lib.OK:
            raise _lib.J2KError(st, L.j2k_status_string(st).decode())
        self.h = h
        self.device = int(device)
        self.L = L
        self._held = []          # tensors in use by work queued on this context's stream (FramePlan stage calls)
        self._capturing = False
        self._ext = None
        self._graphs = weakref.WeakSet()     # closed with the context: a j2k_graph keeps a pointer to its j2k_ctx
        self._comms = weakref.WeakSet()      # and so does a j2k_comm (dist.Comm)
        _live_contexts.add(self)

    def check(self, st):
        if st != _lib.OK:
            raise _lib.J2KError(st, "%s: %s" % (self.L.j2k_status_string(st).decode(),
                                                 self.L.j2k_ctx_last_error(self.h).decode()))

    def set_option(self, name, value):
        """j2k_ctx_set_option: a tuning option of this context (before its plans are made)"""
        self.check(self.L.j2k_ctx_set_option(self.h, name.encode(), int(value)))

    def hold(self, t):
        """Keep `t` alive until the next sync(): kernels queued on the library stream may still read or write it."""
        self._held.append(t)
        if len(self._held) > 8192 and not self._capturing:   # a caller that never synchronises: do it for them rather than grow
            self.sync()                                      # for ever (not while capturing: a synchronising call fails there)

    def sync(self):
        try:
            self.check(self.L.j2k_ctx_sync(self.h))
        finally:
            self._held.clear()

    def capture(self):
        """Record the asynchronous plan calls made inside the `with` block into a HIP graph instead of running them
        (j2k_ctx_capture_begin / _end); the calls must have run once before.  Returns a Graph through `as`:
            with ctx.capture() as g: plan.forward(x, co); plan.encode_stream(co, ...)
            g.launch()          # replays on the context's stream with the same buffers"""
        return _Capture(self)

    def profile_enable(self, on=True):
        """on: False / True (the forward level-0 dispatches) / 2 (every 5-3 transform dispatch, tagged)"""
        self.check(self.L.j2k_ctx_profile_enable(self.h, int(on)))

    def profile_read_tag(self, tag):
        """(launches, total_ms) of the dispatches with this tag (0 forward level 0, 1 forward deeper levels, 2 inverse level 0,
        3 inverse deeper levels) since the last profile_read / profile_enable; does not reset."""
        n = C.c_int64(0); ms = C.c_double(0)
        self.check(self.L.j2k_ctx_profile_read_tag(self.h, int(tag), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_read(self):
        """(launches, total_ms) of the level-0 DWT launches recorded since the last read."""
        n = C.c_int64(0); ms = C.c_double(0)
        self.check(self.L.j2k_ctx_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    @property
    def stream(self):
        return self.L.j2k_ctx_stream(self.h)

    def close(self):
        if self.h:
            for g in list(self._comms) + list(self._graphs):
                g.close()
            try:
                self.L.j2k_ctx_sync(self.h)      # drain the stream before its buffers and the stream itself go away
            finally:
                self._held.clear()
                self.L.j2k_ctx_destroy(self.h)
                self.h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context(0)
    return _default


class Graph:
    """A recorded sequence of plan calls (j2k_graph).  Keeps the tensors the calls used alive."""

    def __init__(self, ctx, handle, held):
        self.ctx, self.h, self._held = ctx, handle, held
        _live_graphs.add(self)
        ctx._graphs.add(self)

    def launch(self):
        """Replay on the context's stream.  Ordering contract: the replay is ordered behind everything queued so far on torch's
        CURRENT stream of the context's device (e.g. a frame.copy_(new) just before) -- the wait that the stage calls issue
        when they run directly is not part of the recorded graph -- and behind earlier work on the context's own stream.
        Results are ready after ctx.sync().  Raises once the context or the graph has been closed."""
        if self.h is None or self.ctx.h is None:
            raise _lib.J2KError(_lib.ERR_INVALID_ARG, "graph launch after the graph or its context was closed")
        import torch
        if self.ctx._ext is None:
            self.ctx._ext = torch.cuda.ExternalStream(int(self.ctx.stream), device=torch.device("cuda", self.ctx.device))
        self.ctx._ext.wait_stream(torch.cuda.current_stream(self.ctx.device))
        self.ctx.check(self.ctx.L.j2k_graph_launch(self.h))

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.j2k_graph_destroy(self.h)
        self.h = None
        self._held = []

    def __del__(self):
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass


class _Capture:
    def __init__(self, ctx):
        self.ctx = ctx
        self.graph = Graph(ctx, None, [])

    def __enter__(self):
        self.mark = len(self.ctx._held)
        self.ctx.check(self.ctx.L.j2k_ctx_capture_begin(self.ctx.h))
        self.ctx._capturing = True
        return self.graph

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        self.ctx._capturing = False
        rc = self.ctx.L.j2k_ctx_capture_end(self.ctx.h, C.byref(h))
        if et is None:
            self.ctx.check(rc)
            self.graph.h = h
            self.graph._held = list(self.ctx._held[self.mark:])
        elif rc == 0 and h:
            self.ctx.L.j2k_graph_destroy(h)
        return False
