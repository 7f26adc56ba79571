"""Multi-GPU fit: column panels of K sharded block-cyclically over the ranks of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The reference has no distributed
code at all; the structure below is the multi-GPU form of libgpx's own two-level
right-looking Cholesky (csrc/chol.hip):

  * the N x N matrix is cut into outer panels of PANEL_BLOCKS x 128 columns; rank r owns panels p = r (mod R);
  * K-build: every rank assembles only the Gram columns of its own panels (no communication);
  * factorisation, per panel p: the owner factors it on high-priority streams (gpx_dev_chol_panel_split: diagonal chain
    with the rows below solved column by column alongside), the panel is BROADCAST (the path's one real exchange step;
    large messages go as scatter + all-gather, which keeps all seven xGMI links of the source busy instead of one
    ring), every rank applies the rank-1024 update to the panels it owns, reading the panel straight from the receive
    buffer.  The message travels in TWO parts: the HEAD -- the rows of the NEXT panel's diagonal square, all its owner
    needs to start its own chain -- as soon as the chain and those rows' solves are through, the TAIL (the rows below,
    the panel's own square, its inverted diagonal blocks and diagonal) when the far rows are solved: the next chain
    runs underneath the far rows' solves and transfer instead of behind them.  Look-ahead: the owner of panel p+1
    updates and factors it first, off the main stream, and its broadcasts are posted from there, so factorisation and
    transfer overlap with the remaining updates of step p on the main stream;
  * three pre-allocated staging buffers per rank; received panels are copied into the rank's own L
    off the critical path, so after the last step all ranks hold the complete factor (34 GB at N = 65536: fits one
    288 GB MI355X) and `estimate_many` shards the QUERIES with no further communication; alpha is solved
    redundantly per rank (two HBM-bound sweeps, no exchange);
  * a non-positive pivot is agreed on by all ranks (all-reduce MAX of the info word) and answered collectively with the
    reference's retry on K + 1e-5 I (skgpuppy/Covariance.py:180-185), exactly like the single-GPU gpx_fit.

The schedule (`panel_cholesky`) is written against a small `ops` interface so that it runs unchanged on CPU
tensors with the gloo backend (tests/test_distributed_gloo.py supplies a torch-CPU `ops`); `GpxOps` is the
product implementation on libgpx device kernels.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

TILE = 128
PANEL_BLOCKS = 8     # outer panel = 8 x 128 = 1024 columns (same as csrc/chol.hip CHOL_NBP)
JITTER = 1e-5        # skgpuppy/Covariance.py:182
INFO_STALLED = 0x3fffffff   # GPX_INFO_STALLED (include/gpx.h): an in-kernel hand-off of a panel step timed out -- not a property of K


class PanelLayout(object):
    """Block-cyclic ownership of outer column panels."""

    def __init__(self, n, world, panel_blocks=PANEL_BLOCKS, split=None):
        # split: the panel message travels as head + tail.  Default: whenever there is a message at all (world > 1; a group of one
        # rank gains nothing from the extra row slice -- 36.6 against 34.8 ms per C3 fit in the one-rank rehearsal);
        # GPX_PANEL_MESSAGE=split / whole overrides
        env = os.environ.get("GPX_PANEL_MESSAGE", "")
        self.split = (env == "split" or (env != "whole" and int(world) > 1)) if split is None else bool(split)
        self.n = int(n)
        self.npad = (self.n + TILE - 1) // TILE * TILE
        self.nblk = self.npad // TILE
        self.world = int(world)
        self.pb = int(panel_blocks)
        self.npanels = (self.nblk + self.pb - 1) // self.pb

    def blocks(self, p):
        b0 = p * self.pb
        return b0, min(b0 + self.pb, self.nblk)

    def owner(self, p):
        return p % self.world

    def owned(self, rank):
        return [p for p in range(self.npanels) if self.owner(p) == rank]

    def message_elems(self, p):
        """doubles in the message of panel p: rows BELOW its square x width | its square | inverted diagonal blocks | diagonal"""
        b0, b1 = self.blocks(p)
        w = (b1 - b0) * TILE
        return (self.npad - b0 * TILE) * w + (b1 - b0) * TILE * TILE + w

    def head_rows(self, p):
        """rows of panel p's message that travel first: those of the NEXT panel's diagonal square (0: the message is not split)"""
        if not self.split or p + 1 >= self.npanels:
            return 0
        n0, n1 = self.blocks(p + 1)
        return (n1 - n0) * TILE

    def parts(self, p):
        """the parts of panel p's message in the order every rank posts them"""
        return ("head", "tail") if self.head_rows(p) > 0 else ("tail",)

    def part_range(self, p, part):
        """[lo, hi) of the part in the message (doubles)"""
        b0, b1 = self.blocks(p)
        cut = self.head_rows(p) * (b1 - b0) * TILE
        return (0, cut) if part == "head" else (cut, self.message_elems(p))


def panel_cholesky(ops, layout, rank, comm):
    """Right-looking panel Cholesky with look-ahead; returns the agreed info word (0 = ok, > 0 = 1-based failing column)
    when this rank's `ops` holds the complete factor.

    ops : build_panel(p)                      assemble an owned panel
          factor_panel(p, prev)               owner: apply panel `prev` (or None) to panel p, factor it, fill its message
          message(p, part)                    the buffer of that part of panel p's message (owner: filled; others: to receive into)
          adopt(p, part, buf, work)           everyone, once per part, in order: `work.wait()`; after the tail the panel is the update operand
          update_panels(qs, p)                apply panel p to the owned panels qs (ascending)
          finish()                            -> this rank's info word
    comm: broadcast(buf, src, ops, part) -> work with .wait()   (asynchronous; posted by every rank in panel order, head before tail)
          max_int(v) -> the maximum of v over the ranks
    """
    P = layout.npanels
    mine = layout.owned(rank)
    for p in mine:
        ops.build_panel(p)

    def post(p, prev):
        """owner: (update with `prev`,) factor, pack; everyone: post the asynchronous broadcasts of panel p's parts."""
        src = layout.owner(p)
        if src == rank:
            ops.factor_panel(p, prev)
        out = []
        for part in layout.parts(p):
            buf = ops.message(p, part)
            out.append((part, buf, comm.broadcast(buf, src, ops, part)))
        return out

    inflight = post(0, None)
    for p in range(P):
        for part, buf, work in inflight:
            ops.adopt(p, part, buf, work)
        nxt = p + 1
        # host order: the (single, cheap to queue) trailing update first, then the next panel's long chain of small
        # launches -- on the device they run side by side on their own streams, ordered by events only: the next owner's chain
        # is gated on the HEAD's arrival alone, the rows below its square and the main stream's updates on the tail's
        ops.update_panels([q for q in mine if q > nxt], p)
        if nxt < P:
            inflight = post(nxt, p)          # the owner of the next panel updates + factors it off the main stream
    return comm.max_int(ops.finish())


# ---------------------------------------------------------------------------------------------------
# torch.distributed plumbing
# ---------------------------------------------------------------------------------------------------
class _Work(object):
    def __init__(self, works, after=None):
        self.works = works
        self.after = after

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        if self.after is not None:
            self.after()


class TorchComm(object):
    """Panel broadcast over torch.distributed.  Messages of at least `split_bytes` travel as scatter + all-gather: the
    source sends a different 1/R of the panel to every rank over its own link, then the ranks exchange their parts --
    2 (R-1)/R of the message per link instead of the whole message hopping round a ring (xGMI is
    point-to-point, a ring broadcast is per-link bound)."""

    def __init__(self, group=None, split_bytes=8 << 20, exercise_single_rank=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.split_bytes = split_bytes
        self.exercise_single_rank = exercise_single_rank     # tests: issue the collectives even in a group of one rank
        self.nccl = dist.get_backend(group) == "nccl"
        # The transport is chosen ONCE, here, by all ranks together, and never changes afterwards: a rank that switched
        # collectives in the middle of the panel sequence would leave its peers inside a different collective (a hang, not an
        # error).  "plain" (GPX_PANEL_BROADCAST=plain, or the probe below failing on any rank): one dist.broadcast per panel.
        self._split_ok = os.environ.get("GPX_PANEL_BROADCAST", "split") != "plain"
        if self._split_ok and self.nccl and (self.world > 1 or exercise_single_rank):
            self._split_ok = self._probe_split()

    def _agree(self, ok):
        """minimum of `ok` over the ranks (a plain all-reduce: the one collective every backend has)"""
        import torch
        t = torch.tensor([int(ok)], dtype=torch.int32, device="cuda")
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _probe_split(self):
        """one tiny scatter + in-place all-gather on device tensors.  The ranks agree on the verdict after EACH of the two
        collectives: a rank on which the scatter raised must not leave its peers inside the all-gather (they would wait in a
        different collective than the all-reduce this rank goes on to -- a hang, not an error)."""
        import torch
        dist = self.dist
        buf = torch.zeros(self.world * 2, dtype=torch.float64, device="cuda")
        if self.rank == 0:
            buf += torch.arange(self.world * 2, dtype=torch.float64, device="cuda") + 1.0
        parts = list(buf.view(self.world, 2).unbind(0))

        def attempt(what, fn):
            try:
                fn()
                torch.cuda.synchronize()
                return 1
            except (RuntimeError, ValueError, TypeError) as exc:
                sys.stderr.write("[skgpuppy_amd.distributed] %s unavailable on rank %d (%s)\n" % (what, self.rank, exc))
                return 0

        ok = self._agree(attempt("scatter", lambda: dist.scatter(parts[self.rank], scatter_list=parts if self.rank == 0 else None,
                                                                  src=0, group=self.group)))
        if ok:
            ok = self._agree(attempt("in-place all-gather", lambda: dist.all_gather_into_tensor(buf, parts[self.rank], group=self.group)))
        if ok:
            ok = self._agree(torch.equal(buf.cpu(), torch.arange(self.world * 2, dtype=torch.float64) + 1.0))
        if not ok and self.rank == 0:
            sys.stderr.write("[skgpuppy_amd.distributed] panel transport: plain broadcast\n")
        return ok

    def broadcast(self, buf, src, ops=None, part="tail"):
        dist = self.dist
        ctx = ops.comm_stream_context(part) if ops is not None and hasattr(ops, "comm_stream_context") else _NullContext()
        with ctx:
            n = buf.numel()
            if self.world == 1 and not self.exercise_single_rank:
                return _Work([])
            if not self._split_ok or n * buf.element_size() < self.split_bytes or n % self.world:
                return _Work([dist.broadcast(buf, src=src, group=self.group, async_op=True)])
            parts = list(buf.view(self.world, n // self.world).unbind(0))
            mine = parts[self.rank]
            if self.nccl:
                # Both collectives run on RCCL's own stream in the order they are posted.  The all-gather is the IN-PLACE form
                # (its input is this rank's chunk of its output): it reads the chunk on that same stream, after the scatter
                # has written it.  (A copy of the chunk made on the posting stream would race with the scatter and spread the
                # slot's previous contents.)  No fallback here: an error after the scatter is posted must surface.
                w1 = dist.scatter(mine, scatter_list=parts if self.rank == src else None, src=src, group=self.group, async_op=True)
                w2 = dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True)
                return _Work([w1, w2])
            # CPU backends (gloo rehearsal) do not order two asynchronous collectives: run them one after the other
            dist.scatter(mine, scatter_list=parts if self.rank == src else None, src=src, group=self.group)
            return _Work([dist.all_gather(parts, mine.clone(), group=self.group, async_op=True)])

    def max_int(self, v):
        import torch
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


class _NullContext(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class HostStagedComm(object):
    """Rehearsal transport: device tensors are broadcast through host memory over a CPU backend (gloo).  Lets the
    N > 1 code path run with several ranks sharing ONE GPU (RCCL refuses duplicate devices); never used by bench.py
    on a real multi-GPU node."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def broadcast(self, buf, src, ops=None, part="tail"):
        comm = self

        class _W(object):
            def wait(self_w):
                if ops is not None and hasattr(ops, "sync_for_host"):
                    ops.sync_for_host()
                h = buf.detach().cpu().contiguous()
                comm.dist.broadcast(h, src=src, group=comm.group)
                if comm.dist.get_rank(comm.group) != src:
                    buf.copy_(h)
        return _W()

    def max_int(self, v):
        import torch
        t = torch.tensor([int(v)], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


# ---------------------------------------------------------------------------------------------------
# product ops: libgpx kernels on this rank's GPU
# ---------------------------------------------------------------------------------------------------
NSLOTS = 3           # staging buffers per rank: panel p's message lives in slot p % 3 (its previous tenant, p - 3, was read steps ago)


class GpxOps(object):
    """Streams per rank: `main` (the caller's current stream) carries the trailing updates; three high-priority streams carry the
    critical path of the NEXT panel on its owner -- `side` the square's update and the chain, `head` the rows of the panel after it
    (update, column solves, pack, the head's broadcast), `far` the rows below (update, column solves, pack, the tail's broadcast) --
    and `copy` the copies of received panels into L.  Three staging buffers hold the messages of panels p-1, p, p+1; events order
    every reuse of a buffer behind its last readers on every stream."""

    def __init__(self, x_dev, theta, layout, device, rank=0, jitter=0.0):
        import torch
        from . import _gpx
        self.torch, self._gpx, self.lib = torch, _gpx, _gpx.lib
        self.layout = layout
        # an owner's chain runs next to 1 / world of the trailing update: the owner's step picks its square launches accordingly (gpx.h)
        _gpx.check(self.lib.gpx_dev_set_panel_share(int(layout.world)), "gpx_dev_set_panel_share")
        self.rank = rank
        self.x = x_dev                                  # [n, d] float64 on `device`
        self.n, self.d = x_dev.shape
        self.theta = np.ascontiguousarray(theta, dtype=np.float64)
        with np.errstate(divide="ignore"):
            self.vt = float(np.exp(self.theta[1])) + jitter
        self.v = float(np.exp(self.theta[0]))
        self.jitter = jitter
        # inputs scaled by sqrt(w) once: every owned panel's Gram block is then a single asynchronous launch
        sw = torch.as_tensor(np.sqrt(np.exp(self.theta[2:2 + self.d])), device=device)
        self.xw = (x_dev * sw[None, :]).contiguous()
        npad, nblk = layout.npad, layout.nblk
        self.L = torch.empty((npad, npad), dtype=torch.float64, device=device)
        self.Dinv = torch.empty((nblk, TILE, TILE), dtype=torch.float64, device=device)
        self.diag = torch.empty(npad, dtype=torch.float64, device=device)
        self.info = torch.zeros(1, dtype=torch.int32, device=device)
        self.main = torch.cuda.current_stream(device)
        self.side = torch.cuda.Stream(device=device, priority=-1)
        self.head = torch.cuda.Stream(device=device, priority=-1)
        self.far = torch.cuda.Stream(device=device, priority=-1)
        self.copy = torch.cuda.Stream(device=device)
        self.stage = [torch.empty(layout.message_elems(0), dtype=torch.float64, device=device) for _ in range(NSLOTS)]
        self._operand = {}                              # panel -> (device pointer, leading dimension, first row) of its update operand
        self._ev_head = {}                              # panel -> event: the rows of the next panel's square are there (no head part: = tail)
        self._ev_tail = {}                              # panel -> event: the whole message is there
        self._ev_lookahead = {}                         # panel -> event: the main stream has applied every panel but the last one to it
        self._readers = {}                              # panel -> events behind every read of its message buffer
        self._local = {}                                # (own panel, part) -> event: that part's rows are solved (before pack and send)
        self._ev_built = None                           # behind the last build_panel on the main stream (see factor_panel)
        # GPX_SHARD_TIMING=1 (bench): event pairs around the owner's chain (side stream; chol_panel_ms), from the point where the far
        # rows may start to the last row's solve (chol_panel_rows_ms = the whole panel step), around the side stream's wait for the head of the
        # panel before an owned one and around the main stream's wait for each panel's tail -- what the first multi-GPU run is judged
        # on (projected timeline)
        self._timed = os.environ.get("GPX_SHARD_TIMING", "0") not in ("", "0")
        self._pairs = {"chol_panel_ms": [], "chol_panel_rows_ms": [], "exposed_head_wait_ms": [], "exposed_wait_ms": []}
        self.timing = {}
        # GPX_SHARD_CHAOS=<seed> (tests): random busy-waits of up to ~1 ms on randomly chosen streams in front of the schedule's steps -- a
        # stream that runs late must never change the result; a missing event edge would
        chaos = os.environ.get("GPX_SHARD_CHAOS", "")
        self._chaos = np.random.RandomState(int(chaos) + 1000 * rank) if chaos not in ("", "0") else None

    def _shake(self):
        if self._chaos is not None and self._chaos.rand() < 0.4:
            st = self._streams()[self._chaos.randint(5)]
            with self.torch.cuda.stream(st):
                self.torch.cuda._sleep(int(self._chaos.randint(100000, 2500000)))

    def _mark(self, stream):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def _event(self, stream):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        return ev

    # ---- helpers ------------------------------------------------------------------------------
    def _stream_ptr(self, st):
        return ctypes.c_void_p(st.cuda_stream)

    @staticmethod
    def _p(t, byte_offset=0):
        return ctypes.c_void_p(t.data_ptr() + byte_offset)

    def _Lptr(self, row, col):
        return ctypes.c_void_p(self.L.data_ptr() + 8 * (row * self.layout.npad + col))

    def _geom(self, p):
        b0, b1 = self.layout.blocks(p)
        return b0, b1, b0 * TILE, (b1 - b0) * TILE, self.layout.npad - b0 * TILE     # blocks, first column, width, rows

    def _slot(self, p):
        return self.stage[p % NSLOTS][:self.layout.message_elems(p)]

    def _views(self, p):
        """(rows below the square [below, w], square [w, w], inverted diagonal blocks, diagonal) of panel p's message buffer"""
        b0, b1, c0, w, rows = self._geom(p)
        buf = self._slot(p)
        a = (rows - w) * w
        b = a + w * w
        c = b + (b1 - b0) * TILE * TILE
        return buf[:a].view(rows - w, w), buf[a:b].view(w, w), buf[b:c].view(b1 - b0, TILE, TILE), buf[c:c + w]

    def _part_stream(self, part):
        return self.head if part == "head" else self.far

    def comm_stream_context(self, part="tail"):
        """the stream a part's broadcast is posted from: on the owner the part is packed there, elsewhere the receive is ordered
        there behind the last readers of its buffer; either way it overlaps with the main stream's trailing updates"""
        return self.torch.cuda.stream(self._part_stream(part))

    def _streams(self):
        return (self.side, self.head, self.far, self.copy, self.main)

    def sync_for_host(self):
        for st in self._streams():
            st.synchronize()

    def _wait_slot_free(self, p, stream):
        """`stream` waits for every read of the message that lived in panel p's slot before"""
        for ev in self._readers.get(p - NSLOTS, ()):
            stream.wait_event(ev)

    # ---- the ops interface ----------------------------------------------------------------------
    def build_panel(self, p):
        """Gram columns of panel p, rows from its diagonal down (+vt on the diagonal, identity padding)."""
        b0, b1, c0, w, rows = self._geom(p)
        n, d, npad = self.n, self.d, self.layout.npad
        if c0 >= n:      # panel entirely in the padding: identity
            self.L[c0:, c0:c0 + w].zero_()
            self.L[c0:c0 + w, c0:c0 + w].fill_diagonal_(1.0)
            self._ev_built = self._event(self.main)
            return
        xi = ctypes.c_void_p(self.xw.data_ptr() + 8 * c0 * d)
        st = self.lib.gpx_dev_gram_scaled(xi, n - c0, xi, min(c0 + w, n) - c0, d, self.v, self.vt, 0, 1,
                                          self._Lptr(c0, c0), npad, npad - c0, w, self._stream_ptr(self.main))
        self._gpx.check(st, "gpx_dev_gram_scaled(panel %d)" % p)
        self._ev_built = self._event(self.main)       # (recorded HERE, behind the builds alone -- not behind the updates main runs later)

    def _gemm(self, q0, q1, p, stream):
        """C[rows >= q0, columns q0..q1) -= P[rows >= q0] P[rows q0..q1)^T with P = the update operand of panel p"""
        ptr, ld, first = self._operand[p]
        npad = self.layout.npad
        _b0, _b1, _c0, wp, _rows = self._geom(p)
        A = ctypes.c_void_p(ptr + 8 * (q0 - first) * ld)
        lower = 1 if q1 == npad else 0            # the run reaches the last column: square, lower tiles only (the bulk SYRK)
        st = self.lib.gpx_dev_gemm_nt(A, ld, A, ld, self._Lptr(q0, q0), npad, npad - q0, q1 - q0, wp, -1.0, 1.0, lower,
                                      self._stream_ptr(stream))
        self._gpx.check(st, "panel update (columns %d..%d <- panel %d)" % (q0, q1, p))

    def factor_panel(self, p, prev):
        b0, b1, c0, w, rows = self._geom(p)
        hr = self.layout.head_rows(p)
        rowstreams = (self.head, self.far)
        self._shake()
        # The panels were assembled on the MAIN stream.  Nothing else orders a rank's first owned panel behind its own Gram launch when
        # that panel is not panel 0 (its chain waits for the previous panel's head, which travels on other streams): found while porting
        # the schedule to csrc/multi.hip in round 6 -- hidden in practice (panel 0's factorisation takes longer than the builds).
        if self._ev_built is not None:
            for st in (self.side,) + rowstreams:
                st.wait_event(self._ev_built)
        if prev is not None:
            # the chain needs the HEAD of `prev` (this panel's square rows of it) and the main stream's earlier updates of this
            # panel; the rows below the square need prev's TAIL as well
            self.side.wait_event(self._ev_head[prev])
            for st in rowstreams:
                st.wait_event(self._ev_tail[prev])
            la = self._ev_lookahead.pop(p, None)
            if la is not None:
                for st in (self.side,) + rowstreams:
                    st.wait_event(la)
        else:
            self.side.wait_stream(self.main)                      # the panel has been assembled on the main stream
        t0 = self._mark(self.side) if self._timed else None
        t0_rows = self._mark(self.far) if self._timed else None   # (behind the far rows' own gates: the tail of `prev`, the look-ahead update)
        if prev is not None:
            ptr, ldp, first = self._operand[prev]
            _pb0, _pb1, _pc0, wp, _prows = self._geom(prev)
            P = ctypes.c_void_p(ptr + 8 * (c0 - first) * ldp)
        else:
            P, ldp, wp = None, 0, 0
        # update with `prev` + factorisation in one native call: square first (the chain starts at once), the rows below on
        # their own streams ahead of their column solves -- the next panel's square rows apart from the rest
        st = self.lib.gpx_dev_chol_panel_split(self._p(self.L), self.layout.npad, self.layout.nblk, b0, b1, hr // TILE, P, ldp, wp,
                                               self._p(self.Dinv), self._p(self.diag), self._p(self.info), self._stream_ptr(self.side),
                                               self._stream_ptr(self.head), self._stream_ptr(self.far))
        self._gpx.check(st, "gpx_dev_chol_panel_split(%d)" % p)
        chain_done = self._event(self.side)
        self.far.wait_event(chain_done)
        # what this rank's own streams wait for: the solves, not the broadcasts
        if hr > 0:
            self._local[(p, "head")] = self._event(self.head)
        self._local[(p, "tail")] = self._event(self.far)
        if self._timed:
            self._pairs["chol_panel_ms"].append((t0, self._mark(self.side)))
            with self.torch.cuda.stream(self.copy):               # (a stream that nothing else waits for)
                for part in self.layout.parts(p):
                    self.copy.wait_event(self._local[(p, part)])
                self._pairs["chol_panel_rows_ms"].append((t0_rows, self._mark(self.copy)))
        if prev is not None:                                      # the step's reads of prev's message buffer
            self._readers.setdefault(prev, []).extend(self._event(st) for st in (self.side,) + rowstreams)
        if self.layout.world > 1:
            lower, square, dinv, diag = self._views(p)
            below = rows - w
            for st in rowstreams:
                self._wait_slot_free(p, st)
            if hr > 0:
                with self.torch.cuda.stream(self.head):
                    lower[:hr].copy_(self.L[c0 + w:c0 + w + hr, c0:c0 + w])
            with self.torch.cuda.stream(self.far):                # (behind the chain: square, inverted blocks and diagonal are its)
                if below > hr:
                    lower[hr:].copy_(self.L[c0 + w + hr:, c0:c0 + w])
                square.copy_(self.L[c0:c0 + w, c0:c0 + w])
                dinv.copy_(self.Dinv[b0:b1])
                diag.copy_(self.diag[c0:c0 + w])
        # the owner's own updates read the panel in place
        self._operand[p] = (self.L.data_ptr() + 8 * ((c0 + w) * self.layout.npad + c0), self.layout.npad, c0 + w)

    def message(self, p, part):
        self._shake()
        lo, hi = self.layout.part_range(p, part)
        if self.layout.owner(p) != self.rank:
            # the receive is posted from the part's stream: behind every read of the slot's previous tenant
            self._wait_slot_free(p, self._part_stream(part))
        return self._slot(p)[lo:hi]

    def adopt(self, p, part, buf, work):
        torch = self.torch
        self._shake()
        b0, b1, c0, w, rows = self._geom(p)
        hr = self.layout.head_rows(p)
        own = self.layout.owner(p) == self.rank
        lower, square, dinv, diag = self._views(p)
        if own:
            # the owner's streams go on as soon as the rows are SOLVED (factor_panel's events); the broadcast is waited for on the copy
            # stream only, as one more reader of the message buffer
            ev = self._local.pop((p, part))
            with torch.cuda.stream(self.copy):
                work.wait()
                self._readers.setdefault(p, []).append(self._event(self.copy))
        if part == "head":
            # the chain stream waits for the head -- exposed only where this rank owns the next panel
            with torch.cuda.stream(self.side):
                t0 = self._mark(self.side) if self._timed and self.layout.owner(p + 1) == self.rank else None
                if own:
                    self.side.wait_event(ev)
                else:
                    work.wait()
                    ev = self._event(self.side)
                if t0 is not None:
                    self._pairs["exposed_head_wait_ms"].append((t0, self._mark(self.side)))
            self._ev_head[p] = ev
            if not own:
                with torch.cuda.stream(self.copy):
                    self.copy.wait_event(ev)
                    self.L[c0 + w:c0 + w + hr, c0:c0 + w].copy_(lower[:hr])
                    self._readers.setdefault(p, []).append(self._event(self.copy))
            return
        if not own:
            with torch.cuda.stream(self.far):
                work.wait()
                ev = self._event(self.far)
        self._ev_tail[p] = ev
        if hr == 0:
            self._ev_head[p] = ev
        with torch.cuda.stream(self.main):
            t0 = self._mark(self.main) if self._timed else None
            self.main.wait_event(ev)                              # no host block on RCCL
            self.main.wait_event(self._ev_head[p])
            if self._timed:                                       # main-stream idle time in front of panel p's updates
                self._pairs["exposed_wait_ms"].append((t0, self._mark(self.main)))
        if not own:
            self._operand[p] = (lower.data_ptr(), w, c0 + w)
            # copy into this rank's L / Dinv / diag: needed for the complete factor only
            with torch.cuda.stream(self.copy):
                self.copy.wait_event(ev)
                if rows - w > hr:
                    self.L[c0 + w + hr:, c0:c0 + w].copy_(lower[hr:])
                self.L[c0:c0 + w, c0:c0 + w].copy_(square)
                self.Dinv[b0:b1].copy_(dinv)
                self.diag[c0:c0 + w].copy_(diag)
                self._readers.setdefault(p, []).append(self._event(self.copy))
        for d_ in (self._ev_head, self._ev_tail, self._readers):
            d_.pop(p - NSLOTS - 1, None)

    def update_panels(self, qs, p):
        self._shake()
        # consecutive owned panels form one launch (world size 1: all of them = the single bulk SYRK of csrc/chol.hip)
        runs = []
        for q in qs:
            _b0, _b1, c0, w, _rows = self._geom(q)
            if runs and runs[-1][1] == c0:
                runs[-1][1] = c0 + w
            else:
                runs.append([c0, c0 + w])
        for i, (q0, q1) in enumerate(runs):
            self._gemm(q0, q1, p, self.main)
            if i == 0 and qs[0] == p + 2:
                # panel p + 2 now lacks panel p + 1 only: its owner's chain waits for THIS, not for the rest of the step's updates
                self._ev_lookahead[p + 2] = self._event(self.main)
        if runs:
            self._readers.setdefault(p, []).append(self._event(self.main))
        self._operand.pop(p - 2, None)

    def finish(self):
        for st in self._streams():
            st.synchronize()
        if self._timed:
            self.timing = {k: float(sum(a.elapsed_time(b) for a, b in v)) for k, v in self._pairs.items()}
            self.timing["panels_owned"] = len(self._pairs["chol_panel_ms"])
        # The raw status word: 0, a 1-based failing column, or GPX_INFO_STALLED.  Only the owner of a stalled panel sees that value, so
        # nothing is raised HERE -- panel_cholesky's max_int carries it to every rank (it is the largest value) and the caller raises
        # on all of them after the collective (a rank that threw alone would leave the others blocked in the all-reduce).
        return int(self.info.item())


def combine_approx_partials(o, Sigma, v, vt):
    """(mean without meant, variance, sigma2, rest) of UncertaintyPropagationApprox.propagate_GA from the summed partials of
    gpx_propagate_approx_rows (skgpuppy/UncertaintyPropagation.py:397-479): o = [beta.C, beta.tr, C.KinvC, KinvC.tr,
    (J_k.KinvJ_k, beta.J_k) for every k]."""
    o = np.asarray(o, dtype=np.float64)
    S = np.asarray(Sigma, dtype=np.float64)
    d = S.shape[0]
    mu = o[0] + 0.5 * o[1]
    s2 = (v + vt) - o[2]
    var2 = -sum(S[k, k] * (o[4 + 2 * k] - o[5 + 2 * k] ** 2) for k in range(d))
    var3 = -o[3]
    return mu, s2 + var2 + var3, s2, var2 + var3


def rhs_shards(nvec, world):
    """[k0, k1) of the nvec right-hand sides each rank solves for: contiguous, as even as possible, empty for surplus ranks"""
    per, extra = divmod(nvec, world)
    out, k = [], 0
    for r in range(world):
        n = per + (1 if r < extra else 0)
        out.append((k, k + n))
        k += n
    return out


def row_shards(n, world, triangular=False):
    """[lo, hi) per rank: 128-aligned row panels of (almost) equal height -- or, with `triangular`, of (almost) equal AREA
    of the lower triangle (row i of the Exact propagation's j <= i double sum costs i + 1 pairs)."""
    nblk = (n + TILE - 1) // TILE
    if triangular:
        cuts = [int(round(nblk * np.sqrt(r / float(world)))) for r in range(world + 1)]
    else:
        per = (nblk + world - 1) // world
        cuts = [min(nblk, r * per) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, nblk
    return [(min(n, cuts[r] * TILE), min(n, max(cuts[r], cuts[r + 1]) * TILE)) for r in range(world)]


class ShardedGaussianProcess(object):
    """GaussianProcess over R GPUs: sharded K-build + panel-broadcast Cholesky, query-sharded estimate_many.
    Must be constructed collectively by every rank of `group` with identical (x, t, theta)."""

    def __init__(self, x, t, theta_min, group=None, device=None, comm=None, split=None):
        import torch
        import torch.distributed as dist
        from . import _gpx
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        # libgpx works on the calling thread's gpx_set_device choice: bind it to this rank's GPU
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _gpx.check(_gpx.lib.gpx_set_device(idx), "gpx_set_device")
        self._on_device = dist.get_backend(group) == "nccl"       # collectives on device tensors (RCCL has no host path)
        self.x = x
        self.n, self.d = np.shape(x)
        self.meant = np.mean(t)
        self.t = np.asarray(t, dtype=np.float64) - self.meant
        self.theta_min = np.ascontiguousarray(theta_min, dtype=np.float64)
        self._x_dev = torch.as_tensor(_gpx.f64(x)).to(self.device)
        self._t_dev = torch.as_tensor(_gpx.f64(self.t)).to(self.device)
        self.layout = PanelLayout(self.n, self.world, split=split)
        self._comm = comm if comm is not None else TorchComm(group)
        # GPX_PANEL_MESSAGE is read per rank: ranks that disagreed would post mismatched collectives (head + tail against one part)
        # and hang.  Agree once, collectively, and fail on every rank if they differ.
        flag = int(self.layout.split)
        if self._comm.max_int(flag) != flag or -self._comm.max_int(-flag) != flag:
            raise RuntimeError("ranks disagree on the panel message form (GPX_PANEL_MESSAGE / split=): set it identically on every rank")
        self._ops = None
        self._h = ctypes.c_void_p()
        self.jitter = 0.0
        self.refit()

    def refit(self):
        """(re)run the sharded fit on the resident inputs -- one benchmark `fit` step.  A non-positive pivot anywhere is
        seen by every rank (max of the info words) and answered by ONE collective retry on K + 1e-5 I, the reference's
        fallback (skgpuppy/Covariance.py:180-185); if that fails too every rank raises LinAlgError."""
        import torch
        from . import _gpx
        self.close()
        with torch.cuda.device(self.device):
            for jitter in (0.0, JITTER):
                ops = GpxOps(self._x_dev, self.theta_min, self.layout, self.device, rank=self.rank, jitter=jitter)
                info = panel_cholesky(ops, self.layout, self.rank, self._comm)
                if info == 0:
                    break
                del ops
                if info == INFO_STALLED:
                    # agreed by the all-reduce inside panel_cholesky: EVERY rank raises, before the jitter retry (a timed-out
                    # hand-off is not a property of K and is never answered with +1e-5 I, include/gpx.h)
                    raise RuntimeError("a hand-off inside a panel step of the sharded factorisation timed out on some rank "
                                       "(GPX_WAIT_LIMIT_MS); the factor is invalid")
            if info > 0:
                raise np.linalg.LinAlgError("covariance matrix not positive definite (leading minor %d), also with +1e-5 jitter" % info)
            self._ops = ops
            self.jitter = jitter
            self.fit_timing = dict(ops.timing)        # GPX_SHARD_TIMING=1: owner-side panel step / exposed message waits of this fit (ms)
            st = _gpx.lib.gpx_adopt_factor(ctypes.c_void_p(self._x_dev.data_ptr()), ctypes.c_void_p(self._t_dev.data_ptr()),
                                           self.n, self.d, _gpx.ptr(self.theta_min), ops._p(ops.L), ops._p(ops.Dinv),
                                           ops._p(ops.diag), jitter, None, ctypes.byref(self._h))
            _gpx.check(st, "gpx_adopt_factor")

    def close(self):
        if getattr(self, "_h", None):
            try:
                from . import _gpx
                _gpx.lib.gpx_free(self._h)
            except Exception:      # interpreter shutdown: the module table is already being torn down
                pass
            self._h = ctypes.c_void_p()
        self._ops = None

    __del__ = close

    def shard(self, m):
        """[lo, hi) of the m queries this rank serves."""
        per = (m + self.world - 1) // self.world
        lo = min(m, self.rank * per)
        return lo, min(m, lo + per)

    def _all_gather(self, mine):
        """all-gather of equally shaped device tensors; the result comes back on the host.  With RCCL every operand stays
        on the GPU (RCCL has no CPU path); a CPU backend (gloo rehearsal) gets host tensors."""
        import torch
        import torch.distributed as dist
        send = mine if self._on_device else mine.cpu()
        gathered = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(gathered, send, group=self.group)
        return [g.cpu() for g in gathered]

    def estimate_local(self, xs_dev, mean_dev, var_dev):
        """predict this rank's resident query shard (device tensors, mean WITHOUT meant)."""
        from . import _gpx
        m = xs_dev.shape[0]
        st = _gpx.lib.gpx_predict(self._h, ctypes.c_void_p(xs_dev.data_ptr()), m, ctypes.c_void_p(mean_dev.data_ptr()),
                                  ctypes.c_void_p(var_dev.data_ptr()))
        _gpx.check(st, "gpx_predict")

    def estimate_many(self, x_stars):
        """Same contract as GaussianProcess.estimate_many (skgpuppy/GaussianProcess.py:68-80); every rank returns
        the full arrays (query shards are all-gathered)."""
        import torch
        from . import _gpx
        xs = _gpx.f64(np.array(x_stars))
        m = xs.shape[0]
        lo, hi = self.shard(m)
        per = (m + self.world - 1) // self.world
        out = torch.zeros((2, per), dtype=torch.float64, device=self.device)
        if hi > lo:
            xs_dev = torch.as_tensor(xs[lo:hi]).to(self.device)
            self.estimate_local(xs_dev, out[0, :hi - lo], out[1, :hi - lo])
        full = torch.cat(self._all_gather(out), dim=1).numpy()[:, :m]
        return full[0] + self.meant, full[1]

    # ---- uncertainty propagation (last row) ------------------------------------------------------
    def propagate_GA(self, u, Sigma):
        """UncertaintyPropagationApprox.propagate_GA on this rank's copy of the factor
        (skgpuppy/UncertaintyPropagation.py:381-523): every rank holds the complete L after the panel broadcasts, so a
        single propagation needs no exchange at all -- it runs as two triangular sweeps on the right-hand-side block
        (no K^-1: 34 GB at N = 65536 stay unallocated).  Returns (mean + meant, variance)."""
        from . import _gpx
        u = _gpx.f64(u)
        S = _gpx.f64(Sigma)
        out = [ctypes.c_double() for _ in range(4)]
        _gpx.check(_gpx.lib.gpx_propagate_approx(self._h, _gpx.ptr(u), _gpx.ptr(S), *[ctypes.byref(o) for o in out]),
                   "gpx_propagate_approx")
        return out[0].value + self.meant, out[1].value

    def propagate_GA_sharded(self, u, Sigma, via="solve"):
        """ONE propagation shared by all ranks (collective); the 4 + 2 d partial sums of
        skgpuppy/UncertaintyPropagation.py:412-479 meet in one all-reduce.

        via="solve" (default): the d + 1 right-hand sides [C, J_1..J_d] are dealt to the ranks, each rank runs the two-sweep
            triangular solver on ITS vectors against its copy of the factor (gpx_propagate_approx_rhs).  No rank ever
            materialises K^-1 (34 GB and ~3 s of N^3 work per rank at N = 65536).
        via="kinv": rank r passes over its row panel of K^-1 only ((K^-1 v)_i and every quadratic form are sums over the rows;
            gpx_propagate_approx_rows).  Only that ROW PANEL of K^-1 is built, on first use (E^T L^-T L^-1 for the panel's unit
            rows: 2 N^3 / R flop and three N / R x N buffers per rank instead of 2 N^3 / 3 and two N x N matrices): the path for
            MANY propagations on one fit (inverse propagation, design studies), where each call then reads 1/R of K^-1 per GPU."""
        import torch
        import torch.distributed as dist
        from . import _gpx
        u = _gpx.f64(u)
        S = _gpx.f64(Sigma)
        part = np.zeros(4 + 2 * self.d)
        if via == "kinv":
            lo, hi = row_shards(self.n, self.world)[self.rank]
            _gpx.check(_gpx.lib.gpx_propagate_approx_rows(self._h, _gpx.ptr(u), _gpx.ptr(S), lo, hi, _gpx.ptr(part)),
                       "gpx_propagate_approx_rows")
        elif via == "solve":
            k0, k1 = rhs_shards(self.d + 1, self.world)[self.rank]
            _gpx.check(_gpx.lib.gpx_propagate_approx_rhs(self._h, _gpx.ptr(u), _gpx.ptr(S), k0, k1, _gpx.ptr(part)),
                       "gpx_propagate_approx_rhs")
        else:
            raise ValueError("via must be 'solve' or 'kinv'")
        tot = torch.as_tensor(part)
        if self._on_device:
            tot = tot.to(self.device)
        if self.world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        mu, var, _s2, _rest = combine_approx_partials(tot.cpu().numpy(), S, float(np.exp(self.theta_min[0])),
                                                      float(np.exp(self.theta_min[1])))
        return mu + self.meant, var

    def propagate_exact_sharded(self, u, Sigma):
        """UncertaintyPropagationExact.propagate_GA (skgpuppy/UncertaintyPropagation.py:246-379) shared by all ranks
        (collective): the j <= i double sum over (Kinv_ij - beta_i beta_j) L_ij is cut into row panels of equal area, each rank
        sums its panel (gpx_propagate_exact_rows) and two partial sums meet in one all-reduce."""
        import torch
        import torch.distributed as dist
        from . import _gpx
        u = _gpx.f64(u)
        S = _gpx.f64(Sigma)
        lo, hi = row_shards(self.n, self.world, triangular=True)[self.rank]
        part = np.zeros(3)
        _gpx.check(_gpx.lib.gpx_propagate_exact_rows(self._h, _gpx.ptr(u), _gpx.ptr(S), lo, hi, _gpx.ptr(part)),
                   "gpx_propagate_exact_rows")
        tot = torch.as_tensor(part[:2].copy())
        if self._on_device:
            tot = tot.to(self.device)
        if self.world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        p0, p1 = [float(v_) for v_ in tot.cpu()]
        v, vt = float(np.exp(self.theta_min[0])), float(np.exp(self.theta_min[1]))
        return p0 + self.meant, (v + vt) - part[2] * p1 - p0 * p0

    def propagate_many(self, us, Sigmas):
        """many independent propagations (the inverse-propagation and design-study workload): the CALLS are sharded
        across the ranks, results all-gathered -- one small collective for the whole batch."""
        import torch
        us = np.asarray(us, dtype=np.float64)
        k = us.shape[0]
        lo, hi = self.shard(k)
        per = (k + self.world - 1) // self.world
        mine = torch.zeros((per, 2), dtype=torch.float64)
        for i in range(lo, hi):
            mine[i - lo, 0], mine[i - lo, 1] = self.propagate_GA(us[i], Sigmas[i])
        full = torch.cat(self._all_gather(mine.to(self.device)), dim=0).numpy()[:k]
        return full[:, 0], full[:, 1]


# ---------------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): strong scaling of fit + estimate_many at config C4
# ---------------------------------------------------------------------------------------------------
def bench_main(args):
    import torch
    import torch.distributed as dist
    from . import _gpx
    import bench as bench_mod

    # RCCL writes a version banner to stdout when the communicator is created; the bench contract is ONE JSON line on
    # stdout, so everything but that line goes to stderr: fd 1 is pointed at fd 2 for the run and the saved descriptor
    # is used for the result
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # a stuck collective must not hold the node: after GPX_BENCH_WATCHDOG seconds (default 1200, 0 = never) every rank writes its
    # Python stacks to stderr and exits non-zero
    # (a STALL detector, re-armed after every step and every leg: a long but progressing run -- many steps, the C4 one-GPU
    # reference leg -- is never cut off)
    watchdog = int(os.environ.get("GPX_BENCH_WATCHDOG", "1200"))
    import faulthandler

    def rearm():
        if watchdog > 0:
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(watchdog, exit=True)
    rearm()
    os.environ.setdefault("GPX_SHARD_TIMING", "1")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local)
    _gpx.check(_gpx.lib.gpx_set_device(local), "gpx_set_device")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    wl = bench_mod.WORKLOADS[args.workload or "c4"]
    N, d, M = wl["N"], wl["d"], wl["M"]
    x, t, xs, theta = bench_mod.recipe(N, d, M)
    dev = torch.device("cuda", local)
    gp = ShardedGaussianProcess(x, t, theta, device=dev)          # first fit = warm-up of allocations / RCCL rings
    lo, hi = gp.shard(M)
    xs_dev = torch.as_tensor(xs[lo:hi]).to(dev)
    mean_d = torch.empty(hi - lo, dtype=torch.float64, device=dev)
    var_d = torch.empty(hi - lo, dtype=torch.float64, device=dev)
    t_fit = t_pred = 0.0

    def step(timed):
        nonlocal t_fit, t_pred
        a = time.perf_counter()
        gp.refit()
        torch.cuda.synchronize()
        b = time.perf_counter()
        gp.estimate_local(xs_dev, mean_d, var_d)
        torch.cuda.synchronize()
        c = time.perf_counter()
        if timed:
            t_fit += b - a
            t_pred += c - b
            for k_, v_ in getattr(gp, "fit_timing", {}).items():
                shard_ms[k_] = shard_ms.get(k_, 0.0) + v_
        rearm()

    shard_ms = {}
    for _ in range(args.warmup):
        step(False)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0, t_fit, t_pred], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    el, tf, tp = [float(v) for v in elapsed.cpu()]

    # Reference for the strong-scaling numbers, outside the timed region: the SAME workload on ONE GPU (rank 0, single-GPU
    # library path), so that the line carries its own 1-GPU baseline (the N=1 bench line is the C3 headline, another workload)
    one_gpu = None
    if (world > 1 or os.environ.get("GPX_BENCH_FORCE_1GPU_REF")) and rank == 0 and not os.environ.get("GPX_BENCH_SKIP_1GPU_REF"):
        xd = torch.as_tensor(x).to(dev)
        td = torch.as_tensor(t - np.mean(t)).to(dev)
        xq = torch.as_tensor(xs).to(dev)
        mq = torch.empty(M, dtype=torch.float64, device=dev)
        vq = torch.empty(M, dtype=torch.float64, device=dev)
        vp = lambda tt: ctypes.c_void_p(tt.data_ptr())  # noqa: E731
        th = np.ascontiguousarray(theta, dtype=np.float64)
        # the same protocol as the sharded legs: args.warmup untimed steps, then args.steps timed ones
        acc = [0.0, 0.0]
        for rep in range(args.warmup + args.steps):
            h1 = ctypes.c_void_p()
            torch.cuda.synchronize()
            a = time.perf_counter()
            _gpx.check(_gpx.lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h1)), "gpx_fit (1-GPU reference)")
            b = time.perf_counter()
            _gpx.check(_gpx.lib.gpx_predict(h1, vp(xq), M, vp(mq), vp(vq)), "gpx_predict (1-GPU reference)")
            torch.cuda.synchronize()
            c = time.perf_counter()
            _gpx.lib.gpx_free(h1)
            rearm()
            if rep >= args.warmup:
                acc[0] += b - a
                acc[1] += c - b
        _gpx.lib.gpx_pool_trim()
        k_ = float(max(1, args.steps))
        one_gpu = {"ms_per_step": (acc[0] + acc[1]) / k_ * 1e3, "fit_ms": acc[0] / k_ * 1e3, "predict_ms": acc[1] / k_ * 1e3,
                   "value": (N + M) * k_ / (acc[0] + acc[1]), "steps": args.steps, "warmup": args.warmup,
                   "note": "same workload, one GPU, single-GPU library path (= bench.py --gpus 1 --workload %s), timed on rank 0 "
                           "after the sharded legs with the same warm-up / step counts" % (args.workload or "c4")}
    rearm()
    dist.barrier()
    # per-rank owner-side panel time and exposed message waits of the timed fits (ms per step), gathered on rank 0
    keys = ["chol_panel_ms", "chol_panel_rows_ms", "exposed_head_wait_ms", "exposed_wait_ms", "panels_owned"]
    mine_t = torch.tensor([shard_ms.get(k_, 0.0) / max(1, args.steps) for k_ in keys], dtype=torch.float64, device=dev)
    all_t = [torch.zeros_like(mine_t) for _ in range(world)]
    dist.all_gather(all_t, mine_t)
    per_rank = [{k_: float(v_) for k_, v_ in zip(keys, row.cpu())} for row in all_t]
    if rank == 0:
        flops = N ** 3 / 3.0 + float(N) * N * M
        line = (json.dumps({
            "metric": "GP fit+predict pts/sec (K+Cholesky, N=%d d=%d)" % (N, d),
            "value": (N + M) * args.steps / el,
            "unit": "pts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: N=%d d=%d M=%d, column panels of K (1024 wide) block-cyclic over %d GPUs, "
                                   "RCCL panel broadcast (%s), query-sharded estimate_many" % (
                                       (args.workload or "c4").upper(), N, d, M, world,
                                       "head + tail per panel" if gp.layout.split else "one message per panel"),
                       "global_batch": N + M, "parallelism": "panel-sharded x%d" % world},
            "fit_ms": tf / args.steps * 1e3,
            "predict_ms": tp / args.steps * 1e3,
            "one_gpu_same_workload": one_gpu,
            # strong scaling against the SAME workload on one GPU (same warm-up / steps, timed after the sharded legs on rank 0)
            "scaling_efficiency": (one_gpu["ms_per_step"] / (world * el / args.steps * 1e3)) if one_gpu else None,
            "per_rank_fit": per_rank,
            "roofline": {"kernel": "gemm_nt_f64_kernel (v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                         "achieved": flops * args.steps / el / 1e12 / world, "peak": bench_mod.FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops * args.steps / el / 1e12 / world / bench_mod.FP64_MFMA_PEAK_TFLOPS,
                         "traffic": None,
                         "note": "whole-step algorithmic flops (N^3/3 + N^2 M) per GPU-second; per-kernel event timing is the N=1 line"},
            "cpu_baseline": None,
            "cpu_baseline_note": "the host-CPU baseline (oracle on the full C3 workload, cores stated) is carried by the N=1 line only: "
                                 "at N=65536 the reference algorithm needs 2 N^3 = 5.6e14 flop and ~100 GB on the host",
        }))
        os.write(result_fd, (line + "\n").encode())
    gp.close()
    dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(result_fd, 1)
    os.close(result_fd)
