"""Tensor-level wrappers over the C ABI (include/arlib_amd.h).

PyTorch is plumbing here: device memory, streams, shapes.  Every wrapper validates operand shapes,
dtypes, devices and contiguity on the host *before* a kernel is launched (a faulting kernel can reset
the whole GPU host), then passes raw pointers + sizes + the current HIP stream to libarlib_amd.so.
No op has a CPU fallback.
"""
import ctypes as C
import os
import time
import numpy as np
import torch

from . import _lib
from ._lib import arl_csr, check

DEFAULT_CHUNK = 512
EVENT_HOOK = None      # bench.py installs an object with begin(tag)/end(token) to bracket SpMM launches with HIP events


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """The caller's current HIP stream (raw handle).  torch.cuda.current_stream() builds a Stream object per call (~8 us, on
    every kernel launch); the raw getter is the same lookup without the object."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s: expected a torch.Tensor' % name)
    if not t.is_cuda:
        raise _lib.ArlError('%s: must live on the GPU (no CPU fallback in arlib_amd)' % name)
    if t.dtype != dtype:
        raise TypeError('%s: dtype %s, expected %s' % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise ValueError('%s: must be contiguous' % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError('%s: expected %d dims, got %s' % (name, ndim, tuple(t.shape)))
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------------------------ graph
class CSRGraph:
    """Device CSR of the normalised (U+I)^2 adjacency plus the long-row plan used by the SpMM kernels.

    rowptr/col/val may be given as numpy arrays or torch tensors; they are validated on the host
    (monotone rowptr, col in range) because the kernels trust them.
    """

    def __init__(self, rowptr, col, val, device, chunk=DEFAULT_CHUNK, validate=True, n_cols=None):
        rp = np.ascontiguousarray(rowptr.cpu().numpy() if isinstance(rowptr, torch.Tensor) else rowptr).astype(np.int64)
        n = len(rp) - 1
        nnz = int(rp[-1])
        if n < 0 or rp[0] != 0 or nnz >= 2 ** 31 or n >= 2 ** 31:
            raise ValueError('CSRGraph: bad rowptr / size out of int32 range')
        if validate:
            if np.any(np.diff(rp) < 0):
                raise ValueError('CSRGraph: rowptr must be non-decreasing')
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.n_rows, self.nnz, self.chunk = n, nnz, int(chunk)
        self.n_cols = n if n_cols is None else int(n_cols)      # rectangular blocks (user-sharded graph) gather from an [n_cols, d] operand
        self.rowptr = torch.as_tensor(rp.astype(np.int32)).to(self.device)
        self.col = (col if isinstance(col, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(col, dtype=np.int32))).to(self.device, torch.int32).contiguous()
        self.val = (val if isinstance(val, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(val, dtype=np.float32))).to(self.device, torch.float32).contiguous()
        if self.col.numel() != nnz or self.val.numel() != nnz:
            raise ValueError('CSRGraph: col/val length != rowptr[-1]')
        if validate and nnz:
            lo, hi = int(self.col.min()), int(self.col.max())
            if lo < 0 or hi >= self.n_cols:
                raise ValueError('CSRGraph: column index out of range [0,%d): %d..%d' % (self.n_cols, lo, hi))
        # long-row plan (host, once per graph)
        deg = np.diff(rp)
        long_rows = np.nonzero(deg > self.chunk)[0] if self.chunk > 0 else np.zeros(0, np.int64)
        if len(long_rows):
            nch = (deg[long_rows] + self.chunk - 1) // self.chunk
            first = np.concatenate([[0], np.cumsum(nch)[:-1]])
            crow = np.repeat(long_rows, nch)
            k = np.arange(int(nch.sum())) - np.repeat(first, nch)
            cbeg = rp[crow] + k * self.chunk
            cend = np.minimum(cbeg + self.chunk, rp[crow + 1])
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
            self.chunk_row, self.chunk_begin, self.chunk_end = t(crow), t(cbeg), t(cend)
            self.long_row, self.long_first, self.long_count = t(long_rows), t(first), t(nch)
            self.n_chunks, self.n_long = int(nch.sum()), len(long_rows)
        else:
            self.chunk_row = self.chunk_begin = self.chunk_end = self.long_row = self.long_first = self.long_count = None
            self.n_chunks = self.n_long = 0
        self._partial = None
        self.dinv = None
        self.blocked = None

    @classmethod
    def from_device(cls, rowptr, col, val, nnz, chunk=DEFAULT_CHUNK, n_cols=None, long_entries=None):
        """CSR graph from DEVICE arrays without any host read (the constructor copies rowptr to the host for its long-row plan: a stream
        synchronisation plus milliseconds of host work -- too much for an operator rebuilt inside an attack step, attack/White/CLeaR.py).
        The long-row plan is built on the device with static shapes: room for nnz / chunk long rows and 2 nnz / chunk chunks, the unused
        tail filled with copies of the LAST real long row (same partial range: it is recomputed with the same value) and with empty chunks.
        The caller guarantees at least one row longer than `chunk`, trusts rowptr / col (no validation), passes nnz = rowptr[-1] as a
        Python int, and uses the graph only for SpMMs whose output does not alias an input (the duplicated long rows are written twice)."""
        dev = rowptr.device
        n = rowptr.numel() - 1
        g = object.__new__(cls)
        g.device, g.n_rows, g.nnz, g.chunk = dev, n, int(nnz), int(chunk)
        g.n_cols = n if n_cols is None else int(n_cols)
        rp = rowptr.to(torch.int64)
        g.rowptr = rowptr.to(torch.int32).contiguous()
        g.col, g.val = col.to(torch.int32).contiguous(), val.to(torch.float32).contiguous()
        deg = rp[1:] - rp[:-1]
        is_long = deg > g.chunk
        # a long row holds more than `chunk` entries; `long_entries` (optional) = an upper bound on the entries that can sit in long rows
        NL = max(1, (g.nnz if long_entries is None else int(long_entries)) // g.chunk)
        NC = 2 * NL                                            # ceil(deg / chunk) <= deg / chunk + 1 per long row
        n_long = is_long.sum()
        order = torch.argsort((~is_long).to(torch.int8), stable=True)[:NL]      # the long rows in ascending order, then filler
        slot = torch.minimum(torch.arange(NL, device=dev), n_long - 1).clamp_(min=0)
        rows = order[slot]                                     # entries past n_long repeat the last real long row
        nch = (deg[rows] + g.chunk - 1) // g.chunk
        real = torch.arange(NL, device=dev) < n_long
        cum = torch.cumsum(torch.where(real, nch, torch.zeros_like(nch)), 0)     # chunks before and including long row l
        first = cum - nch
        first = torch.where(real, first, first[slot])          # the copies share the original's partial range
        c = torch.arange(NC, device=dev)
        l = torch.searchsorted(cum, c, right=True).clamp_(max=NL - 1)
        l = torch.minimum(l, (n_long - 1).clamp(min=0))
        crow = rows[l]
        k = c - first[l]
        cbeg = torch.minimum(rp[crow] + k * g.chunk, rp[crow + 1])              # chunks past the real ones come out empty (begin = end = row end)
        cend = torch.minimum(cbeg + g.chunk, rp[crow + 1])
        i32 = lambda a: a.to(torch.int32).contiguous()
        g.chunk_row, g.chunk_begin, g.chunk_end = i32(crow), i32(cbeg), i32(cend)
        g.long_row, g.long_first, g.long_count = i32(rows), i32(first), i32(nch)
        g.n_chunks, g.n_long = NC, NL
        g._partial = None
        g.dinv = None
        g.blocked = None
        return g

    def enable_blocked(self, split=None, rows_per_wave=32, hub=None, col_block=1024, min_waves=0, unroll=None, split_hubs=True, piece=None, wpg=None, wave_multiple=None, col_order=None):
        """Attach a register-blocked plan (BlockedPlan): full-table SpMMs at d = 64 then run through arl_spmm_blocked_*.
        `split` = number of users of a bipartite adjacency: user rows and item rows get separate launches (they gather from
        different tables).  Returns self."""
        sets = [(0, self.n_rows)] if not split or split >= self.n_rows else [(0, int(split)), (int(split), self.n_rows)]
        self.blocked = BlockedPlan(self, sets, rows_per_wave, hub, col_block, min_waves, unroll, split_hubs, piece, wpg, wave_multiple, col_order)
        return self

    def chunks_only(self, rows, chunk=None):
        """View that computes ONLY the given rows, all of them through the chunk plan (row tasks disabled: n_rows = 0)."""
        chunk = self.chunk if chunk is None else int(chunk)
        rows_h = rows.cpu().numpy().astype(np.int64)
        rp = self.rowptr.cpu().numpy().astype(np.int64)
        deg = rp[rows_h + 1] - rp[rows_h]
        nch = np.maximum((deg + chunk - 1) // chunk, 1)
        first = np.concatenate([[0], np.cumsum(nch)[:-1]])
        crow = np.repeat(rows_h, nch)
        k = np.arange(int(nch.sum())) - np.repeat(first, nch)
        cbeg = rp[crow] + k * chunk
        cend = np.minimum(cbeg + chunk, rp[crow + 1])
        g = object.__new__(CSRGraph)
        g.__dict__.update(self.__dict__)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
        g.chunk = chunk
        g.chunk_row, g.chunk_begin, g.chunk_end = t(crow), t(cbeg), t(cend)
        g.long_row, g.long_first, g.long_count = t(rows_h), t(first), t(nch)
        g.n_chunks, g.n_long = int(nch.sum()), len(rows_h)
        g._partial = None
        g._rows_disabled = True
        g.blocked = None
        g._row_tasks = None
        return g

    def enable_masked_order(self):
        """Attach the (row, begin, end, 0) task table of the flag-masked hop: rows in descending edge-count order, so the four lane groups of a
        wave own rows of (nearly) equal length -- a wave lasts as long as its longest row, and row lengths are heavy-tailed.  Pattern-only:
        shared by with_values() copies."""
        if getattr(self, '_row_tasks', None) is None and self.n_rows > 0 and not getattr(self, '_rows_disabled', False):
            rp = self.rowptr.long()
            deg = rp[1:] - rp[:-1]
            order = torch.sort(deg, descending=True, stable=True)[1]
            self._row_tasks = torch.stack([order, rp[order], rp[order + 1], torch.zeros_like(order)], 1).to(torch.int32).contiguous()
        return self

    def with_values(self, val):
        """Same pattern and plan, different edge values (shares index tensors)."""
        g = object.__new__(CSRGraph)
        g.__dict__.update(self.__dict__)
        g.val = _dev(val, torch.float32, 'val', 1)
        if g.val.numel() != self.nnz:
            raise ValueError('with_values: wrong length')
        if self.blocked is not None:
            g.blocked = self.blocked.with_values(g)
        return g

    def _struct(self, d):
        if self.n_chunks and (self._partial is None or self._partial.numel() < self.n_chunks * d):
            self._partial = torch.empty(self.n_chunks * d, dtype=torch.float32, device=self.device)
        s = arl_csr()
        s.n_rows, s.nnz = (0 if getattr(self, '_rows_disabled', False) else self.n_rows), self.nnz
        s.rowptr, s.col, s.val = self.rowptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr()
        s.chunk, s.n_chunks, s.n_long = self.chunk, self.n_chunks, self.n_long
        if self.n_chunks:
            s.chunk_row, s.chunk_begin, s.chunk_end = self.chunk_row.data_ptr(), self.chunk_begin.data_ptr(), self.chunk_end.data_ptr()
            s.long_row, s.long_first, s.long_count = self.long_row.data_ptr(), self.long_first.data_ptr(), self.long_count.data_ptr()
            s.partial = self._partial.data_ptr()
        rt = getattr(self, '_row_tasks', None)
        s.row_tasks = rt.data_ptr() if rt is not None else None
        return s

    def spmm_bytes(self, d):
        """Algorithmic (compulsory) bytes of one SpMM, SURVEY 8d: 8E + 4(N+1) + 8Nd."""
        return 8 * self.nnz + 4 * (self.n_rows + 1) + 8 * self.n_rows * d


class BlockedPlan:
    """Plan of the register-blocked SpMM (include/arlib_amd.h: arl_blocked) for a CSRGraph, built on the device once per graph.

    Per row set (one launch each): rows with at most `hub` edges are sorted by edge count and dealt longest-first to the least
    loaded of ceil(n / rows_per_wave) waves (arl_lpt_deal), so that every wave carries the same number of edges; a wave's edges
    are sorted by (column block of `col_block` rows, slot) and padded to a multiple of 64 records.  Rows above `hub` edges stay
    with the chunked CSR kernel (`self.hub`: a chunks_only view of the graph), and so does a whole row set that would fill fewer
    than `min_waves` waves (too few streams to cover the memory latency: measured 3x slower at 311 waves)."""

    LONG_WAVES = 3072

    def __init__(self, A, row_sets, rows_per_wave=32, hub=None, col_block=1024, min_waves=0, unroll=None, split_hubs=True, piece=None, wpg=None, wave_multiple=None, col_order=None):
        if rows_per_wave not in (16, 32):
            raise ValueError('BlockedPlan: rows_per_wave must be 16 or 32')
        if A.n_cols >= 1 << 24:
            raise ValueError('BlockedPlan: columns must fit 24 bits')
        if (hub is not None and hub < 1) or col_block < 1 or (piece is not None and piece < 1):
            raise ValueError('BlockedPlan: hub, piece and col_block must be positive')
        self.rpw, self.col_block = int(rows_per_wave), int(col_block)
        # hub / piece / wpg / wave_multiple = None: chosen per row set from its edges per wave (measured on cfg2, tools/hop_experiments.py):
        #   a row longer than the average wave load cannot be balanced, so the threshold is the largest power of two below that load
        #   (capped at 4096: user rows 1024, item rows 4096); a set of long streams (> 4096 edges per wave: ~12 waves per CU, all resident)
        #   is dealt over a multiple of 256 waves and launched one wave per workgroup, so every CU carries the same number of streams
        self._hub_arg, self._piece_arg, self._wpg_arg, self._wm_arg = hub, piece, wpg, wave_multiple
        self.split_hubs = bool(split_hubs)
        self.col_order = col_order
        dev = A.device
        rp_all = A.rowptr.long()
        n_cb = (A.n_cols + self.col_block - 1) // self.col_block
        col_pos = None
        if col_order == 'degree':                 # sweep the operand rows hottest-first: position of a column = its rank by in-plan edge count
            cdeg = torch.bincount(A.col.long(), minlength=A.n_cols)
            col_pos = torch.empty(A.n_cols, dtype=torch.int64, device=dev)
            col_pos[torch.sort(cdeg, descending=True, stable=True)[1]] = torch.arange(A.n_cols, device=dev)
        elif col_order is not None:
            raise ValueError('BlockedPlan: col_order must be None or "degree"')
        self.sets, hub_rows = [], []
        # a set of LONG rows (thousands of edges per wave) runs best with ~3 000 waves per launch -- all resident, 12 per CU, sweeping together
        # (cfg2's item rows: 3 049); larger sets are cut into contiguous row ranges of that many waves (4 M x 400 K: one launch of 12 194
        # waves 3.56 ms, four launches 3.10 ms; cutting cfg2's own 3 049 waves in two costs 0.85 vs 0.54 ms, hence only from 2 x 3 072 on)
        cut = []
        for lo, hi in row_sets:
            if not (0 <= lo <= hi <= A.n_rows):
                raise ValueError('BlockedPlan: bad row set')
            waves = (hi - lo + self.rpw - 1) // self.rpw
            edges = int(rp_all[hi] - rp_all[lo])
            k = waves // self.LONG_WAVES if (waves >= 2 * self.LONG_WAVES and edges >= 4096 * waves) else 1
            cut += [(lo + ((hi - lo) * j) // k, lo + ((hi - lo) * (j + 1)) // k) for j in range(k)]
        row_sets = cut
        for lo, hi in row_sets:
            if not (0 <= lo <= hi <= A.n_rows):
                raise ValueError('BlockedPlan: bad row set')
            rp = rp_all[lo:hi + 1]
            deg = rp[1:] - rp[:-1]
            load = int(rp[-1] - rp[0]) / max(1, (hi - lo + self.rpw - 1) // self.rpw)          # edges per wave
            long_streams = load > 4096
            hub_t = int(self._hub_arg) if self._hub_arg is not None else min(4096, max(256, 1 << max(0, int(load).bit_length() - 1)))
            piece_t = int(self._piece_arg) if self._piece_arg is not None else hub_t
            wpg_t = int(self._wpg_arg) if self._wpg_arg is not None else (1 if long_streams and self.split_hubs else 4)
            wm_t = int(self._wm_arg) if self._wm_arg is not None else (256 if long_streams and self.split_hubs else 1)
            long_ = deg > hub_t
            if self.split_hubs:
                planned = torch.ones_like(long_)
            else:
                planned = ~long_
            n_pl = int(planned.sum())
            # pieces per planned row: 1, or ceil(deg / piece) for a split (hub) row
            npc = torch.where(long_ & planned, (deg + piece_t - 1) // piece_t, torch.ones_like(deg)) * planned.long()
            n_virtual = int(npc.sum())
            if (n_virtual + self.rpw - 1) // self.rpw < min_waves:
                planned = torch.zeros_like(planned); npc = torch.zeros_like(npc); n_virtual = 0
            hub_rows.append(lo + torch.nonzero(~planned).flatten())
            if n_virtual == 0:
                continue
            vfirst = torch.cumsum(npc, 0) - npc                               # first virtual row of every local row
            vrow = torch.repeat_interleave(torch.arange(hi - lo, device=dev), npc, output_size=n_virtual)     # local row of a virtual row
            vk = torch.arange(n_virtual, device=dev) - vfirst[vrow]           # piece number inside its row
            vdeg = (deg[vrow] - vk + npc[vrow] - 1) // npc[vrow]              # piece p of P takes edges p, p + P, ...
            w_desc, o = torch.sort(vdeg, descending=True, stable=True)
            n_waves = (n_virtual + self.rpw - 1) // self.rpw
            if wm_t > 1 and int(deg.sum()) > 4096 * n_waves:
                n_waves = (n_waves + wm_t - 1) // wm_t * wm_t
            w_host = w_desc.to(torch.int32).cpu().contiguous()
            bin_h = torch.empty(n_virtual, dtype=torch.int32); slot_h = torch.empty(n_virtual, dtype=torch.int32)
            check(_lib.lib().arl_lpt_deal(n_virtual, w_host.data_ptr(), n_waves, self.rpw, bin_h.data_ptr(), slot_h.data_ptr()), 'arl_lpt_deal')
            v_wave = torch.empty(n_virtual, dtype=torch.int64, device=dev); v_slot = torch.empty(n_virtual, dtype=torch.int64, device=dev)
            v_wave[o] = bin_h.to(dev).long(); v_slot[o] = slot_h.to(dev).long()
            # split rows: their pieces are numbered in (row, piece) order; a piece's slot entry is -(2 + piece index)
            is_split = (npc > 1)
            split_local = torch.nonzero(is_split).flatten()
            n_split = int(split_local.numel())
            split_cnt = npc[split_local]
            split_first = torch.cumsum(split_cnt, 0) - split_cnt
            piece_base = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
            piece_base[split_local] = split_first
            v_entry = torch.where(is_split[vrow], -(2 + piece_base[vrow] + vk), lo + vrow)
            wave_rows = torch.full((n_waves, self.rpw), -1, dtype=torch.int32, device=dev)
            wave_rows[v_wave, v_slot] = v_entry.to(torch.int32)
            e0, e1 = int(rp[0]), int(rp[-1])
            erow = torch.repeat_interleave(torch.arange(hi - lo, device=dev), deg, output_size=e1 - e0)
            keep = torch.nonzero(planned[erow]).flatten() if n_pl < hi - lo else None
            if keep is not None:
                erow = erow[keep]
                eid = keep + e0
            else:
                eid = torch.arange(e0, e1, device=dev)
            del keep
            ev = vfirst[erow] + (eid - rp[erow]) % npc[erow]                  # virtual row of every edge
            del erow
            ew, es = v_wave[ev], v_slot[ev]; del ev
            ec = A.col[eid].long()
            key = (ew * n_cb + (ec if col_pos is None else col_pos[ec]) // self.col_block) * self.rpw + es
            o = torch.sort(key, stable=True)[1]; del key
            ew, es, ec, eid = ew[o], es[o], ec[o], eid[o]; del o
            cnt = torch.bincount(ew, minlength=n_waves)
            wave_ptr = torch.zeros(n_waves + 1, dtype=torch.int64, device=dev)
            torch.cumsum((cnt + 63) // 64 * 64, 0, out=wave_ptr[1:])
            total = int(wave_ptr[-1])
            if total >= 2 ** 31:
                raise ValueError('BlockedPlan: too many records for int32 offsets')
            start = torch.zeros(n_waves + 1, dtype=torch.int64, device=dev); torch.cumsum(cnt, 0, out=start[1:])
            dst = wave_ptr[ew] + (torch.arange(ew.numel(), device=dev) - start[ew])
            rec_col = torch.zeros(max(total, 64), dtype=torch.int32, device=dev)        # never empty: the C ABI wants real pointers
            rec_src = torch.zeros(max(total, 64), dtype=torch.int32, device=dev)                # edge id of every record (padding: 0, zeroed in _bind)
            rec_col[dst] = (ec | (es << 24)).to(torch.int32)
            rec_src[dst] = eid.to(torch.int32)
            pad = torch.ones(max(total, 64), dtype=torch.bool, device=dev); pad[dst] = False
            pad_pos = torch.nonzero(pad).flatten()
            # a padding record repeats its wave's last real record with value 0: whatever 0 * X[col] gives (NaN for a non-finite
            # operand row) lands on a row that already takes that operand row, never on an unrelated one
            if pad_pos.numel() and total:
                pw = torch.searchsorted(wave_ptr[1:].contiguous(), pad_pos, right=True).clamp_(max=n_waves - 1)
                last = wave_ptr[pw] + cnt[pw] - 1
                ok = cnt[pw] > 0
                rec_col[pad_pos[ok]] = rec_col[last[ok]]
            self.sets.append({'n_waves': n_waves, 'wave_ptr': wave_ptr.to(torch.int32), 'wave_rows': wave_rows, 'rec_col': rec_col, 'rec_src': rec_src, 'pad_pos': pad_pos,
                              'n_rows': n_pl, 'n_edges': int(ew.numel()), 'hub': hub_t, 'piece': piece_t, 'wpg': wpg_t, 'n_split': n_split,
                              'lo': lo, 'hi': hi, 'row_wave': v_wave[vfirst.clamp(max=n_virtual - 1)] * planned.long() - (~planned).long(), 'row_slot': v_slot[vfirst.clamp(max=n_virtual - 1)], 'cnt': cnt, 'n_pieces': int(split_cnt.sum()) if n_split else 0,
                              'split_row': (lo + split_local).to(torch.int32), 'split_first': split_first.to(torch.int32), 'split_count': split_cnt.to(torch.int32),
                              # short streams: more loads per wave; long streams (many edges per wave) run better with 16 (measured, cfg2)
                              'unroll': (16 if self.rpw == 16 or ew.numel() > 4096 * n_waves else 32) if unroll is None else int(unroll)})
        hub_rows = torch.cat(hub_rows) if hub_rows else torch.zeros(0, dtype=torch.int64, device=dev)
        self.n_hub = int(hub_rows.numel())
        self._hub_rows = hub_rows
        self._bind(A)

    def _bind(self, A, hub=None):
        self.structs = []
        for st in self.sets:
            if A.nnz:
                st['rec_val'] = A.val[st['rec_src']]
                st['rec_val'][st['pad_pos']] = 0.0
            else:
                st['rec_val'] = torch.zeros(st['rec_src'].numel(), dtype=torch.float32, device=A.device)
            self.structs.append(None)
        if hub is not None:                       # same chunk plan, new values (no host work)
            self.hub = hub.with_values(A.val)
        else:
            self.hub = A.chunks_only(self._hub_rows) if self.n_hub else None

    def struct(self, k, d):
        """ctypes image of row set k for an operand of width d (the split rows' [n_pieces, d] workspace is sized here)."""
        st = self.sets[k]
        n_sp = st.get('n_split', 0)
        if n_sp and (st.get('partial') is None or st['partial'].numel() < st['n_pieces'] * d):
            st['partial'] = torch.empty(st['n_pieces'] * d, dtype=torch.float32, device=st['rec_col'].device)
        return _lib.arl_blocked(st['n_waves'], self.rpw, st['unroll'], st['wave_ptr'].data_ptr(), st['wave_rows'].data_ptr(), st['rec_col'].data_ptr(),
                                st['rec_val'].data_ptr(), n_sp, st['split_row'].data_ptr() if n_sp else None, st['split_first'].data_ptr() if n_sp else None,
                                st['split_count'].data_ptr() if n_sp else None, st['partial'].data_ptr() if n_sp else None, int(st['wpg']))

    def with_values(self, A):
        """The same plan over a graph with the same pattern and new edge values."""
        p = object.__new__(BlockedPlan)
        p.__dict__.update(self.__dict__)
        p.sets = [dict(st) for st in self.sets]
        p._bind(A, self.hub)
        return p


BLOCKED_MIN_NNZ = 6_000_000      # measured cross-over (tools/blocked_bench.py): 8M edges 0.284 -> 0.238 ms, 3.4M edges 0.141 -> 0.167 ms
BLOCKED_MIN_WAVES = 1024         # a row set with fewer waves stays with the CSR kernel


def auto_blocked(graph, d, split=None, force=False, rows_per_wave=32, small_launch=None):
    """Attach the register-blocked plan to `graph` when it pays (d = 64 or 128, >= BLOCKED_MIN_NNZ edges) or when forced; no-op if the
    graph already has one or cannot take one (other widths, 2^24 columns or more).  Returns the graph.
    small_launch ('user' | 'item' | None): the graph is ONE rectangular block of the user-sharded step (a rank's user rows gathering item rows, or
    the item rows gathering its users).  The plan's defaults were measured on 32 M-edge launches; on the ~4 M-edge blocks of N = 8
    (tools/shard_hop_sweep.py, profiles/r04_c_shard_sweep.txt): item rows run better as twice as many waves of 16 rows, one wave per workgroup,
    while a wave carries under 2 048 edges (108.2 -> 99.0 us); user rows with 16 loads in flight instead of 32 while the launch has under
    8 192 waves (95.9 -> 91.6 us).  Larger blocks (N = 2, 4) keep the defaults."""
    if graph is None or graph.blocked is not None or int(d) not in (64, 128) or graph.n_cols >= (1 << 24):
        return graph
    if force or graph.nnz >= BLOCKED_MIN_NNZ:
        kw = {}
        if small_launch is not None and not split and rows_per_wave == 32:
            waves32 = max(1, (graph.n_rows + 31) // 32)
            if small_launch == 'item' and graph.nnz < 2048 * waves32:
                rows_per_wave, kw = 16, {'wpg': 1}
            elif small_launch == 'user' and waves32 < 8192:
                kw = {'unroll': 16}
        graph.enable_blocked(split=split, rows_per_wave=rows_per_wave, min_waves=0 if force else BLOCKED_MIN_WAVES, **kw)
    return graph


def _spmm_dispatch(A, d, blocked_call, csr_call, rows_from=0):
    """Run one full-table SpMM: through the blocked plan (+ its hub rows through the chunked CSR kernel) when the graph has one
    and d = 64 or 128, else through the CSR kernel.  The callables take the ctypes struct pointer.  rows_from > 0: the caller does not
    read output rows below it -- launches of the plan that only produce such rows are skipped (the CSR schedule ignores the hint)."""
    bp = A.blocked
    if bp is None or d not in (64, 128):
        csr_call(C.byref(A._struct(d)))
        return
    for k in range(len(bp.sets)):
        if bp.sets[k]['hi'] <= rows_from:
            continue
        blocked_call(C.byref(bp.struct(k, d)))
    if bp.hub is not None:
        csr_call(C.byref(bp.hub._struct(d)))


def norm_vals_coo(erow, col, w, dinv):
    """val[e] = (dinv[erow[e]] * w[e]) * dinv[col[e]] for a caller that already has dinv (edge-parallel)."""
    _dev(erow, torch.int32, 'erow', 1); _dev(col, torch.int32, 'col', 1); _dev(w, torch.float32, 'w', 1); _dev(dinv, torch.float32, 'dinv', 1)
    if erow.numel() != w.numel() or col.numel() != w.numel():
        raise ValueError('norm_vals_coo: length mismatch')
    val = torch.empty_like(w)
    check(_lib.lib().arl_norm_vals_coo_f32(_ptr(erow), _ptr(col), _ptr(w), w.numel(), _ptr(dinv), _ptr(val), _stream()), 'arl_norm_vals_coo_f32')
    return val


def norm_adj_values(rowptr, col, w, n_rows, erow=None):
    """val[e] = (dinv[row]*w[e])*dinv[col[e]] on device; returns (val, dinv).  `erow` (int32 row id per edge, optional) selects the
    edge-parallel form for callers that re-normalise the same pattern many times (PGA)."""
    _dev(rowptr, torch.int32, 'rowptr', 1); _dev(col, torch.int32, 'col', 1); _dev(w, torch.float32, 'w', 1)
    if rowptr.numel() != n_rows + 1 or col.numel() != w.numel():
        raise ValueError('norm_adj_values: shape mismatch')
    dinv = torch.empty(n_rows, dtype=torch.float32, device=w.device)
    val = torch.empty_like(w)
    if erow is not None:
        _dev(erow, torch.int32, 'erow', 1)
        if erow.numel() != w.numel():
            raise ValueError('norm_adj_values: erow length')
        check(_lib.lib().arl_norm_adj_values_coo_f32(n_rows, _ptr(rowptr), _ptr(erow), _ptr(col), _ptr(w), w.numel(), _ptr(dinv), _ptr(val), _stream()),
              'arl_norm_adj_values_coo_f32')
        return val, dinv
    check(_lib.lib().arl_norm_adj_values_f32(n_rows, _ptr(rowptr), _ptr(col), _ptr(w), _ptr(dinv), _ptr(val), _stream()), 'arl_norm_adj_values_f32')
    return val, dinv


def bipartite_graph(u, i, n_users, n_items, device=None, weights=None):
    """Normalised symmetric adjacency D^-1/2 [[0, R], [R^T, 0]] D^-1/2 as a CSRGraph, built on the device from the (user, item)
    pairs of R sorted by (user, item) -- what DataLoader.convert_to_laplacian_mat does with scipy on the host
    (util/DataLoader.py:57-87, isolated nodes get weight 0).  u, i: int64/int32 device tensors."""
    dev = u.device if device is None else torch.device(device)
    u = u.to(dev, torch.int64); i = i.to(dev, torch.int64)
    nnz = u.numel()
    U, I = int(n_users), int(n_items)
    w = torch.ones(nnz, dtype=torch.float32, device=dev) if weights is None else weights.to(dev, torch.float32)
    rp_u = torch.searchsorted(u, torch.arange(U + 1, device=dev, dtype=torch.int64))
    it_sorted, order = torch.sort(i, stable=True)                            # users stay ascending inside an item row
    rp_i = torch.searchsorted(it_sorted, torch.arange(I + 1, device=dev, dtype=torch.int64))
    rowptr = torch.cat([rp_u, nnz + rp_i[1:]])
    col = torch.cat([i + U, u[order]]).to(torch.int32)
    ww = torch.cat([w, w[order]])
    val, dinv = norm_adj_values(rowptr.to(torch.int32), col, ww, U + I)
    g = CSRGraph(rowptr, col, val, dev, validate=False)
    g.dinv = dinv
    return g


class IncrementalBipartite:
    """Normalised symmetric adjacency of the stacked interactions [R; B]: R = the REAL users' U x I block, fixed for the life of the object,
    B = the F fake users' rows, replaced at every attack epoch (attack/White/CLeaR.py:66-71,130-145, DLAttack.py:57-61,110-121 rebuild the whole
    (U+F+I)^2 scipy matrix and re-upload it per epoch).  update() assembles the new CSR on the device by MERGING: fake users are the last user
    rows, and in every item row they sort behind all real users, so the new pattern is the old one with entries appended at row ends --
    a prefix sum and two scatters --; degrees change on the touched rows only and the values are re-normalised by the same kernel as a fresh
    build (bit-identical to ops.bipartite_graph on the stacked matrix).  The register-blocked hop plan of R is kept and PATCHED: the new
    records are appended to their rows' waves (streams are re-laid-out by one gather), values re-gathered; no sort of the 6.4e7 records, no host
    dealing.  Everything runs on the device; the host sees a handful of scalars."""

    def __init__(self, u, i, n_real, n_fake, n_items, device=None, weights=None, emb_size=None):
        dev = u.device if device is None else torch.device(device)
        self.device = dev
        self.U, self.F, self.I = int(n_real), int(n_fake), int(n_items)
        U, F, I = self.U, self.F, self.I
        Up = U + F
        self.Up, self.N = Up, Up + I
        u = u.to(dev, torch.int64); i = i.to(dev, torch.int64)
        nnz = u.numel()
        self.nnz = nnz
        w = torch.ones(nnz, dtype=torch.float32, device=dev) if weights is None else weights.to(dev, torch.float32)
        if nnz and (int(u.max()) >= U or int(i.max()) >= I):
            raise ValueError('IncrementalBipartite: real interactions out of range')
        self.rp_u = torch.searchsorted(u, torch.arange(Up + 1, device=dev, dtype=torch.int64))           # fake rows empty
        it_sorted, order = torch.sort(i, stable=True)
        self.rp_i = torch.searchsorted(it_sorted, torch.arange(I + 1, device=dev, dtype=torch.int64))
        self.col_u, self.w_u = (i + Up).to(torch.int32), w
        self.col_i, self.w_i = u[order].to(torch.int32), w[order]
        self.item_of_edge = it_sorted                                                                     # item id of every item-half edge
        self.base = None
        if emb_size is not None and int(emb_size) in (64, 128) and 2 * nnz >= BLOCKED_MIN_NNZ and self.N < (1 << 24):
            rowptr0 = torch.cat([self.rp_u, nnz + self.rp_i[1:]])
            self.base = CSRGraph(rowptr0, torch.cat([self.col_u, self.col_i]), torch.zeros(2 * nnz, dtype=torch.float32, device=dev), dev, validate=False)
            self.base.enable_blocked(split=Up, min_waves=BLOCKED_MIN_WAVES)
            if not self.base.blocked.sets or self.base.blocked.n_hub:
                self.base = None                                                                          # (a set left to the CSR kernel: no plan to patch)

    def __deepcopy__(self, memo):
        return self                                  # immutable base: surrogates forked with copy.deepcopy share it

    def update(self, fu, fi, fw=None):
        """CSRGraph of [R; B] for the fake block given as COO sorted by (fake user, item): fu in [0, F), fi in [0, I), optional weights."""
        dev, U, F, I, Up, N, nnz = self.device, self.U, self.F, self.I, self.Up, self.N, self.nnz
        fu = fu.to(dev, torch.int64); fi = fi.to(dev, torch.int64)
        nf = fu.numel()
        fw = torch.ones(nf, dtype=torch.float32, device=dev) if fw is None else fw.to(dev, torch.float32)
        if nf:
            if int(fu.min()) < 0 or int(fu.max()) >= F or int(fi.min()) < 0 or int(fi.max()) >= I:
                raise ValueError('IncrementalBipartite.update: fake interactions out of range')
            k = fu * I + fi
            if bool((k[1:] <= k[:-1]).any()):
                raise ValueError('IncrementalBipartite.update: fake interactions must be sorted by (user, item) and unique')
        # user half: real user rows as they are, then the fake rows
        rp_user = torch.cat([self.rp_u[:U + 1], nnz + torch.searchsorted(fu, torch.arange(1, F + 1, device=dev, dtype=torch.int64))]) if F else self.rp_u
        col_user = torch.cat([self.col_u, (fi + Up).to(torch.int32)])
        w_user = torch.cat([self.w_u, fw])
        # item half: every item row keeps its real users and gains its fake users at the end
        ins = torch.bincount(fi, minlength=I)
        cumins = torch.cumsum(ins, 0) - ins
        rp_item = self.rp_i.clone(); rp_item[1:] += torch.cumsum(ins, 0)
        new_pos_old = torch.arange(nnz, device=dev, dtype=torch.int64) + cumins[self.item_of_edge]
        fi_s, o = torch.sort(fi, stable=True)                                   # within an item: fake users ascending
        rank = torch.arange(nf, device=dev, dtype=torch.int64) - (torch.cumsum(ins, 0) - ins)[fi_s]
        pos_fake = rp_item[fi_s + 1] - ins[fi_s] + rank
        col_item = torch.empty(nnz + nf, dtype=torch.int32, device=dev); w_item = torch.empty(nnz + nf, dtype=torch.float32, device=dev)
        col_item[new_pos_old] = self.col_i; w_item[new_pos_old] = self.w_i
        col_item[pos_fake] = (U + fu[o]).to(torch.int32); w_item[pos_fake] = fw[o]
        E_user = nnz + nf
        rowptr = torch.cat([rp_user, E_user + rp_item[1:]])
        col = torch.cat([col_user, col_item]); ww = torch.cat([w_user, w_item])
        val, dinv = norm_adj_values(rowptr.to(torch.int32), col, ww, N)
        g = CSRGraph(rowptr, col, val, dev, validate=False)
        g.dinv = dinv
        if self.base is not None:
            g.blocked = self._patched_plan(g, nf, fu, fi, o, fi_s, pos_fake, new_pos_old, E_user)
        return g

    def _patched_plan(self, g, nf, fu, fi, o, fi_s, pos_fake, new_pos_old, E_user):
        dev, U, Up, nnz = self.device, self.U, self.Up, self.nnz
        bp0 = self.base.blocked
        bp = object.__new__(BlockedPlan)
        bp.__dict__.update(bp0.__dict__)
        bp.sets = []
        for st in bp0.sets:
            user_set = st['lo'] == 0
            if user_set:       # new records: the fake users' rows; old records keep their edge ids (real user rows do not move)
                rows = U + fu; cols = fi + Up; eids = nnz + torch.arange(nf, device=dev, dtype=torch.int64)
                src_old = st['rec_src'].long()
            else:              # new records: the items' fake neighbours; an old item-half edge e moved to E_user + new_pos_old[e - nnz]
                rows = Up + fi_s; cols = U + fu[o]; eids = E_user + pos_fake
                src_old = E_user + new_pos_old[(st['rec_src'].long() - nnz).clamp_(min=0)]
            local = rows - st['lo']
            wv, sl = st['row_wave'][local], st['row_slot'][local]
            cnt0, wp0 = st['cnt'], st['wave_ptr'].long()
            n_waves = st['n_waves']
            # new records of a wave keep the (column block, slot) order among themselves; they trail the wave's old records
            okey = torch.sort((wv * (1 << 22) + cols // bp0.col_block) * bp0.rpw + sl, stable=True)[1] if nf else torch.zeros(0, dtype=torch.int64, device=dev)
            wv, sl, cols, eids = wv[okey], sl[okey], cols[okey], eids[okey]
            add = torch.bincount(wv, minlength=n_waves)
            cnt = cnt0 + add
            wpn = torch.zeros(n_waves + 1, dtype=torch.int64, device=dev); torch.cumsum((cnt + 63) // 64 * 64, 0, out=wpn[1:])
            total = int(wpn[-1])
            if total >= 2 ** 31:
                raise ValueError('IncrementalBipartite: too many records for int32 offsets')
            size = max(total, 64)
            pw = torch.repeat_interleave(torch.arange(n_waves, device=dev), wpn[1:] - wpn[:-1], output_size=total)       # wave of every new position
            k = torch.arange(total, device=dev, dtype=torch.int64) - wpn[pw]
            is_old, is_new = k < cnt0[pw], (k >= cnt0[pw]) & (k < cnt[pw])
            rec_col = torch.zeros(size, dtype=torch.int32, device=dev); rec_src = torch.zeros(size, dtype=torch.int32, device=dev)
            po = torch.nonzero(is_old).flatten()
            from_old = wp0[pw[po]] + k[po]
            rec_col[po] = st['rec_col'][from_old]; rec_src[po] = src_old[from_old].to(torch.int32)
            if nf:
                start_new = torch.cumsum(add, 0) - add
                pn = torch.nonzero(is_new).flatten()
                from_new = start_new[pw[pn]] + (k[pn] - cnt0[pw[pn]])
                rec_col[pn] = (cols[from_new] | (sl[from_new] << 24)).to(torch.int32); rec_src[pn] = eids[from_new].to(torch.int32)
            pad_pos = torch.nonzero(~(is_old | is_new)).flatten()
            if pad_pos.numel():
                last = wpn[pw[pad_pos]] + cnt[pw[pad_pos]] - 1
                ok = cnt[pw[pad_pos]] > 0
                rec_col[pad_pos[ok]] = rec_col[last[ok]]
            ns = dict(st)
            ns.update({'wave_ptr': wpn.to(torch.int32), 'rec_col': rec_col, 'rec_src': rec_src, 'pad_pos': pad_pos, 'cnt': cnt, 'n_edges': st['n_edges'] + nf})
            bp.sets.append(ns)
        bp._bind(g, None)
        return bp


def _check_xy(A, X, name='X', rows=None):
    _dev(X, torch.float32, name, 2)
    want = A.n_rows if rows is None else rows
    if X.shape[0] != want:
        raise ValueError('%s: %d rows, expected %d' % (name, X.shape[0], want))
    d = X.shape[1]
    if d % 4 or d > 256:
        raise ValueError('embedding size %d unsupported (multiple of 4, <= 256)' % d)
    if X.device != A.device:
        raise ValueError('%s on %s, graph on %s' % (name, X.device, A.device))
    return d


def spmm(A, X, alpha=1.0, beta=0.0, Z=None, out=None, row_scale=None, rows_from=0):
    """out = alpha * (A @ X) + beta * Z.   X: [A.n_cols, d]; out, Z: [A.n_rows, d].   row_scale [n_rows] (optional): the product's rows are
    multiplied by it in the epilogue, out = alpha * diag(row_scale) (A @ X) + beta * Z.   rows_from (optional hint): output rows below it
    are not needed and MAY be left unwritten (a blocked plan skips the launches that only produce them)."""
    d = _check_xy(A, X, 'X', A.n_cols)
    Y = torch.empty(A.n_rows, d, dtype=torch.float32, device=X.device) if out is None else out
    if _check_xy(A, Y, 'out') != d or Y.data_ptr() == X.data_ptr():
        raise ValueError('spmm: out must be [n_rows, d] and must not alias X')
    if beta != 0.0:
        if Z is None or _check_xy(A, Z, 'Z') != d:
            raise ValueError('spmm: Z [n_rows, d] required when beta != 0')
    L, zp, st = _lib.lib(), (_ptr(Z) if beta != 0.0 else None), _stream()
    tok = EVENT_HOOK.begin('axpby') if EVENT_HOOK is not None else None
    if row_scale is not None:
        _dev(row_scale, torch.float32, 'row_scale', 1)
        if row_scale.numel() != A.n_rows:
            raise ValueError('spmm: row_scale needs one entry per output row')
        _spmm_dispatch(A, d, lambda p: check(L.arl_spmm_blocked_rscale_f32(p, _ptr(X), d, _ptr(row_scale), alpha, beta, zp, _ptr(Y), st), 'arl_spmm_blocked_rscale_f32'),
                       lambda p: check(L.arl_spmm_csr_rscale_f32(p, _ptr(X), d, _ptr(row_scale), alpha, beta, zp, _ptr(Y), st), 'arl_spmm_csr_rscale_f32'),
                       rows_from=rows_from)
    else:
        _spmm_dispatch(A, d, lambda p: check(L.arl_spmm_blocked_f32(p, _ptr(X), d, alpha, beta, zp, None, _ptr(Y), st), 'arl_spmm_blocked_f32'),
                       lambda p: check(L.arl_spmm_csr_f32(p, _ptr(X), d, alpha, beta, zp, _ptr(Y), st), 'arl_spmm_csr_f32'), rows_from=rows_from)
    if tok is not None:
        EVENT_HOOK.end(tok)
    return Y


def spmm_layersum(A, X, S_in, S, Y=None):
    """Y = A @ X (optional store); S = S_in + A @ X."""
    d = _check_xy(A, X)
    for t, nm in ((S_in, 'S_in'), (S, 'S')):
        _check_xy(A, t, nm)
        if t.shape != X.shape:
            raise ValueError('spmm_layersum: %s shape mismatch' % nm)
    if Y is not None:
        _check_xy(A, Y, 'Y')
        if Y.shape != X.shape or Y.data_ptr() == X.data_ptr():
            raise ValueError('spmm_layersum: Y must not alias X')
    if S.data_ptr() == X.data_ptr():
        raise ValueError('spmm_layersum: S must not alias X')
    L, st = _lib.lib(), _stream()
    tok = EVENT_HOOK.begin('layersum') if EVENT_HOOK is not None else None
    _spmm_dispatch(A, d, lambda p: check(L.arl_spmm_blocked_layersum_f32(p, _ptr(X), d, _ptr(S_in), _ptr(S), _ptr(Y), st), 'arl_spmm_blocked_layersum_f32'),
                   lambda p: check(L.arl_spmm_csr_layersum_f32(p, _ptr(X), d, _ptr(S_in), _ptr(S), _ptr(Y), st), 'arl_spmm_csr_layersum_f32'))
    if tok is not None:
        EVENT_HOOK.end(tok)
    return S


def _check_flags(t, n, name):
    _dev(t, torch.uint8, name, 1)
    if t.numel() != n:
        raise ValueError('%s: %d flags, expected %d' % (name, t.numel(), n))
    return t


def spmm_adam(A, X, alpha, beta, Z, P, M, V, lr, step, betas=(0.9, 0.999), eps=1e-8, zflags=None):
    """g = alpha*(A@X) + beta*Z ; Adam update of (P, M, V) with g, fused in the SpMM epilogue.  X: [A.n_cols, d]; rest [A.n_rows, d].
    zflags (uint8 [n_rows], optional): Z is read only on flagged rows (it is zero elsewhere)."""
    d = _check_xy(A, X, 'X', A.n_cols)
    for t, nm in ((P, 'P'), (M, 'M'), (V, 'V')):
        if _check_xy(A, t, nm) != d or t.data_ptr() == X.data_ptr():
            raise ValueError('spmm_adam: %s shape/alias error' % nm)
    if beta != 0.0:
        if Z is None or _check_xy(A, Z, 'Z') != d:
            raise ValueError('spmm_adam: Z shape mismatch')
    if zflags is not None:
        _check_flags(zflags, A.n_rows, 'zflags')
    L, zp, st = _lib.lib(), (_ptr(Z) if beta != 0.0 else None), _stream()
    tok = EVENT_HOOK.begin('adam') if EVENT_HOOK is not None else None
    _spmm_dispatch(A, d, lambda p: check(L.arl_spmm_blocked_adam_f32(p, _ptr(X), d, alpha, beta, zp, _ptr(zflags), _ptr(P), _ptr(M), _ptr(V), lr, betas[0],
                                                                     betas[1], eps, int(step), st), 'arl_spmm_blocked_adam_f32'),
                   lambda p: check(L.arl_spmm_csr_adam_f32(p, _ptr(X), d, alpha, beta, zp, _ptr(zflags), _ptr(P), _ptr(M), _ptr(V), lr, betas[0], betas[1], eps,
                                                           int(step), st), 'arl_spmm_csr_adam_f32'))
    if tok is not None:
        EVENT_HOOK.end(tok)


def spmm_flagged(A, X, xflags=None, alpha=1.0, beta=0.0, Z=None, zflags=None, out=None):
    """out = alpha*(A@X) + beta*Z where X is zero except on rows whose bit is set in the bitmap `xflags` (int32 words; see
    mark_bits_) and Z is read only where the byte flag zflags != 0.  Either may be None (= dense)."""
    d = _check_xy(A, X, 'X', A.n_cols)
    Y = torch.empty(A.n_rows, d, dtype=torch.float32, device=X.device) if out is None else out
    if _check_xy(A, Y, 'out') != d or Y.data_ptr() == X.data_ptr():
        raise ValueError('spmm_flagged: out must be [n_rows, d] and must not alias X')
    if beta != 0.0 and (Z is None or _check_xy(A, Z, 'Z') != d):
        raise ValueError('spmm_flagged: Z [n_rows, d] required when beta != 0')
    if xflags is not None:          # bitmap: int32 words, bit c&31 of word c>>5
        _dev(xflags, torch.int32, 'xflags (bitmap)', 1)
        if xflags.numel() != (A.n_cols + 31) // 32:
            raise ValueError('spmm_flagged: bitmap needs ceil(n_cols/32) int32 words')
    if zflags is not None:
        _check_flags(zflags, A.n_rows, 'zflags')
    L, zp, st = _lib.lib(), (_ptr(Z) if beta != 0.0 else None), _stream()
    tok = EVENT_HOOK.begin('masked' if xflags is not None else 'axpby') if EVENT_HOOK is not None else None
    csr = lambda p: check(L.arl_spmm_csr_flagged_f32(p, _ptr(X), d, _ptr(xflags), alpha, beta, zp, _ptr(zflags), _ptr(Y), st), 'arl_spmm_csr_flagged_f32')
    if xflags is not None:              # the masked hop skips most edges: row-per-group kernel
        csr(C.byref(A._struct(d)))
    else:
        _spmm_dispatch(A, d, lambda p: check(L.arl_spmm_blocked_f32(p, _ptr(X), d, alpha, beta, zp, _ptr(zflags), _ptr(Y), st), 'arl_spmm_blocked_f32'), csr)
    if tok is not None:
        EVENT_HOOK.end(tok)
    return Y


ROWS_NSPLIT = 16     # default number of edge ranges per listed row of the row-subset hop


def spmm_rows(A, X, rows, layers=(), alpha=1.0, nsplit=ROWS_NSPLIT, out=None, workspace=None, check_range=True, row_weight=None):
    """out[t] = alpha * ( sum_k layers[k][rows[t]] + (A @ X)[rows[t]] ) for the listed rows only (duplicates allowed)."""
    d = _check_xy(A, X, 'X', A.n_cols)
    _dev(rows, torch.int32, 'rows', 1)
    n = rows.numel()
    if check_range and n and (int(rows.min()) < 0 or int(rows.max()) >= A.n_rows):
        raise IndexError('spmm_rows: row index out of range')
    if len(layers) > 8:
        raise ValueError('spmm_rows: at most 8 layer tables')
    for t in layers:
        if _check_xy(A, t, 'layer') != d:
            raise ValueError('spmm_rows: layer table shape mismatch')
    if out is None:
        out = torch.empty(n, d, dtype=torch.float32, device=X.device)
    _dev(out, torch.float32, 'out', 2)
    if out.shape != (n, d):
        raise ValueError('spmm_rows: out must be [len(rows), d]')
    need = _lib.lib().arl_spmm_csr_rows_workspace_bytes(n, nsplit, d) // 4
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.float32, device=X.device)
    arr = (C.c_void_p * max(len(layers), 1))(*[t.data_ptr() for t in layers])
    s = A._struct(d)
    tok = EVENT_HOOK.begin('rows') if EVENT_HOOK is not None else None
    if row_weight is not None and (_dev(row_weight, torch.float32, 'row_weight', 1).numel() != n):
        raise ValueError('spmm_rows: row_weight must have one entry per listed row')
    check(_lib.lib().arl_spmm_csr_rows_f32(C.byref(s), _ptr(X), d, _ptr(rows), n, nsplit, C.cast(arr, C.c_void_p), len(layers), alpha, _ptr(row_weight),
                                           _ptr(out), _ptr(workspace), _stream()), 'arl_spmm_csr_rows_f32')
    if tok is not None:
        EVENT_HOOK.end(tok)
    return out


def mark_rows_(flags, idx, value, check_range=True):
    _dev(flags, torch.uint8, 'flags', 1); _dev(idx, torch.int32, 'idx', 1)
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= flags.numel()):
        raise IndexError('mark_rows_: index out of range')
    check(_lib.lib().arl_mark_rows_u8(_ptr(flags), _ptr(idx), idx.numel(), int(value), _stream()), 'arl_mark_rows_u8')
    return flags


def batch_rows_set_(G, flags, bits, idx, src, scale=1.0, check_range=True, row_scale=None, dup_bits=None):
    """G[idx[t]] += scale * row_scale[t] * src[t] (duplicates accumulate in index order: ordered, no float atomics), flags[idx[t]] = 1,
    bit idx[t] of `bits` set.  row_scale: optional [n] per-contribution factors.  dup_bits: optional second all-zero bitmap like `bits`; a first
    launch records the rows named more than once in it and only those take the ordered scan (pass it to batch_rows_clear_ too)."""
    _dev(G, torch.float32, 'G', 2); _dev(flags, torch.uint8, 'flags', 1); _dev(bits, torch.int32, 'bits', 1); _dev(idx, torch.int32, 'idx', 1)
    _dev(src, torch.float32, 'src', 2)
    N, d = G.shape
    if flags.numel() != N or bits.numel() != (N + 31) // 32 or src.shape != (idx.numel(), d):
        raise ValueError('batch_rows_set_: shape mismatch')
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= N):
        raise IndexError('batch_rows_set_: index out of range')
    if row_scale is not None and _dev(row_scale, torch.float32, 'row_scale', 1).numel() != idx.numel():
        raise ValueError('batch_rows_set_: row_scale must have one entry per index')
    if dup_bits is not None and _dev(dup_bits, torch.int32, 'dup_bits', 1).numel() != bits.numel():
        raise ValueError('batch_rows_set_: dup_bits must have the size of bits')
    check(_lib.lib().arl_batch_rows_set_f32(_ptr(G), _ptr(flags), _ptr(bits), _ptr(idx), idx.numel(), d, _ptr(src), scale, _ptr(row_scale), _ptr(dup_bits),
                                            _stream()), 'arl_batch_rows_set_f32')


def batch_rows_clear_(G, flags, bits, idx, check_range=True, dup_bits=None):
    """Rows idx of G zeroed, their byte flags and bitmap bits (and `dup_bits` bits, when given) cleared: one launch."""
    _dev(G, torch.float32, 'G', 2); _dev(flags, torch.uint8, 'flags', 1); _dev(bits, torch.int32, 'bits', 1); _dev(idx, torch.int32, 'idx', 1)
    N, d = G.shape
    if flags.numel() != N or bits.numel() != (N + 31) // 32:
        raise ValueError('batch_rows_clear_: shape mismatch')
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= N):
        raise IndexError('batch_rows_clear_: index out of range')
    if dup_bits is not None and _dev(dup_bits, torch.int32, 'dup_bits', 1).numel() != bits.numel():
        raise ValueError('batch_rows_clear_: dup_bits must have the size of bits')
    check(_lib.lib().arl_batch_rows_clear_f32(_ptr(G), _ptr(flags), _ptr(bits), _ptr(idx), idx.numel(), d, _ptr(dup_bits), _stream()), 'arl_batch_rows_clear_f32')

def mark_bits_(bits, idx, set_, n_nodes, check_range=True):
    """Set / clear bits idx[t] of a node bitmap (int32 words)."""
    _dev(bits, torch.int32, 'bits', 1); _dev(idx, torch.int32, 'idx', 1)
    if bits.numel() != (n_nodes + 31) // 32:
        raise ValueError('mark_bits_: bitmap size')
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= n_nodes):
        raise IndexError('mark_bits_: index out of range')
    check(_lib.lib().arl_mark_rows_bits_u32(_ptr(bits), _ptr(idx), idx.numel(), 1 if set_ else 0, _stream()), 'arl_mark_rows_bits_u32')
    return bits


def zero_rows_(dst, idx, check_range=True):
    _dev(dst, torch.float32, 'dst', 2); _dev(idx, torch.int32, 'idx', 1)
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= dst.shape[0]):
        raise IndexError('zero_rows_: index out of range')
    check(_lib.lib().arl_zero_rows_f32(_ptr(dst), _ptr(idx), idx.numel(), dst.shape[1], _stream()), 'arl_zero_rows_f32')
    return dst


# ------------------------------------------------------------------------------------------------ losses
def _check_idx(t, name, hi, B=None):
    _dev(t, torch.int32, name, 1)
    if B is not None and t.numel() != B:
        raise ValueError('%s: length %d != %d' % (name, t.numel(), B))
    return t


def bpr_l2_fwd_bwd(emb, item_off, u, p, n, reg, G=None, upstream=1.0, workspace=None, loss_out=None, check_range=True, distinct_rows=False):
    """BPR + L2 on rows gathered from the combined table; returns loss_out = [bpr, reg_term, ||U_b||, ||P_b||] (device).
    If G is given, the gradient w.r.t. `emb` rows is added into it, duplicates in sample order (ordered, no float atomics).
    distinct_rows=True: the caller guarantees 3B pairwise distinct rows (the compact [3B, d] form with arange indices): no ownership scan."""
    _dev(emb, torch.float32, 'emb', 2)
    N, d = emb.shape
    B = u.numel()
    for t, nm in ((u, 'u'), (p, 'p'), (n, 'n')):
        _check_idx(t, nm, N, B)
    if B == 0:
        raise ValueError('bpr_l2: empty batch')
    if check_range:   # host-side bounds check (one small sync); hot loops validate once and pass check_range=False
        mx = torch.stack([u.max(), p.max(), n.max(), -u.min(), -p.min(), -n.min()]).tolist()
        if mx[0] >= N or max(mx[1], mx[2]) + item_off >= N or max(mx[3:]) > 0:
            raise IndexError('bpr_l2: batch index out of range')
    if G is not None:
        _dev(G, torch.float32, 'G', 2)
        if G.shape != emb.shape:
            raise ValueError('bpr_l2: G shape mismatch')
    need = _lib.lib().arl_bpr_l2_workspace_bytes(B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=emb.device)
    if loss_out is None:
        loss_out = torch.empty(4, dtype=torch.float32, device=emb.device)
    check(_lib.lib().arl_bpr_l2_fwd_bwd_f32(_ptr(emb), d, item_off, _ptr(u), _ptr(p), _ptr(n), B, reg, upstream, _ptr(loss_out), _ptr(G),
                                            _ptr(workspace), 1 if distinct_rows else 0, _stream()), 'arl_bpr_l2_fwd_bwd_f32')
    return loss_out


def adam_dense(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8):
    for t, nm in ((p, 'p'), (g, 'g'), (m, 'm'), (v, 'v')):
        _dev(t, torch.float32, nm)
        if t.numel() != p.numel():
            raise ValueError('adam_dense: size mismatch on %s' % nm)
    check(_lib.lib().arl_adam_dense_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, betas[0], betas[1], eps, int(step), _stream()), 'arl_adam_dense_f32')


def sgd_dense(p, g, lr):
    _dev(p, torch.float32, 'p'); _dev(g, torch.float32, 'g')
    if g.numel() != p.numel():
        raise ValueError('sgd_dense: size mismatch')
    check(_lib.lib().arl_sgd_dense_f32(_ptr(p), _ptr(g), p.numel(), lr, _stream()), 'arl_sgd_dense_f32')


def gather_rows(src, idx, check_range=True):
    _dev(src, torch.float32, 'src', 2); _dev(idx, torch.int32, 'idx', 1)
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= src.shape[0]):
        raise IndexError('gather_rows: index out of range')
    dst = torch.empty(idx.numel(), src.shape[1], dtype=torch.float32, device=src.device)
    check(_lib.lib().arl_gather_rows_f32(_ptr(src), _ptr(idx), idx.numel(), src.shape[1], _ptr(dst), _stream()), 'arl_gather_rows_f32')
    return dst


def scatter_add_rows(dst, idx, src, scale=1.0, check_range=True):
    _dev(dst, torch.float32, 'dst', 2); _dev(src, torch.float32, 'src', 2); _dev(idx, torch.int32, 'idx', 1)
    if src.shape != (idx.numel(), dst.shape[1]):
        raise ValueError('scatter_add_rows: shape mismatch')
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= dst.shape[0]):
        raise IndexError('scatter_add_rows: index out of range')
    check(_lib.lib().arl_scatter_add_rows_f32(_ptr(dst), _ptr(idx), idx.numel(), dst.shape[1], _ptr(src), scale, _stream()), 'arl_scatter_add_rows_f32')
    return dst


def rows_axpy_unique_(dst, src, idx, alpha=1.0, check_range=True, dup_bits=None):
    """dst[r] += alpha * src[r] once per DISTINCT row r listed in idx (src: a table that is zero outside the listed rows, e.g. the sparse
    batch gradient: its rows reach a dense table without a pass over the whole table)."""
    _dev(dst, torch.float32, 'dst', 2); _dev(src, torch.float32, 'src', 2); _dev(idx, torch.int32, 'idx', 1)
    if src.shape != dst.shape or src.data_ptr() == dst.data_ptr():
        raise ValueError('rows_axpy_unique_: dst and src must be distinct tables of one shape')
    if check_range and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= dst.shape[0]):
        raise IndexError('rows_axpy_unique_: index out of range')
    check(_lib.lib().arl_rows_axpy_unique_f32(_ptr(dst), _ptr(src), _ptr(idx), idx.numel(), dst.shape[1], float(alpha), _ptr(dup_bits), _stream()), 'arl_rows_axpy_unique_f32')
    return dst


def shard_batch_prep(u, p, n, u0, u1, out=None):
    """Bookkeeping of one global batch on a user shard [u0, u1), one launch: returns (lu [B] local row ids, clamped; own [3B] = [1.0 / 0.0 per
    sample | ones]: the factors of the batch's user / positive / negative contributions; item_rows [2B] = [p | n]; rows_l [3B] =
    [lu | Ul + p | Ul + n]).  `out`: the four tensors of a previous call with the same B, reused."""
    for t, nm in ((u, 'u'), (p, 'p'), (n, 'n')):
        _dev(t, torch.int32, nm, 1)
    B = u.numel()
    if p.numel() != B or n.numel() != B or not (0 <= u0 <= u1):
        raise ValueError('shard_batch_prep: u, p, n must have one length and 0 <= u0 <= u1')
    if out is None:
        out = (torch.empty(B, dtype=torch.int32, device=u.device), torch.empty(3 * B, dtype=torch.float32, device=u.device),
               torch.empty(2 * B, dtype=torch.int32, device=u.device), torch.empty(3 * B, dtype=torch.int32, device=u.device))
    lu, own, item_rows, rows_l = out
    check(_lib.lib().arl_shard_batch_prep_i32(_ptr(u), _ptr(p), _ptr(n), B, int(u0), int(u1), _ptr(lu), _ptr(own), _ptr(item_rows), _ptr(rows_l), _stream()),
          'arl_shard_batch_prep_i32')
    return out


def infonce_fwd_bwd(v1, v2, tau, want_grad=True, upstream=1.0):
    _dev(v1, torch.float32, 'v1', 2); _dev(v2, torch.float32, 'v2', 2)
    if v1.shape != v2.shape:
        raise ValueError('infonce: shape mismatch')
    n, d = v1.shape
    if n == 0 or n > 8192 or d % 4 or d > 256:
        raise ValueError('infonce: unsupported shape %s' % (tuple(v1.shape),))
    ws = torch.empty(_lib.lib().arl_infonce_workspace_bytes(n, d) // 4, dtype=torch.float32, device=v1.device)
    loss = torch.empty(1, dtype=torch.float32, device=v1.device)
    d1 = torch.empty_like(v1) if want_grad else None
    d2 = torch.empty_like(v2) if want_grad else None
    check(_lib.lib().arl_infonce_fwd_bwd_f32(_ptr(v1), _ptr(v2), n, d, tau, upstream, _ptr(loss), _ptr(d1), _ptr(d2), _ptr(ws), _stream()), 'arl_infonce_fwd_bwd_f32')
    return loss, d1, d2


NCE_ALLROWS_WIDTHS = (16, 32, 64, 128)
# The all-rows kernels compute exp((s - 1) / tau) with the constant 1 as the shift -- exact for row-NORMALISED operands (|s| <= 1), no running maximum.
# fp32 exp underflows below ~ -87, so a row whose best cosine is under 1 - 87 tau would sum to 0 (lse = -inf, NaN gradients); with tau >= 2 / 87 no
# cosine in [-1, 1] can underflow.  Smaller temperatures take the callers' panel form (running maximum).
NCE_ALLROWS_MIN_TAU = 0.023


def nce_allrows(A, V, tau, want_grad=True, want_dV=True, lse=None):
    """All-rows InfoNCE pieces for row-NORMALISED A [nA, d] (the batch) and V [nV, d] (all users or items; PRECONDITION: unit rows -- the kernel shifts by the
    constant 1 = the largest possible cosine, it keeps no running maximum), tau >= NCE_ALLROWS_MIN_TAU, d in {16, 32, 64, 128}, without an nA x nV logit
    matrix: returns lse [nA] = log sum_j exp(<a_b, v_j>/tau) and, with want_grad, (dA, dV) = (sum_j P_bj v_j, sum_b P_bj a_b) with
    P = exp(<a, v>/tau - lse); want_dV=False skips the table-side sum (dV = None).  The caller applies 1/tau, the positive pairs' terms and
    the upstream gradient (recommender/NCL.py:96-115, attack/White/InfoAttack.py:96-101).  lse: a log-sum-exp from an earlier want_grad=False
    call (forward and backward at different times); without it the gradient call produces it on the way."""
    _dev(A, torch.float32, 'A', 2); _dev(V, torch.float32, 'V', 2)
    nA, d = A.shape
    nV = V.shape[0]
    if V.shape[1] != d or d not in NCE_ALLROWS_WIDTHS or nA == 0 or nV == 0:
        raise ValueError('nce_allrows: A [nA, d], V [nV, d] with d in %s' % (NCE_ALLROWS_WIDTHS,))
    if not float(tau) >= NCE_ALLROWS_MIN_TAU:
        raise ValueError('nce_allrows: tau %g < %g -- exp((s - 1) / tau) of a row-normalised pair can underflow to 0 for every term of a row '
                         '(see NCE_ALLROWS_MIN_TAU); use the panel form' % (float(tau), NCE_ALLROWS_MIN_TAU))
    L = _lib.lib()
    ws = torch.empty(max(L.arl_nce_allrows_workspace_bytes(nA, nV, d) // 4, 4), dtype=torch.float32, device=A.device)
    given = lse is not None
    if given:
        _dev(lse, torch.float32, 'lse', 1)
        if lse.numel() != nA or not want_grad:
            raise ValueError('nce_allrows: lse [nA] goes with want_grad=True')
    else:
        lse = torch.empty(nA, dtype=torch.float32, device=A.device)
    if not want_grad:
        check(L.arl_nce_allrows_lse_f32(_ptr(A), nA, _ptr(V), nV, d, float(tau), _ptr(lse), _ptr(ws), _stream()), 'arl_nce_allrows_lse_f32')
        return lse
    dA, dV = torch.empty_like(A), (torch.empty_like(V) if want_dV else None)
    # lse_given = 0: the log-sum-exp comes out of the dA pass (two passes over V instead of three)
    check(L.arl_nce_allrows_grad_f32(_ptr(A), nA, _ptr(V), nV, d, float(tau), _ptr(lse), 1 if given else 0, _ptr(dA), _ptr(dV), _ptr(ws), _stream()), 'arl_nce_allrows_grad_f32')
    return lse, dA, dV


def normalize_rows(X):
    """(Y, nrm) = (F.normalize(X, dim=1), max(||X_r||, 1e-12)) in one pass (recommender/NCL.py:98-99)."""
    _dev(X, torch.float32, 'X', 2)
    Y, nrm = torch.empty_like(X), torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
    check(_lib.lib().arl_normalize_rows_f32(_ptr(X), X.shape[0], X.shape[1], _ptr(Y), _ptr(nrm), _stream()), 'arl_normalize_rows_f32')
    return Y, nrm


def normalize_rows_bwd(Y, nrm, dY, scale=1.0, out=None, scale_dev=None):
    """Autograd of normalize_rows: scale * (dY - Y <Y, dY>) / nrm; out may be dY (in place); scale_dev: a one-element GPU tensor multiplied into scale."""
    _dev(Y, torch.float32, 'Y', 2); _dev(dY, torch.float32, 'dY', 2); _dev(nrm, torch.float32, 'nrm', 1)
    if out is None:
        out = torch.empty_like(dY)
    _dev(out, torch.float32, 'out', 2)
    if dY.shape != Y.shape or out.shape != Y.shape or nrm.numel() != Y.shape[0]:
        raise ValueError('normalize_rows_bwd: shape mismatch')
    if scale_dev is not None:
        _dev(scale_dev, torch.float32, 'scale_dev')
    check(_lib.lib().arl_normalize_rows_bwd_f32(_ptr(Y), _ptr(nrm), _ptr(dY), Y.shape[0], Y.shape[1], float(scale), _ptr(scale_dev), _ptr(out), _stream()), 'arl_normalize_rows_bwd_f32')
    return out


def simgcl_perturb_rng(src, eps, seed, stream_id, out=None, row_ids=None):
    """out = src + sign(src) * normalize(u) * eps with u ~ uniform[0,1) drawn inside the kernel from (seed, stream_id, row, column): the
    SimGCL perturbation (SimGCL.py:203-205) without a noise table or a clone.  out may be src (in place).  row_ids (int32, optional): src holds
    those rows of a larger table; they get the noise of a full-table call with the same (seed, stream_id)."""
    _dev(src, torch.float32, 'src', 2)
    if out is None:
        out = torch.empty_like(src)
    _dev(out, torch.float32, 'out', 2)
    if out.shape != src.shape:
        raise ValueError('simgcl_perturb_rng: out shape mismatch')
    if row_ids is not None:
        _dev(row_ids, torch.int32, 'row_ids', 1)
        if row_ids.numel() != src.shape[0]:
            raise ValueError('simgcl_perturb_rng: one row id per row')
    check(_lib.lib().arl_simgcl_perturb_rng_f32(_ptr(src), _ptr(out), src.shape[0], src.shape[1], _ptr(row_ids), float(eps), int(seed) & (2 ** 64 - 1),
                                                int(stream_id) & (2 ** 64 - 1), _stream()), 'arl_simgcl_perturb_rng_f32')
    return out


def simgcl_perturb_(E, noise, eps):
    _dev(E, torch.float32, 'E', 2); _dev(noise, torch.float32, 'noise', 2)
    if E.shape != noise.shape:
        raise ValueError('simgcl_perturb_: shape mismatch')
    check(_lib.lib().arl_simgcl_perturb_f32(_ptr(E), _ptr(noise), E.shape[0], E.shape[1], eps, _stream()), 'arl_simgcl_perturb_f32')
    return E


def sfa_l1(X, w, r0, numel_h, want_grad=True, out=None, scale=1.0, accumulate=False):
    """CLeaR's SFA L1 term over H = rows of X repeated w[row] times (attack/White/CLeaR.py:98-125).
    Returns (loss[1], G) with G = (out +)= scale * dloss/dX, or (loss, None)."""
    _dev(X, torch.float32, 'X', 2); _dev(w, torch.float32, 'w', 1); _dev(r0, torch.float32, 'r0', 1)
    n, d = X.shape
    if w.shape[0] != n or r0.shape[0] != d:
        raise ValueError('sfa_l1: w must have one weight per row of X and r0 one entry per column')
    if n == 0 or d > 256 or int(numel_h) <= 0:
        raise ValueError('sfa_l1: unsupported shape %s / numel %d' % (tuple(X.shape), int(numel_h)))
    G = None
    if want_grad:
        G = out if out is not None else torch.empty_like(X)
        _dev(G, torch.float32, 'out', 2)
        if G.shape != X.shape:
            raise ValueError('sfa_l1: out must have the shape of X')
    elif out is not None:
        raise ValueError('sfa_l1: out given but want_grad is False')
    ws = torch.empty(_lib.lib().arl_sfa_workspace_bytes(n, d) // 4, dtype=torch.float32, device=X.device)
    loss = torch.empty(1, dtype=torch.float32, device=X.device)
    check(_lib.lib().arl_sfa_l1_fwd_bwd_f32(_ptr(X), _ptr(w), _ptr(r0), n, d, int(numel_h), float(scale), int(bool(accumulate)), _ptr(loss), _ptr(G),
                                            _ptr(ws), _stream()), 'arl_sfa_l1_fwd_bwd_f32')
    return loss, G


class SfaStages:
    """arl_sfa_stage{1,2,3}_f32: the SFA L1 term of sfa_l1() cut at its two global reductions, for a table whose rows are partitioned over
    ranks.  r = stage1(); all-reduce(r); a_s = stage2(r); all-reduce(a_s); loss, G = stage3(r, a_s)."""

    def __init__(self, X, w, r0):
        _dev(X, torch.float32, 'X', 2); _dev(w, torch.float32, 'w', 1); _dev(r0, torch.float32, 'r0', 1)
        n, d = X.shape
        if w.shape[0] != n or r0.shape[0] != d or n == 0 or d > 256:
            raise ValueError('SfaStages: w must have one weight per row of X, r0 one entry per column, d <= 256')
        self.X, self.w, self.r0, self.n, self.d = X, w, r0, n, d
        self.ws = torch.empty(_lib.lib().arl_sfa_workspace_bytes(n, d) // 4, dtype=torch.float32, device=X.device)

    def stage1(self):
        r = torch.empty(self.d, dtype=torch.float32, device=self.X.device)
        check(_lib.lib().arl_sfa_stage1_f32(_ptr(self.X), _ptr(self.w), _ptr(self.r0), self.n, self.d, _ptr(r), _ptr(self.ws), _stream()), 'arl_sfa_stage1_f32')
        return r

    def stage2(self, r):
        _dev(r, torch.float32, 'r', 1)
        a_s = torch.empty(self.d + 1, dtype=torch.float32, device=self.X.device)
        check(_lib.lib().arl_sfa_stage2_f32(_ptr(self.X), _ptr(self.w), _ptr(r), self.n, self.d, _ptr(a_s), _ptr(self.ws), _stream()), 'arl_sfa_stage2_f32')
        return a_s

    def stage3(self, r, a_s, numel_h, out=None, scale=1.0, accumulate=False):
        _dev(r, torch.float32, 'r', 1); _dev(a_s, torch.float32, 'a_s', 1)
        if r.numel() != self.d or a_s.numel() != self.d + 1 or int(numel_h) <= 0:
            raise ValueError('SfaStages.stage3: r[d], a_s[d + 1], numel_h > 0')
        G = out if out is not None else torch.empty_like(self.X)
        _dev(G, torch.float32, 'out', 2)
        if G.shape != self.X.shape:
            raise ValueError('SfaStages.stage3: out must have the shape of X')
        loss = torch.empty(1, dtype=torch.float32, device=self.X.device)
        check(_lib.lib().arl_sfa_stage3_f32(_ptr(self.X), _ptr(self.w), _ptr(self.r0), _ptr(r), _ptr(a_s), self.n, self.d, int(numel_h), float(scale),
                                            int(bool(accumulate)), _ptr(loss), _ptr(G), _ptr(self.ws), _stream()), 'arl_sfa_stage3_f32')
        return loss, G


# ------------------------------------------------------------------------------------------------ attack primitives
def sddmm_rows_dense(dY, X, rows, col_off, n_cols, out=None):
    _dev(dY, torch.float32, 'dY', 2); _dev(X, torch.float32, 'X', 2); _dev(rows, torch.int32, 'rows', 1)
    if dY.shape[1] != X.shape[1] or col_off < 0 or col_off + n_cols > X.shape[0]:
        raise ValueError('sddmm_rows_dense: shape mismatch')
    if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= dY.shape[0]):
        raise IndexError('sddmm_rows_dense: row out of range')
    if out is None:
        out = torch.zeros(rows.numel(), n_cols, dtype=torch.float32, device=X.device)
    else:
        _dev(out, torch.float32, 'out', 2)
        if out.shape != (rows.numel(), n_cols):
            raise ValueError('sddmm_rows_dense: out shape mismatch')
    check(_lib.lib().arl_sddmm_rows_dense_f32(_ptr(dY), _ptr(X), X.shape[1], _ptr(rows), rows.numel(), col_off, n_cols, _ptr(out), _stream()), 'arl_sddmm_rows_dense_f32')
    return out


def sddmm_csr(A, dY, X, alpha=1.0, out=None):
    """out[e] += alpha * <dY[row(e)], X[col[e]]> over the stored entries of the CSR graph A: the gradient of a loss with respect to the adjacency's values
    given dL/d(A X) = dY (recommender/LightGCN.py:41-43,58-59, requires_adjgrad).  out: [nnz] fp32 (zeros when omitted)."""
    _dev(dY, torch.float32, 'dY', 2); _dev(X, torch.float32, 'X', 2)
    if dY.shape[0] != A.n_rows or dY.shape[1] != X.shape[1] or X.shape[0] < A.n_cols:
        raise ValueError('sddmm_csr: dY [n_rows, d], X [n_cols, d]')
    if out is None:
        out = torch.zeros(A.col.numel(), dtype=torch.float32, device=X.device)
    _dev(out, torch.float32, 'out', 1)
    if out.numel() != A.col.numel():
        raise ValueError('sddmm_csr: one output per stored entry')
    check(_lib.lib().arl_sddmm_csr_f32(_ptr(A.rowptr), _ptr(A.col), A.n_rows, X.shape[1], _ptr(dY), _ptr(X), float(alpha), _ptr(out), _stream()), 'arl_sddmm_csr_f32')
    return out


def tables_sum(tables, alpha=1.0, out=None):
    """alpha * sum of up to 8 equally shaped fp32 tables in one pass (the LightGCN layer mean with alpha = 1/(L+1))."""
    if not 1 <= len(tables) <= 8:
        raise ValueError('tables_sum: 1..8 tables')
    for t in tables:
        _dev(t, torch.float32, 'table')
        if t.shape != tables[0].shape:
            raise ValueError('tables_sum: shape mismatch')
    if tables[0].numel() % 4:
        raise ValueError('tables_sum: element count must be a multiple of 4')
    if out is None:
        out = torch.empty_like(tables[0])
    _dev(out, torch.float32, 'out')
    if out.shape != tables[0].shape:
        raise ValueError('tables_sum: out shape mismatch')
    arr = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
    check(_lib.lib().arl_tables_sum_f32(C.cast(arr, C.c_void_p), len(tables), tables[0].numel(), float(alpha), _ptr(out), _stream()), 'arl_tables_sum_f32')
    return out


_FB_WS = {}


def fake_block_rows_(S, X, Y, rscale=None, alpha=1.0):
    """In place: Y[f] += alpha * rscale[f] * (S @ X)[f]   (S: [F, I] fake-user block, X: [I, d] item rows, Y: [F, d]) -- the fake users' rows of
    the poisoned adjacency product (attack/White/PGA.py:118-134); hand-written fp32 kernel, deterministic."""
    _dev(S, torch.float32, 'S', 2); _dev(X, torch.float32, 'X', 2); _dev(Y, torch.float32, 'Y', 2)
    F, I = S.shape
    d = X.shape[1]
    if X.shape[0] != I or Y.shape != (F, d):
        raise ValueError('fake_block_rows_: shapes S %s X %s Y %s' % (tuple(S.shape), tuple(X.shape), tuple(Y.shape)))
    if rscale is not None:
        _dev(rscale, torch.float32, 'rscale', 1)
        if rscale.numel() != F:
            raise ValueError('fake_block_rows_: rscale length')
    L = _lib.lib()
    need = L.arl_fake_block_rows_workspace_bytes(F, I, d)
    key = (S.device, need)
    ws = _FB_WS.get(key)
    if ws is None:
        _FB_WS.clear()
        ws = _FB_WS[key] = torch.empty(max(need, 4), dtype=torch.uint8, device=S.device)
    check(L.arl_fake_block_rows_f32(_ptr(S), F, I, _ptr(X), d, _ptr(rscale), float(alpha), _ptr(Y), _ptr(ws), _stream()), 'arl_fake_block_rows_f32')
    return Y


def fake_block_cols_(S, Xf, Y, rscale=None, alpha=1.0):
    """In place: Y[i] += alpha * rscale[i] * (S^T @ Xf)[i]   (Xf: [F, d] fake users' rows, Y: [I, d] item rows)."""
    _dev(S, torch.float32, 'S', 2); _dev(Xf, torch.float32, 'Xf', 2); _dev(Y, torch.float32, 'Y', 2)
    F, I = S.shape
    d = Xf.shape[1]
    if Xf.shape[0] != F or Y.shape != (I, d):
        raise ValueError('fake_block_cols_: shapes S %s Xf %s Y %s' % (tuple(S.shape), tuple(Xf.shape), tuple(Y.shape)))
    if rscale is not None:
        _dev(rscale, torch.float32, 'rscale', 1)
        if rscale.numel() != I:
            raise ValueError('fake_block_cols_: rscale length')
    check(_lib.lib().arl_fake_block_cols_f32(_ptr(S), F, I, _ptr(Xf), d, _ptr(rscale), float(alpha), _ptr(Y), _stream()), 'arl_fake_block_cols_f32')
    return Y


def pga_update_(S, grad, dinv_rows=None, dinv_cols=None):
    """In place: S = clamp(S - 0.2*tanh(dinv_rows[r]*grad*dinv_cols[c])), gradient ignored where S == 0 (not in the pattern)."""
    _dev(S, torch.float32, 'S', 2); _dev(grad, torch.float32, 'grad', 2)
    if S.shape != grad.shape:
        raise ValueError('pga_update_: shape mismatch')
    if dinv_rows is not None:
        _dev(dinv_rows, torch.float32, 'dinv_rows', 1)
        if dinv_rows.numel() != S.shape[0]:
            raise ValueError('pga_update_: dinv_rows length')
    if dinv_cols is not None:
        _dev(dinv_cols, torch.float32, 'dinv_cols', 1)
        if dinv_cols.numel() != S.shape[1]:
            raise ValueError('pga_update_: dinv_cols length')
    check(_lib.lib().arl_pga_update_f32(_ptr(S), _ptr(grad), _ptr(dinv_rows), _ptr(dinv_cols), S.shape[0], S.shape[1], _stream()), 'arl_pga_update_f32')
    return S


_CW_WS = {}


def cw_topk_term(X, n_user_rows, n_real, top_idx, targets, c=None, want_w=True, check_range=True):
    """CW term of the attacks' surrogate loss from the users' top-k lists (attack/White/CLeaR.py:83-95, PGA.py:104-116) on the packed table
    X [n_user_rows + I, d]: pairs (real user u < n_real) x (target t), negative = top_idx[u][k - 1 - t].  Returns (loss[1], G [like X], w) with
    loss = c * sum <X_u, X_neg - X_tg> (c defaults to 1 / (n_real T): the reference's mean), G = d loss / d X, and w (want_w) the SFA term's row
    multiplicities (CLeaR.py:98-103).  targets: int64 device tensor of item ids.  Deterministic (64-bit fixed-point item sums), five launches."""
    _dev(X, torch.float32, 'X', 2); _dev(top_idx, torch.int32, 'top_idx', 2); _dev(targets, torch.int64, 'targets', 1)
    N, d = X.shape
    Up, n_real = int(n_user_rows), int(n_real)
    I, k, T = N - Up, top_idx.shape[1], targets.numel()
    if not (0 < Up < N) or not (0 <= n_real <= Up) or top_idx.shape[0] < n_real or not (0 < T <= min(k, 64)) or d > 256:
        raise ValueError('cw_topk_term: X [n_user_rows + I, d], top_idx [>= n_real, k], 1 <= T <= min(k, 64), d <= 256')
    if check_range and n_real and (int(top_idx[:n_real, k - T:].min()) < 0 or int(top_idx[:n_real, k - T:].max()) >= I or int(targets.min()) < 0 or int(targets.max()) >= I):
        raise IndexError('cw_topk_term: item id out of range')
    L = _lib.lib()
    need = L.arl_cw_topk_term_workspace_bytes(I, d, n_real, T)
    key = (X.device, need)
    ws = _CW_WS.get(key)
    if ws is None:
        _CW_WS.clear()
        ws = _CW_WS[key] = torch.empty(max(need, 8), dtype=torch.uint8, device=X.device)
    G = torch.empty_like(X)
    loss = torch.empty(1, dtype=torch.float32, device=X.device)
    w = torch.empty(N, dtype=torch.float32, device=X.device) if want_w else None
    c = 1.0 / (max(n_real, 1) * T) if c is None else float(c)
    check(L.arl_cw_topk_term_f32(_ptr(X), Up, I, d, n_real, _ptr(top_idx), k, _ptr(targets), T, c, _ptr(G), _ptr(loss), _ptr(w), _ptr(ws), _stream()),
          'arl_cw_topk_term_f32')
    return loss, G, w


TOPK_FORM2 = os.environ.get('ARL_TOPK_FORM2', '1') != '0'      # (A/B: ARL_TOPK_FORM2=0 keeps the first form of score_mask_topk's fp16 stream)
TOPK_STATS = {'calls': 0, 'warm': 0, 'cold_repeats': 0}     # counters for benches: warm-started calls and how many of them had to be repeated cold


_EXIT_PROBE = {}          # (U, I, d, k, masked) -> adaptive use of the kernel's early-exit build (see score_mask_topk)
EXIT_REPROBE = 32         # calls between two probes of the exit build on tables where it skipped nothing


def _exit_mode(key, ws, off, nst, device):
    """Which build the next pass over tables of this shape should launch: 1 = let the device pick (exit build where the norm profile allows), 0 = plain.
    The exit build costs ~5 % when nothing is skipped, and whether anything is depends on the tables (thresholds vs norms), which only the pass itself
    finds out: after a pass in mode 1 its 24 bytes of counters are copied to pinned memory asynchronously; a later call that finds the copy complete and
    the exit build picked but nothing skipped switches to the plain build for EXIT_REPROBE calls.  Never waits for the device."""
    st = _EXIT_PROBE.setdefault(key, {'off': 0, 'pending': None})
    p = st['pending']
    if p is not None and p[1].query():
        consumed, wgs = (int(x) for x in p[0][:2].tolist())
        picked = int(p[0][2].item()) & 0xffffffff
        if picked == 1 and wgs > 0 and consumed >= wgs * p[2]:
            st['off'] = EXIT_REPROBE
        st['pending'] = None
    if st['off'] > 0:
        st['off'] -= 1
        return 0, st
    return 1, st


_ORDER_CACHE = {}         # (I, d, device) -> [item order of the last 'norm' call, calls served]
ORDER_REFRESH = 8         # calls that may reuse one norm order


def reset_exit_probe():
    """Forget what earlier passes learnt about the tables of a shape already seen (benches, tests; a caller that switches to unrelated tables):
    the early-exit probe state and the cached item order."""
    _EXIT_PROBE.clear()
    _ORDER_CACHE.clear()


def _exit_probe_record(st, ws, off, nst):
    if st['pending'] is None:
        host = st.get('host')
        if host is None:
            host = st['host'] = torch.empty(3, dtype=torch.int64).pin_memory()       # one pinned slot per shape, reused (no copy is in flight while `pending` is None)
        host.copy_(ws[off:off + 24].view(torch.int64), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st['pending'] = (host, ev, nst)


def score_mask_topk(Pu, Pi, k, mask_rowptr=None, mask_col=None, exact=False, warm_idx=None, item_order='norm'):
    """top-k of Pu @ Pi.T per user with an optional interacted-item mask (CSR over users), streamed.
    exact=True: scores are the exact fp32 contraction; default: split-fp16 matrix path (two fp16 pieces of the power-of-two-scaled operands, three products) for d in {64, 128} (scores within
    ~1e-6 relative, faster), exact otherwise.
    warm_idx: optional int32 [U, k] of DISTINCT candidate items per user (e.g. the previous call's result while the tables
    moved little): pre-sets the thresholds, same result, fewer inserts; if a candidate turned out masked the call is repeated
    cold automatically.
    item_order: 'norm' (default) streams the items by descending row norm on the fp16 matrix path when the stream is long enough for a
    bootstrap pass -- thresholds rise earlier, same result bit for bit (ids, values and tie order are those of the table order); None =
    table order; or an int32 permutation of [0, I)."""
    _dev(Pu, torch.float32, 'Pu', 2); _dev(Pi, torch.float32, 'Pi', 2)
    U, d = Pu.shape
    I = Pi.shape[0]
    if Pi.shape[1] != d or d % 4 or d > 256 or not (0 < k <= min(128, I)):
        raise ValueError('score_mask_topk: unsupported shape / k')
    if mask_rowptr is not None:
        _dev(mask_rowptr, torch.int32, 'mask_rowptr', 1); _dev(mask_col, torch.int32, 'mask_col', 1)
        if mask_rowptr.numel() != U + 1:
            raise ValueError('score_mask_topk: mask_rowptr must have U+1 entries')
        if int(mask_rowptr[-1]) != mask_col.numel():
            raise ValueError('score_mask_topk: mask_rowptr[-1] != len(mask_col)')
    matrix_path = k <= 64 and d in (16, 32, 64, 128)
    flag = None
    if warm_idx is not None and matrix_path:
        _dev(warm_idx, torch.int32, 'warm_idx', 2)
        if warm_idx.shape != (U, k):
            raise ValueError('score_mask_topk: warm_idx must be [U, k]')
        flag = torch.zeros(1, dtype=torch.int32, device=Pu.device)
    else:
        warm_idx = None
    idx = torch.empty(U, k, dtype=torch.int32, device=Pu.device)
    val = torch.empty(U, k, dtype=torch.float32, device=Pu.device)
    ws = None
    order = None
    if not exact and d in (64, 128) and k <= 64:
        ws = torch.empty(_lib.lib().arl_score_mask_topk_workspace_bytes(I, d), dtype=torch.uint8, device=Pu.device)
        if isinstance(item_order, torch.Tensor):
            order = _dev(item_order, torch.int32, 'item_order', 1)
            if order.numel() != I:
                raise ValueError('score_mask_topk: item_order must be a permutation of the I items')
        elif item_order == 'norm':
            if I >= 32768:                                 # shorter streams run without a bootstrap pass: nothing to gain
                # The order is a heuristic of the STREAM only (thresholds rise earlier, the early exit can bite): the result does not depend on it, the
                # exit's bound is computed from the norms of whatever order is streamed.  A loop that scores slowly moving tables again and again
                # (CLeaR's surrogate steps) therefore reuses the last order and re-sorts every ORDER_REFRESH-th call instead of a radix sort per step.
                okey = (I, d, str(Pu.device))
                ent = _ORDER_CACHE.get(okey)
                if ent is None or ent[1] >= ORDER_REFRESH:
                    order = torch.argsort(torch.linalg.vector_norm(Pi, dim=1), descending=True).to(torch.int32)
                    _ORDER_CACHE[okey] = [order, 1]
                else:
                    order = ent[0]; ent[1] += 1
        elif item_order is not None:
            raise ValueError("score_mask_topk: item_order must be 'norm', None or an int32 permutation")
    timed = os.environ.get('ARL_TOPK_TIME') == '1'          # diagnostics: wall time of the pass itself, between device synchronisations
    evs = None
    if TOPK_STATS.get('record_events') or os.environ.get('ARL_TOPK_TIME') == '2':       # diagnostics: device-side span (events on the launch stream, no host synchronisation)
        evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        evs[0].record()
    if timed:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    mode, probe = 0, None
    if ws is not None and order is not None:               # the early exit needs the norm-ordered stream
        soff = _lib.lib().arl_score_mask_topk_stats_offset(I, d)
        nst = (I + (64 if d <= 64 else 32) - 1) // (64 if d <= 64 else 32)
        mode, probe = _exit_mode((U, I, d, k, mask_rowptr is not None), ws, soff, nst, Pu.device)
    uws = None
    if ws is not None and TOPK_FORM2 and (d == 64 or (d == 128 and os.environ.get('ARL_TOPK_FORM2_D128') == '1')):      # (d = 128: developer builds with -DARL_TOPK2_D128=1 only)           # the second form of the stream (32 users per wave, lists kept in the outputs): needs the user workspace
        uws = torch.empty(_lib.lib().arl_score_mask_topk_user_workspace_bytes(U, d), dtype=torch.uint8, device=Pu.device)
    check(_lib.lib().arl_score_mask_topk_f32(_ptr(Pu), _ptr(Pi), U, I, d, _ptr(mask_rowptr), _ptr(mask_col), k, _ptr(idx), _ptr(val), _ptr(ws),
                                             _ptr(warm_idx), _ptr(flag), _ptr(order), mode, _ptr(uws), _stream()), 'arl_score_mask_topk_f32')
    if mode == 1:
        _exit_probe_record(probe, ws, soff, nst)
    if timed:
        torch.cuda.synchronize(); TOPK_STATS.setdefault('ms', []).append(1e3 * (time.perf_counter() - t0))
    if evs is not None:
        evs[1].record(); TOPK_STATS.setdefault('events', []).append(evs)
    TOPK_STATS['calls'] += 1
    TOPK_STATS['warm'] += flag is not None
    if ws is not None and TOPK_STATS.get('record_exit'):      # diagnostics: the pass's early-exit counters (a 16-byte device copy, no synchronisation)
        off = _lib.lib().arl_score_mask_topk_stats_offset(I, d)
        mst = 128 if d <= 16 else (64 if d <= 64 else 32)
        TOPK_STATS.setdefault('exit', []).append((ws[off:off + 16].clone(), (I + mst - 1) // mst))
    # (a warm-started call repeats itself cold on the device when its flag is raised: nothing to wait for here; benches that count the repeats keep the flags)
    if flag is not None and TOPK_STATS.get('record_events'):
        TOPK_STATS.setdefault('flags', []).append(flag)
    return idx, val


def topk_exit_fractions(reset=True):
    """Per recorded pass (TOPK_STATS['record_exit'] = True): the share of (workgroup, stage) pairs of the item stream the early exit skipped.
    Synchronises (reads the counters)."""
    out = []
    for t, nst in TOPK_STATS.get('exit', []):
        consumed, wgs = (int(x) for x in t.view(torch.int64).tolist())
        out.append(1.0 - consumed / float(max(1, wgs * nst)) if wgs else 0.0)
    if reset:
        TOPK_STATS['exit'] = []
    return out


def topn_project_rows(M, n):
    _dev(M, torch.float32, 'M', 2)
    rows, cols = M.shape
    if not (0 <= n <= cols):
        raise ValueError('topn_project_rows: bad n')
    out = torch.empty_like(M)
    idx = torch.empty(rows, max(n, 1), dtype=torch.int32, device=M.device)
    scratch = torch.empty_like(M)
    check(_lib.lib().arl_topn_project_rows_f32(_ptr(M), rows, cols, n, _ptr(out), _ptr(idx), _ptr(scratch), _stream()), 'arl_topn_project_rows_f32')
    return out, idx[:, :n]


# ------------------------------------------------------------------------------------------------ NGCF layer glue
def ngcf_combine(P, E, out=None):
    """[P + E | P * E]  -> [n, 2d] (recommender/NGCF.py:204-206, the operands of the layer's two d x d products)."""
    _dev(P, torch.float32, 'P', 2); _dev(E, torch.float32, 'E', 2)
    if P.shape != E.shape or P.shape[1] % 4:
        raise ValueError('ngcf_combine: P and E must have the same [n, d] shape with d % 4 == 0')
    n, d = P.shape
    ST = out if out is not None else torch.empty(n, 2 * d, dtype=torch.float32, device=P.device)
    check(_lib.lib().arl_ngcf_combine_f32(_ptr(P), _ptr(E), n, d, _ptr(ST), _stream()), 'arl_ngcf_combine_f32')
    return ST


def ngcf_act_(Z, acc=None, slope=0.01):
    """Z <- leaky_relu(Z) in place; acc += Z when given."""
    _dev(Z, torch.float32, 'Z', 2)
    if Z.shape[1] % 4 or (acc is not None and _dev(acc, torch.float32, 'acc', 2).shape != Z.shape):
        raise ValueError('ngcf_act_: bad shapes')
    check(_lib.lib().arl_ngcf_act_f32(_ptr(Z), _ptr(acc), Z.shape[0], Z.shape[1], float(slope), _stream()), 'arl_ngcf_act_f32')
    return Z


def ngcf_act_bwd(gOut, Out, slope=0.01):
    _dev(gOut, torch.float32, 'gOut', 2); _dev(Out, torch.float32, 'Out', 2)
    if gOut.shape != Out.shape or Out.shape[1] % 4:
        raise ValueError('ngcf_act_bwd: bad shapes')
    gZ = torch.empty_like(Out)
    check(_lib.lib().arl_ngcf_act_bwd_f32(_ptr(gOut), _ptr(Out), Out.shape[0], Out.shape[1], float(slope), _ptr(gZ), _stream()), 'arl_ngcf_act_bwd_f32')
    return gZ


def ngcf_combine_bwd(gST, P, E):
    _dev(gST, torch.float32, 'gST', 2); _dev(P, torch.float32, 'P', 2); _dev(E, torch.float32, 'E', 2)
    n, d = P.shape
    if E.shape != P.shape or gST.shape != (n, 2 * d) or d % 4:
        raise ValueError('ngcf_combine_bwd: bad shapes')
    gP, gE = torch.empty_like(P), torch.empty_like(E)
    check(_lib.lib().arl_ngcf_combine_bwd_f32(_ptr(gST), _ptr(P), _ptr(E), n, d, _ptr(gP), _ptr(gE), _stream()), 'arl_ngcf_combine_bwd_f32')
    return gP, gE


# ------------------------------------------------------------------------------------------------ user-sharded loss
NGCF_DENSE_WIDTHS = (16, 32, 64, 128)


def ngcf_dense_fwd(P, E, Wcat, slope=0.01, out=None):
    """leaky_relu((P + E) W1 + (P * E) W2) with Wcat = [W1; W2] ([2d, d]); the [N, 2d] operand is formed in registers (fp32 MFMA)."""
    _dev(P, torch.float32, 'P', 2); _dev(E, torch.float32, 'E', 2); _dev(Wcat, torch.float32, 'Wcat', 2)
    n, d = P.shape
    if E.shape != P.shape or Wcat.shape != (2 * d, d) or d not in NGCF_DENSE_WIDTHS:
        raise ValueError('ngcf_dense_fwd: P, E [n, d], Wcat [2d, d], d in %s' % (NGCF_DENSE_WIDTHS,))
    out = torch.empty_like(P) if out is None else _dev(out, torch.float32, 'out', 2)
    if out.shape != P.shape:
        raise ValueError('ngcf_dense_fwd: out must have the shape of P')
    check(_lib.lib().arl_ngcf_dense_fwd_f32(_ptr(P), _ptr(E), _ptr(Wcat), n, d, float(slope), _ptr(out), _stream()), 'arl_ngcf_dense_fwd_f32')
    return out


def ngcf_dense_bwd(gOut, Out, P, E, Wcat, slope=0.01):
    """Backward of ngcf_dense_fwd: returns (gP, gE, gW [2d, d])."""
    for t, nm in ((gOut, 'gOut'), (Out, 'Out'), (P, 'P'), (E, 'E'), (Wcat, 'Wcat')):
        _dev(t, torch.float32, nm, 2)
    n, d = P.shape
    if gOut.shape != P.shape or Out.shape != P.shape or E.shape != P.shape or Wcat.shape != (2 * d, d) or d not in NGCF_DENSE_WIDTHS:
        raise ValueError('ngcf_dense_bwd: gOut, Out, P, E [n, d], Wcat [2d, d], d in %s' % (NGCF_DENSE_WIDTHS,))
    Wt = Wcat.t().contiguous()
    gZ, gP, gE = torch.empty_like(P), torch.empty_like(P), torch.empty_like(P)
    L = _lib.lib()
    check(L.arl_ngcf_dense_dgrad_f32(_ptr(gOut), _ptr(Out), _ptr(P), _ptr(E), _ptr(Wt), n, d, float(slope), _ptr(gZ), _ptr(gP), _ptr(gE), _stream()),
          'arl_ngcf_dense_dgrad_f32')
    gW = torch.empty(2 * d, d, dtype=torch.float32, device=P.device)
    ws = torch.empty(max(1, L.arl_ngcf_wgrad_workspace_bytes(n, d) // 4), dtype=torch.float32, device=P.device)
    check(L.arl_ngcf_dense_wgrad_f32(_ptr(P), _ptr(E), _ptr(gZ), n, d, _ptr(gW), _ptr(ws), _stream()), 'arl_ngcf_dense_wgrad_f32')
    return gP, gE, gW


def bpr_l2_partial(emb, item_off, u, p, n, B_global, workspace, sums_out):
    """Per-sample BPR coefficients (into `workspace`) + local sums [sum loss terms, sum|u|^2, sum|p|^2] (into sums_out)."""
    _dev(emb, torch.float32, 'emb', 2); _dev(sums_out, torch.float32, 'sums_out', 1); _dev(workspace, torch.float32, 'workspace', 1)
    B = u.numel()
    for t, nm in ((u, 'u'), (p, 'p'), (n, 'n')):
        _check_idx(t, nm, emb.shape[0], B)
    if sums_out.numel() < 3 or workspace.numel() < 4 * max(B, 1) or B_global < B:
        raise ValueError('bpr_l2_partial: workspace / sums_out too small or B_global < B_local')
    check(_lib.lib().arl_bpr_l2_partial_f32(_ptr(emb), emb.shape[1], item_off, _ptr(u), _ptr(p), _ptr(n), B, B_global, _ptr(sums_out), _ptr(workspace),
                                            _stream()), 'arl_bpr_l2_partial_f32')
    return sums_out


def bpr_l2_backward(emb, item_off, u, p, n, reg, norms4, G, workspace, upstream=1.0):
    """Scatter-add the gradient of the local samples into G given the whole batch's norms (norms4[2], norms4[3])."""
    _dev(emb, torch.float32, 'emb', 2); _dev(G, torch.float32, 'G', 2); _dev(norms4, torch.float32, 'norms4', 1); _dev(workspace, torch.float32, 'workspace', 1)
    B = u.numel()
    for t, nm in ((u, 'u'), (p, 'p'), (n, 'n')):
        _check_idx(t, nm, emb.shape[0], B)
    if G.shape != emb.shape or norms4.numel() < 4 or workspace.numel() < 4 * max(B, 1):
        raise ValueError('bpr_l2_backward: shape mismatch')
    check(_lib.lib().arl_bpr_l2_backward_f32(_ptr(emb), emb.shape[1], item_off, _ptr(u), _ptr(p), _ptr(n), B, reg, upstream, _ptr(norms4), _ptr(G),
                                             _ptr(workspace), _stream()), 'arl_bpr_l2_backward_f32')
    return G


# ------------------------------------------------------------------------------------------------ L2-blocked SpMM plan
class TiledPlan:
    """Schedule of the L2-blocked SpMM (include/arlib_amd.h: arl_tiled) for a CSRGraph.

    row_groups: list of (lo, hi) row ranges binned separately (for the bipartite adjacency: [(0, U), (U, N)], so that a bin's
    rows all gather from the same table and every sweep is homogeneous).  Rows are sorted by degree and dealt snake-wise into
    bins of <= cap rows, which gives every bin (nearly) the same number of edges; the hottest rows land in different bins.
    Built with torch ops on the graph's device (one stable sort of the edge list)."""

    def __init__(self, A, row_groups=None, cap=384, col_block=16384, n_slots=256, d=64, hub_threshold=1024):
        dev = A.device
        lpr = 4 if d <= 16 else 8 if d <= 32 else 16 if d <= 64 else 32 if d <= 128 else 64
        self.n_groups = 16 * (64 // lpr)                          # lane groups per 1024-thread workgroup
        N = A.n_rows
        rp = A.rowptr.to(torch.int64)
        deg = (rp[1:] - rp[:-1])
        if row_groups is None:
            row_groups = [(0, N)]
        if cap > 65535 or cap < 1:
            raise ValueError('cap must be in [1, 65535]')
        bin_of_row = torch.full((N,), -1, dtype=torch.int64, device=dev)
        rloc = torch.zeros(N, dtype=torch.int64, device=dev)
        self.hub_threshold = int(hub_threshold)
        bins = 0
        covered = 0
        for lo, hi in row_groups:
            covered += hi - lo
            # rows longer than hub_threshold are left to the chunked CSR kernel (one lane group owning a 100k-edge row would
            # serialise the whole sweep); they are few and carry a small share of the edges
            order = torch.sort(deg[lo:hi], descending=True, stable=True)[1] + lo
            order = order[deg[order] <= self.hub_threshold]
            n = order.numel()
            if n <= 0:
                continue
            nb = -(-n // cap)
            nb = -(-nb // n_slots) * n_slots                      # whole sweeps
            k = torch.arange(n, device=dev)
            cyc, pos = k // nb, k % nb
            b = torch.where(cyc % 2 == 0, pos, nb - 1 - pos)
            bin_of_row[order] = bins + b
            rloc[order] = cyc
            bins += nb
        if covered != N:
            raise ValueError('row_groups must cover every row exactly once')
        self.n_slots, self.cap, self.col_block = int(n_slots), int(cap), int(col_block)
        self.n_sweeps = bins // n_slots
        self.n_cb = -(-A.n_cols // col_block)
        self.n_rows, self.n_cols, self.nnz, self.device = N, A.n_cols, A.nnz, dev
        self._A = A
        binned = bin_of_row >= 0
        self.hub_rows = (~binned).nonzero().squeeze(1)             # handled by the chunked CSR kernel
        bin_rows = torch.full((bins, cap), -1, dtype=torch.int32, device=dev)
        bin_rows[bin_of_row[binned], rloc[binned]] = torch.arange(N, dtype=torch.int32, device=dev)[binned]
        self.bin_rows = bin_rows.contiguous()
        # owner group of a local row: snake over the groups (local rows are in descending-degree order inside a bin)
        ng = self.n_groups
        owner = torch.where((rloc // ng) % 2 == 0, rloc % ng, ng - 1 - rloc % ng)
        row_e = torch.repeat_interleave(torch.arange(N, device=dev), deg)
        keep = binned[row_e]
        e_idx = keep.nonzero().squeeze(1)                          # CSR positions of the edges of binned rows
        row_e = row_e[e_idx]
        seg = bin_of_row[row_e] * ng + owner[row_e]                          # one contiguous edge list per (bin, owner group)
        key = (seg * self.n_cb + (A.col[e_idx].to(torch.int64) // col_block)) * cap + rloc[row_e]
        order = e_idx[torch.sort(key, stable=True)[1]]
        self.order = order.to(torch.int32) if A.nnz < 2 ** 31 else order
        self.e_col = A.col[order].contiguous()
        self.e_val = A.val[order].contiguous()
        self.e_row = rloc[torch.repeat_interleave(torch.arange(N, device=dev), deg)[order]].to(torch.int16).contiguous()
        self.nnz_binned = int(order.numel())
        # hub pass: the graph's chunk plan restricted to the hub rows, row tasks disabled (n_rows = 0)
        self.hub_graph = A.chunks_only(self.hub_rows) if self.hub_rows.numel() else None
        counts = torch.bincount(seg, minlength=bins * self.n_groups)
        flat = torch.zeros(bins * self.n_groups + 1, dtype=torch.int64, device=dev)
        flat[1:] = torch.cumsum(counts, 0)
        self.seg_ptr = flat.to(torch.int32).contiguous()
        self.group_edges_max, self.group_edges_mean = int(counts.max()), float(counts.float().mean())
        idx = None
        del row_e, seg, key, counts, flat, idx

    def update_values(self, val):
        """New edge values in CSR order (same pattern)."""
        _dev(val, torch.float32, 'val', 1)
        if val.numel() != self.nnz:
            raise ValueError('update_values: wrong length')
        self.e_val = val[self.order.long()].contiguous()
        if self.hub_graph is not None:
            self.hub_graph.val = val

    def _struct(self):
        t = _lib.arl_tiled()
        t.n_sweeps, t.n_slots, t.cap, t.n_cb, t.nnz = self.n_sweeps, self.n_slots, self.cap, self.n_cb, self.nnz_binned
        t.n_groups = self.n_groups
        t.bin_rows, t.seg_ptr = self.bin_rows.data_ptr(), self.seg_ptr.data_ptr()
        t.e_col, t.e_val, t.e_row = self.e_col.data_ptr(), self.e_val.data_ptr(), self.e_row.data_ptr()
        return t


def _check_tiled(P, X, name, rows):
    _dev(X, torch.float32, name, 2)
    if X.shape[0] != rows:
        raise ValueError('%s: %d rows, expected %d' % (name, X.shape[0], rows))
    if X.device != P.device:
        raise ValueError('%s on %s, plan on %s' % (name, X.device, P.device))
    d = X.shape[1]
    if d % 4 or d > 256 or P.cap * (d + 4) * 4 > 160 * 1024:
        raise ValueError('embedding size %d does not fit the plan (cap %d rows of LDS accumulators)' % (d, P.cap))
    return d


def spmm_tiled(P, X, alpha=1.0, beta=0.0, Z=None, zflags=None, out=None):
    """out = alpha*(A@X) + beta*Z through the L2-blocked schedule `P` (same numbers as spmm up to fp32 summation order)."""
    d = _check_tiled(P, X, 'X', P.n_cols)
    Y = torch.empty(P.n_rows, d, dtype=torch.float32, device=X.device) if out is None else out
    if _check_tiled(P, Y, 'out', P.n_rows) != d or Y.data_ptr() == X.data_ptr():
        raise ValueError('spmm_tiled: out must be [n_rows, d] and must not alias X')
    if beta != 0.0 and (Z is None or _check_tiled(P, Z, 'Z', P.n_rows) != d):
        raise ValueError('spmm_tiled: Z [n_rows, d] required when beta != 0')
    if zflags is not None:
        _check_flags(zflags, P.n_rows, 'zflags')
    t = P._struct()
    tok = EVENT_HOOK.begin('axpby') if EVENT_HOOK is not None else None
    check(_lib.lib().arl_spmm_tiled_f32(C.byref(t), _ptr(X), d, alpha, beta, _ptr(Z) if beta != 0.0 else None, _ptr(zflags), _ptr(Y), _stream()), 'arl_spmm_tiled_f32')
    if P.hub_graph is not None:          # the few rows longer than hub_threshold: chunked CSR kernel, same epilogue
        s = P.hub_graph._struct(d)
        check(_lib.lib().arl_spmm_csr_flagged_f32(C.byref(s), _ptr(X), d, None, alpha, beta, _ptr(Z) if beta != 0.0 else None, _ptr(zflags), _ptr(Y), _stream()),
              'arl_spmm_csr_flagged_f32 (hub rows)')
    if tok is not None:
        EVENT_HOOK.end(tok)
    return Y


def spmm_tiled_adam(P, X, alpha, beta, Z, Pm, M, V, lr, step, betas=(0.9, 0.999), eps=1e-8, zflags=None):
    d = _check_tiled(P, X, 'X', P.n_cols)
    for t_, nm in ((Pm, 'P'), (M, 'M'), (V, 'V')):
        if _check_tiled(P, t_, nm, P.n_rows) != d or t_.data_ptr() == X.data_ptr():
            raise ValueError('spmm_tiled_adam: %s shape/alias error' % nm)
    if beta != 0.0 and (Z is None or _check_tiled(P, Z, 'Z', P.n_rows) != d):
        raise ValueError('spmm_tiled_adam: Z shape mismatch')
    if zflags is not None:
        _check_flags(zflags, P.n_rows, 'zflags')
    t = P._struct()
    tok = EVENT_HOOK.begin('adam') if EVENT_HOOK is not None else None
    check(_lib.lib().arl_spmm_tiled_adam_f32(C.byref(t), _ptr(X), d, alpha, beta, _ptr(Z) if beta != 0.0 else None, _ptr(zflags), _ptr(Pm), _ptr(M), _ptr(V),
                                             lr, betas[0], betas[1], eps, int(step), _stream()), 'arl_spmm_tiled_adam_f32')
    if P.hub_graph is not None:
        s = P.hub_graph._struct(d)
        check(_lib.lib().arl_spmm_csr_adam_f32(C.byref(s), _ptr(X), d, alpha, beta, _ptr(Z) if beta != 0.0 else None, _ptr(zflags), _ptr(Pm), _ptr(M), _ptr(V),
                                               lr, betas[0], betas[1], eps, int(step), _stream()), 'arl_spmm_csr_adam_f32 (hub rows)')
    if tok is not None:
        EVENT_HOOK.end(tok)
