"""Device-resident CSR of the normalised adjacency + its SpMM schedule."""
import ctypes as C

import torch

from . import _lib


class CsrGraph:
    """rowptr int32[N+1], col int32[nnz], val fp32[nnz] on one HIP device, plus the per-wave work schedule
    rk_spmm_csr uses.  Immutable after construction, cached schedules included: everything here is read-only
    on the device and may be shared by any number of models, handles and streams.  The mutable part of an
    SpMM (arrival counters and partial-sum slots of rows longer than a workgroup) lives in a scratch block
    that every user allocates for itself with new_scratch()."""

    def __init__(self, n_rows, rowptr, col, val, class_split=0):
        self.n_rows, self.rowptr, self.col, self.val = n_rows, rowptr, col, val
        self.class_split = class_split
        self._sched = {}

    def schedule(self, dim):
        """(wave_desc int32 device tensor, n_blocks) of the SpMM work schedule for this dim (cached, read-only)."""
        if dim not in self._sched:
            self._sched[dim] = self._schedule(self.n_rows, self.rowptr, self.class_split, dim)
        return self._sched[dim][:2]

    def new_scratch(self, dim):
        """A fresh, zero-filled scratch block for SpMMs of this dim on ONE stream (None when the graph has no
        row longer than a workgroup).  One per handle / per concurrent stream; never cached here."""
        self.schedule(dim)
        words = self._sched[dim][2]
        return torch.zeros(words, device=self.device, dtype=torch.int32) if words else None

    def lds_plan(self, dim):
        """(plan int32 device tensor, _lib.LdsInfo) of the LDS-resident sliced SpMM (rk_lds_plan_*) for this dim, or None
        when the graph does not qualify (not the bipartite normalised binary adjacency, or a class table does not fit a
        CU's LDS).  Cached, read-only, shareable."""
        if not self.class_split:
            return None
        cache = self.__dict__.setdefault("_lds", {})
        if dim not in cache:
            plan, n_words, info = C.c_void_p(), C.c_int64(0), _lib.LdsInfo()
            _lib.check(_lib.lib().rk_lds_plan_build(self.class_split, self.n_rows - self.class_split, _lib.ptr(self.rowptr),
                                                    _lib.ptr(self.col), _lib.ptr(self.val), dim, _lib.stream_ptr(), C.byref(plan),
                                                    C.byref(n_words), C.byref(info)), "rk_lds_plan_build")
            if n_words.value == 0:
                cache[dim] = None
            else:
                try:
                    buf = torch.zeros(int(n_words.value) + 4, device=self.device, dtype=torch.int32)
                    buf = buf[((-buf.data_ptr() // 4) % 4):][: int(n_words.value)]   # 16-byte aligned view
                    _lib.check(_lib.lib().rk_lds_plan_upload(plan, _lib.ptr(buf), _lib.stream_ptr()), "rk_lds_plan_upload")
                finally:
                    _lib.lib().rk_lds_plan_destroy(plan)
                cache[dim] = (buf, info)
        return cache[dim]

    def spmm_lds(self, x, add=None):
        """y = A.x (+ add) through the LDS-resident kernel (row-major in and out: packs / unpacks around it)."""
        got = self.lds_plan(x.shape[1])
        if got is None:
            raise _lib.HipCallError("this graph has no LDS plan")
        plan, info = got
        x = x.contiguous()
        xs, ys = torch.empty_like(x), torch.empty_like(x)
        L = _lib.lib()
        _lib.check(L.rk_lds_pack(C.byref(info), _lib.ptr(x), _lib.ptr(xs), 1, 0, _lib.stream_ptr()), "rk_lds_pack")
        adds = None
        if add is not None:
            adds = torch.empty_like(x)
            _lib.check(L.rk_lds_pack(C.byref(info), _lib.ptr(add.contiguous()), _lib.ptr(adds), 1, 0, _lib.stream_ptr()), "rk_lds_pack")
        epi = _lib.LdsEpilogue(add=_lib.ptr(adds), y=_lib.ptr(ys), sum_scale=1.0)
        _lib.check(L.rk_spmm_lds(C.byref(info), _lib.ptr(plan), _lib.ptr(xs), C.byref(epi), _lib.stream_ptr()), "rk_spmm_lds")
        y = torch.empty_like(x)
        _lib.check(L.rk_lds_unpack(C.byref(info), _lib.ptr(ys), _lib.ptr(y), 1, 0, _lib.stream_ptr()), "rk_lds_unpack")
        return y

    def transpose_index(self):
        """tpos[e] = position of the transposed entry (col[e], row(e)) in this CSR (int32 device tensor, cached).
        The adjacency is structurally symmetric and sorted by (row, col), so listing the entries in
        (col, row) order enumerates the transposed entries in CSR order: tpos = argsort(col*N + row).
        Used by the graph dropout, whose per-entry mask makes the propagated graph non-symmetric."""
        if getattr(self, "_tpos", None) is None:
            rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.rowptr.device),
                                           (self.rowptr[1:] - self.rowptr[:-1]).long())
            key = self.col.long() * self.n_rows + rows
            self._tpos = torch.argsort(key).to(torch.int32).contiguous()
        return self._tpos

    @property
    def nnz(self):
        return self.col.numel()

    @property
    def device(self):
        return self.rowptr.device

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.rowptr.device == device:
            return self
        return CsrGraph(self.n_rows, self.rowptr.to(device), self.col.to(device), self.val.to(device), self.class_split)

    @staticmethod
    def _schedule(n_rows, rowptr, class_split, dim):
        sched, n_blocks, n_words, s_words = C.c_void_p(), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().rk_csr_schedule_build(n_rows, _lib.ptr(rowptr), class_split, dim, _lib.stream_ptr(), C.byref(sched),
                                                    C.byref(n_blocks), C.byref(n_words), C.byref(s_words)), "rk_csr_schedule_build")
        try:
            desc = torch.zeros(int(n_words.value), device=rowptr.device, dtype=torch.int32)  # descriptors, metas, packed rows
            _lib.check(_lib.lib().rk_csr_schedule_upload(sched, _lib.ptr(desc), _lib.stream_ptr()), "rk_csr_schedule_upload")
        finally:
            _lib.lib().rk_csr_schedule_destroy(sched)
        return desc, int(n_blocks.value), int(s_words.value)

    @classmethod
    def from_torch_coo(cls, coo, device, class_split=0):
        """From the reference's graph tensor: coalesced sparse COO, int64 indices, fp32 values
        (recad/dataset/implicit.py:295-296,320-326)."""
        _lib.require_gpu()
        coo = coo.coalesce()
        n = coo.shape[0]
        idx = coo.indices().to(device)
        row, col64 = idx[0].contiguous(), idx[1].contiguous()
        v = coo.values().to(device=device, dtype=torch.float32).contiguous()
        nnz = v.numel()
        rowptr = torch.empty(n + 1, device=device, dtype=torch.int32)
        col = torch.empty(max(nnz, 1), device=device, dtype=torch.int32)[:nnz]
        val = torch.empty(max(nnz, 1), device=device, dtype=torch.float32)[:nnz]
        _lib.check(_lib.lib().rk_coo_to_csr(n, nnz, _lib.ptr(row), _lib.ptr(col64), _lib.ptr(v), _lib.ptr(rowptr),
                                            _lib.ptr(col), _lib.ptr(val), _lib.stream_ptr()), "rk_coo_to_csr")
        return cls(n, rowptr, col, val, class_split)

    @classmethod
    def from_user_item_csr(cls, n_users, n_items, r_ptr, r_idx, device):
        """D^-1/2 A D^-1/2 built on device from the user->item CSR (item ids sorted per user);
        replaces ImplicitData.getSparseGraph (recad/dataset/implicit.py:243-298)."""
        _lib.require_gpu()
        r_ptr = torch.as_tensor(r_ptr, dtype=torch.int32).to(device).contiguous()
        r_idx = torch.as_tensor(r_idx, dtype=torch.int32).to(device).contiguous()
        E = r_idx.numel()
        N = n_users + n_items
        rowptr = torch.empty(N + 1, device=device, dtype=torch.int32)
        col = torch.empty(max(2 * E, 1), device=device, dtype=torch.int32)[: 2 * E]
        val = torch.empty(max(2 * E, 1), device=device, dtype=torch.float32)[: 2 * E]
        tmp = torch.empty(n_items + 1, device=device, dtype=torch.int32)
        _lib.check(_lib.lib().rk_build_norm_adj(n_users, n_items, _lib.ptr(r_ptr), _lib.ptr(r_idx), _lib.ptr(rowptr),
                                                _lib.ptr(col), _lib.ptr(val), _lib.ptr(tmp), _lib.stream_ptr()),
                   "rk_build_norm_adj")
        return cls(N, rowptr, col, val, n_users)

    def to_torch_coo(self):
        """The graph in the reference's own format (a coalesced torch sparse COO tensor)."""
        counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), counts)
        idx = torch.stack([row, self.col.long()])
        return torch.sparse_coo_tensor(idx, self.val, (self.n_rows, self.n_rows)).coalesce()

    def spmm(self, x, add=None):
        """y = A.x (+ add) for a dense [N,d] fp32 tensor (torch.sparse.mm replacement)."""
        x = x.contiguous()
        y = torch.empty_like(x)
        wave_desc, n_blocks = self.schedule(x.shape[1])
        scratch = self.new_scratch(x.shape[1])  # per call: concurrent calls on different streams stay independent
        _lib.check(_lib.lib().rk_spmm_csr(self.n_rows, _lib.ptr(self.rowptr), _lib.ptr(self.col), _lib.ptr(self.val),
                                          _lib.ptr(wave_desc), n_blocks, _lib.ptr(scratch), x.shape[1], _lib.ptr(x),
                                          _lib.ptr(add), _lib.ptr(y), _lib.stream_ptr()), "rk_spmm_csr")
        return y
