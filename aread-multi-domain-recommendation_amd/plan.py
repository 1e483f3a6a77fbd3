"""Row plan (domain-bucketed, tile-padded row order) -- thin wrapper over aread_plan_build."""
import torch

from . import _lib as L


class RowPlan:
    """Device-side plan for one call.  seg_col < 0 -> one segment (single-domain batch / wo_mask)."""

    def __init__(self, x: torch.Tensor, seg_col: int, n_seg: int, buf: torch.Tensor = None):
        """buf: reuse a plan buffer of the same (B, n_seg) (AREAD.prepare_batch double-buffers its plans)"""
        L.require_device(x)
        L.require(x, torch.int32, "x")
        self.B, self.f_in = int(x.shape[0]), int(x.shape[1])
        self.n_seg = int(n_seg) if seg_col >= 0 else 1
        self.seg_col = int(seg_col)
        lay = L.PlanLayout()
        L.check(L.lib().aread_plan_layout_get(self.B, self.n_seg, lay))
        self.layout = lay
        self.max_rows, self.max_tiles = int(lay.max_rows), int(lay.max_tiles)
        if buf is not None and (buf.numel() != int(lay.words) or buf.dtype != torch.int32 or buf.device != x.device):
            raise ValueError("RowPlan: the buffer to reuse does not fit this (B, n_seg)")
        self.buf = buf if buf is not None else torch.empty(int(lay.words), dtype=torch.int32, device=x.device)
        L.check(L.lib().aread_plan_build(L.ptr(x), self.B, self.f_in, self.seg_col, self.n_seg, L.ptr(self.buf),
                                         L.stream()))

    def _view(self, off, n):
        return self.buf[int(off):int(off) + int(n)]

    @property
    def row_sample(self): return self._view(self.layout.off_row_sample, self.max_rows)
    @property
    def sample_row(self): return self._view(self.layout.off_sample_row, self.B)
    @property
    def seg_count(self): return self._view(self.layout.off_seg_count, 64)
    @property
    def seg_start(self): return self._view(self.layout.off_seg_start, 64)
    @property
    def tile_seg(self): return self._view(self.layout.off_tile_seg, self.max_tiles)
    @property
    def tile_valid(self): return self._view(self.layout.off_tile_valid, self.max_tiles)
    @property
    def header(self): return self.buf[:16]
