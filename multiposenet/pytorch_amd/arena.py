"""Flat parameter / gradient arenas.

All parameters of ``poseNet`` live in ONE f32 buffer; each ``nn.Parameter`` is a view into it.
Conv weights keep the reference's logical shape [Cout, Cin, R, S] (so ``state_dict`` keys and shapes
equal network/net_utils.py:32-34's HDF5 layout) but are stored [Cout][R][S][Cin] (channels_last
strides) — the K-contiguous order the MFMA kernels consume, so no per-step layout transform of the
master weights is ever needed.  The gradient arena mirrors it: wgrad kernels accumulate straight
into ``param.grad`` storage, and data-parallel all-reduce runs on contiguous slices of it (no
bucket copies).  Offsets are 256-byte aligned.
"""
import torch

ALIGN = 64  # floats


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class ParamArena(object):
    def __init__(self, named_params, device):
        """named_params: ordered list of (name, Parameter).  Re-points every parameter into a new arena."""
        self.device = torch.device(device)
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets, self.sizes = [], []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            self.sizes.append(p.numel())
            off += _align(p.numel())
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad_flat = None
        self.index = {}
        with torch.no_grad():
            for i, p in enumerate(self.params):
                v = self._view(self.flat, i, p.shape)
                v.copy_(p.data.to(self.device, dtype=torch.float32))
                p.data = v
                p.grad = None
                self.index[id(p)] = i
        self.lowp = {}          # 16-bit operand copies of the whole arena, by dtype (refreshed once per forward)

    def _view(self, flat, i, shape):
        seg = flat[self.offsets[i]: self.offsets[i] + self.sizes[i]]
        if len(shape) == 4:
            O, I, R, S = shape
            return seg.view(O, R, S, I).permute(0, 3, 1, 2)
        return seg.view(*shape)

    def owns(self, p):
        i = self.index.get(id(p))
        if i is None:
            return False
        return p.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[i] and p.device == self.flat.device

    def consistent(self):
        return all(self.owns(p) for p in self.params)

    # ---- gradients ----------------------------------------------------------------------------
    def ensure_grads(self):
        """Attach .grad views (zeroing the arena if any grad was dropped / replaced)."""
        fresh = False
        if self.grad_flat is None:
            self.grad_flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            fresh = True
        need_zero = False
        for i, p in enumerate(self.params):
            if not p.requires_grad:
                continue
            g = p.grad
            if g is None or g.data_ptr() != self.grad_flat.data_ptr() + 4 * self.offsets[i]:
                need_zero = True
                break
        if need_zero and not fresh:
            # optimizer.zero_grad(set_to_none=True) dropped the views: one memset re-zeroes everything
            self.grad_flat.zero_()
        if need_zero or fresh:
            for i, p in enumerate(self.params):
                if p.requires_grad:
                    p.grad = self._view(self.grad_flat, i, p.shape)
        return self.grad_flat

    def grad_seg(self, p):
        """Contiguous f32 [numel] slice of the gradient arena for parameter p ([Cout][R][S][Cin] order)."""
        i = self.index[id(p)]
        return self.grad_flat[self.offsets[i]: self.offsets[i] + self.sizes[i]]

    def data_seg(self, p, flat=None):
        i = self.index[id(p)]
        f = self.flat if flat is None else flat
        return f[self.offsets[i]: self.offsets[i] + self.sizes[i]]

    def trainable_runs(self):
        """Maximal contiguous [start, end) float ranges of the arena covered by trainable params."""
        runs = []
        cur = None
        for i, p in enumerate(self.params):
            if p.requires_grad:
                s, e = self.offsets[i], self.offsets[i] + _align(self.sizes[i])
                if cur is not None and cur[1] == s:
                    cur[1] = e
                else:
                    if cur is not None:
                        runs.append(tuple(cur))
                    cur = [s, e]
            else:
                if cur is not None:
                    runs.append(tuple(cur))
                    cur = None
        if cur is not None:
            runs.append(tuple(cur))
        return runs
