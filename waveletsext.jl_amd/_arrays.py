"""Array plumbing between Python callers and the C ABI.

The C ABI takes Julia-layout (column-major, batch last) buffers.  Callers pass either
  * numpy arrays of the Julia shape (any order; converted to Fortran order) -> host pointers, the
    library stages H2D/D2H itself, results come back as Fortran-ordered numpy arrays; or
  * torch tensors on a HIP device with the Julia shape and column-major strides (see `jl_empty`,
    `to_device`) -> device pointers, asynchronous on torch's current stream, results are torch
    tensors of the same kind.
torch is used for device memory and streams only.
"""
import ctypes

import numpy as np

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_NP2SUF = {np.dtype(np.float64): "_f64", np.dtype(np.float32): "_f32"}


def is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _colmajor_strides(shape):
    st, acc = [], 1
    for s in shape:
        st.append(acc)
        acc *= int(s)
    return tuple(st)


def jl_empty(shape, dtype, device):
    """Uninitialised device tensor with Julia shape `shape` and column-major layout."""
    shape = tuple(int(s) for s in shape)
    base = torch.empty(tuple(reversed(shape)), dtype=dtype, device=device)
    return base.permute(*reversed(range(len(shape)))) if len(shape) > 1 else base


def is_colmajor(t):
    shape = tuple(t.shape)
    if t.numel() == 0:
        return True
    want = _colmajor_strides(shape)
    return all(s == 1 or st == w for s, st, w in zip(shape, t.stride(), want))


def to_colmajor(t):
    if is_colmajor(t):
        return t
    out = jl_empty(t.shape, t.dtype, t.device)
    out.copy_(t)
    return out


def to_device(a, device="cuda"):
    """numpy array (Julia shape) -> column-major device tensor."""
    a = np.asfortranarray(a)
    t = torch.from_numpy(np.ascontiguousarray(a.T)).to(device)
    return t.permute(*reversed(range(a.ndim))) if a.ndim > 1 else t


def to_numpy(t):
    """device tensor (Julia shape) -> Fortran-ordered numpy array."""
    if not is_torch(t):
        return np.asfortranarray(t)
    t = t.detach()
    nbytes = t.numel() * t.element_size()
    if t.is_cuda and (1 << 20) <= nbytes <= (256 << 20):
        # through page-locked memory (torch's caching host allocator; empty_like keeps the strides, so column-major stays column-major): .cpu()
        # copies into pageable memory at a fraction of the link's rate
        import torch
        host = torch.empty_like(t, device="cpu", pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return np.asfortranarray(host.numpy())
    return np.asfortranarray(t.cpu().numpy())


class Arg:
    """One array argument prepared for the C ABI."""

    def __init__(self, x, dtype=None):
        if is_torch(x) and x.device.type != "cpu":
            if x.dtype not in (torch.float64, torch.float32):
                raise TypeError("element type must be Float64 or Float32")
            self.kind = "torch"
            self.arr = to_colmajor(x)
            self.dtype = np.dtype(np.float64 if x.dtype == torch.float64 else np.float32)
            self.ptr = ctypes.c_void_p(self.arr.data_ptr())
            self.device = x.device
        else:
            if is_torch(x):
                x = x.numpy()
            a = np.asarray(x)
            if dtype is not None:
                a = a.astype(dtype, copy=False)
            if a.dtype not in _NP2SUF:
                if np.issubdtype(a.dtype, np.integer) or a.dtype == np.float16:
                    a = a.astype(np.float64)
                else:
                    raise TypeError("element type must be Float64 or Float32")
            self.kind = "numpy"
            self.arr = np.asfortranarray(a)
            self.dtype = self.arr.dtype
            self.ptr = ctypes.c_void_p(self.arr.ctypes.data)
            self.device = None
        self.shape = tuple(int(s) for s in self.arr.shape)
        self.suffix = _NP2SUF[np.dtype(self.dtype)]

    def new(self, shape, dtype=None):
        """Uninitialised output of the same kind (and element type, unless `dtype` says otherwise)."""
        shape = tuple(int(s) for s in shape)
        dt = self.dtype if dtype is None else np.dtype(dtype)
        if self.kind == "torch":
            td = torch.float64 if dt == np.float64 else torch.float32
            return Arg(jl_empty(shape, td, self.device))
        return Arg(np.empty(shape, dtype=dt, order="F"))

    def stream(self):
        if self.kind == "torch":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return ctypes.c_void_p(0)


def out_arg(y, like):
    """Wrap a caller-provided output (`!` methods): must already be column-major, same kind."""
    if like.kind == "torch":
        if not (is_torch(y) and y.device == like.device and is_colmajor(y)):
            raise TypeError("output must be a column-major tensor on the same device as the input")
        a = Arg(y)
    else:
        if not (isinstance(y, np.ndarray) and y.flags.f_contiguous and y.flags.writeable):
            raise TypeError("output must be a writeable Fortran-ordered numpy array")
        a = Arg.__new__(Arg)
        a.kind, a.arr, a.dtype, a.device = "numpy", y, y.dtype, None
        a.ptr = ctypes.c_void_p(y.ctypes.data)
        a.shape = tuple(int(s) for s in y.shape)
        a.suffix = _NP2SUF[np.dtype(y.dtype)]
    if a.dtype != like.dtype:
        raise TypeError("output element type differs from the input's")
    return a


def tree_arg(tree):
    """BitVector -> (uint8 buffer kept alive, pointer, length); None -> NULL."""
    if tree is None:
        return None, ctypes.c_void_p(0), 0
    t = np.ascontiguousarray(np.asarray(tree).astype(np.uint8))
    return t, ctypes.c_void_p(t.ctypes.data), int(t.size)


def qmf_arg(wt):
    q = np.ascontiguousarray(np.asarray(wt.qmf, dtype=np.float64))
    return q, ctypes.c_void_p(q.ctypes.data), int(q.size)
