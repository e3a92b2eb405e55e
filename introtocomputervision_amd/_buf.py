"""Argument plumbing: numpy arrays take the `_host` entry points, torch CUDA tensors the
`_dev` ones.  Nothing here computes anything."""
import numpy as np

try:  # torch is plumbing for device memory / streams only
    import torch
except Exception:  # pragma: no cover
    torch = None


def is_dev(a):
    return torch is not None and isinstance(a, torch.Tensor)


def check2d(a, dtype_np, dtype_t=None, name="image"):
    if is_dev(a):
        if not a.is_cuda:
            raise ValueError(f"{name}: torch tensors must live on the GPU (numpy arrays take the host path)")
        if a.dim() != 2 or a.stride(1) != 1:
            raise ValueError(f"{name}: need a 2-D tensor with unit column stride")
        want = dtype_t if dtype_t is not None else getattr(torch, np.dtype(dtype_np).name)
        if a.dtype != want:
            raise ValueError(f"{name}: dtype {a.dtype}, expected {want}")
    else:
        if not isinstance(a, np.ndarray) or a.ndim != 2 or a.dtype != np.dtype(dtype_np):
            raise ValueError(f"{name}: need a 2-D numpy array of {np.dtype(dtype_np).name}")
        if a.strides[1] != a.itemsize:
            raise ValueError(f"{name}: need unit column stride")


def ptr(a):
    return a.data_ptr() if is_dev(a) else a.ctypes.data


def stride_bytes(a):
    if is_dev(a):
        return a.stride(0) * a.element_size() if a.shape[0] > 1 else a.shape[1] * a.element_size()
    return a.strides[0] if a.shape[0] > 1 else a.shape[1] * a.itemsize


def empty_like_shape(ref, shape, dtype_np=np.float32):
    if is_dev(ref):
        return torch.empty(shape, dtype=getattr(torch, np.dtype(dtype_np).name), device=ref.device)
    return np.empty(shape, dtype=dtype_np)


def zeros_like_shape(ref, shape, dtype_np=np.float32):
    if is_dev(ref):
        return torch.zeros(shape, dtype=getattr(torch, np.dtype(dtype_np).name), device=ref.device)
    return np.zeros(shape, dtype=dtype_np)


def stream_of(a):
    """Current torch stream handle for device calls (None for host calls)."""
    if is_dev(a):
        return torch.cuda.current_stream(a.device).cuda_stream
    return None


_pinned_words = {}


def pinned_count(ref):
    """One pinned, device-visible int64 word per (device, stream) for a count the host reads right after the call:
    the kernel stores it straight into host memory (no device-to-host copy) and read_count() polls the word -- a
    stream synchronise alone costs ~25 us here, as much as torch's .item(); the poll sees the store after ~3.
    The word starts at -1; every kernel that owns a count stores a value >= 0.  Safe to reuse: every caller reads
    it before it returns."""
    key = (ref.device.index or 0, torch.cuda.current_stream(ref.device).cuda_stream)
    ent = _pinned_words.get(key)
    if ent is None:
        if len(_pinned_words) > 64:
            _pinned_words.clear()
        t = torch.zeros((1,), dtype=torch.int64).pin_memory()
        ent = (t, t.numpy())
        _pinned_words[key] = ent
    ent[1][0] = -1
    return ent[0]


def read_count(word, ref):
    """The count a kernel stored into `word` (pinned_count).  Outputs the same launch sequence wrote are ordered
    behind it for every later operation on the stream; a host reader of those goes through torch's own copies,
    which synchronise."""
    import time
    view = word.numpy()
    t_end = time.perf_counter() + 2e-3
    while view[0] < 0 and time.perf_counter() < t_end:
        pass
    if view[0] < 0:  # not there after 2 ms: wait for the stream the ordinary way (and surface any launch error)
        torch.cuda.current_stream(ref.device).synchronize()
    return max(int(view[0]), 0)
