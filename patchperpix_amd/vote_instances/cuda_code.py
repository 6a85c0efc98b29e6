"""Device shim with the reference's names (PatchPerPix/vote_instances/cuda_code.py:5-59).

The reference JIT-compiles templated CUDA source with pycuda and allocates CUDA managed
memory.  Here the kernels are pre-built HIP code objects inside libppp_mi355x.so
(``patchperpix_amd.backend``) and buffers are torch-ROCm tensors, so:

* ``init_cuda`` / ``delete_cuda`` / ``sync`` / ``get_cuda_stream`` keep their meaning
  (context = the torch device);
* ``alloc_zero_array`` returns a zero-filled DEVICE tensor (not a host-visible ndarray);
* ``make_kernel`` has no equivalent -- nothing is compiled at run time -- and raises.
"""
import numpy as np

from .. import backend

_NP2TORCH = {"float32": "float32", "float16": "float16", "uint32": "int32", "int32": "int32",
             "uint8": "uint8", "bool": "uint8", "int64": "int64"}


def make_kernel(code, options=None):
    raise RuntimeError(
        "patchperpix_amd does not JIT kernels: the gfx950 kernels are compiled ahead of time "
        "into libppp_mi355x.so (see include/ppp_mi355x.h)")


def init_cuda():
    """Returns the 'context': the torch device of this process (one process per GPU)."""
    import torch
    backend.lib()  # fail loudly if the HIP library is missing
    if backend.device_count() < 1 or not torch.cuda.is_available():
        raise RuntimeError("no HIP device available (patchperpix_amd has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def alloc_zero_array(shape, dtype):
    import torch
    if np.isscalar(shape):
        shape = (int(shape),)
    name = _NP2TORCH[np.dtype(dtype).name]
    return torch.zeros(tuple(int(s) for s in shape), dtype=getattr(torch, name), device="cuda")


def sync(context=None):
    import torch
    torch.cuda.synchronize()


def delete_cuda(context=None):
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def get_cuda_stream():
    import torch
    return torch.cuda.Stream()
