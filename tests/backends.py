"""Two ways to drive the C ABI from tests:
  EmuBackend : the product .hip sources compiled for the host (tests/emu) - numpy buffers;
               checks kernel logic in the GPU-less build container (`-m "not gpu"`).
  GpuBackend : the real libdynaboa_hip.so on cuda:0 - torch tensors own the memory (`-m gpu`).
Both expose the same tiny interface so tests/kernel_cases.py runs unchanged on either."""
from __future__ import annotations

import ctypes

import numpy as np


class EmuBackend:
    name = "emu"

    def __init__(self):
        from emu.build_emu import build
        from dynaboa_amd import _abi
        self.raw = ctypes.CDLL(build())      # also carries the emulator's own hooks (emu_lazy / emu_flush: stream-ordering checks)
        self.lib = _abi.bind(self.raw)
        self.stream = None
        self._keep = []          # buffers stay alive for the backend's lifetime (module-scoped fixture)

    def dev(self, a, dtype=np.float32):
        b = np.ascontiguousarray(np.asarray(a), dtype=dtype).copy()
        self._keep.append(b)
        if len(self._keep) > 4096:
            del self._keep[:2048]
        return b

    def empty(self, shape, dtype=np.float32):
        return np.full(shape, np.nan if dtype == np.float32 else 0, dtype=dtype)

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype=dtype)

    def ptr(self, b):
        return None if b is None else b.ctypes.data

    def host(self, b):
        return np.array(b)

    def nbytes(self, b):
        return b.nbytes

    def ptr_array(self, bufs):
        arr = (ctypes.c_void_p * len(bufs))(*[self.ptr(b) for b in bufs])
        return arr, ctypes.cast(arr, ctypes.c_void_p)

    def sync(self):
        pass

    def aux_stream(self):
        return None


class GpuBackend:
    name = "gpu"

    def __init__(self):
        import torch
        from dynaboa_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device("cuda:0")
        self.stream = torch.cuda.current_stream().cuda_stream
        self._keep = []

    def dev(self, a, dtype=np.float32):
        t = self.torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dtype)).to(self.device)
        self._keep.append(t)
        if len(self._keep) > 4096:
            del self._keep[:2048]
        return t

    def empty(self, shape, dtype=np.float32):
        td = {np.float32: self.torch.float32, np.int32: self.torch.int32, np.uint32: self.torch.int32}[dtype]
        t = self.torch.empty(shape, dtype=td, device=self.device)
        if dtype == np.float32:
            t.fill_(float("nan"))
        return t

    def zeros(self, shape, dtype=np.float32):
        td = {np.float32: self.torch.float32, np.int32: self.torch.int32, np.uint32: self.torch.int32}[dtype]
        return self.torch.zeros(shape, dtype=td, device=self.device)

    def ptr(self, b):
        return None if b is None else b.data_ptr()

    def host(self, b):
        self.torch.cuda.synchronize()
        return b.detach().cpu().numpy()

    def nbytes(self, b):
        return b.numel() * b.element_size()

    def ptr_array(self, bufs):
        arr = (ctypes.c_void_p * len(bufs))(*[self.ptr(b) for b in bufs])
        return arr, ctypes.cast(arr, ctypes.c_void_p)

    def sync(self):
        self.torch.cuda.synchronize()

    def aux_stream(self):
        if not hasattr(self, "_aux"):
            self._aux = self.torch.cuda.Stream()
        return self._aux.cuda_stream
