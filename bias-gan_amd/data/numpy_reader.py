"""`numpy_reader` with the reference's pybind surface (src/numpy_reader/cpp/
numpy_reader.cpp:501-536) on top of the asynchronous staging ring
(csrc/staging_ring.hip).

Same calls -- numpy_reader(split_axis=False, device=-1), .parse, .init_file,
.get_sample, .get_batch, .finalize_file, .num_samples/.shape/.strides,
.num_inter_threads/.num_intra_threads, .set_batchsize, .print_file_info,
.enable_p2p/.disable_p2p -- plus .prefetch(filename) / .get_prefetched(): the
next file's payload is read by the ring's threads and copied to HBM on the
ring's own stream while the current step runs.  Returned tensors are owned by
PyTorch (a copy out of the ring slot), like the reference's _sample.clone().
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from .. import _lib as L

_DTYPES = {0: torch.float32, 1: torch.float64, 2: torch.int32, 3: torch.int64}


class numpy_reader:
    def __init__(self, split_axis: bool = False, device: int = -1, ring_slots: int = 3):
        self._split_axis = bool(split_axis)
        self._device = int(device)
        self.num_inter_threads = 1   # kept for API parity: the ring has one pool of reader threads
        self.num_intra_threads = 1   # chunks one payload is split into
        self._batchsize = 1
        self._ring_slots = max(2, int(ring_slots))
        self._ring = None
        self._ring_bytes = 0
        self._info: Optional[L.NpyInfo] = None
        self._file: Optional[str] = None
        self._numsample = 0
        self._shape: List[int] = []
        self._strides: List[int] = []
        self._numelem = 0
        self._pending = []  # (ticket, kind) of prefetched samples

    # ------------------------------------------------------------------ metadata
    @property
    def num_samples(self):
        return self._numsample

    @property
    def shape(self):
        return list(self._shape)

    @property
    def strides(self):
        return list(self._strides)

    def set_batchsize(self, batch_size: int):
        if batch_size == 0 or batch_size > self._numsample:
            raise IndexError("NumpyReader: the batch size has to be a positive number and must not be bigger than "
                             "the total number of samples.")
        if not self._split_axis and batch_size > 1:
            raise IndexError("NumpyReader: in order to use batching, you must have more than one sample per file. "
                             "Otherwise batch externally.")
        self._batchsize = int(batch_size)

    def parse(self, filename: str):
        info = L.NpyInfo()
        L.host_call("bg_npy_parse", filename.encode(), C.byref(info))
        shape = [int(info.shape[i]) for i in range(info.ndim)] or [1]
        if self._split_axis:
            if info.fortran_order:
                raise RuntimeError("NumpyReader: reading column-major arrays (Fortran order) is currently only "
                                   "supported if the split_axis option is false.")
            self._numsample = shape[0]
            shape = shape[1:] or [1]
        else:
            self._numsample = 1
        self._shape = shape
        n = len(shape)
        st = [1] * n
        if not info.fortran_order:
            for i in range(n - 2, -1, -1):
                st[i] = shape[i + 1] * st[i + 1]
        else:
            for i in range(1, n):
                st[i] = shape[i - 1] * st[i - 1]
        self._strides = st
        self._numelem = 1
        for s in shape:
            self._numelem *= s
        self._info = info
        return self

    def print_file_info(self):
        i = self._info
        print("Fortran Order:", "Yes" if i.fortran_order else "No")
        print("Endianess: little endian")
        print("Number of Samples:", self._numsample)
        print("Number of Elements:", self._numelem)
        print("Size per Element:", i.typesize)
        print("Type id:", _DTYPES[i.dtype_code])
        print("Shape:", tuple(self._shape))
        print("Stride:", tuple(self._strides))

    def enable_p2p(self):   # peer access is a cuFile/GDS concern of the reference; nothing to do here
        pass

    def disable_p2p(self):
        pass

    # ---------------------------------------------------------------------- ring
    def _ensure_ring(self, nbytes: int):
        if self._ring is not None and nbytes <= self._ring_bytes:
            return
        self._close_ring()
        ring = C.c_void_p()
        threads = max(1, int(self.num_intra_threads) * max(1, int(self.num_inter_threads)))
        L.host_call("bg_ring_create", self._device, self._ring_slots, nbytes, min(threads, 16), C.byref(ring))
        self._ring, self._ring_bytes = ring, nbytes

    def _close_ring(self):
        if self._ring is not None:
            L.load().bg_ring_destroy(self._ring)
            self._ring = None
            self._pending = []

    def __del__(self):
        try:
            self._close_ring()
        except Exception:
            pass

    def _stream(self):
        return L.stream() if self._device >= 0 else None

    def _submit(self, filename: str, sample: int) -> int:
        info = self._info
        bsize = self._numelem * info.typesize
        self._ensure_ring(bsize * (1 if self._batchsize == 1 else 1))
        ticket = C.c_int64()
        sample = (sample + self._numsample) % self._numsample
        L.host_call("bg_ring_submit", self._ring, filename.encode(), info.data_offset + bsize * sample, bsize,
                    max(1, int(self.num_intra_threads)), C.byref(ticket))
        return ticket.value

    def _collect(self, ticket: int, out: torch.Tensor):
        nbytes = out.numel() * out.element_size()
        L.host_call("bg_ring_copy_out", self._ring, ticket, out.data_ptr(), nbytes, self._stream())
        L.host_call("bg_ring_release", self._ring, ticket, self._stream())

    def _empty(self, lead=()):
        dev = torch.device("cuda", self._device) if self._device >= 0 else torch.device("cpu")
        dt = _DTYPES[self._info.dtype_code]
        if lead:
            return torch.empty(tuple(lead) + tuple(self._shape), dtype=dt, device=dev)
        return torch.empty_strided(tuple(self._shape), tuple(self._strides), dtype=dt, device=dev)

    # --------------------------------------------------------------- file access
    def init_file(self, filename: str):
        if self._info is None:
            self.parse(filename)
        self._file = filename

    def finalize_file(self):
        self._file = None

    def get_sample(self, element_id: int) -> torch.Tensor:
        if self._batchsize > 1:
            raise RuntimeError("NumpyReader: please use getBatch to load a batch if batch-size > 1.")
        out = self._empty()
        self._collect(self._submit(self._file, element_id), out)
        return out

    def get_batch(self, element_ids) -> torch.Tensor:
        ids = list(element_ids)
        if len(ids) != self._batchsize:
            raise RuntimeError("NumpyReader: please make sure that the number of items matches the batchsize.")
        out = self._empty(lead=(self._batchsize,))
        # as many reads in flight as the ring has slots
        tickets = []
        for k, i in enumerate(ids):
            if len(tickets) == self._ring_slots:
                j, t = tickets.pop(0)
                self._collect(t, out[j])
            tickets.append((k, self._submit(self._file, i)))
        for j, t in tickets:
            self._collect(t, out[j])
        return out

    # ------------------------------------------------------------ asynchronous use
    def prefetch(self, filename: str, element_id: int = 0):
        """Start reading (and staging to HBM) a sample of `filename`; returns immediately."""
        if self._info is None:
            self.parse(filename)
        if len(self._pending) >= self._ring_slots:
            raise RuntimeError("numpy_reader.prefetch: every ring slot is in flight; call get_prefetched() first")
        self._pending.append(self._submit(filename, element_id))

    def get_prefetched(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The oldest prefetched sample (the current stream waits for its H2D copy, the host does not).  `out`: a
        contiguous tensor of the sample's shape and dtype to receive it (a row of a batch buffer: no stacking copy)."""
        if not self._pending:
            raise RuntimeError("numpy_reader.get_prefetched: nothing was prefetched")
        if out is not None:
            if tuple(out.shape) != tuple(self._shape) or out.dtype != _DTYPES[self._info.dtype_code] or not out.is_contiguous():
                raise ValueError("numpy_reader.get_prefetched: `out` must be a contiguous tensor of the sample's shape and dtype")
            self._collect(self._pending.pop(0), out)
            return out
        out = self._empty()
        self._collect(self._pending.pop(0), out)
        return out
