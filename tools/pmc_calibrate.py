#!/usr/bin/env python3
"""Known-byte-count workload for calibrating FETCH_SIZE / WRITE_SIZE on gfx950 in THIS code's access
pattern: cdrl_gather_rows copies ROWS x ROW_ELEMS floats with 16-byte lanes (reads N bytes, writes N
bytes, N >> 256 MiB Infinity Cache), ITER times."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd import _lib  # noqa: E402

lib = _lib.load()
ROWS, ROW_ELEMS, ITER = 2048, 262144, 10          # 2 GiB per direction per iteration
src = torch.rand(ROWS, ROW_ELEMS, device='cuda:0')
dst = torch.empty_like(src)
idx = torch.arange(ROWS, dtype=torch.int32, device='cuda:0')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(ITER):
    _lib.check(lib.cdrl_gather_rows(C.c_void_p(src.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()),
                                    ROWS, ROW_ELEMS, st))
torch.cuda.synchronize()
print('bytes per direction', ROWS * ROW_ELEMS * 4 * ITER)
