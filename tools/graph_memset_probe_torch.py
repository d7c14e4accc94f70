#!/usr/bin/env python3
"""Second half of the hipGraph-memset root-cause probe (tools/graph_memset_probe.hip is the plain-HIP half, which
replays memset nodes correctly): the SAME sequence captured the way the failing round-1 test captured it -- through
torch.cuda.graph on a torch stream, hipMemsetAsync issued via ctypes on that stream -- followed by torch kernels that
consume and then dirty the buffer.  Replayed three times on changing inputs."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
from ls_amd import _lib  # noqa: E402  (loads torch's libamdhip64 first, exactly like the product does)

_lib.load()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int

dev = torch.device("cuda:0")
for nbytes in (32, 8, 4096):
    words = nbytes // 8
    buf = torch.zeros(words, dtype=torch.int64, device=dev)
    inp = torch.zeros(words, dtype=torch.int64, device=dev)
    out = torch.zeros(words, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)

    def seq():
        rc = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), 0xFF, nbytes, C.c_void_p(stream.cuda_stream))
        assert rc == 0, rc
        out.copy_(buf + inp)      # consume: -1 + inp
        buf.fill_(7)              # dirty it

    with torch.cuda.stream(stream):
        seq()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        seq()
    bad = 0
    for rep in range(3):
        inp.copy_(torch.arange(words, device=dev) + 100 * (rep + 1))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ok = bool((out == inp - 1).all())
        print(f"torch-captured hipMemsetAsync {nbytes:5d} B replay {rep}: {'ok' if ok else 'MEMSET NOT SEEN'} out[0]={int(out[0])} expected {int(inp[0]) - 1}")
        bad += not ok
    print(f"  -> {bad} bad replay(s) for {nbytes} B")
